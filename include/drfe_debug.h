/* drfe_debug.h - test hooks of libdrfe.so: entry points that exist so that tests can hold an internal routine (the correctly
 * rounded sin / cos, the restated introsort, the device sort, the vectorised AHC trial solver) to its reference.  Not part of the
 * drop-in boundary (include/drfe.h): a DR-SLAM build never includes this file. */
#ifndef DRFE_DEBUG_H
#define DRFE_DEBUG_H
#include "drfe.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Test hook of dr_slam_amd/csrc/cr_sincos.h: correctly rounded sin / cos of n doubles in [0, 64) (host build of the routine the
 * device path uses for region2rect's direction and region_grow's seed direction); ok[i] = 0 where the rounding could not be
 * certified.  Host code. */
int drfe_debug_cr_sincos(const double* x, int n, double* s, double* c, int32_t* ok);

/* Test hook of dr_slam_amd/csrc/lsd_order_kernels.hip: n LSD ordering keys (gradient bin << 22 | y << 11 | x) sorted in place on the
 * device into std::sort's order under lsd.cpp's compare_norm (larger bins first, the order of equal bins = libstdc++'s
 * introsort's, heap-sort branch included).  *status: 0, or 1 if a range above 1024 keys exhausted introsort's depth limit (one
 * lane heap-sorts shorter ones; the caller orders such a frame on the host).  _depth: depth_limit >= 0 replaces 2 lg n, so that
 * tests reach the heap-sort branch (compare with drfe_debug_order_sort, mode 3, same depth_limit). */
int drfe_debug_device_order_sort(drfe_ctx* ctx, uint32_t* keys, size_t n, int* status);
int drfe_debug_device_order_sort_depth(drfe_ctx* ctx, uint32_t* keys, size_t n, int depth_limit, int* status);

/* Test hook of the vectorised trial-merge solver of the AHC clustering (dr_slam_amd/csrc/ahc_math_simd.h): plane fits of n (nine
 * sums, N) records by the scalar routine (mode 0), its 8-lane AVX2 (1) or 8-lane AVX-512F (2) instantiation; out8 = center,
 * normal, mse, curvature per record.  DRFE_ERR_STATE if this CPU lacks the mode.  Host code. */
int drfe_debug_ahc_trials(const double* sums9, const int32_t* N, int n, int mode, double* out8);

/* Test hook of dr_slam_amd/csrc/introsort_restated.h, the two std::sort calls whose permutation of equal keys reaches the output:
 * kind 0 = LSD's pseudo-ordering (uint32 keys: gradient bin << 22 | y << 11 | x, larger bins first: lsd.cpp compare_norm), kind 1 =
 * pcl::VoxelGrid's index sort (uint64 records: leaf << 32 | point, smaller leaves first).  recs[n] sorted in place.  mode 0:
 * std::sort with the reference's comparator; 1 / 2: the product's restatement with scalar / AVX2 stopper masks; 3: the plain
 * transcription of libstdc++'s introsort.  depth_limit >= 0 replaces the 2 lg n of modes 1-3 (reaches the heap-sort branch).
 * skip_below > 0 (kind 0, modes 1 / 2): only the keys whose bin is >= skip_below are wanted - they form a prefix of the result
 * and come out in std::sort's order, the rest follows unsorted within its bins' ranges (what the product asks for: pixels
 * without a level-line angle never seed a region).  DRFE_ERR_STATE if this CPU lacks AVX2 (mode 2).  Host code. */
int drfe_debug_order_sort(void* recs, size_t n, int kind, int mode, int depth_limit, uint32_t skip_below);

#ifdef __cplusplus
}
#endif
#endif /* DRFE_DEBUG_H */
