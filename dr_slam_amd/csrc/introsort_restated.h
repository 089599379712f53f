/* introsort_restated.h — the two `std::sort` calls of the path whose PERMUTATION OF EQUAL KEYS reaches the output, executed with
 * the element moves libstdc++'s std::sort makes and a fraction of its branch mispredictions:
 *   lsd_order::sort    LSD's pseudo-ordering, std::sort(ordered_points, compare_norm) of cv::LineSegmentDetectorImpl::ll_angle
 *                      (modules/imgproc/src/lsd.cpp of OpenCV 3.4): the comparator looks at the gradient bin only (descending),
 *                      so 196 000 pixels share 1024 keys and the order of equal bins decides the seed order of region growing,
 *                      hence the segments;
 *   voxel_order::sort  pcl::VoxelGrid::applyFilter's std::sort(index_vector) (filters/impl/voxel_grid.hpp of PCL 1.9): the
 *                      comparator looks at the leaf index only (ascending), and the order of a leaf's points is the order of
 *                      its float centroid sums.
 * The oracle calls std::sort; this file restates what std::sort does (libstdc++ bits/stl_algo.h: introsort = median-of-three
 * Hoare partitions down to ranges of 16, heap sort below a depth limit of 2 lg n, one final insertion sort) and executes the
 * same moves differently:
 *   - a Hoare partition swaps the k-th record from the left that does not go before the pivot with the k-th from the right the
 *     pivot does not go before, for as long as the former lies left of the latter, and cuts at min(L[K], R[K-1]) (K swaps done).
 *     The two stopper sets are found 64 records at a time as bit masks (vector compares), the swaps read them off with ctz / clz:
 *     one unpredictable branch per 64 records instead of one per record (introsort_restated.inc);
 *   - the final insertion sort moves a record left past records it goes before only: it is THE stable sort of the array the
 *     partitions left - a counting sort over the 1024 bins for LSD, the insertion sort itself for the voxel records (runs of
 *     <= 16 already in order among themselves);
 *   - the heap-sort fallback (never reached on image data, kept for exactness) is libstdc++'s own std::partial_sort.
 * tests/test_host_cpu.py compares the results with std::sort's on random, constant, sorted, organ-pipe and image-like key arrays
 * of many sizes (drfe_debug_order_sort); the GPU parity tests compare segments and plane coefficients with the oracle's. */
#ifndef DRFE_INTROSORT_RESTATED_H
#define DRFE_INTROSORT_RESTATED_H

#include <stdint.h>
#include <stddef.h>
#include <algorithm>
#include <cstring>
#include <vector>
#include <immintrin.h>


/* ---- LSD: key = bin << 22 | y << 11 | x (lines_lsd.cpp), larger bins first ---- */
#define ISR_REC uint32_t
#define ISR_SHIFT 22
#define ISR_DESC 1
#define ISR_NS lsd_order_scalar
#define ISR_TARGET
#define ISR_AVX2 0
#include "introsort_restated.inc"
#undef ISR_NS
#undef ISR_TARGET
#undef ISR_AVX2
#define ISR_NS lsd_order_avx2
#define ISR_TARGET __attribute__((target("avx2,bmi,bmi2,lzcnt")))
#define ISR_AVX2 1
#include "introsort_restated.inc"
#undef ISR_NS
#undef ISR_TARGET
#undef ISR_AVX2
#undef ISR_REC
#undef ISR_SHIFT
#undef ISR_DESC

/* ---- VoxelGrid: record = leaf index << 32 | point index, smaller leaves first ---- */
#define ISR_REC uint64_t
#define ISR_SHIFT 32
#define ISR_DESC 0
#define ISR_NS voxel_order_scalar
#define ISR_TARGET
#define ISR_AVX2 0
#include "introsort_restated.inc"
#undef ISR_NS
#undef ISR_TARGET
#undef ISR_AVX2
#define ISR_NS voxel_order_avx2
#define ISR_TARGET __attribute__((target("avx2,bmi,bmi2,lzcnt")))
#define ISR_AVX2 1
#include "introsort_restated.inc"
#undef ISR_NS
#undef ISR_TARGET
#undef ISR_AVX2
#undef ISR_REC
#undef ISR_SHIFT
#undef ISR_DESC

namespace isr {
static inline bool have_avx2()
{
    static const bool v = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
    return v;
}
}  // namespace isr

namespace lsd_order {

enum { BIN_SHIFT = 22, NBINS = 1024 };
typedef lsd_order_scalar::Before Before;
static inline uint32_t bin_of(uint32_t k) { return k >> BIN_SHIFT; }

/* the final insertion sort = the stable sort by descending bin.  The array arrives in runs of <= 16 that are already in order
 * among themselves, so neighbours share bins and one counter per bin would serialise the scatter on its store-to-load chain:
 * four quarters of the array scatter side by side from their own start offsets. */
static inline void stable_by_bin(uint32_t* a, size_t n, std::vector<uint32_t>& tmp)
{
    enum { Q = 4 };
    static thread_local uint32_t cnt[Q][NBINS];
    std::memset(cnt, 0, sizeof(cnt));
    const size_t q = (n + Q - 1) / Q;
    size_t lo[Q], hi[Q];
    for (int s = 0; s < Q; s++) { lo[s] = std::min(n, (size_t)s * q); hi[s] = std::min(n, lo[s] + q); }
    for (size_t k = 0; k < q; k++)
        for (int s = 0; s < Q; s++)
            if (lo[s] + k < hi[s]) cnt[s][bin_of(a[lo[s] + k])]++;
    uint32_t run = 0;
    for (int b = NBINS - 1; b >= 0; b--)
        for (int s = 0; s < Q; s++) { const uint32_t c = cnt[s][b]; cnt[s][b] = run; run += c; }
    tmp.resize(n);
    uint32_t* out = tmp.data();
    for (size_t k = 0; k < q; k++)
        for (int s = 0; s < Q; s++)
            if (lo[s] + k < hi[s]) { const uint32_t v = a[lo[s] + k]; out[cnt[s][bin_of(v)]++] = v; }
    std::memcpy(a, out, n * sizeof(uint32_t));
}

/* a[0..n) into std::sort(a, a + n, Before())'s order.  mode: -1 best for this CPU, 0 scalar masks, 1 AVX2 masks.
 * depthOverride >= 0 replaces the depth limit 2 lg n (tests reach the heap-sort branch with it).
 * skipBelow > 0: only the keys with bin >= skipBelow are wanted - they come out in std::sort's order (a prefix of the array,
 * bins descend); ranges known to hold smaller bins only are not sorted among themselves.  LSD's region growing never seeds from
 * a pixel without a level-line angle, and those are the small bins: a third of the array. */
static inline void sort(uint32_t* a, size_t n, std::vector<uint32_t>& tmp, int mode = -1, int depthOverride = -1, uint32_t skipBelow = 0)
{
    if (n < 2) return;
    if (mode < 0) mode = isr::have_avx2() ? 1 : 0;
    if (mode == 1 && isr::have_avx2()) lsd_order_avx2::partition_phase(a, n, depthOverride, skipBelow);
    else lsd_order_scalar::partition_phase(a, n, depthOverride, skipBelow);
    stable_by_bin(a, n, tmp);
}
static inline void reference_sort(uint32_t* a, size_t n, int depthOverride) { lsd_order_scalar::reference_sort(a, n, depthOverride); }

}  // namespace lsd_order

namespace voxel_order {

typedef voxel_order_scalar::Before Before;
static inline uint64_t record(uint32_t leaf, uint32_t point) { return (uint64_t)leaf << 32 | point; }
static inline uint32_t leaf_of(uint64_t r) { return (uint32_t)(r >> 32); }
static inline uint32_t point_of(uint64_t r) { return (uint32_t)r; }

/* a[0..n) into std::sort(a, a + n, Before())'s order (arguments as lsd_order::sort) */
static inline void sort(uint64_t* a, size_t n, int mode = -1, int depthOverride = -1)
{
    if (n < 2) return;
    if (mode < 0) mode = isr::have_avx2() ? 1 : 0;
    if (mode == 1 && isr::have_avx2()) voxel_order_avx2::partition_phase(a, n, depthOverride);
    else voxel_order_scalar::partition_phase(a, n, depthOverride);
    voxel_order_scalar::insertion_pass(a, n);
}
static inline void reference_sort(uint64_t* a, size_t n, int depthOverride) { voxel_order_scalar::reference_sort(a, n, depthOverride); }

}  // namespace voxel_order
#endif
