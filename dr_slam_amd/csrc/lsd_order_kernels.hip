/* lsd_order_kernels.hip — cv::LineSegmentDetectorImpl's pseudo-ordering on the device: std::sort(ordered_points, compare_norm)
 * of OpenCV 3.4 lsd.cpp (behind reference src/LSDextractor.cpp:12-43).
 *
 * compare_norm looks at the gradient bin only, 196 000 pixels share 1024 bins, and std::sort is not stable: the order of equal
 * bins - which is the seed order of region growing - is whatever libstdc++'s introsort leaves.  Only the same algorithm gives
 * the same permutation: introsort_device.h executes introsort's element moves (the way introsort_restated.h restates them for
 * the host).  One workgroup of 1024 threads per frame; keys and results stay in HBM, the device region growing reads them there. */
#include "drfe_internal.h"
#include "lines_internal.h"
#ifdef ORD_PROFILE
/* phase times summed over frames (100 MHz ticks of thread 0): 0 workgroup partitions, 1 wavefront phase, 2 counting passes, 3 = frames */
__device__ unsigned long long g_ordProf[4], g_ordWave[5];
#define ISD_WT0() do { if (threadIdx.x == 0) wtMark = wall_clock64(); } while (0)
#define ISD_WT(k) do { if (threadIdx.x == 0) { const unsigned long long t__ = wall_clock64(); atomicAdd(&g_ordWave[k], t__ - wtMark); wtMark = t__; } } while (0)
#define ISD_TP(k) do { if (threadIdx.x == 0) { const unsigned long long t__ = wall_clock64(); atomicAdd(&g_ordProf[k], t__ - sh.tp); sh.tp = t__; } } while (0)
#endif
#include "introsort_device.h"
#ifndef ORD_T
#define ORD_T 256
#endif

namespace {
/* key bin << 22 | y << 11 | x; compare_norm: a goes before b iff its bin is larger */
struct LsdKeyTraits {
    typedef uint32_t Rec;
    static __device__ __forceinline__ uint32_t key(uint32_t v) { return 1023u - (v >> 22); }
};
} // namespace

#ifdef ORD_WAVES_PER_EU                 /* experiment builds: 8 -> 64 VGPRs (2 spilled): seven workgroups per CU (the LDS bound) instead of six */
#define ORD_OCC __attribute__((amdgpu_waves_per_eu(ORD_WAVES_PER_EU, ORD_WAVES_PER_EU)))
#else
#define ORD_OCC
#endif
extern "C" __global__ __launch_bounds__(ORD_T) ORD_OCC void k_lsd_order(uint32_t* __restrict__ keysBase, size_t keyStride, int n,
                                                                uint32_t* __restrict__ posLBase, uint32_t* __restrict__ posRBase,
                                                                size_t posStride, int* __restrict__ statusBase, int statusStride,
                                                                int depthLimit)
{
    extern __shared__ uint32_t dyn[];                  /* counting passes: 32 x ORD_T counters */
    __shared__ isd::Shared<ORD_T> sh;
    uint32_t* a = keysBase + keyStride * blockIdx.x;
    uint32_t* posL = posLBase + posStride * blockIdx.x;
    uint32_t* posR = posRBase + posStride * blockIdx.x;
#ifdef ORD_PROFILE
    if (threadIdx.x == 0) { sh.tp = wall_clock64(); atomicAdd(&g_ordProf[3], 1ull); }
#endif
    /* the counting passes ping-pong between the keys and posL: ten key bits = two passes, the result lands in the keys */
    const int st = isd::sort<ORD_T, LsdKeyTraits>(a, n, posL, posR, posL, dyn, sh, depthLimit, 10);
    if (threadIdx.x == 0) statusBase[(size_t)statusStride * blockIdx.x] = st;
}

hipError_t drfe_launch_lsd_order(uint32_t* d_keys, size_t keyStride, int n, uint32_t* d_posL, uint32_t* d_posR, size_t posStride,
                                 int* d_status, int statusStride, int nframes, hipStream_t s, int depthOverride)
{
    if (nframes <= 0) return hipSuccess;
    int lg = 0;
    for (size_t v = (size_t)n; v > 1; v >>= 1) lg++;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)k_lsd_order, hipFuncAttributeMaxDynamicSharedMemorySize, ORD_DYN_LDS_BYTES(ORD_T));
        if (e != hipSuccess) return e;
        configured = true;
    }
    hipLaunchKernelGGL(k_lsd_order, dim3(nframes), dim3(ORD_T), ORD_DYN_LDS_BYTES(ORD_T), s, d_keys, keyStride, n, d_posL, d_posR, posStride, d_status,
                       statusStride, depthOverride >= 0 ? depthOverride : 2 * lg);       /* the override: tests of the heap-sort branch */
    return hipGetLastError();
}

#ifdef ORD_PROFILE
extern "C" int drfe_debug_order_profile(unsigned long long* out9)
{
    unsigned long long z[5] = {0};
    if (hipMemcpyFromSymbol(out9, HIP_SYMBOL(g_ordProf), 4 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out9 + 4, HIP_SYMBOL(g_ordWave), sizeof(z)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ordWave), z, sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_ordProf), z, 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
