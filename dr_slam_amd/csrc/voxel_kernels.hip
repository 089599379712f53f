/* voxel_kernels.hip — pcl::VoxelGrid<PointXYZRGB>::applyFilter (filters/impl/voxel_grid.hpp of PCL 1.9.1, leaf 0.05 m) on the
 * device, for the per-plane loop of Frame::ComputePlanes (reference src/Frame.cc:977-1016; restated for the host in
 * planes_post.cpp: voxel_downsample, semantics in DESIGN.md section 9).
 *
 * VoxelGrid sorts (leaf index, point index) records with std::sort ON THE LEAF INDEX ALONE and sums a leaf's points in that
 * order in float: the centroid's last bits are introsort's permutation of equal keys.  introsort_device.h reproduces it; the
 * rest is parallel by construction.  One workgroup of 256 threads works on a plane ("job") at a time; the launch holds as many
 * workgroups as the device keeps resident and each takes the next job from a list ordered by size, largest first
 * (k_voxel_jobs_order) - a grid of one workgroup per job, walked in index order, left the planes of equal index (the large
 * first plane of every frame) on one XCD and seven XCDs nearly idle (measured: 134 ms for the 4974 clouds of 512 frames whose
 * workgroup times add up to 9.3 s).  Per job:
 *   bounds (float min / max, order-free) -> grid origin and divisions as the host computes them -> leaf index per point ->
 *   std::sort's order (introsort's moves + stable counting passes) -> run heads by prefix sums -> one thread per leaf adds
 *   its points in sorted order and divides by the count.
 * A job whose grid would overflow int32 (PCL returns the input cloud there) or whose sort needs libstdc++'s heap-sort
 * branch reports a negative count: the caller runs that plane on the host. */
#include "drfe_internal.h"
#include <cstdlib>
#include <algorithm>
#include "post_internal.h"
#ifdef VOX_PROFILE
/* phase times summed over workgroups (100 MHz ticks of thread 0): 0 bounds + keys, 1 workgroup partitions, 2 wavefront phase,
 * 3 counting passes, 4 leaf heads, 5 centroids, 6 = workgroups, 7 = points */
__device__ unsigned long long g_voxProf[8];
/* 0 = longest workgroup (ticks), 1 = workgroups above 5 ms, 2 = above 20 ms, 3 = ticks of those that ended in the heap-sort flag, 4 = their number */
__device__ unsigned long long g_voxTail[8];
/* [0..7] workgroup ticks per XCD, [8..15] non-empty workgroups per XCD, [16] sum over workgroups of the number running when each began, [17] running now */
__device__ unsigned long long g_voxXcd[18];
#define ISD_TP(k) do { if (threadIdx.x == 0) { const unsigned long long t__ = wall_clock64(); atomicAdd(&g_voxProf[(k) + 1], t__ - sh.tp); sh.tp = t__; } } while (0)
#define VOX_TP(k) do { if (threadIdx.x == 0) { const unsigned long long t__ = wall_clock64(); atomicAdd(&g_voxProf[k], t__ - sh.tp); sh.tp = t__; } } while (0)
#else
#define VOX_TP(k)
#endif
#include "introsort_device.h"
#define VOX_T 256                 /* threads per plane: 21.8 KB of LDS per workgroup (introsort_device.h), six fit a CU by registers */
#define VOX_WAVES (VOX_T / 64)

namespace {

struct VoxTraits {
    typedef unsigned long long Rec;                      /* leaf index << 32 | point index */
    static __device__ __forceinline__ uint32_t key(unsigned long long v) { return (uint32_t)(v >> 32); }
};

/* std::floor(float) exactly as planes_post.cpp's floor_f */
__device__ __forceinline__ float floor_f(float v)
{
    if (!(fabsf(v) < 8388608.0f)) return v;
    const float t = (float)(int)v;
    return t > v ? t - 1.0f : t;
}

} // namespace

/* the non-empty jobs in descending order of size (by power-of-two class); empty jobs get their count of 0 here.
 * ctl[0] = the next list entry to take (0), ctl[1] = entries.  One workgroup. */
extern "C" __global__ __launch_bounds__(1024) void k_voxel_jobs_order(const int2* __restrict__ jobs, int njobs, int* __restrict__ list,
                                                                      int* __restrict__ ctl, int* __restrict__ counts)
{
    __shared__ int hist[32], cursor[32];
    const int tid = threadIdx.x;
    if (tid < 32) hist[tid] = 0;
    __syncthreads();
    for (int j = tid; j < njobs; j += 1024) {
        const int n = jobs[j].y;
        if (n > 0) atomicAdd(&hist[31 - __clz(n)], 1);
        else counts[j] = 0;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int b = 31; b >= 0; b--) { cursor[b] = run; run += hist[b]; }
        ctl[0] = 0; ctl[1] = run;
    }
    __syncthreads();
    for (int j = tid; j < njobs; j += 1024) {
        const int n = jobs[j].y;
        if (n > 0) list[atomicAdd(&cursor[31 - __clz(n)], 1)] = j;
    }
}

/* 96 VGPRs (four of them spilled) instead of the 112 the compiler takes: five workgroups per CU instead of four; the planes path at
 * saturation +1.6 ... +9 % depending on the box (four alternations each, profiles/r05_occupancy3_variants.txt).  -DVOX_WAVES_PER_EU=4 / 6: A/B */
#ifndef VOX_WAVES_PER_EU
#define VOX_WAVES_PER_EU 5
#endif
#define VOX_OCC __attribute__((amdgpu_waves_per_eu(VOX_WAVES_PER_EU, VOX_WAVES_PER_EU)))
extern "C" __global__ __launch_bounds__(VOX_T) VOX_OCC void k_voxel_grid(const float* __restrict__ pts, const int2* __restrict__ jobs,
                                                                 const int* __restrict__ list, int* __restrict__ ctl,
                                                                 unsigned long long* __restrict__ recs, unsigned long long* __restrict__ tmp,
                                                                 uint32_t* __restrict__ posL, uint32_t* __restrict__ posR,
                                                                 float* __restrict__ out, int* __restrict__ counts, float leafSize)
{
    extern __shared__ uint32_t dyn[];
    __shared__ isd::Shared<VOX_T> sh;
    __shared__ float red[6][VOX_WAVES];
    __shared__ int grid[8];                 /* bx by bz sx sxy keyBits ok total */
    __shared__ int nextJob;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nList = __builtin_amdgcn_readfirstlane(ctl[1]);
  for (;;) {
    /* the next job of the list; every workgroup reaches the end of the list and leaves */
    __syncthreads();
    if (tid == 0) nextJob = atomicAdd(&ctl[0], 1);
    __syncthreads();
    const int li = __builtin_amdgcn_readfirstlane(nextJob);          /* wave-uniform scalars: the branches around this loop's barriers are scalar */
    if (li >= nList) break;
    const int jb = __builtin_amdgcn_readfirstlane(list[li]);
    const int2 job = jobs[jb];
    const int off = __builtin_amdgcn_readfirstlane(job.x), n = __builtin_amdgcn_readfirstlane(job.y);
    const float* P = pts + 3 * (size_t)off;
    unsigned long long* a = recs + off;
    if (n <= 0) { if (tid == 0) counts[jb] = 0; continue; }
#ifdef VOX_PROFILE
    const unsigned long long voxT0 = wall_clock64();
    if (tid == 0) { const unsigned long long r = atomicAdd(&g_voxXcd[17], 1ull); atomicAdd(&g_voxXcd[16], r); }
    if (tid == 0) { sh.tp = voxT0; atomicAdd(&g_voxProf[6], 1ull); atomicAdd(&g_voxProf[7], (unsigned long long)n); }
#endif
    const float inv = 1.0f / leafSize;
    /* getMinMax3D */
    float lo[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, hi[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (int i = tid; i < n; i += VOX_T)
        for (int k = 0; k < 3; k++) { const float v = P[3 * (size_t)i + k]; lo[k] = fminf(lo[k], v); hi[k] = fmaxf(hi[k], v); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], o)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], o)); }
    if (lane == 0) for (int k = 0; k < 3; k++) { red[k][wv] = lo[k]; red[3 + k][wv] = hi[k]; }
    __syncthreads();
    if (tid == 0) {
        for (int k = 0; k < 3; k++) { lo[k] = red[k][0]; hi[k] = red[3 + k][0]; }
        for (int w = 1; w < VOX_WAVES; w++)
            for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], red[k][w]); hi[k] = fmaxf(hi[k], red[3 + k][w]); }
        const long long nx = (long long)((hi[0] - lo[0]) * inv) + 1, ny = (long long)((hi[1] - lo[1]) * inv) + 1, nz = (long long)((hi[2] - lo[2]) * inv) + 1;
        const int bx = (int)floor_f(lo[0] * inv), by = (int)floor_f(lo[1] * inv), bz = (int)floor_f(lo[2] * inv);
        const int ex = (int)floor_f(hi[0] * inv), ey = (int)floor_f(hi[1] * inv), ez = (int)floor_f(hi[2] * inv);
        const int sx = ex - bx + 1, sxy = sx * (ey - by + 1);
        grid[0] = bx; grid[1] = by; grid[2] = bz; grid[3] = sx; grid[4] = sxy;
        const long long maxKey = (long long)sxy * (long long)(ez - bz + 1) - 1;
        grid[6] = (nx * ny * nz > 2147483647LL || maxKey < 0 || maxKey > 2147483647LL) ? 0 : 1;
        int bits = 1;
        while (bits < 31 && (maxKey >> bits) != 0) bits++;
        grid[5] = bits;
    }
    __syncthreads();
    if (!__builtin_amdgcn_readfirstlane(grid[6])) { if (tid == 0) counts[jb] = -1; continue; }        /* "leaf size too small": PCL keeps the input */
    const int bx = grid[0], by = grid[1], bz = grid[2], sx = grid[3], sxy = grid[4], keyBits = grid[5];
    for (int i = tid; i < n; i += VOX_T) {
        const int ia = (int)(floor_f(P[3 * (size_t)i] * inv) - (float)bx);
        const int ib = (int)(floor_f(P[3 * (size_t)i + 1] * inv) - (float)by);
        const int ic = (int)(floor_f(P[3 * (size_t)i + 2] * inv) - (float)bz);
        a[i] = (unsigned long long)(unsigned)(ia + ib * sx + ic * sxy) << 32 | (unsigned)i;
    }
    int lg = 0;
    for (unsigned v = (unsigned)n; v > 1; v >>= 1) lg++;
    __syncthreads();
    VOX_TP(0);
    const int st = __builtin_amdgcn_readfirstlane(isd::sort<VOX_T, VoxTraits>(a, n, posL + off, posR + off, tmp + off, dyn, sh, 2 * lg, keyBits));
#ifdef VOX_PROFILE
    if (st != 0 && tid == 0) { atomicAdd(&g_voxTail[3], wall_clock64() - voxT0); atomicAdd(&g_voxTail[4], 1ull); atomicAdd(&g_voxXcd[17], ~0ull); }
#endif
    if (st != 0) { if (tid == 0) counts[jb] = (st & ~3) ? -(8 + (st >> 2)) : -2; continue; }       /* -2: heap-sort branch; <= -9: a loop bound */
    /* leaves = runs of equal keys: heads numbered by prefix sums, head positions to posL */
    uint32_t* heads = posL + off;
    int total = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int base = 0; base < n; base += VOX_T) {
        const int p = base + tid;
        const bool h = p < n && (p == 0 || VoxTraits::key(a[p]) != VoxTraits::key(a[p - 1]));
        const unsigned long long m = __ballot(h);
        __syncthreads();
        if (lane == 0) sh.wcnt[wv] = __popcll(m);
        __syncthreads();
        int before = 0, all = 0;
        for (int k = 0; k < VOX_WAVES; k++) { const int c = sh.wcnt[k]; if (k < wv) before += c; all += c; }
        if (h) heads[total + before + __popcll(m & lt)] = (uint32_t)p;
        total += all;
    }
    __syncthreads();
    VOX_TP(4);
    /* centroid of a leaf: its points added in sorted order (float), divided by the count */
    float* O = out + 3 * (size_t)off;
    for (int r = tid; r < total; r += VOX_T) {
        const uint32_t first = heads[r];
        uint32_t last = r + 1 < total ? heads[r + 1] : (uint32_t)n;
        if (last > (uint32_t)n) last = (uint32_t)n;
        float ax = 0.f, ay = 0.f, az = 0.f;
        for (uint32_t p = first; p < last; p++) {
            const uint32_t i = (uint32_t)a[p];
            ax += P[3 * (size_t)i]; ay += P[3 * (size_t)i + 1]; az += P[3 * (size_t)i + 2];
        }
        const float cnt = (float)(last - first);
        O[3 * (size_t)r] = ax / cnt; O[3 * (size_t)r + 1] = ay / cnt; O[3 * (size_t)r + 2] = az / cnt;
    }
    if (tid == 0) counts[jb] = total;
    __syncthreads();
    VOX_TP(5);
#ifdef VOX_PROFILE
    if (tid == 0) { const unsigned xcc = __builtin_amdgcn_s_getreg(6164) & 7u; atomicAdd(&g_voxXcd[xcc], wall_clock64() - voxT0); atomicAdd(&g_voxXcd[8 + xcc], 1ull); atomicAdd(&g_voxXcd[17], ~0ull); }
    if (tid == 0) { const unsigned long long d = wall_clock64() - voxT0; atomicMax(&g_voxTail[0], d); if (d > 500000ull) atomicAdd(&g_voxTail[1], 1ull); if (d > 2000000ull) atomicAdd(&g_voxTail[2], 1ull); }
#endif
  }
}

hipError_t drfe_launch_voxel_grid(const float* d_pts, const int2* d_jobs, int njobs, int* d_list, unsigned long long* d_recs, unsigned long long* d_tmp,
                                  uint32_t* d_posL, uint32_t* d_posR, float* d_out, int* d_counts, float leafSize, hipStream_t s)
{
    if (njobs <= 0) return hipSuccess;
    static int resident = 0;
    if (!resident) {
        hipError_t e = hipFuncSetAttribute((const void*)k_voxel_grid, hipFuncAttributeMaxDynamicSharedMemorySize, ORD_DYN_LDS_BYTES(VOX_T));
        if (e != hipSuccess) return e;
        int dev = 0, cus = 0;
        if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        /* 21.8 KB of LDS and 4 x 80 registers each: six per CU would fit, three are launched - with several steps in flight the
         * other long kernels want the LDS (measured, planes path at 1 / 4 / 5 steps in flight: 4 550 / 11 400 / 12 600 frames/s with
         * three per CU, 5 040 / 9 250 / 10 980 with six); DRFE_VOXEL_RESIDENT=<per CU> for experiments */
        const char* er = std::getenv("DRFE_VOXEL_RESIDENT");
        const int perCu = er ? std::max(1, std::min(6, std::atoi(er))) : 3;
        resident = perCu * (cus > 0 ? cus : 256);
    }
    hipLaunchKernelGGL(k_voxel_jobs_order, dim3(1), dim3(1024), 0, s, d_jobs, njobs, d_list + 2, d_list, d_counts);
    hipLaunchKernelGGL(k_voxel_grid, dim3(njobs < resident ? njobs : resident), dim3(VOX_T), ORD_DYN_LDS_BYTES(VOX_T), s, d_pts, d_jobs, (const int*)(d_list + 2), d_list,
                       d_recs, d_tmp, d_posL, d_posR, d_out, d_counts, leafSize);
    return hipGetLastError();
}

#ifdef VOX_PROFILE
extern "C" int drfe_debug_voxel_profile(unsigned long long* out8 /* 34 */)
{
    unsigned long long z[8] = {0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_voxProf), sizeof(z)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out8 + 8, HIP_SYMBOL(g_voxTail), sizeof(z)) != hipSuccess) return -1;
    unsigned long long z18[18] = {0};
    if (hipMemcpyFromSymbol(out8 + 16, HIP_SYMBOL(g_voxXcd), sizeof(z18)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_voxXcd), z18, sizeof(z18)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_voxTail), z, sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_voxProf), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
