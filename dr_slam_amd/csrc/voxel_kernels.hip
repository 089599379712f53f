/* voxel_kernels.hip — pcl::VoxelGrid<PointXYZRGB>::applyFilter (filters/impl/voxel_grid.hpp of PCL 1.9.1, leaf 0.05 m) on the
 * device, for the per-plane loop of Frame::ComputePlanes (reference src/Frame.cc:977-1016; restated for the host in
 * planes_post.cpp: voxel_downsample, semantics in DESIGN.md section 9).
 *
 * VoxelGrid sorts (leaf index, point index) records with std::sort ON THE LEAF INDEX ALONE and sums a leaf's points in that
 * order in float: the centroid's last bits are introsort's permutation of equal keys.  introsort_device.h reproduces it; the
 * rest is parallel by construction.  One workgroup of 256 threads per plane ("job"):
 *   bounds (float min / max, order-free) -> grid origin and divisions as the host computes them -> leaf index per point ->
 *   std::sort's order (introsort's moves + stable counting passes) -> run heads by prefix sums -> one thread per leaf adds
 *   its points in sorted order and divides by the count.
 * A job whose grid would overflow int32 (PCL returns the input cloud there) or whose sort needs libstdc++'s heap-sort
 * branch reports a negative count: the caller runs that plane on the host. */
#include "drfe_internal.h"
#include "post_internal.h"
#include "introsort_device.h"
#define VOX_T 256                 /* threads per plane: 47 KB of LDS per workgroup, so three fit a CU beside other kernels' wavefronts */
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

extern "C" __global__ __launch_bounds__(VOX_T) void k_voxel_grid(const float* __restrict__ pts, const int2* __restrict__ jobs,
                                                                 unsigned long long* __restrict__ recs, unsigned long long* __restrict__ tmp,
                                                                 uint32_t* __restrict__ posL, uint32_t* __restrict__ posR,
                                                                 float* __restrict__ out, int* __restrict__ counts, float leafSize)
{
    extern __shared__ uint32_t dyn[];
    __shared__ isd::Shared<VOX_T> sh;
    __shared__ float red[6][VOX_WAVES];
    __shared__ int grid[8];                 /* bx by bz sx sxy keyBits ok total */
    const int2 job = jobs[blockIdx.x];
    const int off = job.x, n = job.y;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* P = pts + 3 * (size_t)off;
    unsigned long long* a = recs + off;
    if (n <= 0) { if (tid == 0) counts[blockIdx.x] = 0; return; }
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
    if (!grid[6]) { if (tid == 0) counts[blockIdx.x] = -1; return; }        /* "leaf size too small": PCL keeps the input */
    const int bx = grid[0], by = grid[1], bz = grid[2], sx = grid[3], sxy = grid[4], keyBits = grid[5];
    for (int i = tid; i < n; i += VOX_T) {
        const int ia = (int)(floor_f(P[3 * (size_t)i] * inv) - (float)bx);
        const int ib = (int)(floor_f(P[3 * (size_t)i + 1] * inv) - (float)by);
        const int ic = (int)(floor_f(P[3 * (size_t)i + 2] * inv) - (float)bz);
        a[i] = (unsigned long long)(unsigned)(ia + ib * sx + ic * sxy) << 32 | (unsigned)i;
    }
    int lg = 0;
    for (unsigned v = (unsigned)n; v > 1; v >>= 1) lg++;
    const int st = isd::sort<VOX_T, VoxTraits>(a, n, posL + off, posR + off, tmp + off, dyn, sh, 2 * lg, keyBits);
    if (st != 0) { if (tid == 0) counts[blockIdx.x] = -2; return; }
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
    /* centroid of a leaf: its points added in sorted order (float), divided by the count */
    float* O = out + 3 * (size_t)off;
    for (int r = tid; r < total; r += VOX_T) {
        const uint32_t first = heads[r], last = r + 1 < total ? heads[r + 1] : (uint32_t)n;
        float ax = 0.f, ay = 0.f, az = 0.f;
        for (uint32_t p = first; p < last; p++) {
            const uint32_t i = (uint32_t)a[p];
            ax += P[3 * (size_t)i]; ay += P[3 * (size_t)i + 1]; az += P[3 * (size_t)i + 2];
        }
        const float cnt = (float)(last - first);
        O[3 * (size_t)r] = ax / cnt; O[3 * (size_t)r + 1] = ay / cnt; O[3 * (size_t)r + 2] = az / cnt;
    }
    if (tid == 0) counts[blockIdx.x] = total;
}

hipError_t drfe_launch_voxel_grid(const float* d_pts, const int2* d_jobs, int njobs, unsigned long long* d_recs, unsigned long long* d_tmp,
                                  uint32_t* d_posL, uint32_t* d_posR, float* d_out, int* d_counts, float leafSize, hipStream_t s)
{
    if (njobs <= 0) return hipSuccess;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)k_voxel_grid, hipFuncAttributeMaxDynamicSharedMemorySize, ORD_DYN_LDS_BYTES(VOX_T));
        if (e != hipSuccess) return e;
        configured = true;
    }
    hipLaunchKernelGGL(k_voxel_grid, dim3(njobs), dim3(VOX_T), ORD_DYN_LDS_BYTES(VOX_T), s, d_pts, d_jobs, d_recs, d_tmp, d_posL, d_posR, d_out, d_counts, leafSize);
    return hipGetLastError();
}
