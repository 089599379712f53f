/* planes_kernels.hip — device stage of the AHC depth-plane path (SURVEY.md §8a a-17 + the PlaneSeg
 * constructor part of a-18).
 *
 *   k_ahc_blocks   PlaneDetection::readDepthImage            reference src/PlaneExtractor.cpp:28-55
 *                  + ahc::PlaneSeg(points, rid, ...) ctor     include/peac/AHCPlaneSeg.hpp:210-285
 *                  + Stats::compute                          include/peac/AHCPlaneSeg.hpp:125-156
 *                  + the T_mse(P_INIT) gate of initGraph     include/peac/AHCPlaneFitter.hpp:808-810
 *
 * One lane per 10x10 init block (3072 blocks at 640x480).  The organised float64 cloud (7.4 MB per
 * frame in the reference) is never materialised: x,y,z are recomputed from the raw CV_16U depth.  The
 * nine sums are accumulated in the reference's row-major order by a single lane because their last
 * bits feed strict comparisons later.
 */
#include "drfe_internal.h"
#include "planes_internal.h"
#include "ahc_math.h"

__device__ __forceinline__ double ahc_z(const uint16_t* d, size_t rowStride, int i, int j, double factor)
{
    double z = (double)d[(size_t)i * rowStride + j] * factor;
    if (z > 5.0) z = 0.0;          /* src/PlaneExtractor.cpp:44-48: far points become (0,0,0) */
    return z;
}

__global__ __launch_bounds__(128) void k_ahc_blocks(const uint16_t* __restrict__ depth, size_t frameStride,
                                                    size_t rowStride, int w, int h, float fx, float fy, float cx,
                                                    float cy, float depthFactor, int Nw, int Nh,
                                                    AhcBlockRec* __restrict__ out)
{
    const int b = blockIdx.x * 128 + threadIdx.x;
    const int slot = blockIdx.y;
    if (b >= Nw * Nh) return;
    const uint16_t* d = depth + (size_t)slot * frameStride;
    const int i0 = (b / Nw) * AHC_WIN, j0 = (b % Nw) * AHC_WIN;
    const double factor = (double)depthFactor;
    const double dfx = (double)fx, dfy = (double)fy, dcx = (double)cx, dcy = (double)cy;
    bool valid = true;
    double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < AHC_WIN; r++) {
        const int i = i0 + r;
        for (int c = 0; c < AHC_WIN; c++) {
            const int j = j0 + c;
            const double z = ahc_z(d, rowStride, i, j, factor);
            if (z == 0.0) valid = false;                       /* INIT_STRICT: no missing depth */
            const double tdz = 0.04 * fabs(z) + 0.02;          /* ParamSet::T_dz */
            if (j + 1 < w) {
                const double zn = ahc_z(d, rowStride, i, j + 1, factor);
                if (zn != 0.0 && fabs(z - zn) > tdz) valid = false;
            }
            if (i + 1 < h) {
                const double zn = ahc_z(d, rowStride, i + 1, j, factor);
                if (zn != 0.0 && fabs(z - zn) > tdz) valid = false;
            }
            const double x = ((double)j - dcx) * z / dfx;
            const double y = ((double)i - dcy) * z / dfy;
            S[0] += x; S[1] += y; S[2] += z;
            S[3] += x * x; S[4] += y * y; S[5] += z * z;
            S[6] += x * y; S[7] += y * z; S[8] += x * z;
        }
    }
    AhcBlockRec rec;
    rec.valid = 0;
    rec.N = 0;
    for (int k = 0; k < 9; k++) rec.sums[k] = 0.0;
    for (int k = 0; k < 3; k++) { rec.center[k] = 0.0; rec.normal[k] = 0.0; }
    rec.mse = rec.curvature = __builtin_nan("");   /* N < 4: quiet NaN, AHCPlaneSeg.hpp:268-269 */
    if (valid) {
        AhcFit f;
        ahc_plane_from_sums(S, AHC_WIN * AHC_WIN, &f);
        for (int k = 0; k < 9; k++) rec.sums[k] = S[k];
        for (int k = 0; k < 3; k++) { rec.center[k] = f.center[k]; rec.normal[k] = f.normal[k]; }
        rec.mse = f.mse; rec.curvature = f.curvature;
        rec.N = AHC_WIN * AHC_WIN;
        const double t = 1.6e-6 * f.center[2] * f.center[2] + 5.0;     /* T_mse(P_INIT, z) = t^2 */
        rec.valid = (f.mse < t * t) ? 1 : 0;
    }
    out[(size_t)slot * Nw * Nh + b] = rec;
}

hipError_t drfe_launch_ahc_blocks(const uint16_t* d_depth, size_t frameStride, size_t rowStride, int w, int h,
                                  const float K4[4], float depthFactor, int nframes, AhcBlockRec* d_out, hipStream_t s)
{
    const int Nw = w / AHC_WIN, Nh = h / AHC_WIN;
    hipLaunchKernelGGL(k_ahc_blocks, dim3((Nw * Nh + 127) / 128, nframes), dim3(128), 0, s, d_depth, frameStride,
                       rowStride, w, h, K4[0], K4[1], K4[2], K4[3], depthFactor, Nw, Nh, d_out);
    return hipGetLastError();
}
