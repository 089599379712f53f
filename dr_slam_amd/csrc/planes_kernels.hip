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

/* ------------------------------------------------------------------------------------------------ */
/* CAPE: one lane per PATCH x PATCH cell.
 *   PlaneDetection_CAPE::runPlaneDetection cloud + cell-major layout   src/PlaneExtractor.cpp:117-152
 *   PlaneSeg::PlaneSeg (validity, cross-search jump test, float32 sums) src/CAPE/PlaneSeg.cpp:8-94
 *   PlaneSeg::fitPlane                                                  src/CAPE/PlaneSeg.cpp:110-142
 *   cell_distance_tols                                                  src/CAPE/CAPE.cpp:70-75
 * The cell-major copy of the cloud is never built: cell-local index k = lr*PATCH + lc addresses pixel
 * (r0+lr, c0+lc) directly.  The float32 sums run sequentially in k (the canonical order of
 * oracle/cape_oracle.cpp; Eigen's own order is SIMD-width dependent). */
__device__ __forceinline__ float cape_z(const float* d, size_t rowStride, int r0, int c0, int patch, int k)
{
    const int lr = k / patch, lc = k - lr * patch;
    return d[(size_t)(r0 + lr) * rowStride + c0 + lc];
}

__global__ __launch_bounds__(64) void k_cape_cells(const float* __restrict__ depth, size_t rowStride, int w, int h,
                                                   float fx, float fy, float cx, float cy, int patch, float sinCos,
                                                   float maxMergeDist, CapeCellRec* __restrict__ out, size_t frameStride)
{
    const int nh = w / patch, nv = h / patch;
    const int cell = blockIdx.x * 64 + threadIdx.x;
    if (cell >= nh * nv) return;
    depth += frameStride * blockIdx.y; out += (size_t)nh * nv * blockIdx.y;       /* blockIdx.y = frame of a batch (frameStride 0: one frame) */
    const int r0 = (cell / nh) * patch, c0 = (cell % nh) * patch;
    const int n = patch * patch;
    CapeCellRec rec;
    for (int k = 0; k < 9; k++) rec.acc[k] = 0.0;
    for (int k = 0; k < 3; k++) { rec.mean[k] = 0.0; rec.normal[k] = 0.0; }
    rec.d = 0.0; rec.MSE = 0.f; rec.score = 0.f; rec.tol = 0.f;
    rec.planar = 1;
    int cnt = 0;
    for (int k = 0; k < n; k++) cnt += cape_z(depth, rowStride, r0, c0, patch, k) > 0 ? 1 : 0;
    rec.nr_pts = cnt;
    if (cnt < n / 2) rec.planar = 0;
    if (rec.planar) {   /* horizontal scan through the middle row */
        int jumps = 0;
        int i = patch * (patch / 2);
        const int j = i + patch;
        float zl = fmaxf(cape_z(depth, rowStride, r0, c0, patch, i), cape_z(depth, rowStride, r0, c0, patch, i + 1));
        i++;
        while (i < j) {
            const float z = cape_z(depth, rowStride, r0, c0, patch, i);
            if (z > 0 && (double)fabsf(z - zl) < 100.0) zl = z;
            else if (z > 0) jumps++;
            i++;
        }
        if (jumps > 1) rec.planar = 0;
    }
    if (rec.planar) {   /* vertical scan through the middle column */
        int jumps = 0;
        int i = patch / 2;
        const int j = n - i;
        float zl = fmaxf(cape_z(depth, rowStride, r0, c0, patch, i), cape_z(depth, rowStride, r0, c0, patch, i + patch));
        i += patch;
        while (i < j) {
            const float z = cape_z(depth, rowStride, r0, c0, patch, i);
            if (z > 0 && (double)fabsf(z - zl) < 100.0) zl = z;
            else if (z > 0) jumps++;
            i += patch;
        }
        if (jumps > 1) rec.planar = 0;
    }
    if (rec.planar) {
        const double dfx = (double)fx, dfy = (double)fy, dcx = (double)cx, dcy = (double)cy;
        float sx = 0, sy = 0, sz = 0, sxx = 0, syy = 0, szz = 0, sxy = 0, sxz = 0, syz = 0;
        float X0 = 0, Y0 = 0, Z0 = 0, X1 = 0, Y1 = 0, Z1 = 0;
        for (int lr = 0; lr < patch; lr++)
            for (int lc = 0; lc < patch; lc++) {
                const double z = (double)depth[(size_t)(r0 + lr) * rowStride + c0 + lc];
                const float X = (float)(((double)(c0 + lc) - dcx) * z / dfx);
                const float Y = (float)(((double)(r0 + lr) - dcy) * z / dfy);
                const float Z = (float)z;
                sx += X; sy += Y; sz += Z;
                sxx += X * X; syy += Y * Y; szz += Z * Z;
                sxy += X * Y; sxz += X * Z; syz += Y * Z;
                if (lr == 0 && lc == 0) { X0 = X; Y0 = Y; Z0 = Z; }
                X1 = X; Y1 = Y; Z1 = Z;
            }
        rec.acc[0] = sx; rec.acc[1] = sy; rec.acc[2] = sz; rec.acc[3] = sxx; rec.acc[4] = syy; rec.acc[5] = szz;
        rec.acc[6] = sxy; rec.acc[7] = sxz; rec.acc[8] = syz;
        const double np = (double)cnt;
        rec.mean[0] = rec.acc[0] / np; rec.mean[1] = rec.acc[1] / np; rec.mean[2] = rec.acc[2] / np;
        double ev[3], Q[9];
        ahc_eig3(rec.acc[3] - rec.acc[0] * rec.acc[0] / np, rec.acc[6] - rec.acc[0] * rec.acc[1] / np,
                 rec.acc[7] - rec.acc[0] * rec.acc[2] / np, rec.acc[4] - rec.acc[1] * rec.acc[1] / np,
                 rec.acc[8] - rec.acc[1] * rec.acc[2] / np, rec.acc[5] - rec.acc[2] * rec.acc[2] / np, ev, Q);
        double dd = -(Q[0] * rec.mean[0] + Q[1] * rec.mean[1] + Q[2] * rec.mean[2]);
        if (dd > 0) { rec.normal[0] = Q[0]; rec.normal[1] = Q[1]; rec.normal[2] = Q[2]; }
        else { rec.normal[0] = -Q[0]; rec.normal[1] = -Q[1]; rec.normal[2] = -Q[2]; dd = -dd; }
        rec.d = dd;
        rec.MSE = (float)(ev[0] / np);
        rec.score = (float)(ev[1] / ev[0]);
        const double t = 0.000001425 * rec.mean[2] * rec.mean[2] + 10;
        if ((double)rec.MSE > t * t) rec.planar = 0;
        if (rec.planar) {
            const float dx = X1 - X0, dy = Y1 - Y0, dz = Z1 - Z0;
            float sq = dx * dx;
            sq += dy * dy;
            sq += dz * dz;
            const float diameter = sqrtf(sq);
            const float tt = fminf(fmaxf(diameter * sinCos, 20.0f), maxMergeDist);
            rec.tol = (float)((double)tt * (double)tt);
        }
    }
    out[cell] = rec;
}

hipError_t drfe_launch_cape_cells(const float* d_depth, size_t rowStride, int w, int h, const float K4[4], int patch,
                                  float sinCos, float maxMergeDist, CapeCellRec* d_out, hipStream_t s)
{
    const int ncell = (w / patch) * (h / patch);
    hipLaunchKernelGGL(k_cape_cells, dim3((ncell + 63) / 64), dim3(64), 0, s, d_depth, rowStride, w, h, K4[0], K4[1],
                       K4[2], K4[3], patch, sinCos, maxMergeDist, d_out, (size_t)0);
    return hipGetLastError();
}

/* nframes depth images frameStride floats apart: cell records of frame f at d_out[f * ncell ..] */
hipError_t drfe_launch_cape_cells_batch(const float* d_depth, size_t frameStride, size_t rowStride, int w, int h, const float K4[4], int patch,
                                        float sinCos, float maxMergeDist, int nframes, CapeCellRec* d_out, hipStream_t s)
{
    if (nframes <= 0) return hipSuccess;
    const int ncell = (w / patch) * (h / patch);
    hipLaunchKernelGGL(k_cape_cells, dim3((ncell + 63) / 64, nframes), dim3(64), 0, s, d_depth, rowStride, w, h, K4[0], K4[1],
                       K4[2], K4[3], patch, sinCos, maxMergeDist, d_out, frameStride);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_cape_refine(const float* __restrict__ depth, size_t rowStride, int w, int h, float fx,
                                                     float fy, float cx, float cy, int patch, const CapeRefinePlane* __restrict__ planes,
                                                     int nplanes, const uint8_t* __restrict__ gridEroded,
                                                     const uint8_t* __restrict__ boundary, uint8_t* __restrict__ seg,
                                                     const CapeFrameOut* __restrict__ frameOut, size_t frameStride, size_t tabStride)
{
    const int pc = blockIdx.x * 256 + threadIdx.x, pr = blockIdx.y;
    if (pc >= w) return;
    const int nh = w / patch, ncell = nh * (h / patch);
    if (frameOut) {
        /* batch form (k_cape_frame's tables): blockIdx.z = frame; its table block holds CapeRefinePlane[CAPE_DEV_MAXP] |
         * gridEroded[ncell] | boundary[n][ncell] */
        const int f = blockIdx.z;
        if (frameOut[f].status != 0) return;                      /* the host finishes this frame */
        nplanes = frameOut[f].nPlanes;
        const uint8_t* tab = reinterpret_cast<const uint8_t*>(planes) + tabStride * f;
        planes = reinterpret_cast<const CapeRefinePlane*>(tab);
        gridEroded = tab + CAPE_DEV_MAXP * sizeof(CapeRefinePlane);
        boundary = gridEroded + ncell;
        depth += frameStride * f; seg += (size_t)w * h * f;
    }
    const int cell = (pr / patch) * nh + pc / patch;
    uint8_t v = gridEroded[cell];
    if (v == 0) {
        /* distances_stacked starts as memset(.., 100, ..): every float is 0x64646464 */
        float best = __uint_as_float(0x64646464u);
        float X = 0.f, Y = 0.f, Z = 0.f;
        bool have = false;
        for (int p = 0; p < nplanes; p++) {
            if (!boundary[(size_t)p * ncell + cell]) continue;
            if (!have) {
                const double z = (double)depth[(size_t)pr * rowStride + pc];
                X = (float)(((double)pc - cx) * z / fx); Y = (float)(((double)pr - cy) * z / fy); Z = (float)z;
                have = true;
            }
            const CapeRefinePlane P = planes[p];
            const float dv = X * P.nx + Y * P.ny + Z * P.nz + P.d;
            const float dist = (float)((double)dv * (double)dv);
            if (dist < P.maxDist && dist < best) { best = dist; v = (uint8_t)(p + 1); }
        }
    }
    seg[(size_t)pr * w + pc] = v;
}

hipError_t drfe_launch_cape_refine(const float* d_depth, size_t rowStride, int w, int h, const float K4[4], int patch,
                                   const CapeRefinePlane* d_planes, int nplanes, const uint8_t* d_gridEroded,
                                   const uint8_t* d_boundary, uint8_t* d_seg, hipStream_t s)
{
    hipLaunchKernelGGL(k_cape_refine, dim3((w + 255) / 256, h), dim3(256), 0, s, d_depth, rowStride, w, h, K4[0], K4[1], K4[2],
                       K4[3], patch, d_planes, nplanes, d_gridEroded, d_boundary, d_seg, (const CapeFrameOut*)nullptr, (size_t)0, (size_t)0);
    return hipGetLastError();
}

hipError_t drfe_launch_cape_refine_batch(const float* d_depth, size_t frameStride, size_t rowStride, int w, int h, const float K4[4], int patch,
                                         const uint8_t* d_tabs, size_t tabStride, const CapeFrameOut* d_frameOut, int nframes, uint8_t* d_seg,
                                         hipStream_t s)
{
    if (nframes <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_cape_refine, dim3((w + 255) / 256, h, nframes), dim3(256), 0, s, d_depth, rowStride, w, h, K4[0], K4[1], K4[2],
                       K4[3], patch, reinterpret_cast<const CapeRefinePlane*>(d_tabs), 0, (const uint8_t*)nullptr, (const uint8_t*)nullptr, d_seg,
                       d_frameOut, frameStride, tabStride);
    return hipGetLastError();
}

hipError_t drfe_launch_ahc_blocks(const uint16_t* d_depth, size_t frameStride, size_t rowStride, int w, int h,
                                  const float K4[4], float depthFactor, int nframes, AhcBlockRec* d_out, hipStream_t s)
{
    const int Nw = w / AHC_WIN, Nh = h / AHC_WIN;
    hipLaunchKernelGGL(k_ahc_blocks, dim3((Nw * Nh + 127) / 128, nframes), dim3(128), 0, s, d_depth, frameStride,
                       rowStride, w, h, K4[0], K4[1], K4[2], K4[3], depthFactor, Nw, Nh, d_out);
    return hipGetLastError();
}
