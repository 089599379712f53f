/* normals_kernels.hip — the surface-normal pass of Frame::ComputePlanes (reference src/Frame.cc:1025-1090) on gfx950:
 * 3x-subsampled point cloud -> pcl::IntegralImageNormalEstimation (AVERAGE_3D_GRADIENT, MaxDepthChangeFactor 0.05,
 * NormalSmoothingSize 10) -> every other normal as a SurfaceNormal record.  PCL 1.9.1 semantics as written down in
 * DESIGN.md section 9; bit-exact against oracle/post_oracle.cpp (tests/test_gpu_post.py).
 *
 * PCL's three raster recurrences are order-defined (float / double rounding at every step), so they keep their order and
 * are parallelised across ROWS with a skew instead:
 *   - chamfer distance to the nearest depth discontinuity (two passes): row r may finish column c once row r-1 has finished
 *     column c+1 -> lane = row, lane r works on column t - 2r at step t; one value crosses lanes per step (LDS, double
 *     buffered, one barrier per step);
 *   - the double-precision integral images I(r+1,c+1) = (I(r,c+1) + I(r+1,c)) - I(r,c) + x: skew 1, six sums and two
 *     finite-value counts cross lanes per step.
 * One workgroup per frame, frames of a batch side by side (blockIdx.x = frame): 2 x 534 + 374 dependent steps per frame.
 * The cloud, the depth-change seeds and the normals are plain thread-per-point kernels. */
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include "post_internal.h"

#define SN_MAX_ROWS 512       /* lanes of the row-skewed kernels (ceil(h/3) rows: 160 at 640x480, 320 at 1280x960) */

struct SnDepth {
    const void* ptr; size_t frameStride, rowStride; float factor; int isU16;
    /* imDepth.convertTo(CV_32F, factor): float(raw) * factor in float32 (src/Frame.cc:113-115), as k_stereo does */
    __device__ __forceinline__ float at(int f, int y, int x) const
    {
        const size_t o = (size_t)f * frameStride + (size_t)y * rowStride + x;
        return isU16 ? (float)static_cast<const uint16_t*>(ptr)[o] * factor : static_cast<const float*>(ptr)[o];
    }
};

/* z of cloud point (r, c): the reference zeroes depths beyond Point.MaxDistance (src/Frame.cc:1036-1039) */
__device__ __forceinline__ float sn_z(const SnDepth& D, int f, int r, int c, float maxDist)
{
    const float d = D.at(f, 3 * r, 3 * c);
    return d > maxDist ? 0.f : d;
}

__device__ __forceinline__ bool sn_jump(float d, float dn)
{
    const float lim = (0.05f * (fabsf(d) + 1.0f)) * 2.0f;       /* max_depth_change_factor_ * (|depth| + 1) * 2 */
    return fabsf(d - dn) > lim || !isfinite(d) || !isfinite(dn);
}

__global__ __launch_bounds__(256) void k_sn_prepare(SnDepth D, int W, int H, float fx, float fy, float cx, float cy, float maxDist,
                                                    float* __restrict__ cloud, float* __restrict__ dist)
{
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= W * H) return;
    const int r = i / W, c = i - r * W;
    const float z = sn_z(D, f, r, c, maxDist);
    float* p = cloud + ((size_t)f * W * H + i) * 3;
    p[0] = ((float)(3 * c) - cx) * z / fx;
    p[1] = ((float)(3 * r) - cy) * z / fy;
    p[2] = z;
    /* depthChangeMap, gathered: the reference's loop over (ri < H-1, ci < W-1) clears both ends of a jumping pair */
    bool seed = false;
    if (r < H - 1 && c < W - 1) seed = sn_jump(z, sn_z(D, f, r, c + 1, maxDist)) || sn_jump(z, sn_z(D, f, r + 1, c, maxDist));
    if (!seed && c >= 1 && r < H - 1) seed = sn_jump(sn_z(D, f, r, c - 1, maxDist), z);
    if (!seed && r >= 1 && c < W - 1) seed = sn_jump(sn_z(D, f, r - 1, c, maxDist), z);
    dist[(size_t)f * W * H + i] = seed ? 0.0f : (float)(W + H);
}

/* one raster pass of the 3-4 chamfer transform (weights 1.0 / 1.4), in place.  mirror = 0: top-left to bottom-right over rows
 * 1..H-1, columns 1..W-1; mirror = 1: the reverse pass, which is the same recurrence in (H-1-r, W-1-c) coordinates -
 * including the reference's read one element past the row end (previous_row[W] is current_row[0]; next_row[-1] is
 * current_row[W-1]). */
__device__ void sn_chamfer_pass(float* __restrict__ d, int W, int H, int mirror, float* xch /* [2][blockDim.x] */)
{
    const int lane = threadIdx.x, nl = blockDim.x;
    const int r = lane;                                            /* row in pass coordinates */
    const bool on = r < H;
    const int rr = mirror ? H - 1 - r : r;
    float* row = d + (size_t)(on ? rr : 0) * W;
    auto at = [&](int c) -> float& { return row[mirror ? W - 1 - c : c]; };
    const float cur0 = on ? at(0) : 0.f;
    float pm1 = 0.f, p0 = 0.f, pn = 0.f, left = cur0;
    const int T = (W - 1) + 2 * (H - 1) + 1;
    for (int t = 0; t < T; t++) {
        const int c = t - 2 * r;
        /* what the row above produced one step ago is its column c + 1 */
        pm1 = p0; p0 = pn;
        pn = (r >= 1 && t >= 1) ? xch[((t - 1) & 1) * nl + lane - 1] : 0.f;
        float out = 0.f;
        if (on && c >= 0 && c < W) {
            if (c == 0) out = cur0;
            else if (r == 0) out = at(c);
            else {
                const float ur = (c + 1 < W ? pn : cur0) + 1.4f;
                const float mv = fminf(fminf(pm1 + 1.4f, p0 + 1.0f), fminf(left + 1.0f, ur));
                const float ce = at(c);
                out = mv < ce ? mv : ce;
                if (mv < ce) at(c) = mv;
            }
            left = out;
        }
        xch[(t & 1) * nl + lane] = out;
        __syncthreads();
    }
}

__global__ __launch_bounds__(SN_MAX_ROWS) void k_sn_chamfer(float* __restrict__ dist, int W, int H)
{
    extern __shared__ float snx[];
    float* d = dist + (size_t)blockIdx.x * W * H;
    sn_chamfer_pass(d, W, H, 0, snx);
    __syncthreads();
    sn_chamfer_pass(d, W, H, 1, snx);
}

/* IntegralImage2D<float, 3>::computeIntegralImages for the two gradient images at once (double sums + finite counts).
 * integ: [(H+1) x (W+1)] entries of 6 doubles (dX xyz, dY xyz); cnt: same grid, 2 unsigned.  Row 0 / column 0 are the
 * zero frame of the recurrence and are never read by the normal kernel (its windows start at >= 5): not stored. */
struct SnXch { double s[6]; unsigned n[2]; };

__global__ __launch_bounds__(SN_MAX_ROWS) void k_sn_integral(const float* __restrict__ cloud, int W, int H,
                                                             double* __restrict__ integ, unsigned* __restrict__ cnt)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char snraw[];
    SnXch* xch = reinterpret_cast<SnXch*>(snraw);
    const int lane = threadIdx.x, nl = blockDim.x, f = blockIdx.x;
    const int r = lane;
    const bool on = r < H;
    const float* P = cloud + (size_t)f * W * H * 3;
    const int IW = W + 1;
    double* I = integ + (size_t)f * IW * (H + 1) * 6;
    unsigned* C = cnt + (size_t)f * IW * (H + 1) * 2;
    SnXch up, upleft, left;
    for (int k = 0; k < 6; k++) { up.s[k] = upleft.s[k] = left.s[k] = 0.0; }
    up.n[0] = up.n[1] = upleft.n[0] = upleft.n[1] = left.n[0] = left.n[1] = 0;
    const int T = W + H;
    for (int t = 0; t < T; t++) {
        const int c = t - r;
        upleft = up;
        if (r >= 1 && t >= 1) up = xch[((t - 1) & 1) * nl + lane - 1];
        SnXch out = left;
        if (on && c >= 0 && c < W) {
            /* initAverage3DGradientMethod: central differences, zero on the one-point frame */
            float e[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (r >= 1 && r < H - 1 && c >= 1 && c < W - 1) {
                const float* rg = P + ((size_t)r * W + c + 1) * 3;
                const float* lf = P + ((size_t)r * W + c - 1) * 3;
                const float* dn = P + ((size_t)(r + 1) * W + c) * 3;
                const float* uq = P + ((size_t)(r - 1) * W + c) * 3;
                e[0] = rg[0] - lf[0]; e[1] = rg[1] - lf[1]; e[2] = rg[2] - lf[2];
                e[3] = dn[0] - uq[0]; e[4] = dn[1] - uq[1]; e[5] = dn[2] - uq[2];
            }
            const bool okx = isfinite((e[0] + e[1]) + e[2]), oky = isfinite((e[3] + e[4]) + e[5]);
#pragma unroll
            for (int k = 0; k < 6; k++) {
                double v = (up.s[k] + left.s[k]) - upleft.s[k];
                if (k < 3 ? okx : oky) v += (double)e[k];
                out.s[k] = v;
            }
            out.n[0] = up.n[0] + left.n[0] - upleft.n[0] + (okx ? 1u : 0u);
            out.n[1] = up.n[1] + left.n[1] - upleft.n[1] + (oky ? 1u : 0u);
            const size_t o = (size_t)(r + 1) * IW + c + 1;
#pragma unroll
            for (int k = 0; k < 6; k++) I[o * 6 + k] = out.s[k];
            C[o * 2] = out.n[0]; C[o * 2 + 1] = out.n[1];
            left = out;
        }
        xch[(t & 1) * nl + lane] = out;
        __syncthreads();
    }
}

/* computeFeatureFull (BORDER_POLICY_IGNORE, fixed smoothing) + computePointNormal (AVERAGE_3D_GRADIENT) +
 * flipNormalTowardsViewpoint; every (odd row, odd column) point also becomes a SurfaceNormal record (:1069-1090) */
__global__ __launch_bounds__(256) void k_sn_normals(const float* __restrict__ cloud, const float* __restrict__ dist,
                                                    const double* __restrict__ integ, const unsigned* __restrict__ cnt, int W,
                                                    int H, float smoothing, float* __restrict__ normals,
                                                    drfe_surface_normal* __restrict__ recs)
{
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= W * H) return;
    const int r = i / W, c = i - r * W;
    const size_t fi = (size_t)f * W * H + i;
    const float* p = cloud + fi * 3;
    const float qnan = __builtin_nanf("");
    float nx = qnan, ny = qnan, nz = qnan;
    const int border = (int)smoothing;
    if (r >= border && r < H - border && c >= border && c < W - border && isfinite(p[2])) {
        const float sm = fminf(dist[fi], smoothing);
        if (sm > 2.0f) {
            const int rw = (int)sm, rw2 = rw >> 1;
            const int IW = W + 1;
            const size_t base = (size_t)f * IW * (H + 1);
            const size_t ul = base + (size_t)(r - rw2) * IW + (c - rw2), ur = ul + rw, ll = ul + (size_t)rw * IW, lr = ll + rw;
            const unsigned cx_ = cnt[ul * 2] + cnt[lr * 2] - cnt[ur * 2] - cnt[ll * 2];
            const unsigned cy_ = cnt[ul * 2 + 1] + cnt[lr * 2 + 1] - cnt[ur * 2 + 1] - cnt[ll * 2 + 1];
            if (cx_ != 0 && cy_ != 0) {
                double g[6];
#pragma unroll
                for (int k = 0; k < 6; k++) g[k] = ((integ[lr * 6 + k] + integ[ul * 6 + k]) - integ[ur * 6 + k]) - integ[ll * 6 + k];
                /* gradient_y.cross (gradient_x) */
                const double vx = g[4] * g[2] - g[5] * g[1], vy = g[5] * g[0] - g[3] * g[2], vz = g[3] * g[1] - g[4] * g[0];
                const double len = (vx * vx + vy * vy) + vz * vz;
                if (len != 0.0) {
                    const double sl = sqrt(len);
                    nx = (float)(vx / sl); ny = (float)(vy / sl); nz = (float)(vz / sl);
                    const float ax = 0.f - p[0], ay = 0.f - p[1], az = 0.f - p[2];
                    const float ct = (ax * nx + ay * ny) + az * nz;
                    if (ct < 0) { nx *= -1; ny *= -1; nz *= -1; }
                }
            }
        }
    }
    normals[fi * 3] = nx; normals[fi * 3 + 1] = ny; normals[fi * 3 + 2] = nz;
    if ((r & 1) && (c & 1)) {
        drfe_surface_normal& o = recs[(size_t)f * (W / 2) * (H / 2) + (size_t)(r >> 1) * (W / 2) + (c >> 1)];
        o.normal[0] = nx; o.normal[1] = ny; o.normal[2] = nz;
        o.camera_position[0] = p[0]; o.camera_position[1] = p[1]; o.camera_position[2] = p[2];
        o.frame_x = c * 3; o.frame_y = r * 3;
    }
}

hipError_t drfe_launch_surface_normals(const void* d_depth, int isU16, float factor, size_t frameStride, size_t rowStride, int w,
                                       int h, const float K4[4], float maxDist, int nframes, const SnBuffers& b, hipStream_t s)
{
    const int W = drfe_sn_w(w), H = drfe_sn_h(h);
    if (H > SN_MAX_ROWS) return hipErrorInvalidValue;
    SnDepth D{d_depth, frameStride, rowStride, factor, isU16};
    const int nb = (W * H + 255) / 256;
    const int rows = (H + 63) / 64 * 64;
    hipLaunchKernelGGL(k_sn_prepare, dim3(nb, nframes), dim3(256), 0, s, D, W, H, K4[0], K4[1], K4[2], K4[3], maxDist, b.d_cloud,
                       b.d_dist);
    hipLaunchKernelGGL(k_sn_chamfer, dim3(nframes), dim3(rows), 2 * rows * sizeof(float), s, b.d_dist, W, H);
    hipLaunchKernelGGL(k_sn_integral, dim3(nframes), dim3(rows), 2 * rows * sizeof(SnXch), s, b.d_cloud, W, H, b.d_integ, b.d_cnt);
    hipLaunchKernelGGL(k_sn_normals, dim3(nb, nframes), dim3(256), 0, s, b.d_cloud, b.d_dist, b.d_integ, b.d_cnt, W, H, 10.0f,
                       b.d_normals, b.d_recs);
    return hipGetLastError();
}
