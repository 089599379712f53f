/* lines_kernels.hip — device image passes of the line-feature path.
 *
 * The arithmetic of LineSegment::ExtractLineSegment (reference src/LSDextractor.cpp:12-43) lives in
 * OpenCV 3.4 imgproc/lsd.cpp and opencv_contrib line_descriptor (not vendored; see DESIGN.md §5 for the
 * parity status).  Its image-sized, order-free passes run here; region growing, rectangle fitting and
 * the NFA test are sequential and stay on the host (lines_lsd.cpp):
 *   k_gauss_h / k_gauss_v   cv::GaussianBlur CV_8U fixed-point path (8.8 taps), BORDER_REFLECT_101
 *   k_resize_exact08        cv::resize(..., 0.8, 0.8, INTER_LINEAR_EXACT)
 *   k_ll_angle              LineSegmentDetectorImpl::ll_angle: gradient magnitude (f64), level-line
 *                           angle (fastAtan2 in f32, widened), running maximum of the magnitude
 *   k_sobel3                cv::Sobel 3x3 CV_16S (dx and dy) for the LBD descriptor
 */
#include "drfe_internal.h"
#include "lines_internal.h"
#include "lsd_rect_walk.h"
#include "../../include/drfe_math.h"

__device__ __forceinline__ int refl(int p, int n)
{
    if (p < 0) p = -p;
    if (p >= n) p = 2 * (n - 1) - p;
    return p;
}

__global__ __launch_bounds__(256) void k_gauss_h(const uint8_t* __restrict__ src, int w, int h, LineTaps taps,
                                                 uint16_t* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    src += (size_t)blockIdx.z * w * h; dst += (size_t)blockIdx.z * w * h;      /* frame */
    const uint8_t* row = src + (size_t)y * w;
    uint32_t acc = 0;
    const int r = taps.n / 2;
    for (int k = 0; k < taps.n; k++) acc += (uint32_t)taps.t[k] * row[refl(x + k - r, w)];
    dst[(size_t)y * w + x] = (uint16_t)min(acc, 65535u);
}

__global__ __launch_bounds__(256) void k_gauss_v(const uint16_t* __restrict__ src, int w, int h, LineTaps taps,
                                                 uint8_t* __restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    src += (size_t)blockIdx.z * w * h; dst += (size_t)blockIdx.z * w * h;      /* frame */
    uint32_t acc = 0;
    const int r = taps.n / 2;
    for (int k = 0; k < taps.n; k++) acc += (uint32_t)taps.t[k] * src[(size_t)refl(y + k - r, h) * w + x];
    dst[(size_t)y * w + x] = (uint8_t)min(255u, (acc + 32768u) >> 16);
}

/* source position (d + 0.5) * 1.25 - 0.5: offset = floor, 8.8 weight of the next sample = frac * 256
 * (exactly 32, 96, 160 or 224 for this ratio) */
__device__ __forceinline__ void exact_tap(int d, int srcN, int* o, int* c1)
{
    const double v = (d + 0.5) * (1.0 / 0.8) - 0.5;
    int oo = (int)floor(v);
    int cc = (int)rint((v - oo) * 256.0);
    if (oo < 0) { oo = 0; cc = 0; }
    if (oo >= srcN - 1) { oo = srcN - 1; cc = 0; }
    *o = oo; *c1 = cc;
}

__global__ __launch_bounds__(256) void k_resize_exact08(const uint8_t* __restrict__ src, int sw, int sh,
                                                        uint8_t* __restrict__ dst, int dw, int dh)
{
    const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
    if (dx >= dw) return;
    src += (size_t)blockIdx.z * sw * sh; dst += (size_t)blockIdx.z * dw * dh;  /* frame */
    int xo, xc1, yo, yc1;
    exact_tap(dx, sw, &xo, &xc1);
    exact_tap(dy, sh, &yo, &yc1);
    const int xb = min(xo + 1, sw - 1), yb = min(yo + 1, sh - 1);
    const uint32_t xc0 = 256 - xc1, yc0 = 256 - yc1;
    const uint32_t h0 = xc0 * src[(size_t)yo * sw + xo] + (uint32_t)xc1 * src[(size_t)yo * sw + xb];
    const uint32_t h1 = xc0 * src[(size_t)yb * sw + xo] + (uint32_t)xc1 * src[(size_t)yb * sw + xb];
    const uint32_t acc = yc0 * h0 + (uint32_t)yc1 * h1;
    dst[(size_t)dy * dw + dx] = (uint8_t)min(255u, (acc + 32768u) >> 16);
}

__global__ __launch_bounds__(256) void k_ll_angle(const uint8_t* __restrict__ img, int W, int H, double threshold,
                                                  double* __restrict__ modgrad, double* __restrict__ angles,
                                                  float2* __restrict__ cs, unsigned long long* __restrict__ maxGradBits)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    {   /* frame */
        const size_t fo = (size_t)blockIdx.z * W * H;
        img += fo; modgrad += fo; angles += fo; cs += fo; maxGradBits += 2 * (size_t)blockIdx.z;
    }
    const size_t o = (size_t)y * W + x;
    cs[o] = make_float2(0.f, 0.f);
    if (x == W - 1 || y == H - 1) { angles[o] = -1024.0; modgrad[o] = 0.0; return; }   /* NOTDEF border */
    const int DA = img[o + W + 1] - img[o];
    const int BC = img[o + 1] - img[o + W];
    const int gx = DA + BC, gy = DA - BC;
    const double norm = sqrt((double)(gx * gx + gy * gy) / 4.0);
    modgrad[o] = norm;
    if (norm <= threshold) angles[o] = -1024.0;
    else {
        const double a = (double)drfe_fast_atan2((float)gx, (float)(-gy)) * (3.14159265358979323846 / 180.0);
        angles[o] = a;
        /* cos(float(angle)), sin(float(angle)) that region_grow adds up for every pixel it accepts (lsd.cpp): computed
         * here once per pixel with the shared routine instead of ~2 x 10^5 libm calls per frame on the host */
        float sn, cn;
        drfe_sincos((float)a, &sn, &cn);
        cs[o] = make_float2(cn, sn);
        atomicMax(maxGradBits, (unsigned long long)__double_as_longlong(norm));   /* positive doubles order as integers */
    }
}

__global__ __launch_bounds__(256) void k_sobel3(const uint8_t* __restrict__ src, int w, int h, int16_t* __restrict__ gx,
                                                int16_t* __restrict__ gy)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    src += (size_t)blockIdx.z * w * h; gx += (size_t)blockIdx.z * w * h; gy += (size_t)blockIdx.z * w * h;   /* frame */
    const int xm = refl(x - 1, w), xp = refl(x + 1, w);
    const uint8_t* r0 = src + (size_t)refl(y - 1, h) * w;
    const uint8_t* r1 = src + (size_t)y * w;
    const uint8_t* r2 = src + (size_t)refl(y + 1, h) * w;
    const int a = r0[xm], b = r0[x], c = r0[xp], d = r1[xm], f = r1[xp], g = r2[xm], hh = r2[x], i = r2[xp];
    gx[(size_t)y * w + x] = (int16_t)((c + 2 * f + i) - (a + 2 * d + g));
    gy[(size_t)y * w + x] = (int16_t)((g + 2 * hh + i) - (a + 2 * b + c));
}

/* The pixel loop of cv::LineSegmentDetectorImpl::rect_nfa (OpenCV 3.4 imgproc/src/lsd.cpp) for one rectangle per wavefront.
 * The corner bookkeeping and the scan-line walk are order-defined double arithmetic (lx += step per row) and run
 * wave-uniformly; the pixels of a row are spread over the lanes (coalesced reads of the angle row), the two counters meet
 * in a wave reduction.
 * rectMode 0 (default) is the literal source: `struct edge { cv::Point p; bool taken; }` has INTEGER corners, the four edge
 * steps are integer quotients, and the second steps divide by (y - tailp->p.x), the x / y mix their guards test (so never
 * by zero).  rectMode 1 is the real-valued reading of round 3 (double quotients, (y - tailp->p.y) denominators, a step
 * that would divide by zero taken as 0), kept selectable until a pin against a real OpenCV 3.4.4 decides. */
/* (pixels, aligned pixels) of one rectangle, lanes over a row's pixels; every lane returns the wave's totals */
__device__ __forceinline__ int2 rect_walk_count(const RectCand& rc, const RectWalk& w, const double* __restrict__ ang, int W, int H,
                                                int lane)
{
    const double kNotDef = -1024.0, kTwoPi = 2.0 * 3.14159265358979323846, kThreeHalfPi = 3.0 * 3.14159265358979323846 / 2.0;
    double lstep = w.fl, rstep = w.fr, lx = (double)w.loX, rx = (double)w.loX;
    int total = 0, alg = 0;
    /* rows outside the image change nothing in rect_nfa (its `continue` comes before the edge steps) and pixels outside are
     * not counted: both loops are clamped to the image, which also bounds the work of a degenerate rectangle */
    const int yBeg = max(w.loY, 0), yEnd = min(w.hiY, H - 1);
    for (int y = yBeg; y <= yEnd; ++y) {
        const int xs = max((int)lx, 0), xe = min((int)rx, W - 1);
        const double* row = ang + (size_t)y * W;
        for (int x = xs + lane; x <= xe; x += 64) {
            ++total;
            const double a = row[x];
            if (a != kNotDef) {
                double d = rc.theta - a;
                if (d < 0) d = -d;
                if (d > kThreeHalfPi) { d -= kTwoPi; if (d < 0) d = -d; }
                if (d <= rc.prec) ++alg;
            }
        }
        if (y >= w.leftY) lstep = w.sl;
        if (y >= w.rightY) rstep = w.sr;
        lx += lstep;
        rx += rstep;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { total += __shfl_xor(total, o); alg += __shfl_xor(alg, o); }
    return make_int2(total, alg);
}

__global__ __launch_bounds__(64) void k_rect_counts(const RectCand* __restrict__ cands, int n, const double* __restrict__ ang, int W,
                                                    int H, int rectMode, int2* __restrict__ out)
{
    const int id = blockIdx.x, lane = threadIdx.x;
    if (id >= n) return;
    const RectCand rc = cands[id];
    const RectWalk w = rect_walk_setup(rc, rectMode);
    const int2 r = rect_walk_count(rc, w, ang, W, H, lane);
    if (lane == 0) out[id] = r;
}

hipError_t drfe_launch_rect_counts(const RectCand* d_cands, int n, const double* d_angles, int W, int H, int rectMode, int2* d_counts,
                                   hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_rect_counts, dim3(n), dim3(64), 0, s, d_cands, n, d_angles, W, H, rectMode, d_counts);
    return hipGetLastError();
}

/* BinaryDescriptor::computeLBD + binaryConversion (opencv_contrib 3.4 line_descriptor/src/binary_descriptor.cpp) for one key
 * line per wavefront.  The sample positions walk by repeated float addition (sx += dL0; x0 -= dL1), so a lane that owns band
 * row hID replays the hID row steps and then the len pixel steps in order; the 8 x 9 band moments are sums over at most 21 rows
 * in row order (lane = one moment of one band); the normalisation chain is a single lane. */
__global__ __launch_bounds__(64) void k_lbd(const LbdLine* __restrict__ lines, int n, const int16_t* __restrict__ gx,
                                            const int16_t* __restrict__ gy, int realW, int realH, LbdTables tab,
                                            uint8_t* __restrict__ out, const int* __restrict__ frameCounts, int klCap)
{
    const int NB = 9, WB = 7, height = 63;
    __shared__ float rowv[63][8];
    __shared__ float acc[8][9];
    __shared__ float des[72];
    const int id = blockIdx.x, lane = threadIdx.x;
    if (frameCounts) {
        /* batch form (k_lsd_keylines' outputs): blockIdx.y = frame, its line count at frameCounts[4 f], klCap line slots and
         * descriptor rows per frame, the Sobel images of the frames realW x realH apart */
        const int f = blockIdx.y;
        n = frameCounts[4 * (size_t)f];
        lines += (size_t)f * klCap; out += (size_t)f * klCap * 32;
        gx += (size_t)f * realW * realH; gy += (size_t)f * realW * realH;
    }
    if (id >= n) return;
    const LbdLine L = lines[id];
    const short maxX = (short)(realW - 1), maxY = (short)(realH - 1);
    const short len = (short)L.len;
    const short halfH = (height - 1) / 2, halfW = (short)((len - 1) / 2);
    const float dL0 = L.dL0, dL1 = L.dL1, dO0 = -dL1, dO1 = dL0;
    if (lane < height) {
        float x0 = -dL0 * halfW + dL1 * halfH + L.midX;
        float y0 = -dL1 * halfW - dL0 * halfH + L.midY;
        for (int r = 0; r < lane; r++) { x0 -= dL1; y0 += dL0; }
        float sx = x0, sy = y0, pL = 0, nL = 0, pO = 0, nO = 0;
        for (short wID = 0; wID < len; wID++) {
            short t = (short)roundf(sx);
            const short xc = (t < 0) ? (short)0 : (t > maxX) ? maxX : t;
            t = (short)roundf(sy);
            const short yc = (t < 0) ? (short)0 : (t > maxY) ? maxY : t;
            const short dx = gx[yc * realW + xc], dy = gy[yc * realW + xc];
            const float gDL = dx * dL0 + dy * dL1, gDO = dx * dO0 + dy * dO1;
            if (gDL > 0) pL += gDL; else nL -= gDL;
            if (gDO > 0) pO += gDO; else nO -= gDO;
            sx += dL0;
            sy += dL1;
        }
        const float cg = tab.coefG[lane];
        pL = cg * pL; nL = cg * nL; pO = cg * pO; nO = cg * nO;
        rowv[lane][0] = pL; rowv[lane][1] = nL; rowv[lane][2] = pL * pL; rowv[lane][3] = nL * nL;
        rowv[lane][4] = pO; rowv[lane][5] = nO; rowv[lane][6] = pO * pO; rowv[lane][7] = nO * nO;
    }
    __syncthreads();
    for (int item = lane; item < 72; item += 64) {
        const int q = item / NB, b = item - q * NB;
        float a = 0.f;
        /* rows of band b-1 reach b as their "next" band (coefL[i]), rows of b as their own (coefL[i + 7]), rows of b+1 as their
         * "previous" band (coefL[i + 14]): ascending row order */
        for (int hID = max(0, (b - 1) * WB); hID < min(height, (b + 2) * WB); hID++) {
            const int own = hID / WB, i = hID - own * WB;
            const float cl = tab.coefL[own == b ? i + WB : own == b + 1 ? i + 2 * WB : i];
            a += ((q & 2) ? cl * cl : cl) * rowv[hID][q];
        }
        acc[q][b] = a;
    }
    __syncthreads();
    if (lane == 0) {
        const float invN2 = (float)(1.0 / (WB * 2.0)), invN3 = (float)(1.0 / (WB * 3.0));
        for (int b = 0; b < NB; b++) {
            const float invN = (b == 0 || b == NB - 1) ? invN2 : invN3;
            float* d = des + 8 * b;
            float m = acc[0][b] * invN; d[0] = m; d[4] = sqrtf(acc[2][b] * invN - m * m);
            m = acc[1][b] * invN; d[1] = m; d[5] = sqrtf(acc[3][b] * invN - m * m);
            m = acc[4][b] * invN; d[2] = m; d[6] = sqrtf(acc[6][b] * invN - m * m);
            m = acc[5][b] * invN; d[3] = m; d[7] = sqrtf(acc[7][b] * invN - m * m);
        }
        float tm = 0, ts = 0;
        for (int b = 0; b < NB; b++) {
            const float* d = des + 8 * b;
            tm += d[0] * d[0]; tm += d[1] * d[1]; tm += d[2] * d[2]; tm += d[3] * d[3];
            ts += d[4] * d[4]; ts += d[5] * d[5]; ts += d[6] * d[6]; ts += d[7] * d[7];
        }
        tm = 1 / sqrtf(tm);
        ts = 1 / sqrtf(ts);
        for (int b = 0; b < NB; b++) {
            float* d = des + 8 * b;
            d[0] *= tm; d[1] *= tm; d[2] *= tm; d[3] *= tm; d[4] *= ts; d[5] *= ts; d[6] *= ts; d[7] *= ts;
        }
        for (int i = 0; i < 72; i++) if ((double)des[i] > 0.4) des[i] = (float)0.4;
        float nrm = 0;
        for (int i = 0; i < 72; i++) nrm += des[i] * des[i];
        nrm = 1 / sqrtf(nrm);
        for (int i = 0; i < 72; i++) des[i] = des[i] * nrm;
    }
    __syncthreads();
    if (lane < 32) {
        /* binaryConversion: 32 band pairs x 8 comparisons */
        const unsigned char pa[32] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 6, 6, 7};
        const unsigned char pb[32] = {1, 2, 3, 4, 5, 6, 2, 3, 4, 5, 6, 3, 4, 5, 6, 7, 8, 4, 5, 6, 7, 8, 5, 6, 7, 8, 6, 7, 8, 7, 8, 8};
        const float *a = des + 8 * pa[lane], *b = des + 8 * pb[lane];
        unsigned v = 0;
        for (int i = 0; i < 8; i++) if (a[i] > b[i]) v += 1u << i;
        out[(size_t)id * 32 + lane] = (uint8_t)v;
    }
}

hipError_t drfe_launch_lbd(const LbdLine* d_lines, int n, const int16_t* d_gx, const int16_t* d_gy, int w, int h,
                           const LbdTables& tab, uint8_t* d_out, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_lbd, dim3(n), dim3(64), 0, s, d_lines, n, d_gx, d_gy, w, h, tab, d_out, (const int*)nullptr, 0);
    return hipGetLastError();
}

/* the same for nframes frames at once: frame f's lines at d_lines[f * klCap ..], its count at d_frameCounts[4 f], its Sobel
 * images at d_gx / d_gy + f * w * h, its descriptor rows at d_out[f * klCap * 32 ..] */
hipError_t drfe_launch_lbd_batch(const LbdLine* d_lines, const int* d_frameCounts, int klCap, int nframes, const int16_t* d_gx, const int16_t* d_gy,
                                 int w, int h, const LbdTables& tab, uint8_t* d_out, hipStream_t s)
{
    if (nframes <= 0 || klCap <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_lbd, dim3(klCap, nframes), dim3(64), 0, s, d_lines, 0, d_gx, d_gy, w, h, tab, d_out, d_frameCounts, klCap);
    return hipGetLastError();
}

hipError_t drfe_launch_lines_passes(const uint8_t* d_img, int w, int h, const LineTaps& lsdTaps, const LineTaps& lbdTaps,
                                    LinesScratch* sc, int frame0, int nframes, double threshold, hipStream_t s)
{
    if (nframes <= 0) return hipSuccess;
    const dim3 g((w + 255) / 256, h, nframes), b(256);
    const int W = sc->sw, H = sc->sh;
    const size_t n = (size_t)w * h * frame0, ns = (size_t)W * H * frame0;
    uint16_t* tmp16 = sc->d_tmp16 + n;
    uint8_t* blur = sc->d_blur + n;
    /* LSD input: Gaussian(7x7, sigma 0.75) then exact 0.8 downscale */
    hipLaunchKernelGGL(k_gauss_h, g, b, 0, s, d_img, w, h, lsdTaps, tmp16);
    hipLaunchKernelGGL(k_gauss_v, g, b, 0, s, tmp16, w, h, lsdTaps, blur);
    hipLaunchKernelGGL(k_resize_exact08, dim3((W + 255) / 256, H, nframes), b, 0, s, blur, w, h, sc->d_scaled + ns, W, H);
    (void)hipMemsetAsync(sc->d_meta + 2 * (size_t)frame0, 0, 16 * (size_t)nframes, s);
    hipLaunchKernelGGL(k_ll_angle, dim3((W + 255) / 256, H, nframes), b, 0, s, sc->d_scaled + ns, W, H, threshold, sc->d_modgrad + ns,
                       sc->d_angles + ns, sc->d_cs + ns, sc->d_meta + 2 * (size_t)frame0);
    /* LBD input: Gaussian(5x5, sigma 1) then Sobel */
    hipLaunchKernelGGL(k_gauss_h, g, b, 0, s, d_img, w, h, lbdTaps, tmp16);
    hipLaunchKernelGGL(k_gauss_v, g, b, 0, s, tmp16, w, h, lbdTaps, blur);
    hipLaunchKernelGGL(k_sobel3, g, b, 0, s, blur, w, h, sc->d_gx + n, sc->d_gy + n);
    return hipGetLastError();
}
