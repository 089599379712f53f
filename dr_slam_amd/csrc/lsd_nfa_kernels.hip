/* lsd_nfa_kernels.hip - cv::LineSegmentDetectorImpl::rect_improve + rect_nfa + nfa (OpenCV 3.4 imgproc/src/lsd.cpp, the detector
 * behind reference src/LSDextractor.cpp:14-17) on the device, for the rectangles k_lsd_grow accepted.
 *
 * Only DECISIONS leave rect_improve: which candidate of a refinement stage replaces the rectangle (`v > log_nfa`) and whether
 * the segment is kept (`log_nfa > LOG_EPS`).  The reference takes them on doubles its host's libm computed (exp, log10, pow).
 * The device evaluates the same expression tree - log-gamma values, log(p), log(1 - p), log10(p) come from tables the HOST's
 * libm filled, every + - * / is the same IEEE operation - and only exp / log10 / pow are its own (<= 1 ulp each).  Each value
 * carries a bound on how far the host's can lie from it; a decision is taken when the margin exceeds the bounds (or when both
 * sides are bit-identical by construction: equal inputs, the closed-form branches), and the frame is flagged for the host's
 * validation (lines_lsd.cpp, RectValidator) otherwise.  The flagged rate is reported (DRFE_TRACE_LINES, bench.py).
 *
 * One wavefront per rectangle.  The up to five candidates of a stage depend on the stage's starting rectangle only, so their
 * pixel loops run side by side - 12 lanes each, a scan line's pixels across the lanes - and their NFAs in five lanes; the
 * choice among them follows the reference's order. */
#include <hip/hip_runtime.h>

#include "lines_internal.h"
#include "lsd_rect_walk.h"

namespace {

#define NFA_GROUP 12                      /* lanes per candidate: 5 x 12 = 60 of the 64 */
#define NFA_MAX_CAND 5

struct NfaVal { double v, e; int n, k, pj; int unc; };   /* value, bound on |host - device|, its inputs, 1 = a branch inside could not be certified */

/* nfa(n, k, p) of lsd.cpp with p = p0 / 2^pj.  T.lgamma[i] = log_gamma(double(i)) by the host (the table covers every count a
 * W x H field can produce). */
__device__ NfaVal nfa_device(int n, int k, int pj, const LsdNfaTables& T)
{
    NfaVal r;
    r.n = n; r.k = k; r.pj = pj; r.unc = 0; r.e = 0;
    if (n == 0 || k == 0) { r.v = -T.logNT; return r; }
    if (n == k) { r.v = -T.logNT - (double)n * T.log10P[pj]; return r; }
    const double p = T.p[pj];
    const double pTerm = p / (1 - p);
    const double log1 = T.lgamma[n + 1] - T.lgamma[k + 1] - T.lgamma[n - k + 1] + (double)k * T.logP[pj] + (double)(n - k) * T.log1mP[pj];
    /* term = exp(log1); double_equal(term, 0) holds iff term <= 100 eps x DBL_MIN, i.e. iff exp lands on one of the 99 smallest
     * subnormals: log1 < -739.84.  Clear of that threshold the closed form of the branch is the same IEEE expression on both
     * sides (bound 0); around it, and while the first term is subnormal at all (its bits are the libm's rounding onto the
     * subnormal grid), the value is only known to a few units - such a candidate has -log10(NFA) near 300 and does not meet
     * a comparison that could go either way (if it does, the frame goes to the host) */
    if (log1 < -739.84) {
        r.v = ((double)k > (double)n * p) ? -log1 / 2.30258509299404568402 - T.logNT : -T.logNT;
        if (log1 >= -741.0) r.e = 12.0;          /* the host's exp may land on either side of the 100 eps x DBL_MIN threshold */
        return r;
    }
    double term = exp(log1);
    if (log1 < -708.0) r.e = 12.0;               /* a subnormal first term: its bits are the libm's rounding onto the subnormal grid */
    double tail = term;
    for (int i = k + 1; i <= n; ++i) {
        const double binTerm = (double)(n - i + 1) / (double)i, mult = binTerm * pTerm;
        term *= mult;
        tail += term;
        if (binTerm < 1) {
            const double err = term * ((1 - pow(mult, (double)(n - i + 1))) / (1 - mult) - 1);
            const double rhs = 0.1 * fabs(-log10(tail) - T.logNT) * tail;
            if (fabs(err - rhs) <= 1e-9 * (fabs(err) + fabs(rhs))) r.unc = 1;       /* the host may leave the loop elsewhere */
            if (err < rhs) break;
        }
    }
    r.v = -log10(tail) - T.logNT;
    /* exp and log10 within an ulp of the host's, then identical IEEE operations: relative 1e-9 covers the accumulation of a
     * loop of 10^6 terms with three orders of magnitude to spare */
    const double e = 1e-9 * fmax(1.0, fabs(r.v));
    if (e > r.e) r.e = e;
    return r;
}

/* `a > b` as the host will decide it; sets unc when the margin does not cover the two bounds */
__device__ __forceinline__ bool certain_greater(const NfaVal& a, const NfaVal& b, int& unc)
{
    if (a.n == b.n && a.k == b.k && a.pj == b.pj) return false;       /* the host computes one value for both */
    if (a.e == 0 && b.e == 0) return a.v > b.v;                         /* bit-identical on both sides */
    if (fabs(a.v - b.v) <= a.e + b.e) unc = 1;
    return a.v > b.v;
}

__device__ __forceinline__ bool certain_positive(const NfaVal& a, int& unc)
{
    if (a.e != 0 && fabs(a.v) <= a.e) unc = 1;
    return a.v > 0;
}

struct ImproveShared {
    RectCand cand[NFA_MAX_CAND];
    int pj[NFA_MAX_CAND];
    int total[64], alg[64];
    NfaVal val[NFA_MAX_CAND];
};

} // namespace

/* One wavefront per rectangle; blockIdx.y = frame of the launch, blockIdx.x strides over the frame's rectangles. */
__global__ __launch_bounds__(64) void k_rect_improve(const LsdGrowFrame* __restrict__ frames, int W, int H, int rectMode, LsdNfaTables T,
                                                     int rectCap, LsdSegOut* __restrict__ segs)
{
    __shared__ ImproveShared S;
    const LsdGrowFrame F = frames[blockIdx.y];
    const int lane = threadIdx.x, g = lane / NFA_GROUP, gl = lane - g * NFA_GROUP;
    if (F.out[1] != 0) return;                      /* the growth already hands this frame to the host */
    const int count = min(F.out[0], rectCap);
    LsdSegOut* out = segs + (size_t)blockIdx.y * rectCap;
    const double delta = 0.5, d2 = delta / 2.0;
    for (int id = blockIdx.x; id < count; id += gridDim.x) {
        LsdRect rec = F.rects[id];
        int recPj = 0, unc = 0;
        NfaVal best;
        best.v = 0; best.e = 0; best.n = best.k = best.pj = -1; best.unc = 0;
        bool done = false;
        for (int stage = 0; stage < 6 && !done; stage++) {
            /* the candidates of this stage: rect_improve's cumulative modifications of a copy of the current rectangle */
            int nc = 0;
            if (lane == 0) {
                LsdRect r = rec;
                int pj = recPj;
                if (stage == 0) { S.cand[0] = RectCand{r.x1, r.y1, r.x2, r.y2, r.width, r.dx, r.dy, r.theta, r.prec}; S.pj[0] = pj; nc = 1; }
                else
                    for (int n = 0; n < 5; ++n) {
                        if (stage == 1) { r.p /= 2; r.prec = r.p * 3.14159265358979323846; ++pj; }
                        else {
                            if (!((r.width - delta) >= 0.5)) continue;        /* guards the last precision stage too */
                            if (stage == 5) { r.p /= 2; r.prec = r.p * 3.14159265358979323846; ++pj; }
                            else if (stage == 2) r.width -= delta;
                            else if (stage == 3) { r.x1 += -r.dy * d2; r.y1 += r.dx * d2; r.x2 += -r.dy * d2; r.y2 += r.dx * d2; r.width -= delta; }
                            else { r.x1 -= -r.dy * d2; r.y1 -= r.dx * d2; r.x2 -= -r.dy * d2; r.y2 -= r.dx * d2; r.width -= delta; }
                        }
                        S.cand[nc] = RectCand{r.x1, r.y1, r.x2, r.y2, r.width, r.dx, r.dy, r.theta, r.prec};
                        S.pj[nc] = pj;
                        ++nc;
                    }
            }
            nc = __shfl(nc, 0);
            __syncthreads();
            if (nc == 0) continue;
            /* pixel loops: candidate g on lanes [12 g, 12 g + 12) */
            int total = 0, alg = 0;
            if (g < nc) {
                const RectCand rc = S.cand[g];
                const RectWalk w = rect_walk_setup(rc, rectMode);
                const double kNotDef = -1024.0, kTwoPi = 2.0 * 3.14159265358979323846, kThreeHalfPi = 3.0 * 3.14159265358979323846 / 2.0;
                double lstep = w.fl, rstep = w.fr, lx = (double)w.loX, rx = (double)w.loX;
                const int yBeg = max(w.loY, 0), yEnd = min(w.hiY, H - 1);
                for (int y = yBeg; y <= yEnd; ++y) {
                    const int xs = max((int)lx, 0), xe = min((int)rx, W - 1);
                    const double* row = F.ang + (size_t)y * W;
                    for (int x = xs + gl; x <= xe; x += NFA_GROUP) {
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
            }
            S.total[lane] = total; S.alg[lane] = alg;
            __syncthreads();
            if (lane < nc) {
                int t = 0, a = 0;
                for (int q = 0; q < NFA_GROUP; q++) { t += S.total[lane * NFA_GROUP + q]; a += S.alg[lane * NFA_GROUP + q]; }
                S.val[lane] = nfa_device(t, a, S.pj[lane], T);
            }
            __syncthreads();
            /* the reference's order: a later candidate replaces an earlier one only when strictly greater */
            for (int c = 0; c < nc; c++) {
                const NfaVal v = S.val[c];
                unc |= v.unc;
                bool take;
                if (stage == 0) take = true;
                else take = certain_greater(v, best, unc);
                if (take) {
                    best = v;
                    if (stage != 0) {
                        const RectCand rc = S.cand[c];
                        rec.x1 = rc.x1; rec.y1 = rc.y1; rec.x2 = rc.x2; rec.y2 = rc.y2; rec.width = rc.width; rec.prec = rc.prec;
                        recPj = S.pj[c]; rec.p = T.p[recPj];
                    }
                }
            }
            if (stage < 5 && certain_positive(best, unc)) done = true;
            __syncthreads();
        }
        int fin = 0;
        const bool keep = certain_positive(best, fin);
        unc |= fin;
        if (lane == 0) {
            LsdSegOut o;
            /* LineSegmentDetectorImpl::detect: back to the input image's scale */
            double x1 = rec.x1 + 0.5, y1 = rec.y1 + 0.5, x2 = rec.x2 + 0.5, y2 = rec.y2 + 0.5;
            x1 /= 0.8; y1 /= 0.8; x2 /= 0.8; y2 /= 0.8;
            o.x1 = (float)x1; o.y1 = (float)y1; o.x2 = (float)x2; o.y2 = (float)y2;
            o.flag = keep ? 1 : 0;
            out[id] = o;
            if (unc) atomicOr(&F.out[2], 1);
        }
    }
}

hipError_t drfe_launch_rect_improve(const LsdGrowFrame* d_frames, int nframes, int W, int H, int rectMode, const LsdNfaTables& tab,
                                    int rectCap, LsdSegOut* d_segs, hipStream_t s)
{
    if (nframes <= 0) return hipSuccess;
    /* 192 wavefronts per frame stride over its rectangles (a 640 x 480 frame has ~1500; each takes tens of microseconds) */
    hipLaunchKernelGGL(k_rect_improve, dim3(192, nframes), dim3(64), 0, s, d_frames, W, H, rectMode, tab, rectCap, d_segs);
    return hipGetLastError();
}
