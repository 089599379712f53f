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
 * One LANE per candidate.  A rectangle of a 640 x 480 frame is tested ~17 times, on 24 pixels and 6 tail-loop iterations a time:
 * far too little for a wavefront (the first version of this kernel gave every rectangle one, a scan line's pixels across 12
 * lanes per candidate, and spent 2800 cycles per call on set-up, barriers and its one busy lane: 15 ms per 512 frames, a
 * tenth of the full front-end's SIMD time).  The up to five candidates of a stage depend on the stage's starting rectangle
 * only, so a wavefront takes TWELVE rectangles: lane 5 r + c builds candidate c of rectangle r (the reference's cumulative
 * modifications, replayed), walks its scan lines, counts, and evaluates nfa(); lane 5 r then makes rectangle r's choice among
 * them in the reference's order.  The twelve advance stage by stage, each at its own stage. */
#include <hip/hip_runtime.h>

#include "lines_internal.h"
#include "lsd_rect_walk.h"
#include "cr_sincos.h"

namespace {

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
    /* the first term as the configured reading has it (lines_internal.h, drfe_lsd_configure_rect): the library's `double(n) + 1`
     * (default) or the paper's log_gamma(n + 1).  Either way log1 is the same chain of IEEE operations on the same table values as
     * the host's, and everything below depends on log1 alone: the thresholds on it are properties of exp(), not of how it was summed */
    const double first = T.lgammaFirst ? T.lgamma[n + 1] : (double)n + 1.0;
    const double log1 = first - T.lgamma[k + 1] - T.lgamma[n - k + 1] + (double)k * T.logP[pj] + (double)(n - k) * T.log1mP[pj];
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
    int iters = 0;
    for (int i = k + 1; i <= n; ++i) {
        const double binTerm = (double)(n - i + 1) / (double)i, mult = binTerm * pTerm;
        term *= mult;
        tail += term;
        ++iters;
        if (binTerm < 1) {
            const double err = term * ((1 - pow(mult, (double)(n - i + 1))) / (1 - mult) - 1);
            const double rhs = 0.1 * fabs(-log10(tail) - T.logNT) * tail;
            if (fabs(err - rhs) <= 1e-9 * (fabs(err) + fabs(rhs))) r.unc = 1;       /* reason 1 */
                  /* the host may leave the loop elsewhere */
            if (err < rhs) break;
        }
    }
    const double lt = log10(tail);
    r.v = -lt - T.logNT;
    /* How far the host's value can lie from this one.  log1 is bit-identical on both sides (host tables, IEEE operations); the
     * two exp results are within an ulp of the true value each; every loop iteration multiplies and adds with identical
     * factors, so the relative difference of the two tails grows by at most an ulp per iteration: (iters + 2) ulps.  Through
     * log10 that is an absolute (iters + 2) ulp(1) / ln 10, plus an ulp of each side's log10 result and of the final subtraction.
     * Sixty-four times that sum: ~1e-12 for a typical candidate.  (The first version used 1e-9 max(1, |v|) and sent a sixth of
     * the soak's frames to the host: candidates with next to no aligned pixels all sit within 1e-9 of -log NT.) */
    const double e = 64.0 * 0x1p-52 * ((double)iters + 8.0 + fabs(lt) + fabs(r.v));
    if (e > r.e) r.e = e;
    return r;
}

/* `a > b` as the host will decide it; sets unc when the margin does not cover the two bounds */
__device__ __forceinline__ bool certain_greater(const NfaVal& a, const NfaVal& b, int& unc)
{
    if (a.n == b.n && a.k == b.k && a.pj == b.pj) return false;       /* the host computes one value for both */
    if (a.e == 0 && b.e == 0) return a.v > b.v;                         /* bit-identical on both sides */
    if (fabs(a.v - b.v) <= a.e + b.e) unc |= (a.e >= 1.0 || b.e >= 1.0) ? 8 : 2;      /* reasons 2 (close values) / 8 (a subnormal-regime value involved) */
    return a.v > b.v;
}

__device__ __forceinline__ bool certain_positive(const NfaVal& a, int& unc)
{
    if (a.e != 0 && fabs(a.v) <= a.e) unc |= 4;                                         /* reason 4 */
    return a.v > 0;
}

/* One hypothesis about a rectangle's state inside rect_improve: the fields the refinement stages modify, the index of its
 * precision, the best value so far and the stage it continues with */
struct Hyp {
    double x1, y1, x2, y2, width, prec;
    NfaVal best;
    int pj, stage;
};
#ifndef NFA_MAXQ
#define NFA_MAXQ 7                         /* hypotheses per rectangle (queued + explored).  Measured on 256 frames of the four scene kinds: frames
                                            * handed to the host 4 with 12, 8 or 7 entries, 6 with 5 (and 8 with NFA_MAXLIVE 4 whatever this is); with 7 a
                                            * wavefront's twelve slots hold 22.4 KB of LDS instead of 26.7: seven per CU */
#endif
#ifndef NFA_MAXLIVE
#define NFA_MAXLIVE 6                      /* hypotheses alive inside one stage's selection */
#endif

#define NFA_RECTS 12                       /* rectangles per wavefront: 12 x 5 candidate lanes */
struct ImproveShared {                     /* one rectangle's slot */
    RectCand cand[NFA_MAX_CAND];
    int pj[NFA_MAX_CAND], ok[NFA_MAX_CAND];
    NfaVal val[NFA_MAX_CAND];
    Hyp q[NFA_MAXQ];                       /* q[qi] = the hypothesis being explored; entries behind it wait */
    Hyp live[NFA_MAXLIVE];
    double dx, dy, theta;
    int nq, qi, stage, active, id, finished, nOut, outKeep, flag, why;
    float outSeg[4];
};

/* the segment LineSegmentDetectorImpl::detect emits for a kept rectangle (input-image scale), or "rejected": every explored
 * hypothesis must end in the same one */
__device__ void improve_outcome(ImproveShared& S, bool keep, const Hyp& h)
{
    double x1 = h.x1 + 0.5, y1 = h.y1 + 0.5, x2 = h.x2 + 0.5, y2 = h.y2 + 0.5;
    x1 /= 0.8; y1 /= 0.8; x2 /= 0.8; y2 /= 0.8;
    const float f[4] = {(float)x1, (float)y1, (float)x2, (float)y2};
    if (S.nOut == 0) { S.outKeep = keep ? 1 : 0; for (int k = 0; k < 4; k++) S.outSeg[k] = f[k]; }
    else {
        bool same = S.outKeep == (keep ? 1 : 0);
        if (same && keep) for (int k = 0; k < 4; k++) same = same && __float_as_uint(S.outSeg[k]) == __float_as_uint(f[k]);
        if (!same) { S.flag = 1; S.why |= 64; }
    }
    S.nOut++;
}

/* The choice among a stage's candidates and the stage's exit test for the hypothesis q[qi], by one lane.  A comparison the
 * bounds do not settle SPLITS the hypothesis: both answers are followed (a rejected rectangle is rejected whichever of two
 * all but equal hopeless candidates "won" - the common case on low-texture frames, where half of the frames had such a pair;
 * a kept one may or may not depend on it), and only outcomes that differ flag the frame. */
__device__ void improve_select(ImproveShared& S, int qi, int stage, int nc, const LsdNfaTables& T)
{
    int nl = 1;
    S.live[0] = S.q[qi];
    for (int c = 0; c < nc; c++) {
        const NfaVal v = S.val[c];
        if (v.unc) { S.flag = 1; S.why |= 1; }
        const int nl0 = nl;
        for (int l = 0; l < nl0; l++) {
            bool take = true;
            if (stage != 0) {
                int u = 0;
                take = certain_greater(v, S.live[l].best, u);
                if (u) {
                    /* the other answer becomes a hypothesis of its own - unless one in that state exists already (a hypothesis
                     * that takes candidate c IS candidate c: at most nc + 1 distinct states per stage) */
                    bool dup = false;
                    if (!take)
                        for (int m = 0; m < nl; m++) dup = dup || (S.live[m].best.n == v.n && S.live[m].best.k == v.k && S.live[m].best.pj == v.pj && S.live[m].pj == S.pj[c] &&
                                                                   S.live[m].width == S.cand[c].width && S.live[m].x1 == S.cand[c].x1 && S.live[m].y1 == S.cand[c].y1);
                    if (dup) { /* nothing to add */ }
                    else if (nl < NFA_MAXLIVE && !(u & 8)) {
                        S.live[nl] = S.live[l];
                        if (!take) {
                            Hyp& o = S.live[nl];
                            const RectCand rc = S.cand[c];
                            o.x1 = rc.x1; o.y1 = rc.y1; o.x2 = rc.x2; o.y2 = rc.y2; o.width = rc.width; o.prec = rc.prec; o.pj = S.pj[c]; o.best = v;
                        }
                        nl++;
                    } else { S.flag = 1; S.why |= (u & 8) ? 8 : 128; }
                }
            }
            if (take) {
                Hyp& h = S.live[l];
                h.best = v;
                if (stage != 0) {
                    const RectCand rc = S.cand[c];
                    h.x1 = rc.x1; h.y1 = rc.y1; h.x2 = rc.x2; h.y2 = rc.y2; h.width = rc.width; h.prec = rc.prec; h.pj = S.pj[c];
                }
            }
        }
    }
    /* exit test of the stage: log_nfa > LOG_EPS ends rect_improve (kept); after the last stage it is the segment's verdict */
    bool haveNext = false;
    for (int l = 0; l < nl; l++) {
        int u = 0;
        const bool pos = certain_positive(S.live[l].best, u);
        if (stage == 5) {
            improve_outcome(S, pos, S.live[l]);
            if (u) improve_outcome(S, !pos, S.live[l]);           /* differs by construction: flags */
            continue;
        }
        if (pos || u) improve_outcome(S, true, S.live[l]);
        if (!pos || u) {
            Hyp h = S.live[l];
            h.stage = stage + 1;
            if (!haveNext) { S.q[qi] = h; haveNext = true; }
            else if (S.nq < NFA_MAXQ) S.q[S.nq++] = h;
            else { S.flag = 1; S.why |= 16; }
        }
    }
    S.finished = haveNext ? 0 : 1;
}

} // namespace

/* (pixels, aligned pixels) of one rectangle by ONE lane: rect_nfa's pixel loop as rect_walk_count walks it, without the lanes */
__device__ __forceinline__ int2 rect_walk_count_lane(const RectCand& rc, const RectWalk& w, const double* __restrict__ ang, int W, int H)
{
    const double kNotDef = -1024.0, kTwoPi = 2.0 * 3.14159265358979323846, kThreeHalfPi = 3.0 * 3.14159265358979323846 / 2.0;
    double lstep = w.fl, rstep = w.fr, lx = (double)w.loX, rx = (double)w.loX;
    int total = 0, alg = 0;
    const int yBeg = max(w.loY, 0), yEnd = min(w.hiY, H - 1);
    /* A lane's walk is otherwise one L2 round trip per pixel: the first four pixels of TWO scan lines are fetched together (the
     * loads depend on the edge walk only, not on the counts), counted without branches; what a scan line holds beyond four
     * pixels follows four at a time. */
    auto count4 = [&](const double* row, int x, int xe) {
        double a[4];
#pragma unroll
        for (int u = 0; u < 4; u++) a[u] = x + u <= xe ? row[x + u] : kNotDef;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            double d = rc.theta - a[u];
            if (d < 0) d = -d;
            if (d > kThreeHalfPi) { d -= kTwoPi; if (d < 0) d = -d; }
            total += x + u <= xe ? 1 : 0;
            alg += (a[u] != kNotDef && d <= rc.prec) ? 1 : 0;
        }
    };
    for (int y = yBeg; y <= yEnd; y += 2) {
        const int xs0 = max((int)lx, 0), xe0 = min((int)rx, W - 1);
        if (y >= w.leftY) lstep = w.sl;
        if (y >= w.rightY) rstep = w.sr;
        lx += lstep;
        rx += rstep;
        const bool two = y + 1 <= yEnd;
        int xs1 = 0, xe1 = -1;
        if (two) {
            xs1 = max((int)lx, 0); xe1 = min((int)rx, W - 1);
            if (y + 1 >= w.leftY) lstep = w.sl;
            if (y + 1 >= w.rightY) rstep = w.sr;
            lx += lstep;
            rx += rstep;
        }
        const double* row0 = ang + (size_t)y * W;
        const double* row1 = ang + (size_t)(two ? y + 1 : y) * W;
        double a0[4], a1[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { a0[u] = xs0 + u <= xe0 ? row0[xs0 + u] : kNotDef; a1[u] = xs1 + u <= xe1 ? row1[xs1 + u] : kNotDef; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            double d = rc.theta - a0[u];
            if (d < 0) d = -d;
            if (d > kThreeHalfPi) { d -= kTwoPi; if (d < 0) d = -d; }
            total += xs0 + u <= xe0 ? 1 : 0;
            alg += (a0[u] != kNotDef && d <= rc.prec) ? 1 : 0;
            double e = rc.theta - a1[u];
            if (e < 0) e = -e;
            if (e > kThreeHalfPi) { e -= kTwoPi; if (e < 0) e = -e; }
            total += xs1 + u <= xe1 ? 1 : 0;
            alg += (a1[u] != kNotDef && e <= rc.prec) ? 1 : 0;
        }
        for (int x = xs0 + 4; x <= xe0; x += 4) count4(row0, x, xe0);
        for (int x = xs1 + 4; x <= xe1; x += 4) count4(row1, x, xe1);
    }
    return make_int2(total, alg);
}

/* blockIdx.y = frame of the launch; the frame's gridDim.x wavefronts take its rectangles one by one, NFA_RECTS at a time each */
__global__ __launch_bounds__(64) void k_rect_improve(const LsdGrowFrame* __restrict__ frames, int W, int H, int rectMode, LsdNfaTables T,
                                                     int rectCap, LsdSegOut* __restrict__ segs)
{
    __shared__ ImproveShared SS[NFA_RECTS];
    const LsdGrowFrame F = frames[blockIdx.y];
    const int lane = threadIdx.x, r = lane / NFA_MAX_CAND, c = lane - r * NFA_MAX_CAND;
    const bool candLane = lane < NFA_RECTS * NFA_MAX_CAND, selLane = candLane && c == 0;
    if (F.out[1] != 0) return;                      /* the growth already hands this frame to the host */
    const int count = min(F.out[0], rectCap);
    LsdSegOut* out = segs + (size_t)blockIdx.y * rectCap;
    const double delta = 0.5, d2 = delta / 2.0;
#ifdef NFA_PROFILE
    unsigned long long tp[5] = {0, 0, 0, 0, 0};
#define NTP(k, t0) tp[k] += wall_clock64() - (t0)
#else
#define NTP(k, t0) (void)(t0)
#endif
    /* A slot takes the frame's next rectangle as soon as its own is decided (a counter in the frame's out words, zeroed by
     * k_lsd_grow): rectangles need one to a dozen stage rounds, and twelve marching as a group waited for their slowest. */
    auto next_rect = [&](ImproveShared& S) {                   /* the slot's selection lane */
        const int id = atomicAdd(&F.out[DRFE_LSD_OUT_NEXT_RECT], 1);
        S.id = id; S.active = id < count ? 1 : 0;
        if (S.active) {
            const LsdRect rec0 = F.rects[id];
            Hyp h;
            h.x1 = rec0.x1; h.y1 = rec0.y1; h.x2 = rec0.x2; h.y2 = rec0.y2; h.width = rec0.width; h.prec = rec0.prec;
            h.best.v = 0; h.best.e = 0; h.best.n = h.best.k = h.best.pj = -1; h.best.unc = 0;
            h.pj = 0; h.stage = 0;
            S.q[0] = h; S.nq = 1; S.qi = 0; S.stage = 0; S.nOut = 0; S.outKeep = 0; S.flag = 0; S.why = 0;
            S.dx = rec0.dx; S.dy = rec0.dy; S.theta = rec0.theta;
            for (int k = 0; k < 4; k++) S.outSeg[k] = 0.f;
        }
    };
    {
        if (selLane) next_rect(SS[r]);
        __syncthreads();
        while (__ballot(candLane && SS[candLane ? r : 0].active != 0)) {
            /* candidate c of this stage of rectangle r: rect_improve's cumulative modifications of a copy of the hypothesis'
             * rectangle, each behind its guard (a guard that fails once fails for the rest of the stage) */
            const unsigned long long t0 = wall_clock64();
            if (candLane && SS[r].active) {
                ImproveShared& S = SS[r];
                const Hyp h = S.q[S.qi];
                const int stage = S.stage;
                double x1 = h.x1, y1 = h.y1, x2 = h.x2, y2 = h.y2, width = h.width, prec = h.prec, p = T.p[h.pj];
                int pj = h.pj;
                bool ok = stage == 0 ? c == 0 : true;
                if (stage != 0)
                    for (int m = 0; m <= c && ok; ++m) {
                        if (stage == 1) { p /= 2; prec = p * 3.14159265358979323846; ++pj; }
                        else {
                            if (!((width - delta) >= 0.5)) { ok = false; break; }        /* guards the last precision stage too */
                            if (stage == 5) { p /= 2; prec = p * 3.14159265358979323846; ++pj; }
                            else if (stage == 2) width -= delta;
                            else if (stage == 3) { x1 += -S.dy * d2; y1 += S.dx * d2; x2 += -S.dy * d2; y2 += S.dx * d2; width -= delta; }
                            else { x1 -= -S.dy * d2; y1 -= S.dx * d2; x2 -= -S.dy * d2; y2 -= S.dx * d2; width -= delta; }
                        }
                    }
                S.ok[c] = ok ? 1 : 0;
                if (ok) {
                    const RectCand rc = RectCand{x1, y1, x2, y2, width, S.dx, S.dy, S.theta, prec};
                    const RectWalk w = rect_walk_setup(rc, rectMode);
                    NTP(0, t0);
                    const unsigned long long t1 = wall_clock64();
                    const int2 cnt = rect_walk_count_lane(rc, w, F.ang, W, H);
                    NTP(1, t1);
                    const unsigned long long t2 = wall_clock64();
                    S.cand[c] = rc; S.pj[c] = pj;
                    S.val[c] = nfa_device(cnt.x, cnt.y, pj, T);
                    NTP(2, t2);
                }
            }
            __syncthreads();
            const unsigned long long t3 = wall_clock64();
            /* rectangle r's choice among them and the stage's exit test; then its next stage, its next hypothesis, or its result */
            if (selLane && SS[r].active) {
                ImproveShared& S = SS[r];
                int nc = 0;
                while (nc < NFA_MAX_CAND && S.ok[nc]) nc++;
                improve_select(S, S.qi, S.stage, nc, T);
                if (!S.finished) S.stage = S.stage + 1;
                else if (S.qi + 1 < S.nq) { S.qi = S.qi + 1; S.stage = S.q[S.qi].stage; }
                else {
                    LsdSegOut o;
                    o.x1 = S.outSeg[0]; o.y1 = S.outSeg[1]; o.x2 = S.outSeg[2]; o.y2 = S.outSeg[3];
                    o.flag = S.outKeep;
                    out[S.id] = o;
                    if (S.flag || S.nOut == 0) { atomicOr(&F.out[2], 1); atomicOr(&F.out[3], S.why ? S.why : 32); }      /* out[3]: why (DRFE_TRACE_LINES) */
                    next_rect(S);
                }
            }
            __syncthreads();
            NTP(3, t3);
            NTP(4, t0);
        }
        __syncthreads();
    }
#ifdef NFA_PROFILE
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) for (int k = 0; k < 5; k++) ((unsigned long long*)(F.out + 24))[k] = tp[k];
#endif
}

hipError_t drfe_launch_rect_improve(const LsdGrowFrame* d_frames, int nframes, int W, int H, int rectMode, const LsdNfaTables& tab,
                                    int rectCap, LsdSegOut* d_segs, hipStream_t s)
{
    if (nframes <= 0) return hipSuccess;
    /* wavefronts per frame, twelve rectangles in flight each (a 640 x 480 frame has 1500-2500 rectangles) */
    static const int perFrame = [] { const char* e = std::getenv("DRFE_NFA_BLOCKS"); const int v = e ? std::atoi(e) : 128; return v < 1 ? 1 : v > 1024 ? 1024 : v; }();
    hipLaunchKernelGGL(k_rect_improve, dim3(perFrame, nframes), dim3(64), 0, s, d_frames, W, H, rectMode, tab, rectCap, d_segs);
    return hipGetLastError();
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * From the validated segments to the caller's key lines on the device: LSDDetector::detect's KeyLine fields (octave 0), the
 * reference's response cut (src/LSDextractor.cpp:18-28: std::sort by response, keep lsdNFeatures, renumber class_id), the
 * LBD sampling parameters k_lbd reads and the line equations (:32-42).  One wavefront per frame.
 *
 * std::sort is not stable and the comparator sees the response only, so the survivors of a tie at the cut - and their order -
 * are whatever libstdc++'s introsort leaves: its element moves are executed here by one lane on (response, index) records in
 * LDS (median of three to the front, unguarded Hoare partition, recursion on the right part, final insertion sort; a range
 * that exhausts the depth limit - heap sort in libstdc++ - sends the frame to the host).
 * atan2 (KeyLine::angle) and cos / sin (the LBD direction) are the host libm's in the reference, rounded to float: the device
 * certifies that rounding (the double moved by 2^-45 either way gives the same float; cr_sincos.h for cos / sin) and flags
 * the frame for the host otherwise. */
namespace {

#define KL_MAX_KEPT 6144           /* kept segments per frame the LDS sort holds (a 640 x 480 frame has ~250, a 1280 x 960 one more than 2000) */

struct KlShared {
    float resp[KL_MAX_KEPT];
    uint16_t idx[KL_MAX_KEPT];      /* position in the frame's rectangle list */
    int stackLo[64], stackHi[64], stackDepth[64];
};

/* comp(a, b) of the reference's lambda: a.response > b.response */
__device__ __forceinline__ bool kl_before(float a, float b) { return a > b; }

__device__ __forceinline__ void kl_swap(KlShared& S, int a, int b)
{
    const float r = S.resp[a]; S.resp[a] = S.resp[b]; S.resp[b] = r;
    const uint16_t i = S.idx[a]; S.idx[a] = S.idx[b]; S.idx[b] = i;
}

/* std::sort(first, last, comp) of libstdc++ (bits/stl_algo.h: __sort -> __introsort_loop + __final_insertion_sort) on
 * [0, n); returns false when a range runs out of its depth limit (the __partial_sort heap branch is not restated) */
__device__ bool kl_std_sort(KlShared& S, int n)
{
    if (n < 2) return true;
    int lg = 0;
    for (int v = n; v > 1; v >>= 1) lg++;
    int sp = 0;
    S.stackLo[0] = 0; S.stackHi[0] = n; S.stackDepth[0] = 2 * lg; sp = 1;
    /* __introsort_loop(first, last, depth): while (last - first > 16) { cut = partition; __introsort_loop(cut, last, depth); last = cut; }
     * - the recursion on the right part runs to completion before the left part is touched: a stack of pending LEFT parts
     * would reverse that, so the loop below keeps (first, last, depth) of the current call and pushes the left remainder. */
    while (sp > 0) {
        --sp;
        int first = S.stackLo[sp], last = S.stackHi[sp], depth = S.stackDepth[sp];
        while (last - first > 16) {
            if (depth == 0) return false;
            --depth;
            /* __unguarded_partition_pivot: __move_median_to_first(first, first + 1, mid, last - 1), then partition (first + 1, last) around *first */
            const int mid = first + (last - first) / 2, a = first + 1, b = mid, c = last - 1;
            const float ra = S.resp[a], rb = S.resp[b], rc = S.resp[c];
            if (kl_before(ra, rb)) {
                if (kl_before(rb, rc)) kl_swap(S, first, b);
                else if (kl_before(ra, rc)) kl_swap(S, first, c);
                else kl_swap(S, first, a);
            } else if (kl_before(ra, rc)) kl_swap(S, first, a);
            else if (kl_before(rb, rc)) kl_swap(S, first, c);
            else kl_swap(S, first, b);
            const float pivot = S.resp[first];
            int lo = first + 1, hi = last;
            for (;;) {
                while (kl_before(S.resp[lo], pivot)) ++lo;
                --hi;
                while (kl_before(pivot, S.resp[hi])) --hi;
                if (!(lo < hi)) break;
                kl_swap(S, lo, hi);
                ++lo;
            }
            const int cut = lo;
            /* recursion on [cut, last) first: continue with it, remember [first, cut) for afterwards */
            if (sp >= 64) return false;
            S.stackLo[sp] = first; S.stackHi[sp] = cut; S.stackDepth[sp] = depth; ++sp;
            first = cut;
        }
    }
    /* __final_insertion_sort: __insertion_sort on the first 16, __unguarded_insertion_sort on the rest */
    const int head = n > 16 ? 16 : n;
    for (int i = 1; i < head; ++i) {
        const float v = S.resp[i]; const uint16_t vi = S.idx[i];
        if (kl_before(v, S.resp[0])) {
            for (int j = i; j > 0; --j) { S.resp[j] = S.resp[j - 1]; S.idx[j] = S.idx[j - 1]; }      /* move_backward(first, i, i + 1) */
            S.resp[0] = v; S.idx[0] = vi;
        } else {
            int j = i;
            while (kl_before(v, S.resp[j - 1])) { S.resp[j] = S.resp[j - 1]; S.idx[j] = S.idx[j - 1]; --j; }
            S.resp[j] = v; S.idx[j] = vi;
        }
    }
    for (int i = head; i < n; ++i) {
        const float v = S.resp[i]; const uint16_t vi = S.idx[i];
        int j = i;
        while (kl_before(v, S.resp[j - 1])) { S.resp[j] = S.resp[j - 1]; S.idx[j] = S.idx[j - 1]; --j; }
        S.resp[j] = v; S.idx[j] = vi;
    }
    return true;
}

struct KlEnds { float e0, e1, e2, e3; };

/* checkLineExtremes of LSDDetector::detectImpl */
__device__ __forceinline__ KlEnds kl_clamp(const LsdSegOut& s, int w, int h)
{
    KlEnds e = {s.x1, s.y1, s.x2, s.y2};
    if (e.e0 < 0) e.e0 = 0;
    if (e.e0 >= w) e.e0 = (float)w - 1.0f;
    if (e.e2 < 0) e.e2 = 0;
    if (e.e2 >= w) e.e2 = (float)w - 1.0f;
    if (e.e1 < 0) e.e1 = 0;
    if (e.e1 >= h) e.e1 = (float)h - 1.0f;
    if (e.e3 < 0) e.e3 = 0;
    if (e.e3 >= h) e.e3 = (float)h - 1.0f;
    return e;
}

__device__ __forceinline__ float kl_length(const KlEnds& e)
{
    const double dx = (double)(e.e0 - e.e2), dy = (double)(e.e1 - e.e3);       /* std::pow(float, 2): the exact square in double */
    return (float)sqrt(dx * dx + dy * dy);
}

} // namespace

__global__ __launch_bounds__(64) void k_lsd_keylines(const LsdGrowFrame* __restrict__ frames, const LsdSegOut* __restrict__ segs, int rectCap,
                                                     int w, int h, int maxLines, int klCap, drfe_keyline* __restrict__ klOut,
                                                     double* __restrict__ lineFOut, LbdLine* __restrict__ lbdOut, int* __restrict__ frameOut)
{
    __shared__ KlShared S;
    __shared__ int sortOk;
    const int f = blockIdx.x, lane = threadIdx.x;
    const LsdGrowFrame F = frames[f];
    int* fo = frameOut + 4 * (size_t)f;                 /* nLines, nDetected, status, pad */
    if (F.out[1] != 0 || F.out[2] != 0) { if (lane == 0) { fo[0] = 0; fo[1] = 0; fo[2] = 1; } return; }      /* the host validates this frame */
    const int count = min(F.out[0], rectCap);
    const LsdSegOut* sg = segs + (size_t)f * rectCap;
    /* kept segments in seed order */
    int nKept = 0;
    int status = 0;
    for (int base = 0; base < count; base += 64) {
        const int i = base + lane;
        const bool keep = i < count && sg[i].flag != 0;
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int pos = nKept + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < KL_MAX_KEPT) {
                S.idx[pos] = (uint16_t)i;
                const KlEnds e = kl_clamp(sg[i], w, h);
                S.resp[pos] = kl_length(e) / (float)max(w, h);
            }
        }
        nKept += __popcll(m);
    }
    if (nKept > KL_MAX_KEPT) status = 2;
    __syncthreads();
    const bool cut = nKept > maxLines;
    if (cut && status == 0) {
        if (lane == 0) sortOk = kl_std_sort(S, nKept) ? 1 : 0;
        __syncthreads();
        if (!sortOk) status = 2;
    }
    const int nl = cut ? maxLines : nKept;
    if (nl > klCap) status = 2;
    int unc = 0;
    if (status == 0)
        for (int k = lane; k < nl; k += 64) {
            const int i = S.idx[k];
            const KlEnds e = kl_clamp(sg[i], w, h);
            drfe_keyline kl;
            kl.start_point_x = e.e0; kl.start_point_y = e.e1; kl.end_point_x = e.e2; kl.end_point_y = e.e3;
            kl.s_point_in_octave_x = e.e0; kl.s_point_in_octave_y = e.e1; kl.e_point_in_octave_x = e.e2; kl.e_point_in_octave_y = e.e3;
            kl.line_length = kl_length(e);
            const int x0 = (int)rintf(e.e0), y0 = (int)rintf(e.e1), x1 = (int)rintf(e.e2), y1 = (int)rintf(e.e3);
            kl.num_of_pixels = max(abs(x1 - x0), abs(y1 - y0)) + 1;               /* LineIterator(...).count */
            const double a = atan2((double)(e.e3 - e.e1), (double)(e.e2 - e.e0));
            const float af = (float)a;
            if ((float)(a * (1.0 - 0x1p-45)) != af || (float)(a * (1.0 + 0x1p-45)) != af) unc = 1;
            kl.angle = af;
            kl.class_id = k;          /* the running index before the cut == the position when nothing is cut; the position after it */
            kl.octave = 0;
            kl.size = (e.e2 - e.e0) * (e.e3 - e.e1);
            kl.response = kl.line_length / (float)max(w, h);
            kl.pt_x = (e.e2 + e.e0) / 2; kl.pt_y = (e.e3 + e.e1) / 2;
            klOut[(size_t)f * klCap + k] = kl;
            /* BinaryDescriptor::computeLBD's line: midpoint, direction (cos, sin of the float angle by the libm, rounded to float), length */
            LbdLine L;
            L.midX = (float)(0.5 * (e.e0 + e.e2)); L.midY = (float)(0.5 * (e.e1 + e.e3));
            float sn, cn;
            const double aa = fabs((double)af);
            if (!drfe_cr_sincos_f(aa, &sn, &cn)) unc = 1;
            L.dL0 = cn; L.dL1 = (af < 0 || (af == 0 && signbit(af))) ? -sn : sn;      /* sin is odd, and so is its rounding */
            L.len = kl.num_of_pixels; L.pad = 0;
            lbdOut[(size_t)f * klCap + k] = L;
            /* keylineFunctions: normalised cross product of the homogeneous end points */
            const double sx = e.e0, sy = e.e1, ex = e.e2, ey = e.e3;
            const double l0 = sy * 1.0 - 1.0 * ey, l1 = 1.0 * ex - sx * 1.0, l2 = sx * ey - sy * ex;
            const double nrm = sqrt(l0 * l0 + l1 * l1 + l2 * l2);
            double* lf = lineFOut + ((size_t)f * klCap + k) * 3;
            lf[0] = l0 / nrm; lf[1] = l1 / nrm; lf[2] = l2 / nrm;
        }
    if (__ballot(unc)) status |= 1;
    if (lane == 0) { fo[0] = status ? 0 : nl; fo[1] = nKept; fo[2] = status; fo[3] = 0; }
}

hipError_t drfe_launch_lsd_keylines(const LsdGrowFrame* d_frames, const LsdSegOut* d_segs, int rectCap, int nframes, int w, int h, int maxLines,
                                    int klCap, drfe_keyline* d_kl, double* d_lineF, LbdLine* d_lbd, int* d_frameOut, hipStream_t s)
{
    if (nframes <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_lsd_keylines, dim3(nframes), dim3(64), 0, s, d_frames, d_segs, rectCap, w, h, maxLines, klCap, d_kl, d_lineF, d_lbd, d_frameOut);
    return hipGetLastError();
}
