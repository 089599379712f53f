/* lines_lsd.cpp — LineSegment::ExtractLineSegment behind drfe_lsd_extract (include/drfe.h).
 *
 * reference src/LSDextractor.cpp:12-43 = LSDDetector::detect (octave 0, LSD_REFINE_ADV) -> keep the 40
 * highest-response lines -> BinaryDescriptor::compute (LBD, 256 bit) -> normalised line equation.
 * Split: the image passes run on the device (lines_kernels.hip); this file holds the sequential
 * parts of OpenCV's LineSegmentDetectorImpl (pseudo-ordering, region growing, rectangle fit,
 * refinement, NFA validation), the KeyLine bookkeeping of LSDDetector::detect, the response sort and
 * cut of the reference, and computeLBD's band accumulation + binarisation.  Parity status: unpinned
 * (the OpenCV sources are not in the reference), see DESIGN.md §5.
 */
#include "drfe_internal.h"
#include "lines_internal.h"
#include "../../include/drfe_math.h"
#include "introsort_restated.h"
#include "cr_sincos.h"

#include <algorithm>
#include <cfloat>
#include <functional>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#define HIPCHK(c, call)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (c)->err = std::string(#call) + ": " + hipGetErrorString(e__);                      \
            return DRFE_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

namespace {

const double kNotDef = -1024.0, kTwoPi = 2.0 * M_PI, kThreeHalfPi = 3.0 * M_PI / 2.0, kDeg2Rad = M_PI / 180.0;

struct RPt { int x, y; double angle, modgrad; };
/* one pixel of the pseudo-ordering: gradient bin << 32 | y << 16 | x.  std::sort's permutation depends on the comparator's
 * answers only (it looks at the bin), so sorting these 4-byte keys orders the pixels exactly as sorting OpenCV's 12-byte
 * normPoint records does, with a third less memory traffic */
typedef uint32_t OPt;       /* gradient bin (10 bits) << 22 | y (11 bits) << 11 | x (11 bits): 4-byte keys, the array of a 512 x 384
                               field fits the L2; fields up to 2048 x 2048 (checked by the caller) */
#define LSD_ORDER_IDX_BITS 22
typedef LsdRect RectD;             /* rect of lsd.cpp: what region2rect fills (lines_internal.h) */

/* cv::LineSegmentDetectorImpl's validation half: rect_improve + rect_nfa + nfa over the rectangles region growing accepted.
 * It reads the angle field only through the pixel counts (device: k_rect_counts), so it serves the host finder and the device
 * one (lsd_grow_kernels.hip) alike. */
class RectValidator {
public:
    /* counts(cands, out): (pixels, aligned pixels) of every rectangle - the device kernel k_rect_counts */
    typedef std::function<bool(const std::vector<RectCand>&, std::vector<int2>&)> CountFn;

    RectValidator(int W, int H, int rectMode = 0) : lgammaFirst_(lsd_lgamma_first(rectMode) != 0) { logNT_ = 5 * (std::log10(double(W)) + std::log10(double(H))) / 2 + std::log10(11.0); }
    void setRectMode(int rectMode) { lgammaFirst_ = lsd_lgamma_first(rectMode) != 0; }
    double logNT() const { return logNT_; }
    size_t minReg(double p) const { return size_t(-logNT_ / std::log10(p)); }

    /* rect_improve for all rectangles, then the segments whose NFA passes, as LineSegmentDetectorImpl::detect emits them */
    bool emit(std::vector<RectD>& pending, std::vector<float>& lines, const CountFn& counts) const
    {
        const double scale = 0.8, logEps = 0;
        std::vector<double> logNfa;
        if (!improveAll(pending, logNfa, counts)) return false;
        for (size_t i = 0; i < pending.size(); i++) {
            if (logNfa[i] <= logEps) continue;
            RectD rec = pending[i];
            rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
            rec.x1 /= scale; rec.y1 /= scale; rec.x2 /= scale; rec.y2 /= scale;
            lines.push_back(float(rec.x1)); lines.push_back(float(rec.y1));
            lines.push_back(float(rec.x2)); lines.push_back(float(rec.y2));
        }
        return true;
    }
    /* the constants nfa() takes from the libm, for the device's rect_improve (lsd_nfa_kernels.hip): log10 / log of the up to
     * eleven precisions p0 / 2^j, and log_gamma at every integer a W x H field can ask for - the host's values, so the
     * device's NFA differs from this class's only where exp / log10 / pow enter */
    void fillTables(double p0, int W, int H, LsdNfaTables& t, std::vector<double>* lg) const
    {
        t.logNT = logNT_;
        double p = p0;
        for (int j = 0; j < 11; j++) {
            t.p[j] = p; t.logP[j] = std::log(p); t.log1mP[j] = std::log(1.0 - p); t.log10P[j] = std::log10(p);
            p /= 2;
        }
        t.lgamma = nullptr; t.lgammaN = W * H + 2;
        t.lgammaFirst = lgammaFirst_ ? 1 : 0;
        if (lg) {                                  /* 13 ms of libm calls at 512 x 384: once per arena, not per call */
            lg->resize((size_t)t.lgammaN);
            for (size_t i = 0; i < lg->size(); i++) (*lg)[i] = logGammaInt((int)i);
        }
    }
private:
    double logNT_;
    bool lgammaFirst_;        /* nfa()'s first term: false = the library's `double(n) + 1`, true = log_gamma(n + 1) */
    static double logGamma(double x)
    {
        if (x > 15.0)
            return 0.918938533204673 + (x - 0.5) * std::log(x) - x +
                   0.5 * x * std::log(x * std::sinh(1 / x) + 1 / (810.0 * std::pow(x, 6.0)));
        static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
        double a = (x + 0.5) * std::log(x + 5.5) - (x + 5.5), b = 0;
        for (int n = 0; n < 7; ++n) { a -= std::log(x + double(n)); b += q[n] * std::pow(x, double(n)); }
        return a + std::log(b);
    }
    /* logGamma at the integer arguments nfa() asks for, from a table filled once by logGamma itself (identical values; the
     * seven log + seven pow per call were most of the NFA stage's host time) */
    static double logGammaInt(int x)
    {
        static const int kTable = 1 << 16;
        static std::vector<double> table;
        static std::once_flag once;
        std::call_once(once, [] {
            table.resize(kTable);
            for (int i = 0; i < kTable; i++) table[i] = logGamma(double(i));
        });
        return (x >= 0 && x < kTable) ? table[x] : logGamma(double(x));
    }
    static bool nearlyEqual(double a, double b)
    {
        if (a == b) return true;
        const double aa = std::fabs(a), bb = std::fabs(b);
        double m = aa > bb ? aa : bb;
        if (m < DBL_MIN) m = DBL_MIN;
        return (std::fabs(a - b) / m) <= (100.0 * DBL_EPSILON);
    }
    double nfa(int n, int k, double p) const
    {
        if (n == 0 || k == 0) return -logNT_;
        if (n == k) return -logNT_ - double(n) * std::log10(p);
        const double pTerm = p / (1 - p);
        /* OpenCV 3.4 lsd.cpp: `double log1term = (double(n) + 1) - log_gamma(double(k) + 1) - log_gamma(double(n-k) + 1) + ...` -
         * the paper's log_gamma(n + 1) lost its function call in the library (library bugs are preserved) */
        const double first = lgammaFirst_ ? logGammaInt(n + 1) : (double(n) + 1);
        const double log1 = first - logGammaInt(k + 1) - logGammaInt(n - k + 1) +
                            double(k) * std::log(p) + double(n - k) * std::log(1.0 - p);
        double term = std::exp(log1);
        if (nearlyEqual(term, 0)) return (k > n * p) ? -log1 / M_LN10 - logNT_ : -logNT_;
        double tail = term;
        for (int i = k + 1; i <= n; ++i) {
            const double binTerm = double(n - i + 1) / double(i), mult = binTerm * pTerm;
            term *= mult;
            tail += term;
            if (binTerm < 1) {
                const double err = term * ((1 - std::pow(mult, double(n - i + 1))) / (1 - mult) - 1);
                if (err < 0.1 * std::fabs(-std::log10(tail) - logNT_) * tail) break;
            }
        }
        return -std::log10(tail) - logNT_;
    }
    static RectCand cand(const RectD& r) { return RectCand{r.x1, r.y1, r.x2, r.y2, r.width, r.dx, r.dy, r.theta, r.prec}; }

    /* cv::LineSegmentDetectorImpl::rect_improve for every rectangle of the frame, level-synchronous: the five candidates of
     * a refinement stage depend only on the rectangle the stage starts from, so a stage is ONE counting launch over all
     * live rectangles (device) followed by the NFA comparisons in the reference's order (host, the caller's libm).
     * Stages: the rectangle itself; 5 x precision halved; 5 x width reduced; 5 x one side; 5 x the other side; 5 x precision. */
    bool improveAll(std::vector<RectD>& rects, std::vector<double>& best, const CountFn& counts) const
    {
        const double delta = 0.5, d2 = delta / 2.0, logEps = 0;
        const size_t R = rects.size();
        best.assign(R, 0.0);
        std::vector<char> done(R, 0);
        std::vector<RectCand> cands;
        std::vector<RectD> trial;              /* the candidate rectangles of the current stage */
        std::vector<int> owner;                /* rectangle a candidate belongs to */
        std::vector<int2> cnt;
        for (int stage = 0; stage < 6; stage++) {
            cands.clear(); trial.clear(); owner.clear();
            for (size_t i = 0; i < R; i++) {
                if (done[i]) continue;
                RectD r = rects[i];
                if (stage == 0) { trial.push_back(r); owner.push_back((int)i); continue; }
                for (int n = 0; n < 5; ++n) {
                    if (stage == 1) { r.p /= 2; r.prec = r.p * M_PI; }
                    else {
                        if (!((r.width - delta) >= 0.5)) continue;        /* guards the last precision stage too */
                        if (stage == 5) { r.p /= 2; r.prec = r.p * M_PI; }
                        else if (stage == 2) r.width -= delta;
                        else if (stage == 3) { r.x1 += -r.dy * d2; r.y1 += r.dx * d2; r.x2 += -r.dy * d2; r.y2 += r.dx * d2; r.width -= delta; }
                        else { r.x1 -= -r.dy * d2; r.y1 -= r.dx * d2; r.x2 -= -r.dy * d2; r.y2 -= r.dx * d2; r.width -= delta; }
                    }
                    trial.push_back(r); owner.push_back((int)i);
                }
            }
            if (trial.empty()) continue;
            cands.reserve(trial.size());
            for (const RectD& r : trial) cands.push_back(cand(r));
            if (!counts(cands, cnt)) return false;
            for (size_t k = 0; k < trial.size(); k++) {
                const int i = owner[k];
                const double v = nfa(cnt[k].x, cnt[k].y, trial[k].p);
                if (stage == 0) best[i] = v;
                else if (v > best[i]) { best[i] = v; rects[i] = trial[k]; }
            }
            if (stage < 5)
                for (size_t i = 0; i < R; i++)
                    if (!done[i] && best[i] > logEps) done[i] = 1;
        }
        return true;
    }
};

/* sequential half of cv::LineSegmentDetectorImpl, fed with the device-computed gradient fields */
class SegmentFinder {
public:
    /* used / order: caller-owned buffers that survive between frames (a lane reuses them: no multi-megabyte
     * allocation, hence no mmap/page-fault traffic, per frame) */
    SegmentFinder(int W, int H, const double* modgrad, const double* angles, const float* cs, double maxGrad,
                  std::vector<uint8_t>& used, std::vector<OPt>& order, std::vector<OPt>& orderTmp)
        : val_(W, H), W_(W), H_(H), mod_(modgrad), ang_(angles), cs_(cs), used_(used), order_(order), orderTmp_(orderTmp)
    {
        /* 0 = free, 1 = claimed, 2 = no level-line angle (never joins a region): the probe of a neighbour then reads the
         * compact byte map only, not the angle field, for the third of the pixels that can never pass */
        used_.resize((size_t)W * H + 8);             /* + 8: grow() reads the state bytes four at a time */
        for (size_t i = 0; i < (size_t)W * H; i++) used_[i] = angles[i] == kNotDef ? 2 : 0;
        const double binCoef = (maxGrad > 0) ? double(1024 - 1) / maxGrad : 0;
        order_.clear();
        order_.reserve((size_t)(W - 1) * (H - 1));
        uint32_t minSeedBin = 1024;               /* smallest bin of a pixel that can seed a region (has an angle) */
        for (int y = 0; y < H - 1; ++y)
            for (int x = 0; x < W - 1; ++x) {
                const uint32_t bin = (uint32_t)int(mod_[(size_t)y * W + x] * binCoef);
                if (used_[(size_t)y * W + x] == 0 && bin < minSeedBin) minSeedBin = bin;
                order_.push_back((bin << LSD_ORDER_IDX_BITS) | ((uint32_t)y << 11) | (uint32_t)x);
            }
        minSeedBin_ = minSeedBin;
        /* std::sort, as OpenCV: the order of equal bins is whatever libstdc++'s introsort leaves - reproduced move for move by
         * lsd_order::sort (introsort_restated.h) without the per-element branch mispredictions; DRFE_LSD_STD_SORT=1 calls std::sort */
        static const bool stdSort = std::getenv("DRFE_LSD_STD_SORT") != nullptr;
        /* pixels without an angle never seed (the loop below skips them): the ranges that hold only bins below the smallest
         * seed bin are left unsorted and the seed loop stops where they begin */
        if (stdSort) std::sort(order_.begin(), order_.end(), lsd_order::Before());
        else lsd_order::sort(order_.data(), order_.size(), orderTmp_, -1, -1, minSeedBin_);
    }

    typedef RectValidator::CountFn CountFn;
    void setRectMode(int rectMode) { val_.setRectMode(rectMode); }

    bool run(std::vector<float>& lines, const CountFn& counts)
    {
        /* rect_improve only READS the angle field and decides whether the segment is kept: it is taken out of the seed loop
         * (whose `used` bookkeeping is the sequential part) and evaluated for all rectangles of the frame at once */
        std::vector<RectD> pending;
        findRects(pending);
        const auto t2 = timed_ ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        if (!val_.emit(pending, lines, counts)) return false;
        if (timed_) tImprove_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count();
        return true;
    }

    /* the seed loop: region_grow -> region2rect -> refine for every seed of the ordering; accepted rectangles in seed order */
    void findRects(std::vector<RectD>& pending)
    {
        const double angTh = 22.5, densityTh = 0.7;
        const double prec = M_PI * angTh / 180, p = angTh / 180;
        const size_t minReg = val_.minReg(p);
        std::vector<RPt> reg;
        /* seeds jump across the image in bin order: a seed's 3 x 3 neighbourhood of the state bytes and of the angles is six cache
         * lines nobody has touched lately.  Asked for a few seeds ahead they are there when the seed's turn comes (round 6; the fields
         * never change and a prefetch decides nothing).  Measured on the single-frame entry: 11.2 -> 11.0 ms per frame at 12 seeds ahead,
         * 11.6 at 24 - the loop is bound by its arithmetic (a fastAtan2 per join), not by these misses */
        const size_t nOrd = order_.size();
        const OPt* const ord = order_.data();
#ifndef LSD_HOST_PREFETCH
#define LSD_HOST_PREFETCH 12            /* seeds ahead; 0 = off (A/B builds) */
#endif
        constexpr size_t kAhead = LSD_HOST_PREFETCH;
        for (size_t oi = 0; oi < nOrd; oi++) {
            const OPt key = ord[oi];
            if (kAhead && oi + kAhead < nOrd) {
                const OPt k2 = ord[oi + kAhead];
                const int x2 = (int)(k2 & 0x7FFu), y2 = (int)((k2 >> 11) & 0x7FFu);
                if (y2 >= 1 && y2 + 1 < H_) {
                    const size_t at = (size_t)(y2 - 1) * W_ + (size_t)(x2 > 0 ? x2 - 1 : 0);
                    const uint8_t* u = used_.data() + at;
                    __builtin_prefetch(u); __builtin_prefetch(u + W_); __builtin_prefetch(u + 2 * (size_t)W_);
                    const double* a = ang_ + at;
                    __builtin_prefetch(a); __builtin_prefetch(a + W_); __builtin_prefetch(a + 2 * (size_t)W_);
                }
            }
            if ((key >> LSD_ORDER_IDX_BITS) < minSeedBin_) break;          /* bins descend: no seed from here on */
            const struct { int x, y; } s = {(int)(key & 0x7FFu), (int)((key >> 11) & 0x7FFu)};
            if (used_[(size_t)s.y * W_ + s.x]) continue;          /* claimed, or no angle */
            double regAngle;
            if (!timed_) {
                grow(s.x, s.y, reg, regAngle, prec);
                if (reg.size() < minReg) continue;
                fillModgrad(reg);
            } else {   /* DRFE_TRACE_LINES: the same steps with wall-clock accounting */
                const auto t0 = std::chrono::steady_clock::now();
                grow(s.x, s.y, reg, regAngle, prec);
                tGrow_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                nGrow_++; nGrowPx_ += (long)reg.size();
                if (reg.size() < minReg) continue;
                fillModgrad(reg);
            }
            RectD rec;
            const auto t1 = timed_ ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
            toRect(reg, regAngle, prec, p, rec);
            const bool okr = refine(reg, regAngle, prec, p, rec, densityTh);
            if (timed_) { tRefine_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count(); nRect_++; }
            if (!okr) continue;
            pending.push_back(rec);
        }
    }

    double tGrow_ = 0, tRefine_ = 0, tImprove_ = 0; long nGrow_ = 0, nRect_ = 0, nGrowPx_ = 0; bool timed_ = false;
private:
    RectValidator val_;
    int W_, H_;
    const double *mod_, *ang_;
    const float* cs_;                /* device-computed (cos, sin) of float(angle) per pixel */
    std::vector<uint8_t>& used_;
    std::vector<OPt>& order_;
    std::vector<OPt>& orderTmp_;
    uint32_t minSeedBin_ = 0;

    static double sq(double v) { return v * v; }
    static double dist(double x1, double y1, double x2, double y2) { return std::sqrt(sq(x2 - x1) + sq(y2 - y1)); }
    static double diffSigned(double a, double b)
    {
        double d = a - b;
        while (d <= -M_PI) d += kTwoPi;
        while (d > M_PI) d -= kTwoPi;
        return d;
    }
    bool aligned(int x, int y, double theta, double prec) const
    {
        if (x < 0 || y < 0 || x >= W_ || y >= H_) return false;
        const double a = ang_[(size_t)y * W_ + x];
        if (a == kNotDef) return false;
        double n = theta - a;
        if (n < 0) n = -n;
        if (n > kThreeHalfPi) { n -= kTwoPi; if (n < 0) n = -n; }
        return n <= prec;
    }
    /* `float(cos(reg_angle))`, `float(sin(reg_angle))` at the start of region_grow and `cos(theta)`, `sin(theta)` of region2rect:
     * the reference takes them from the host's libm (glibc >= 2.28: within 0.55 ulp, i.e. one time in ten thousand the
     * neighbour of the correctly rounded double).  Canonical here, on the host and on the device: the correctly rounded value
     * (cr_sincos.h; what glibc <= 2.27 returned), libm only if the routine cannot certify its rounding. */
    static void seedDirection(double a, float& c, float& s)
    {
        if (!drfe_cr_sincos_f(a, &s, &c)) { c = float(std::cos(a)); s = float(std::sin(a)); }
    }
    static void rectDirection(double theta, double& c, double& s)
    {
        if (!drfe_cr_sincos(theta, &s, &c)) { c = std::cos(theta); s = std::sin(theta); }
    }
    /* region_grow.  The gradient magnitude of a member is filled in by fillModgrad() for the regions that reach the minimum
     * size only (one in twenty: the rest are dropped without ever reading it). */
    void grow(int sx, int sy, std::vector<RPt>& reg, double& regAngle, double prec)
    {
        reg.clear();
        regAngle = ang_[(size_t)sy * W_ + sx];
        reg.push_back({sx, sy, regAngle, 0.0});
        /* the seed's own direction enters the running sums when the first neighbour joins: most of the ~35 000 seeds of a
         * frame never get one, and these are the only two libm calls of the loop */
        float sumdx = 0.f, sumdy = 0.f;
        bool seeded = false;
        const double seedAngle = regAngle;
        uint8_t* const used = used_.data();
        used[(size_t)sy * W_ + sx] = 1;
        auto probe = [&](int xx, int yy, uint8_t& u) {
            /* isAligned() on a pixel known to be inside the image and to have an angle */
            const size_t at = (size_t)yy * W_ + xx;
            const double a = ang_[at];
            double dn = regAngle - a;
            if (dn < 0) dn = -dn;
            if (dn > kThreeHalfPi) { dn -= kTwoPi; if (dn < 0) dn = -dn; }
            if (dn <= prec) {
                u = 1;
                reg.push_back({xx, yy, a, 0.0});
                if (!seeded) { seedDirection(seedAngle, sumdx, sumdy); seeded = true; }
                sumdx += cs_[2 * at];        /* cos(float(angle)), shared routine (device) */
                sumdy += cs_[2 * at + 1];    /* sin(float(angle)) */
                regAngle = drfe_fast_atan2(sumdy, sumdx) * kDeg2Rad;
            }
        };
        /* the three state bytes of a row as bits: byte == 0 (free; 1 = claimed, 2 = no angle) -> bit c of the result */
        auto freeBits = [](const uint8_t* p) {
            uint32_t r;
            std::memcpy(&r, p, 4);
            r &= 0x00FFFFFFu;
            r = ~(r | (r >> 1)) & 0x00010101u;
            return (r | (r >> 7) | (r >> 14)) & 7u;
        };
        for (size_t i = 0; i < reg.size(); i++) {
            const int px = reg[i].x, py = reg[i].y;
            if (px >= 1 && py >= 1 && px <= W_ - 2 && py <= H_ - 2) {
                /* interior member: the 3 x 3 neighbourhood's state as nine bits in visiting order (rows, then columns), read
                 * with three loads; nothing inside it changes behind the scan except the neighbour just claimed, so the bits
                 * stay valid for the whole visit and only the free neighbours (few) are looked at */
                uint8_t* u0 = used + (size_t)(py - 1) * W_ + (px - 1);
                uint32_t f = freeBits(u0) | (freeBits(u0 + W_) << 3) | (freeBits(u0 + 2 * (size_t)W_) << 6);
                while (f) {
                    const int k = __builtin_ctz(f);
                    f &= f - 1;
                    const int r = (k * 11) >> 5, c = k - 3 * r;          /* k / 3, k % 3 for k < 9 */
                    probe(px - 1 + c, py - 1 + r, u0[(size_t)r * W_ + c]);
                }
                continue;
            }
            for (int yy = std::max(py - 1, 0); yy <= std::min(py + 1, H_ - 1); ++yy)
                for (int xx = std::max(px - 1, 0); xx <= std::min(px + 1, W_ - 1); ++xx) {
                    uint8_t& u = used[(size_t)yy * W_ + xx];
                    if (u != 0) continue;
                    probe(xx, yy, u);
                }
        }
    }
    void fillModgrad(std::vector<RPt>& reg) const
    {
        for (RPt& r : reg) r.modgrad = mod_[(size_t)r.y * W_ + r.x];
    }
    double thetaOf(const std::vector<RPt>& reg, double x, double y, double regAngle, double prec) const
    {
        double Ixx = 0, Iyy = 0, Ixy = 0;
        for (const RPt& r : reg) {
            const double dx = double(r.x) - x, dy = double(r.y) - y;
            Ixx += dy * dy * r.modgrad;
            Iyy += dx * dx * r.modgrad;
            Ixy -= dx * dy * r.modgrad;
        }
        const double lambda = 0.5 * (Ixx + Iyy - std::sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
        double theta = (std::fabs(Ixx) > std::fabs(Iyy)) ? double(drfe_fast_atan2(float(lambda - Ixx), float(Ixy)))
                                                         : double(drfe_fast_atan2(float(Ixy), float(lambda - Iyy)));
        theta *= kDeg2Rad;
        if (std::fabs(diffSigned(theta, regAngle)) > prec) theta += M_PI;
        return theta;
    }
    void toRect(const std::vector<RPt>& reg, double regAngle, double prec, double p, RectD& rec) const
    {
        double x = 0, y = 0, sum = 0;
        for (const RPt& r : reg) { x += double(r.x) * r.modgrad; y += double(r.y) * r.modgrad; sum += r.modgrad; }
        x /= sum; y /= sum;
        const double theta = thetaOf(reg, x, y, regAngle, prec);
        double dx, dy;
        rectDirection(theta, dx, dy);
        double lmin = 0, lmax = 0, wmin = 0, wmax = 0;
        for (const RPt& r : reg) {
            const double rx = double(r.x) - x, ry = double(r.y) - y;
            const double l = rx * dx + ry * dy, w = -rx * dy + ry * dx;
            if (l > lmax) lmax = l; else if (l < lmin) lmin = l;
            if (w > wmax) wmax = w; else if (w < wmin) wmin = w;
        }
        rec = {x + lmin * dx, y + lmin * dy, x + lmax * dx, y + lmax * dy, wmax - wmin, x, y, theta, dx, dy, prec, p};
        if (rec.width < 1.0) rec.width = 1.0;
    }
    bool shrink(std::vector<RPt>& reg, double regAngle, double prec, double p, RectD& rec, double density, double densityTh)
    {
        const double xc = double(reg[0].x), yc = double(reg[0].y);
        const double r1 = sq(rec.x1 - xc) + sq(rec.y1 - yc), r2 = sq(rec.x2 - xc) + sq(rec.y2 - yc);
        double radSq = r1 > r2 ? r1 : r2;
        while (density < densityTh) {
            radSq *= 0.75 * 0.75;
            for (size_t i = 0; i < reg.size(); ++i)
                if (sq(double(reg[i].x) - xc) + sq(double(reg[i].y) - yc) > radSq) {
                    used_[(size_t)reg[i].y * W_ + reg[i].x] = 0;
                    std::swap(reg[i], reg[reg.size() - 1]);
                    reg.pop_back();
                    --i;
                }
            if (reg.size() < 2) return false;
            toRect(reg, regAngle, prec, p, rec);
            density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        }
        return true;
    }
    bool refine(std::vector<RPt>& reg, double regAngle, double prec, double p, RectD& rec, double densityTh)
    {
        double density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        if (density >= densityTh) return true;
        const double xc = double(reg[0].x), yc = double(reg[0].y), angC = reg[0].angle;
        double sum = 0, ssum = 0;
        int n = 0;
        for (const RPt& r : reg) {
            used_[(size_t)r.y * W_ + r.x] = 0;
            if (dist(xc, yc, r.x, r.y) < rec.width) {
                const double d = diffSigned(r.angle, angC);
                sum += d; ssum += d * d; ++n;
            }
        }
        const double mean = sum / double(n);
        const double tau = 2.0 * std::sqrt((ssum - 2.0 * mean * sum) / double(n) + mean * mean);
        const int sx = reg[0].x, sy = reg[0].y;
        grow(sx, sy, reg, regAngle, tau);
        if (reg.size() < 2) return false;
        fillModgrad(reg);
        toRect(reg, regAngle, prec, p, rec);
        density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        if (density < densityTh) return shrink(reg, regAngle, prec, p, rec, density, densityTh);
        return true;
    }
public:
    /* rect_nfa's pixel loop on the host: DRFE_LSD_CHECK=1 cross-checks k_rect_counts with it, drfe_lsd_segments_host counts
     * with it.  rectMode (lsd_walk_mode of the configured mode) 0: the literal OpenCV 3.4 source (integer corners and step quotients, (y - tailp->p.x) in the
     * second steps' guards and denominators); 1: the real-valued reading of round 3 (lines_kernels.hip, rect_walk_setup) */
    void countHost(const RectCand& rec, int rectMode, int& total, int& alg) const
    {
        struct Corner { int x, y; bool taken; };
        const double hw = rec.width / 2.0, dyhw = rec.dy * hw, dxhw = rec.dx * hw;
        Corner c[4] = {{int(rec.x1 - dyhw), int(rec.y1 + dxhw), false},
                       {int(rec.x2 - dyhw), int(rec.y2 + dxhw), false},
                       {int(rec.x2 + dyhw), int(rec.y2 - dxhw), false},
                       {int(rec.x1 + dyhw), int(rec.y1 - dxhw), false}};
        std::sort(c, c + 4, [](const Corner& a, const Corner& b) { return a.x == b.x ? a.y < b.y : a.x < b.x; });
        Corner *lo = &c[0], *hi = &c[0];
        for (int i = 1; i < 4; ++i) { if (lo->y > c[i].y) lo = &c[i]; if (hi->y < c[i].y) hi = &c[i]; }
        lo->taken = true;
        Corner *left = 0, *right = 0, *tail = 0;
        for (int i = 0; i < 4; ++i) if (!c[i].taken) { if (!left || left->x > c[i].x) left = &c[i]; }
        left->taken = true;
        for (int i = 0; i < 4; ++i) if (!c[i].taken) { if (!right || right->x < c[i].x) right = &c[i]; }
        right->taken = true;
        for (int i = 0; i < 4; ++i) if (!c[i].taken) { if (!tail || tail->x > c[i].x) tail = &c[i]; }
        double fl, sl, fr, sr;
        if (rectMode == 0) {
            fl = (lo->y != left->y) ? (lo->x - left->x) / (lo->y - left->y) : 0;
            sl = (left->y != tail->x) ? (left->x - tail->x) / (left->y - tail->x) : 0;
            fr = (lo->y != right->y) ? (lo->x - right->x) / (lo->y - right->y) : 0;
            sr = (right->y != tail->x) ? (right->x - tail->x) / (right->y - tail->x) : 0;
        } else {
            fl = (lo->y != left->y) ? double(lo->x - left->x) / double(lo->y - left->y) : 0;
            sl = (left->y != tail->x) ? double(left->x - tail->x) / double(left->y - tail->y) : 0;
            fr = (lo->y != right->y) ? double(lo->x - right->x) / double(lo->y - right->y) : 0;
            sr = (right->y != tail->x) ? double(right->x - tail->x) / double(right->y - tail->y) : 0;
            if (!std::isfinite(sl)) sl = 0;
            if (!std::isfinite(sr)) sr = 0;
        }
        double lstep = fl, rstep = fr, lx = lo->x, rx = lo->x;
        total = 0; alg = 0;
        for (int y = lo->y; y <= hi->y; ++y) {
            if (y < 0 || y >= H_) continue;
            for (int x = int(lx); x <= int(rx); ++x) {
                if (x < 0 || x >= W_) continue;
                ++total;
                if (aligned(x, y, rec.theta, rec.prec)) ++alg;
            }
            if (y >= left->y) lstep = sl;
            if (y >= right->y) rstep = sr;
            lx += lstep;
            rx += rstep;
        }
    }
};

/* the Gaussian weights of BinaryDescriptor::computeLBD: local (3 x 7 rows, sigma 7) and global (63 rows, sigma 31), cast to
 * float where the reference casts them */
static const LbdTables& lbdTables()
{
    static LbdTables t;
    static std::once_flag once;
    std::call_once(once, [] {
        const int NB = 9, WB = 7;
        double u = (WB * 3 - 1) / 2, sigma = (WB * 2 + 1) / 2, inv = -1 / (2 * sigma * sigma);
        for (int i = 0; i < WB * 3; i++) t.coefL[i] = (float)std::exp((i - u) * (i - u) * inv);
        u = (NB * WB - 1) / 2; sigma = u; inv = -1 / (2 * sigma * sigma);
        for (int i = 0; i < NB * WB; i++) t.coefG[i] = (float)std::exp((i - u) * (i - u) * inv);
    });
    return t;
}

static LineTaps gaussTaps(int n, double sigma)
{
    LineTaps t;
    t.n = n;
    double v[9], sum = 0;
    const double s2 = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; i++) { const double x = i - (n - 1) * 0.5; v[i] = std::exp(s2 * x * x); sum += v[i]; }
    sum = 1.0 / sum;
    for (int i = 0; i < 9; i++) t.t[i] = i < n ? (int)std::rint(v[i] * sum * 256.0) : 0;
    return t;
}

} // namespace

/* One line-extraction lane: stream, NFA / LBD scratch, host buffers, its own error string.  The context owns one (the
 * single-frame entry) and, for drfe_lsd_extract_batch, a pool of them, one per host thread.  `ls` of a lane holds the image
 * arrays of ONE frame slot (host-grow path); the batch's device-grow path keeps the frames' arrays in the context's batch
 * arena and uses a lane for its stream and small scratch only. */
struct LineHost {                     /* per-lane host buffers reused across frames */
    std::vector<double> modgrad, angles;
    std::vector<float> cs;
    std::vector<uint8_t> used;
    std::vector<OPt> order, orderTmp;
};
struct LineWorker {
    std::string err;
    LinesScratch* ls = nullptr;
    hipStream_t stream = nullptr;
    bool ownsStream = false;
    LineHost* host = nullptr;
    hipEvent_t pollEv = nullptr;      /* pool lanes: drfe_pool_sync sleeps between polls instead of spinning */
    int rectMode = 0;                 /* rect_nfa's reading (drfe_lsd_configure_rect): 0 literal OpenCV 3.4, 1 real-valued */
};

static hipError_t lane_sync(LineWorker* c)
{
    return c->pollEv ? drfe_pool_sync(c->stream, c->pollEv) : hipStreamSynchronize(c->stream);
}

static void scratch_free(LinesScratch*& s)
{
    if (!s) return;
    void* ptrs[] = {s->d_img, s->d_blur, s->d_scaled, s->d_tmp16, s->d_modgrad, s->d_angles, s->d_cs, s->d_cs0, s->d_meta, s->d_gx, s->d_gy, s->d_cands, s->d_counts,
                    s->d_lbdLines, s->d_lbdOut, s->d_order, s->d_reg, s->d_tmp, s->d_notdef, s->d_regMw, s->d_tmpMw, s->d_gbmMw, s->d_rects, s->d_out, s->d_frames, s->d_ordStatus, s->d_segs, s->d_lgamma, s->d_kl, s->d_klLineF, s->d_klLbd, s->d_klDesc, s->d_klOut};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    void* hptrs[] = {s->h_order, s->h_meta, s->h_rects, s->h_out, s->h_frames, s->h_ordStatus, s->h_cands, s->h_counts, s->h_segs, s->h_kl, s->h_klLineF, s->h_klDesc, s->h_klOut};
    for (void* p : hptrs) if (p) (void)hipHostFree(p);
    delete s;
    s = nullptr;
}

void drfe_lines_free(drfe_ctx* c)
{
    scratch_free(c->ls);
    scratch_free(c->lsBatch);
    delete static_cast<LineHost*>(c->lineHost);
    c->lineHost = nullptr;
    auto* pool = static_cast<std::vector<LineWorker>*>(c->lineWorkers);
    if (pool) {
        for (LineWorker& w : *pool) {
            scratch_free(w.ls);
            delete w.host;
            if (w.ownsStream && w.stream) (void)hipStreamDestroy(w.stream);
            if (w.pollEv) (void)hipEventDestroy(w.pollEv);
        }
        delete pool;
        c->lineWorkers = nullptr;
    }
}

#define LSD_RECT_CAP 4096            /* accepted regions per frame the device path can hold (a 640 x 480 frame has ~1500) */

/* image arrays for `frames` slots of w x h; grow = the device region-growing buffers and their pinned mirrors as well;
 * images = false: a lane that only needs the NFA / LBD scratch (its frames live in the batch arena) */
static int ensure_lines(std::string& err, LinesScratch*& ls, int w, int h, int frames, bool images, bool grow)
{
    struct E { std::string& err; } e{err};
#define LCHK(call)                                                                              \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) { e.err = std::string(#call) + ": " + hipGetErrorString(e__); return DRFE_ERR_HIP; } \
    } while (0)
    if (ls && ls->w == w && ls->h == h && ls->frames >= frames && (!images || ls->d_img) && (!grow || ls->d_order)) return DRFE_OK;
    scratch_free(ls);
    LinesScratch* s = new (std::nothrow) LinesScratch();
    if (!s) return DRFE_ERR_INVALID;
    std::memset(s, 0, sizeof(*s));
    ls = s;
    s->w = w; s->h = h; s->frames = frames;
    s->sw = (int)std::rint(w * 0.8); s->sh = (int)std::rint(h * 0.8);   /* saturate_cast<int>(size * inv_scale) */
    const size_t F = (size_t)frames, n = (size_t)w * h * F, ns = (size_t)s->sw * s->sh * F;
    if (images) {
        LCHK(hipMalloc((void**)&s->d_img, n));
        LCHK(hipMalloc((void**)&s->d_blur, n));
        LCHK(hipMalloc((void**)&s->d_scaled, ns));
        LCHK(hipMalloc((void**)&s->d_tmp16, n * 2));
        LCHK(hipMalloc((void**)&s->d_modgrad, ns * 8));
        LCHK(hipMalloc((void**)&s->d_angles, ns * 8));
        LCHK(hipMalloc((void**)&s->d_cs, ns * 8));
        LCHK(hipMalloc((void**)&s->d_meta, 16 * F));
        LCHK(hipMalloc((void**)&s->d_gx, n * 2));
        LCHK(hipMalloc((void**)&s->d_gy, n * 2));
    }
    if (grow) {
        const size_t nk = (size_t)(s->sw - 1) * (s->sh - 1) * F;
        /* accepted regions scale with the field: 4096 at 512 x 384 (~1500 seen), 16384 at 1024 x 768 (~5800 seen on a 1280 x 960 frame) */
        s->rectCap = std::max(LSD_RECT_CAP, s->sw * s->sh / 48);
        LCHK(hipMalloc((void**)&s->d_order, nk * 4));
        LCHK(hipMalloc((void**)&s->d_reg, ns * 4));
        LCHK(hipMalloc((void**)&s->d_cs0, ns * 8));
        LCHK(hipMalloc((void**)&s->d_tmp, ns * 4));
        LCHK(hipMalloc((void**)&s->d_notdef, (((size_t)s->sw * s->sh + 31) / 32) * 4 * F));
        /* the multi-wave growth: each of a frame's four wavefronts owns half a field's worth of member-list entries (a region
         * beyond that hands the frame to the host) */
        s->regCapMw = s->sw * s->sh / 2;
        LCHK(hipMalloc((void**)&s->d_regMw, F * 4 * (size_t)s->regCapMw * 4));
        LCHK(hipMalloc((void**)&s->d_tmpMw, F * 4 * (size_t)s->regCapMw * 4));
        LCHK(hipMalloc((void**)&s->d_gbmMw, F * (((size_t)s->sw * s->sh + 31) / 32) * 4));
        LCHK(hipMalloc((void**)&s->d_rects, F * s->rectCap * sizeof(LsdRect)));
        LCHK(hipMalloc((void**)&s->d_out, F * DRFE_LSD_OUT_INTS * sizeof(int)));
        LCHK(hipMalloc((void**)&s->d_frames, F * sizeof(LsdGrowFrame)));
        LCHK(hipMalloc((void**)&s->d_ordStatus, F * sizeof(int)));
        LCHK(hipHostMalloc((void**)&s->h_ordStatus, F * sizeof(int), hipHostMallocDefault));
        LCHK(hipHostMalloc((void**)&s->h_meta, 16 * F, hipHostMallocDefault));
        LCHK(hipHostMalloc((void**)&s->h_rects, F * s->rectCap * sizeof(LsdRect), hipHostMallocDefault));
        LCHK(hipHostMalloc((void**)&s->h_out, F * DRFE_LSD_OUT_INTS * sizeof(int), hipHostMallocDefault));
        LCHK(hipHostMalloc((void**)&s->h_frames, F * sizeof(LsdGrowFrame), hipHostMallocDefault));
        LCHK(hipMalloc((void**)&s->d_segs, F * s->rectCap * sizeof(LsdSegOut)));
        LCHK(hipHostMalloc((void**)&s->h_segs, F * s->rectCap * sizeof(LsdSegOut), hipHostMallocDefault));
        /* log_gamma at the integers, by the host's libm (RectValidator::fillTables fills and uploads it on first use) */
        s->lgammaN = 0;
        LCHK(hipMalloc((void**)&s->d_lgamma, ((size_t)s->sw * s->sh + 2) * sizeof(double)));
    }
#undef LCHK
    return DRFE_OK;
}

struct LsdParams {
    LineTaps lsdTaps, lbdTaps;
    double rho;
    LsdParams()
    {
        /* LineSegmentDetector defaults: scale 0.8, sigma_scale 0.6 -> sigma 0.75, 7x7 kernel; quant 2, ang_th 22.5 */
        const double sigma = 0.6 / 0.8;
        const int hk = (int)std::ceil(sigma * std::sqrt(2 * 3.0 * std::log(10.0)));
        lsdTaps = gaussTaps(1 + 2 * hk, sigma);
        lbdTaps = gaussTaps(5, 1.0);
        rho = 2.0 / std::sin(M_PI * 22.5 / 180);
    }
};

/* the device arrays of one frame that the validation / descriptor stages read */
struct FrameView { int w, h, sw, sh; const double* d_angles; const int16_t* d_gx; const int16_t* d_gy; };

/* DRFE_TRACE_LINES accounting of the counting rounds: wall and CPU time of this thread inside them */
static thread_local double g_countsWallUs = 0, g_countsCpuUs = 0;
static inline double thread_cpu_us() { struct timespec t; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }

/* rect_nfa's pixel counts on the device (k_rect_counts) through lane c's stream and grow-only scratch */
static RectValidator::CountFn device_counts(LineWorker* c, const FrameView& v, int& countRc, const SegmentFinder* check)
{
    return [c, v, &countRc, check](const std::vector<RectCand>& cands, std::vector<int2>& out) -> bool {
        LinesScratch* s = c->ls;
        hipStream_t st = c->stream;
        const size_t nc = cands.size();
        const auto tw0 = std::chrono::steady_clock::now();
        const double tc0 = thread_cpu_us();
        struct Acc { std::chrono::steady_clock::time_point w; double c; ~Acc() { g_countsWallUs += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w).count(); g_countsCpuUs += thread_cpu_us() - c; } } acc{tw0, tc0};
        out.resize(nc);
        if (nc > s->candCap) {
            if (s->d_cands) (void)hipFree(s->d_cands);
            if (s->d_counts) (void)hipFree(s->d_counts);
            if (s->h_cands) (void)hipHostFree(s->h_cands);
            if (s->h_counts) (void)hipHostFree(s->h_counts);
            s->d_cands = nullptr; s->d_counts = nullptr; s->h_cands = nullptr; s->h_counts = nullptr;
            s->candCap = std::max<size_t>(nc * 2, 4096);
            if (hipMalloc((void**)&s->d_cands, s->candCap * sizeof(RectCand)) != hipSuccess ||
                hipMalloc((void**)&s->d_counts, s->candCap * sizeof(int2)) != hipSuccess ||
                hipHostMalloc((void**)&s->h_cands, s->candCap * sizeof(RectCand), hipHostMallocDefault) != hipSuccess ||
                hipHostMalloc((void**)&s->h_counts, s->candCap * sizeof(int2), hipHostMallocDefault) != hipSuccess) {
                s->candCap = 0; c->err = "lsd_extract: hipMalloc of the NFA scratch failed"; countRc = DRFE_ERR_HIP; return false;
            }
        }
        std::memcpy(s->h_cands, cands.data(), nc * sizeof(RectCand));
        hipError_t e = hipMemcpyAsync(s->d_cands, s->h_cands, nc * sizeof(RectCand), hipMemcpyHostToDevice, st);
        /* the download is issued only when the kernel has finished: queued behind it, it would sit in a DMA ring until then and
         * hold up the other lanes' copies behind it (a kernel writing straight into pinned host memory is worse: measured 2.3x
         * slower for the whole front-end - every such kernel ends in a system-scope write-back) */
        if (e == hipSuccess) e = drfe_launch_rect_counts(s->d_cands, (int)nc, v.d_angles, v.sw, v.sh, lsd_walk_mode(c->rectMode), s->d_counts, st);
        if (e == hipSuccess) e = lane_sync(c);
        if (e == hipSuccess) e = hipMemcpyAsync(s->h_counts, s->d_counts, nc * sizeof(int2), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = lane_sync(c);
        if (e == hipSuccess) std::memcpy(out.data(), s->h_counts, nc * sizeof(int2));
        if (e != hipSuccess) { c->err = std::string("lsd_extract: rectangle counting: ") + hipGetErrorString(e); countRc = DRFE_ERR_HIP; return false; }
        if (check)
            for (size_t k = 0; k < nc; k++) {
                int t = 0, a = 0;
                check->countHost(cands[k], lsd_walk_mode(c->rectMode), t, a);
                if (t != out[k].x || a != out[k].y)
                    std::fprintf(stderr, "k_rect_counts mismatch: cand %zu device (%d, %d) host (%d, %d)  x1 %.17g y1 %.17g x2 %.17g y2 %.17g w %.17g dx %.17g dy %.17g theta %.17g prec %.17g\n",
                                 k, out[k].x, out[k].y, t, a, cands[k].x1, cands[k].y1, cands[k].x2, cands[k].y2, cands[k].width, cands[k].dx, cands[k].dy, cands[k].theta, cands[k].prec);
            }
        return true;
    };
}

/* from the detector's segments to the caller's buffers: LSDDetector::detect's KeyLine fields (octave 0), the reference's
 * response cut (src/LSDextractor.cpp:23-28), LBD descriptors on the device (k_lbd), line equations (:32-42) */
static int keylines_and_descriptors(LineWorker* c, const FrameView& v, const std::vector<float>& segs, int max_lines, drfe_keyline* lines,
                                    uint8_t* ldesc, double* line_f, int cap, int* n_lines, int* n_detected)
{
    const int w = v.w, h = v.h;
    LinesScratch* s = c->ls;
    hipStream_t st = c->stream;
    std::vector<drfe_keyline> kls;
    int classCounter = -1;
    for (size_t k = 0; k + 3 < segs.size(); k += 4) {
        float e[4] = {segs[k], segs[k + 1], segs[k + 2], segs[k + 3]};
        if (e[0] < 0) e[0] = 0;
        if (e[0] >= w) e[0] = (float)w - 1.0f;
        if (e[2] < 0) e[2] = 0;
        if (e[2] >= w) e[2] = (float)w - 1.0f;
        if (e[1] < 0) e[1] = 0;
        if (e[1] >= h) e[1] = (float)h - 1.0f;
        if (e[3] < 0) e[3] = 0;
        if (e[3] >= h) e[3] = (float)h - 1.0f;
        drfe_keyline kl;
        kl.start_point_x = e[0]; kl.start_point_y = e[1]; kl.end_point_x = e[2]; kl.end_point_y = e[3];
        kl.s_point_in_octave_x = e[0]; kl.s_point_in_octave_y = e[1]; kl.e_point_in_octave_x = e[2]; kl.e_point_in_octave_y = e[3];
        kl.line_length = (float)std::sqrt(std::pow(e[0] - e[2], 2) + std::pow(e[1] - e[3], 2));
        const int x0 = drfe_round_half_even(e[0]), y0 = drfe_round_half_even(e[1]);
        const int x1 = drfe_round_half_even(e[2]), y1 = drfe_round_half_even(e[3]);
        kl.num_of_pixels = std::max(std::abs(x1 - x0), std::abs(y1 - y0)) + 1;   /* LineIterator(...).count */
        kl.angle = (float)std::atan2((double)(e[3] - e[1]), (double)(e[2] - e[0]));
        kl.class_id = ++classCounter;
        kl.octave = 0;
        kl.size = (e[2] - e[0]) * (e[3] - e[1]);
        kl.response = kl.line_length / std::max(w, h);
        kl.pt_x = (e[2] + e[0]) / 2; kl.pt_y = (e[3] + e[1]) / 2;
        kls.push_back(kl);
    }
    if (n_detected) *n_detected = (int)kls.size();
    if ((int)kls.size() > max_lines) {   /* src/LSDextractor.cpp:23-28 */
        std::sort(kls.begin(), kls.end(), [](const drfe_keyline& a, const drfe_keyline& b) { return a.response > b.response; });
        kls.resize(max_lines);
        for (int i = 0; i < max_lines; i++) kls[i].class_id = i;
    }
    const int nl = (int)kls.size();
    *n_lines = nl;
    if (nl > cap) { c->err = "lsd_extract: line buffer too small"; return DRFE_ERR_CAPACITY; }
    /* LBD descriptors of the kept lines on the device (k_lbd); the direction cosines come from this host's libm, as the
     * reference's do */
    if (nl > 0 && ldesc) {
        if ((size_t)nl > s->lbdCap) {
            if (s->d_lbdLines) (void)hipFree(s->d_lbdLines);
            if (s->d_lbdOut) (void)hipFree(s->d_lbdOut);
            s->d_lbdLines = nullptr; s->d_lbdOut = nullptr;
            s->lbdCap = std::max<size_t>((size_t)nl, 64);
            HIPCHK(c, hipMalloc((void**)&s->d_lbdLines, s->lbdCap * sizeof(LbdLine)));
            HIPCHK(c, hipMalloc((void**)&s->d_lbdOut, s->lbdCap * 32));
        }
        std::vector<LbdLine> ll(nl);
        for (int i = 0; i < nl; i++) {
            const drfe_keyline& kl = kls[i];
            ll[i].midX = (float)(0.5 * (kl.s_point_in_octave_x + kl.e_point_in_octave_x));
            ll[i].midY = (float)(0.5 * (kl.s_point_in_octave_y + kl.e_point_in_octave_y));
            ll[i].dL0 = (float)std::cos((double)kl.angle);
            ll[i].dL1 = (float)std::sin((double)kl.angle);
            ll[i].len = kl.num_of_pixels; ll[i].pad = 0;
        }
        HIPCHK(c, hipMemcpyAsync(s->d_lbdLines, ll.data(), nl * sizeof(LbdLine), hipMemcpyHostToDevice, st));
        HIPCHK(c, drfe_launch_lbd(s->d_lbdLines, nl, v.d_gx, v.d_gy, w, h, lbdTables(), s->d_lbdOut, st));
        HIPCHK(c, hipMemcpyAsync(ldesc, s->d_lbdOut, (size_t)nl * 32, hipMemcpyDeviceToHost, st));
        HIPCHK(c, lane_sync(c));
    }
    for (int i = 0; i < nl; i++) {
        if (lines) lines[i] = kls[i];
        if (line_f) {   /* normalised cross product of the homogeneous end points, :32-42 */
            const double sx = kls[i].start_point_x, sy = kls[i].start_point_y, ex = kls[i].end_point_x, ey = kls[i].end_point_y;
            const double l0 = sy * 1.0 - 1.0 * ey, l1 = 1.0 * ex - sx * 1.0, l2 = sx * ey - sy * ex;
            const double nrm = std::sqrt(l0 * l0 + l1 * l1 + l2 * l2);
            line_f[3 * i] = l0 / nrm; line_f[3 * i + 1] = l1 / nrm; line_f[3 * i + 2] = l2 / nrm;
        }
    }
    return DRFE_OK;
}

/* the sequential half on the HOST for one frame whose fields lie at slot `slot` of `fields` (copied back over lane c's stream),
 * then validation + key lines + descriptors */
static int host_grow_and_finish(LineWorker* c, LinesScratch* fields, int slot, int max_lines, drfe_keyline* lines, uint8_t* ldesc,
                                double* line_f, int cap, int* n_lines, int* n_detected, std::chrono::steady_clock::time_point tStart)
{
    hipStream_t st = c->stream;
    const bool trace = std::getenv("DRFE_TRACE_LINES") != nullptr;
    const size_t ns = (size_t)fields->sw * fields->sh, n = (size_t)fields->w * fields->h;
    if (!c->host) c->host = new LineHost();
    LineHost& H = *c->host;
    std::vector<double>&modgrad = H.modgrad, &angles = H.angles;
    modgrad.resize(ns); angles.resize(ns); H.cs.resize(2 * ns);
    unsigned long long meta[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(modgrad.data(), fields->d_modgrad + ns * slot, ns * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(angles.data(), fields->d_angles + ns * slot, ns * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(H.cs.data(), fields->d_cs + ns * slot, ns * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(meta, fields->d_meta + 2 * (size_t)slot, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(c, lane_sync(c));
    double maxGrad = -1;
    if (meta[0]) std::memcpy(&maxGrad, &meta[0], 8);

    const auto tDev = std::chrono::steady_clock::now();
    std::vector<float> segs;
    SegmentFinder finder(fields->sw, fields->sh, modgrad.data(), angles.data(), H.cs.data(), maxGrad, H.used, H.order, H.orderTmp);
    const auto tSort = std::chrono::steady_clock::now();
    finder.timed_ = trace;
    finder.setRectMode(c->rectMode);
    int countRc = DRFE_OK;
    const FrameView v = {fields->w, fields->h, fields->sw, fields->sh, fields->d_angles + ns * slot, fields->d_gx + n * slot, fields->d_gy + n * slot};
    const bool checkCounts = std::getenv("DRFE_LSD_CHECK") != nullptr;
    if (!finder.run(segs, device_counts(c, v, countRc, checkCounts ? &finder : nullptr))) return countRc;
    const auto tSeg = std::chrono::steady_clock::now();
    if (trace) std::fprintf(stderr, "drfe_lsd_extract: pixel ordering (bins + std::sort) %.2f ms; grow %.2f ms (%ld regions, %ld pixels); rect+refine %.2f ms (%ld); improve/NFA %.2f ms\n",
                            std::chrono::duration<double, std::milli>(tSort - tDev).count(), finder.tGrow_, finder.nGrow_, finder.nGrowPx_, finder.tRefine_, finder.nRect_, finder.tImprove_);
    struct TraceAtExit {   /* DRFE_TRACE_LINES=1: where a call spends its time (device passes + copies | LSD host | LBD host) */
        bool on; std::chrono::steady_clock::time_point a, b, cc;
        ~TraceAtExit() {
            if (!on) return;
            const auto e = std::chrono::steady_clock::now();
            auto ms = [](auto x, auto y) { return std::chrono::duration<double, std::milli>(y - x).count(); };
            std::fprintf(stderr, "drfe_lsd_extract: device+copies %.2f ms, LSD host %.2f ms, keylines+LBD host %.2f ms\n", ms(a, b), ms(b, cc), ms(cc, e));
        }
    } traceAtExit{trace, tStart, tDev, tSeg};
    return keylines_and_descriptors(c, v, segs, max_lines, lines, ldesc, line_f, cap, n_lines, n_detected);
}

/* LineSegment::ExtractLineSegment for one frame on one lane, region growing on the host (the low-latency path) */
static int lsd_extract_core(LineWorker* c, int device, const uint8_t* gray, int w, int h, size_t stride, int max_lines,
                            drfe_keyline* lines, uint8_t* ldesc, double* line_f, int cap, int* n_lines, int* n_detected)
{
    *n_lines = 0;
    HIPCHK(c, hipSetDevice(device));
    int rc = ensure_lines(c->err, c->ls, w, h, 1, true, false);
    if (rc != DRFE_OK) return rc;
    LinesScratch* s = c->ls;
    if (s->sw > 2048 || s->sh > 2048) { c->err = "lsd_extract: image larger than 2560 x 2560 (pixel-ordering keys)"; return DRFE_ERR_INVALID; }
    static const LsdParams P;
    hipStream_t st = c->stream;
    const auto tStart = std::chrono::steady_clock::now();
    HIPCHK(c, hipMemcpy2DAsync(s->d_img, (size_t)w, gray, stride, (size_t)w, (size_t)h, hipMemcpyHostToDevice, st));
    HIPCHK(c, drfe_launch_lines_passes(s->d_img, w, h, P.lsdTaps, P.lbdTaps, s, 0, 1, P.rho, st));
    return host_grow_and_finish(c, s, 0, max_lines, lines, ldesc, line_f, cap, n_lines, n_detected, tStart);
}

/* ---- the batch entry with region growing on the device -------------------------------------------------------------------
 * Per chunk of frames: image passes + ordering keys (device) -> keys to the host -> the ordering (host: std::sort's
 * permutation, introsort_restated.h, one task per frame on the pool) -> orderings to the device -> k_lsd_grow, one
 * wavefront per frame -> accepted rectangles to the host -> per frame on the pool: rect_improve / NFA with k_rect_counts,
 * key lines, k_lbd.  Chunks overlap: while one chunk's frames grow on the device, the pool orders the next and validates
 * the previous. */
namespace {

struct BatchJob {
    drfe_ctx* c;
    LinesScratch* A;                   /* the batch arena (frame slots) */
    std::vector<LineWorker>* pool;
    const uint8_t* gray; size_t frameStride, stride;
    int w, h, nframes, maxLines, cap;
    drfe_keyline* lines; uint8_t* ldesc; double* lineF; int* nLines; int* nDetected;
    int chunk, nChunks;
    std::vector<hipStream_t> chunkStream;
    std::vector<hipEvent_t> keysReady, growDone;
    std::vector<std::atomic<int>> sortedInChunk;
    std::vector<int> chunkState;         /* 0 = ordering, 1 = growing on the device, 3 = a worker fetches its status words, 2 = released to validation (under mu) */
    std::mutex mu;
    std::condition_variable cv;
    std::deque<int> sortQ, finishQ;      /* frame indices */
    int pendingFinish = 0;               /* frames not yet finished */
    int firstRc = DRFE_OK; std::string firstErr;
    bool abort = false;
    double prec, p; int minReg;
    bool deviceNfa = true;              /* rect_improve + NFA decisions by k_rect_improve (default) or on the pool with the host's libm (DRFE_LSD_HOST_NFA=1) */
    bool deviceKl = true;               /* key lines, response cut, LBD and line equations on the device too (k_lsd_keylines + k_lbd): the workers only copy */
    std::atomic<long> klToHost{0};
    LsdNfaTables nfaTab;
    int rectMode = 0;
    std::atomic<long> nfaToHost{0};     /* frames whose NFA decisions the device could not certify */
    std::atomic<int> nfaWhy{0}; std::atomic<long> nfaWhyCount[8] = {{0}, {0}, {0}, {0}, {0}, {0}, {0}, {0}};      /* DRFE_TRACE_LINES: which certification failed (stopping rule | close values | sign | subnormal regime) */
    bool deviceOrder = true;            /* the ordering by k_lsd_order (default) or by the pool (DRFE_LSD_HOST_ORDER=1: A/B, tests) */
    hipEvent_t stageEv[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   /* DRFE_TRACE_LINES, chunk 0: start | upload | passes | keys | ordering | growth */
    hipEvent_t clk[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   /* drfe_long_kernel_clock, chunk 0: start | upload | passes | k_lsd_keys | k_lsd_order | k_lsd_grow | k_rect_improve | key lines + LBD */
    std::atomic<long> usSort{0}, usFinish{0}, usWait{0}, usNfa{0}, usKeyl{0}, usRectDl{0}, usCountsWall{0}, usCountsCpu{0}, nCountRounds{0}, handedBack{0};   /* DRFE_TRACE_LINES: task time by kind, summed over the workers */
    std::chrono::steady_clock::time_point t0, tLastSort, tFirstFinish; std::atomic<int> nFirst{0};
    BatchJob(int nChunks_) : sortedInChunk(nChunks_), chunkState(nChunks_, 0) {}
};

} // namespace

static int batch_launch_grow(BatchJob& J, int ch, std::string& err)
{
    LinesScratch* A = J.A;
    const int f0 = ch * J.chunk, nf = std::min(J.chunk, J.nframes - f0);
    const size_t ns = (size_t)A->sw * A->sh, nk = (size_t)(A->sw - 1) * (A->sh - 1);
    hipStream_t st = J.chunkStream[ch];
    for (int f = f0; f < f0 + nf; f++) {
        LsdGrowFrame& g = A->h_frames[f];
        g.ang = A->d_angles + ns * f; g.cs = A->d_cs + ns * f; g.cs0 = A->d_cs0 + ns * f; g.mod = A->d_modgrad + ns * f;
        g.order = A->d_order + nk * f; g.reg = A->d_reg + ns * f; g.tmp = A->d_tmp + ns * f;
        g.notdef = A->d_notdef + ((ns + 31) / 32) * f;
        g.regMw = A->d_regMw + 4 * (size_t)A->regCapMw * f; g.tmpMw = A->d_tmpMw + 4 * (size_t)A->regCapMw * f;
        g.gbm = A->d_gbmMw + ((ns + 31) / 32) * f;
        g.rects = A->d_rects + (size_t)A->rectCap * f; g.out = A->d_out + DRFE_LSD_OUT_INTS * (size_t)f;
        g.nOrder = (int)nk;
        g.meta = J.deviceOrder ? A->d_meta + 2 * (size_t)f : nullptr;
        g.minSeedBin = J.deviceOrder ? 0u : 1024u - (uint32_t)(A->h_meta[2 * (size_t)f + 1] & 0xFFFFFFFFull);
    }
#define BCHK(call)                                                                              \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e__); return DRFE_ERR_HIP; } \
    } while (0)
    if (J.deviceOrder) {
        /* std::sort's permutation on the device, in place; the member-list arrays serve as its scratch (the growth that
         * follows on the same stream overwrites them) */
        BCHK(drfe_launch_lsd_order(A->d_order + nk * f0, nk, (int)nk, A->d_reg + ns * f0, A->d_tmp + ns * f0, ns, A->d_ordStatus + f0, 1, nf, st));
        if (ch == 0 && J.stageEv[4]) (void)hipEventRecord(J.stageEv[4], st);
        if (ch == 0 && J.clk[4]) (void)hipEventRecord(J.clk[4], st);
    } else
        BCHK(hipMemcpyAsync(A->d_order + nk * f0, A->h_order + nk * f0, nk * 4 * nf, hipMemcpyHostToDevice, st));
    BCHK(hipMemcpyAsync(A->d_frames + f0, A->h_frames + f0, sizeof(LsdGrowFrame) * nf, hipMemcpyHostToDevice, st));
    /* Region growing: one wavefront per frame (k_lsd_grow: the least wave-time per frame - what counts when calls of hundreds of
     * frames run side by side and the device is full) or four (k_lsd_grow_mw: speculation with in-order commit, 75 -> 45 ms per
     * frame - what counts when the call cannot fill the device by itself).  Measured on one MI355X (profiles/r05a_path_saturation.txt):
     * one 512-frame call at a time 6 100 frames/s with four waves against 4 600, five calls in flight 7 700 against 8 900.
     * Default: four waves up to 256 frames per call (fewer than one wavefront per CU each); DRFE_LSD_GROW_WAVES=1 / 4 forces one. */
    static const int growWavesEnv = [] { const char* e = std::getenv("DRFE_LSD_GROW_WAVES"); return e ? std::atoi(e) : 0; }();
    const int growMode = J.c->lsdDeviceGrow;          /* drfe_lsd_configure: 2 / 3 force a kernel */
    const bool growMw = growMode == 3 || growWavesEnv == 4 || (growMode != 2 && growWavesEnv != 1 && J.nframes <= 256);
    BCHK(drfe_launch_lsd_grow(A->d_frames + f0, nf, A->sw, A->sh, J.prec, J.p, J.minReg, 0.7, A->rectCap, st, growMw ? A->regCapMw : 0));
    if (ch == 0 && J.clk[5]) (void)hipEventRecord(J.clk[5], st);
    /* rect_improve + the NFA decisions of every accepted rectangle, behind the growth on the same stream: no host round trip */
    if (J.deviceNfa)
        BCHK(drfe_launch_rect_improve(A->d_frames + f0, nf, A->sw, A->sh, lsd_walk_mode(J.rectMode), J.nfaTab, A->rectCap, A->d_segs + (size_t)A->rectCap * f0, st));
    if (ch == 0 && J.clk[6] && J.deviceNfa) (void)hipEventRecord(J.clk[6], st);
    if (J.deviceNfa && J.deviceKl) {
        const size_t k0 = (size_t)A->klCap * f0, px = (size_t)A->w * A->h;
        BCHK(drfe_launch_lsd_keylines(A->d_frames + f0, A->d_segs + (size_t)A->rectCap * f0, A->rectCap, nf, A->w, A->h, J.maxLines, A->klCap,
                                      A->d_kl + k0, A->d_klLineF + 3 * k0, A->d_klLbd + k0, A->d_klOut + 4 * (size_t)f0, st));
        BCHK(drfe_launch_lbd_batch(A->d_klLbd + k0, A->d_klOut + 4 * (size_t)f0, A->klCap, nf, A->d_gx + px * f0, A->d_gy + px * f0, A->w, A->h, lbdTables(),
                                   A->d_klDesc + 32 * k0, st));
        if (ch == 0 && J.clk[7]) (void)hipEventRecord(J.clk[7], st);
    }
    /* no download behind the growth: a copy queued on a DMA ring waits there for its kernel and holds up every other stream's
     * copies behind it (measured: the plane path's kernels and CAPE's transfers stalled for the whole growth); the worker that
     * sees the event fetches the chunk's status words */
    if (ch == 0 && J.stageEv[5]) (void)hipEventRecord(J.stageEv[5], st);
    BCHK(hipEventRecord(J.growDone[ch], st));
#undef BCHK
    return DRFE_OK;
}

static void batch_fail(BatchJob& J, int rc, const std::string& err)
{
    std::lock_guard<std::mutex> lk(J.mu);
    if (J.firstRc == DRFE_OK) { J.firstRc = rc; J.firstErr = err; }
    J.abort = true;
    J.cv.notify_all();
}

static void batch_worker(BatchJob& J, LineWorker* lw)
{
    LinesScratch* A = J.A;
    const size_t ns = (size_t)A->sw * A->sh, n = (size_t)A->w * A->h, nk = (size_t)(A->sw - 1) * (A->sh - 1);
    (void)hipSetDevice(J.c->device);
    std::vector<OPt> tmp;
    const RectValidator val(A->sw, A->sh, J.rectMode);
    for (;;) {
        int f = -1, waitCh = -1, fetchCh = -1; bool fin = false;
        {
            std::unique_lock<std::mutex> lk(J.mu);
            for (;;) {
                if (J.abort) return;
                /* chunks whose regions have grown: this worker fetches their status words (outside the lock), then their frames
                 * become validation tasks */
                for (int ch = 0; ch < J.nChunks && fetchCh < 0; ch++)
                    if (J.chunkState[ch] == 1 && hipEventQuery(J.growDone[ch]) == hipSuccess) { J.chunkState[ch] = 3; fetchCh = ch; }
                if (fetchCh >= 0) break;
                if (!J.finishQ.empty()) { f = J.finishQ.front(); J.finishQ.pop_front(); fin = true; break; }
                if (!J.sortQ.empty()) { f = J.sortQ.front(); J.sortQ.pop_front(); break; }
                if (J.pendingFinish == 0) return;
                /* nothing to do until a chunk has grown: sleep on the oldest one (outside the lock), or on the other workers */
                for (int ch = 0; ch < J.nChunks && waitCh < 0; ch++) if (J.chunkState[ch] == 1) waitCh = ch;
                if (waitCh >= 0) break;
                J.cv.wait(lk);
            }
        }
        if (fetchCh >= 0) {
            const int f0 = fetchCh * J.chunk, nf = std::min(J.chunk, J.nframes - f0);
            hipError_t e = hipMemcpyAsync(A->h_out + DRFE_LSD_OUT_INTS * (size_t)f0, A->d_out + DRFE_LSD_OUT_INTS * (size_t)f0, DRFE_LSD_OUT_INTS * sizeof(int) * nf,
                                          hipMemcpyDeviceToHost, lw->stream);
            if (e == hipSuccess && J.deviceOrder) e = hipMemcpyAsync(A->h_ordStatus + f0, A->d_ordStatus + f0, sizeof(int) * nf, hipMemcpyDeviceToHost, lw->stream);
            if (e == hipSuccess && J.deviceNfa && J.deviceKl) {
                /* the chunk's finished key lines, descriptors and line equations: four copies for all of its frames */
                const size_t k0 = (size_t)A->klCap * f0, kn = (size_t)A->klCap * nf;
                e = hipMemcpyAsync(A->h_klOut + 4 * (size_t)f0, A->d_klOut + 4 * (size_t)f0, 4 * sizeof(int) * nf, hipMemcpyDeviceToHost, lw->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(A->h_kl + k0, A->d_kl + k0, kn * sizeof(drfe_keyline), hipMemcpyDeviceToHost, lw->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(A->h_klLineF + 3 * k0, A->d_klLineF + 3 * k0, kn * 3 * sizeof(double), hipMemcpyDeviceToHost, lw->stream);
                if (e == hipSuccess) e = hipMemcpyAsync(A->h_klDesc + 32 * k0, A->d_klDesc + 32 * k0, kn * 32, hipMemcpyDeviceToHost, lw->stream);
            }
            if (e == hipSuccess) e = lane_sync(lw);
            if (e != hipSuccess) { batch_fail(J, DRFE_ERR_HIP, "lsd_extract_batch: status words of a chunk"); return; }
            std::lock_guard<std::mutex> lk(J.mu);
            J.chunkState[fetchCh] = 2;
            for (int k = 0; k < nf; k++) J.finishQ.push_back(f0 + k);
            J.cv.notify_all();
            continue;
        }
        if (waitCh >= 0) {
            const auto tw = std::chrono::steady_clock::now();
            if (drfe_event_wait_sleeping(J.growDone[waitCh]) != hipSuccess) { batch_fail(J, DRFE_ERR_HIP, "lsd_extract_batch: grow"); return; }
            J.usWait += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tw).count();
            continue;
        }
        const auto tTask = std::chrono::steady_clock::now();
        const int ch = f / J.chunk;
        if (!fin) {
            /* the ordering of frame f: std::sort's permutation of its keys */
            if (hipEventSynchronize(J.keysReady[ch]) != hipSuccess) { batch_fail(J, DRFE_ERR_HIP, "lsd_extract_batch: keys"); return; }
            OPt* keys = A->h_order + nk * f;
            const uint32_t minSeedBin = 1024u - (uint32_t)(A->h_meta[2 * (size_t)f + 1] & 0xFFFFFFFFull);
            static const bool stdSort = std::getenv("DRFE_LSD_STD_SORT") != nullptr;
            if (stdSort) std::sort(keys, keys + nk, lsd_order::Before());
            else lsd_order::sort(keys, nk, tmp, -1, -1, minSeedBin);
            J.usSort += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tTask).count();
            const int nf = std::min(J.chunk, J.nframes - ch * J.chunk);
            if (J.sortedInChunk[ch].fetch_add(1) + 1 == nf) {
                std::string err;
                const int rc = batch_launch_grow(J, ch, err);
                if (rc != DRFE_OK) { batch_fail(J, rc, err); return; }
                std::lock_guard<std::mutex> lk(J.mu);
                J.chunkState[ch] = 1;
                J.cv.notify_all();
            }
            continue;
        }
        /* validation + key lines + descriptors of frame f (its chunk has grown) */
        int rc = ensure_lines(lw->err, lw->ls, J.w, J.h, 1, false, false);
        int nd = 0;
        if (rc == DRFE_OK) {
            const int nRects = A->h_out[DRFE_LSD_OUT_INTS * (size_t)f];
            const int status = A->h_out[DRFE_LSD_OUT_INTS * (size_t)f + 1] | (J.deviceOrder ? A->h_ordStatus[f] << 8 : 0);
            drfe_keyline* lo = J.lines ? J.lines + (size_t)f * J.cap : nullptr;
            uint8_t* dout = J.ldesc ? J.ldesc + (size_t)f * J.cap * 32 : nullptr;
            double* lf = J.lineF ? J.lineF + (size_t)f * J.cap * 3 : nullptr;
            if (status != 0) {
                J.handedBack++;
                /* a rounding the device could not certify, or more regions than the rectangle list holds: this frame's
                 * sequential half again on the host, from the fields the device still has */
                rc = ensure_lines(lw->err, lw->ls, J.w, J.h, 1, false, false);
                if (rc == DRFE_OK) rc = host_grow_and_finish(lw, A, f, J.maxLines, lo, dout, lf, J.cap, &J.nLines[f], &nd, std::chrono::steady_clock::now());
            } else if (J.deviceNfa && J.deviceKl && A->h_klOut[4 * (size_t)f + 2] == 0) {
                /* everything happened on the device: copy the frame's results out of the chunk's pinned mirrors */
                const int nl = A->h_klOut[4 * (size_t)f];
                nd = A->h_klOut[4 * (size_t)f + 1];
                J.nLines[f] = nl;
                if (nl > J.cap) { lw->err = "lsd_extract: line buffer too small"; rc = DRFE_ERR_CAPACITY; }
                else {
                    const size_t k0 = (size_t)A->klCap * f;
                    if (lo) std::memcpy(lo, A->h_kl + k0, sizeof(drfe_keyline) * nl);
                    if (dout) std::memcpy(dout, A->h_klDesc + 32 * k0, (size_t)nl * 32);
                    if (lf) std::memcpy(lf, A->h_klLineF + 3 * k0, (size_t)nl * 3 * sizeof(double));
                }
            } else if (J.deviceNfa && A->h_out[DRFE_LSD_OUT_INTS * (size_t)f + 2] == 0) {
                if (J.deviceKl) J.klToHost++;
                /* the device validated the rectangles (k_rect_improve): fetch the segments, keep the accepted ones in seed order */
                LsdSegOut* hs = A->h_segs + (size_t)A->rectCap * f;
                if (nRects > 0 && hipMemcpyAsync(hs, A->d_segs + (size_t)A->rectCap * f, sizeof(LsdSegOut) * nRects, hipMemcpyDeviceToHost, lw->stream) != hipSuccess) rc = DRFE_ERR_HIP;
                if (rc == DRFE_OK && nRects > 0 && lane_sync(lw) != hipSuccess) rc = DRFE_ERR_HIP;
                if (rc != DRFE_OK) lw->err = "lsd_extract_batch: segment download";
                if (rc == DRFE_OK) {
                    std::vector<float> segs;
                    segs.reserve((size_t)nRects * 4);
                    for (int i = 0; i < nRects; i++)
                        if (hs[i].flag) { segs.push_back(hs[i].x1); segs.push_back(hs[i].y1); segs.push_back(hs[i].x2); segs.push_back(hs[i].y2); }
                    const FrameView v = {A->w, A->h, A->sw, A->sh, A->d_angles + ns * f, A->d_gx + n * f, A->d_gy + n * f};
                    const auto tk = std::chrono::steady_clock::now();
                    J.usRectDl += std::chrono::duration_cast<std::chrono::microseconds>(tk - tTask).count();
                    rc = keylines_and_descriptors(lw, v, segs, J.maxLines, lo, dout, lf, J.cap, &J.nLines[f], &nd);
                    J.usKeyl += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tk).count();
                }
            } else {
                if (J.deviceNfa) { J.nfaToHost++; J.nfaWhy |= A->h_out[DRFE_LSD_OUT_INTS * (size_t)f + 3]; for (int b = 0; b < 8; b++) if (A->h_out[DRFE_LSD_OUT_INTS * (size_t)f + 3] & (1 << b)) J.nfaWhyCount[b]++; }
                std::vector<RectD> pending(nRects);
                /* through the arena's pinned mirror: a download into pageable memory stages inside the copy call */
                LsdRect* hr = A->h_rects + (size_t)A->rectCap * f;
                if (nRects > 0 && hipMemcpyAsync(hr, A->d_rects + (size_t)A->rectCap * f, sizeof(LsdRect) * nRects, hipMemcpyDeviceToHost, lw->stream) != hipSuccess) rc = DRFE_ERR_HIP;
                if (rc == DRFE_OK && nRects > 0 && lane_sync(lw) != hipSuccess) rc = DRFE_ERR_HIP;
                if (rc == DRFE_OK && nRects > 0) std::memcpy(pending.data(), hr, sizeof(LsdRect) * nRects);
                if (rc != DRFE_OK) lw->err = "lsd_extract_batch: rectangle download";
                if (rc == DRFE_OK) {
                    const FrameView v = {A->w, A->h, A->sw, A->sh, A->d_angles + ns * f, A->d_gx + n * f, A->d_gy + n * f};
                    int countRc = DRFE_OK;
                    std::vector<float> segs;
                    const auto tn = std::chrono::steady_clock::now();
                    J.usRectDl += std::chrono::duration_cast<std::chrono::microseconds>(tn - tTask).count();
                    g_countsWallUs = g_countsCpuUs = 0;
                    const bool okE = val.emit(pending, segs, device_counts(lw, v, countRc, nullptr));
                    const auto tk = std::chrono::steady_clock::now();
                    J.usCountsWall += (long)g_countsWallUs; J.usCountsCpu += (long)g_countsCpuUs;
                    J.usNfa += std::chrono::duration_cast<std::chrono::microseconds>(tk - tn).count();
                    if (!okE) rc = countRc;
                    else rc = keylines_and_descriptors(lw, v, segs, J.maxLines, lo, dout, lf, J.cap, &J.nLines[f], &nd);
                    J.usKeyl += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tk).count();
                }
            }
        }
        if (J.nDetected) J.nDetected[f] = nd;
        J.usFinish += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tTask).count();
        if (rc != DRFE_OK) { batch_fail(J, rc, lw->err); return; }
        {
            std::lock_guard<std::mutex> lk(J.mu);
            if (--J.pendingFinish == 0) J.cv.notify_all();
        }
    }
}

/* 1 if k_lsd_grow can take frames of this size: its `used` bitmap + member ring must fit the LDS of a CU, coordinates the
 * 11-bit fields of the ordering keys */
static bool device_grow_fits(int w, int h)
{
    const int sw = (int)std::rint(w * 0.8), sh = (int)std::rint(h * 0.8);
    return sw <= 2048 && sh <= 2048 && sw >= 8 && sh >= 8 && drfe_lsd_grow_lds_bytes(sw, sh) <= 160 * 1024;
}

static int lsd_extract_batch_device(drfe_ctx* c, std::vector<LineWorker>* pool, int T, const uint8_t* gray, size_t frame_stride, int w, int h,
                                    size_t stride, int nframes, int max_lines, drfe_keyline* lines, uint8_t* ldesc, double* line_f, int cap,
                                    int* n_lines, int* n_detected)
{
    const auto tBegin = std::chrono::steady_clock::now();
    DrfeRange range("drfe:lines batch (upload, image passes, order, grow, rect_improve, keylines, LBD)");
    int rc = ensure_lines(c->err, c->lsBatch, w, h, nframes, true, true);
    if (rc != DRFE_OK) return rc;
    LinesScratch* A = c->lsBatch;
    static const LsdParams P;
    /* at most four chunks: one hardware queue each (streams that share a queue run one behind the other) */
    /* chunks (= low-priority streams = hardware queues) of this call: the runtime has four queues per priority, and the line and the
     * plane batch of a front-end step run side by side - two each (DRFE_BATCH_CHUNKS overrides) */
    static const int nch = [] { const char* e = std::getenv("DRFE_BATCH_CHUNKS"); const int v = e ? std::atoi(e) : 1; return v < 1 ? 1 : v > 16 ? 16 : v; }();
    const int chunk = std::max(1, std::min(nframes, std::max(16, (nframes + nch - 1) / nch)));
    const int nChunks = (nframes + chunk - 1) / chunk;
    BatchJob J(nChunks);
    J.c = c; J.A = A; J.pool = pool; J.gray = gray; J.frameStride = frame_stride; J.stride = stride;
    J.w = w; J.h = h; J.nframes = nframes; J.maxLines = max_lines; J.cap = cap;
    J.lines = lines; J.ldesc = ldesc; J.lineF = line_f; J.nLines = n_lines; J.nDetected = n_detected;
    J.chunk = chunk; J.nChunks = nChunks;
    J.pendingFinish = nframes;
    J.deviceOrder = std::getenv("DRFE_LSD_HOST_ORDER") == nullptr;
    if (!J.deviceOrder && !A->h_order)           /* the host-ordering experiment's pinned key mirror (0.4 GB per 512 frames): on demand */
        HIPCHK(c, hipHostMalloc((void**)&A->h_order, (size_t)(A->sw - 1) * (A->sh - 1) * (size_t)A->frames * 4, hipHostMallocDefault));
    J.rectMode = c->lsdRectMode;
    const RectValidator val(A->sw, A->sh, J.rectMode);
    J.prec = M_PI * 22.5 / 180; J.p = 22.5 / 180; J.minReg = (int)val.minReg(J.p);
    J.deviceNfa = c->lsdDeviceNfa && std::getenv("DRFE_LSD_HOST_NFA") == nullptr;
    J.deviceKl = J.deviceNfa && std::getenv("DRFE_LSD_HOST_KEYLINES") == nullptr;
    if (J.deviceKl && (A->klCap < max_lines || A->klCap == 0)) {
        void* dp[] = {A->d_kl, A->d_klLineF, A->d_klLbd, A->d_klDesc, A->d_klOut};
        for (void* q : dp) if (q) (void)hipFree(q);
        void* hp[] = {A->h_kl, A->h_klLineF, A->h_klDesc, A->h_klOut};
        for (void* q : hp) if (q) (void)hipHostFree(q);
        A->d_kl = nullptr; A->d_klLineF = nullptr; A->d_klLbd = nullptr; A->d_klDesc = nullptr; A->d_klOut = nullptr;
        A->h_kl = nullptr; A->h_klLineF = nullptr; A->h_klDesc = nullptr; A->h_klOut = nullptr;
        A->klCap = 0;                 /* committed below, once every buffer exists: a failed allocation leaves "no buffers" behind */
        const size_t kn = (size_t)max_lines * A->frames;
        HIPCHK(c, hipMalloc((void**)&A->d_kl, kn * sizeof(drfe_keyline)));
        HIPCHK(c, hipMalloc((void**)&A->d_klLineF, kn * 3 * sizeof(double)));
        HIPCHK(c, hipMalloc((void**)&A->d_klLbd, kn * sizeof(LbdLine)));
        HIPCHK(c, hipMalloc((void**)&A->d_klDesc, kn * 32));
        HIPCHK(c, hipMalloc((void**)&A->d_klOut, (size_t)A->frames * 4 * sizeof(int)));
        HIPCHK(c, hipHostMalloc((void**)&A->h_kl, kn * sizeof(drfe_keyline), hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void**)&A->h_klLineF, kn * 3 * sizeof(double), hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void**)&A->h_klDesc, kn * 32, hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void**)&A->h_klOut, (size_t)A->frames * 4 * sizeof(int), hipHostMallocDefault));
        A->klCap = max_lines;
    }
    if (J.deviceNfa) {
        std::vector<double> lg;
        const bool need = A->lgammaN != A->sw * A->sh + 2;
        val.fillTables(J.p, A->sw, A->sh, J.nfaTab, need ? &lg : nullptr);
        if (need) {                                  /* once per arena: 1.5 MB at 640 x 480 */
            HIPCHK(c, hipMemcpy(A->d_lgamma, lg.data(), lg.size() * sizeof(double), hipMemcpyHostToDevice));
            A->lgammaN = (int)lg.size();
        }
        J.nfaTab.lgamma = A->d_lgamma;
    }
    for (int ch = 0; ch < nChunks; ch++) J.sortedInChunk[ch].store(0);
    J.chunkStream.resize(nChunks); J.keysReady.resize(nChunks); J.growDone.resize(nChunks);
    /* one stream per chunk: kernels of different chunks overlap, the work of one chunk stays ordered.  LOW priority: the
     * runtime keeps a pool of hardware queues per priority, so the lanes' normal-priority streams (NFA counts, descriptors:
     * microsecond kernels a host thread waits for) never queue behind a chunk that grows for a hundred milliseconds */
    int prLow = 0, prHigh = 0;
    HIPCHK(c, hipDeviceGetStreamPriorityRange(&prLow, &prHigh));
    if (std::getenv("DRFE_TRACE_LINES")) for (hipEvent_t& e : J.stageEv) HIPCHK(c, hipEventCreate(&e));
    if (c->longClock) for (hipEvent_t& e : J.clk) HIPCHK(c, hipEventCreate(&e));
    for (int ch = 0; ch < nChunks; ch++) {
        HIPCHK(c, drfe_long_kernel_stream(&J.chunkStream[ch], 0));
        HIPCHK(c, hipEventCreateWithFlags(&J.keysReady[ch], hipEventDisableTiming | hipEventBlockingSync));
        HIPCHK(c, hipEventCreateWithFlags(&J.growDone[ch], hipEventDisableTiming | hipEventBlockingSync));
    }
    const size_t n = (size_t)w * h, ns = (size_t)A->sw * A->sh, nk = (size_t)(A->sw - 1) * (A->sh - 1);
    int launchRc = DRFE_OK;
    for (int ch = 0; ch < nChunks && launchRc == DRFE_OK; ch++) {
        const int f0 = ch * chunk, nf = std::min(chunk, nframes - f0);
        hipStream_t st = J.chunkStream[ch];
        hipError_t e = hipSuccess;
        const bool tr = ch == 0 && J.stageEv[0];
        const bool ck = ch == 0 && J.clk[0];
        if (tr) (void)hipEventRecord(J.stageEv[0], st);
        if (ck) (void)hipEventRecord(J.clk[0], st);
        if (stride == (size_t)w && frame_stride == n) e = hipMemcpyAsync(A->d_img + n * f0, gray + frame_stride * f0, n * nf, hipMemcpyHostToDevice, st);
        else
            for (int f = f0; f < f0 + nf && e == hipSuccess; f++)
                e = hipMemcpy2DAsync(A->d_img + n * f, (size_t)w, gray + frame_stride * f, stride, (size_t)w, (size_t)h, hipMemcpyHostToDevice, st);
        if (tr) (void)hipEventRecord(J.stageEv[1], st);
        if (ck) (void)hipEventRecord(J.clk[1], st);
        if (e == hipSuccess) e = drfe_launch_lines_passes(A->d_img + n * f0, w, h, P.lsdTaps, P.lbdTaps, A, f0, nf, P.rho, st);
        if (tr) (void)hipEventRecord(J.stageEv[2], st);
        if (ck) (void)hipEventRecord(J.clk[2], st);
        if (e == hipSuccess) e = drfe_launch_lsd_keys(A->d_modgrad + ns * f0, A->d_angles + ns * f0, A->sw, A->sh, A->d_meta + 2 * (size_t)f0, A->d_order + nk * f0, A->d_cs0 + ns * f0, A->d_notdef + ((ns + 31) / 32) * f0, nf, st);
        if (tr) (void)hipEventRecord(J.stageEv[3], st);
        if (ck) (void)hipEventRecord(J.clk[3], st);
        if (e == hipSuccess && !J.deviceOrder) {
            e = hipMemcpyAsync(A->h_order + nk * f0, A->d_order + nk * f0, nk * 4 * nf, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipMemcpyAsync(A->h_meta + 2 * (size_t)f0, A->d_meta + 2 * (size_t)f0, 16 * (size_t)nf, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipEventRecord(J.keysReady[ch], st);
        }
        if (e != hipSuccess) { c->err = std::string("lsd_extract_batch: image passes: ") + hipGetErrorString(e); launchRc = DRFE_ERR_HIP; }
        if (launchRc == DRFE_OK && J.deviceOrder) {
            /* ordering and growth follow on the same stream: nothing returns to the host before the rectangles */
            std::string err;
            launchRc = batch_launch_grow(J, ch, err);
            if (launchRc != DRFE_OK) c->err = err;
            else J.chunkState[ch] = 1;
        }
    }
    const auto tLaunched = std::chrono::steady_clock::now();
    if (launchRc == DRFE_OK) {
        if (!J.deviceOrder) for (int f = 0; f < nframes; f++) J.sortQ.push_back(f);
        std::vector<std::thread> th;
        th.reserve(T);
        for (int k = 0; k < T; k++) th.emplace_back([&J, pool, k]() { DrfePoolCpuScope cpu(0); batch_worker(J, &(*pool)[k]); });
        for (std::thread& t : th) t.join();
    }
    if (J.stageEv[0] && J.deviceOrder && launchRc == DRFE_OK) {
        float ms[5] = {0, 0, 0, 0, 0};
        for (int k = 0; k < 5; k++) (void)hipEventElapsedTime(&ms[k], J.stageEv[k], J.stageEv[k + 1]);
        std::fprintf(stderr, "drfe_lsd_extract_batch stages on the device (chunk 0, waiting for resources included): upload %.1f ms, image passes %.1f, k_lsd_keys %.1f, k_lsd_order %.1f, k_lsd_grow %.1f\n",
                     ms[0], ms[1], ms[2], ms[3], ms[4]);
    }
    for (hipEvent_t& e : J.stageEv) if (e) (void)hipEventDestroy(e);
    if (J.clk[0]) {
        /* the workers have fetched every chunk's results: chunk 0's stream is idle.  Intervals whose closing event was never recorded
         * (ordering / NFA / key lines on the host) stay 0 */
        if (launchRc == DRFE_OK && J.deviceOrder) {
            (void)hipStreamSynchronize(J.chunkStream[0]);
            const int last = J.deviceNfa ? (J.deviceKl ? 7 : 6) : 5;
            for (int k = 0; k < 7; k++) { float ms = 0; if (k < last) (void)hipEventElapsedTime(&ms, J.clk[k], J.clk[k + 1]); c->longMs[k] = ms; }
        }
        for (hipEvent_t& e : J.clk) if (e) (void)hipEventDestroy(e);
    }
    c->lsdStats[0] += nframes; c->lsdStats[1] += J.handedBack.load(); c->lsdStats[2] += J.nfaToHost.load(); c->lsdStats[3] += J.klToHost.load();
    if (std::getenv("DRFE_TRACE_LINES"))
        std::fprintf(stderr, "drfe_lsd_extract_batch: rect_improve / NFA %s; %ld of %d frames back to the host's validation (a decision too close to certify)\n",
                     J.deviceNfa ? "on the device (k_rect_improve)" : "on the host pool", J.nfaToHost.load(), nframes);
    if (std::getenv("DRFE_TRACE_LINES") && J.nfaToHost.load())
        std::fprintf(stderr, "drfe_lsd_extract_batch: uncertified frames by kind: stopping rule %ld, subnormal regime %ld, hypothesis queue full %ld, no outcome %ld, outcomes differ %ld, too many hypotheses in a stage %ld\n",
                     J.nfaWhyCount[0].load(), J.nfaWhyCount[3].load(), J.nfaWhyCount[4].load(), J.nfaWhyCount[5].load(), J.nfaWhyCount[6].load(), J.nfaWhyCount[7].load());
    if (std::getenv("DRFE_TRACE_LINES"))
        std::fprintf(stderr, "drfe_lsd_extract_batch (device grow): %d frames, %d chunks of %d, %d threads: enqueue %.1f ms, total %.1f ms; per frame: ordering %.2f ms, rect download %.2f, NFA rounds %.2f (of which in the counting round trips: %.2f wall, %.2f CPU), key lines + LBD %.2f (validation task %.2f); workers slept %.1f ms each waiting for the device; %ld frames redone on the host\n",
                     nframes, nChunks, chunk, T, std::chrono::duration<double, std::milli>(tLaunched - tBegin).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tBegin).count(), J.usSort / 1e3 / nframes,
                     J.usRectDl / 1e3 / nframes, J.usNfa / 1e3 / nframes, J.usCountsWall / 1e3 / nframes, J.usCountsCpu / 1e3 / nframes, J.usKeyl / 1e3 / nframes, J.usFinish / 1e3 / nframes, J.usWait / 1e3 / T, J.handedBack.load());
    if (std::getenv("DRFE_NFA_PROFILE")) {       /* -DNFA_PROFILE build of k_rect_improve: phase clocks (100 MHz) of block 0 of frame 0, lane 0 */
        const unsigned long long* pr = (const unsigned long long*)(A->h_out + 24);
        std::fprintf(stderr, "k_rect_improve block 0 (lane 0): candidate + walk set-up %.3f ms, pixel walk %.3f, nfa %.3f, selection %.3f, all %.3f\n", pr[0] / 1e5, pr[1] / 1e5, pr[2] / 1e5, pr[3] / 1e5, pr[4] / 1e5);
    }
    if (std::getenv("DRFE_LSD_PROFILE") && std::getenv("DRFE_LSD_GROW_WAVES") && std::getenv("DRFE_LSD_GROW_WAVES")[0] == '4') {
        /* LSD_PROFILE builds of k_lsd_grow_mw, frame 0: times summed over its four wavefronts */
        const unsigned long long* pr = (const unsigned long long*)(A->h_out + 4);
        std::fprintf(stderr, "k_lsd_grow_mw frame 0: %.2f ms wall; summed over 4 waves: drain %.2f ms (of it run-at-head %.2f: %llu deferred, %llu conflicts, %llu direct-mode), scanning %.2f, take incl. scanning and waits %.2f, "
                     "small speculation %.2f (%llu), big speculation %.2f (%llu, %llu dropped), waiting for a table %.2f, (unused %.2f), idle at the end %.2f\n",
                     pr[14] / 1e5, pr[0] / 1e5, pr[12] / 1e5, pr[10], pr[11], pr[13], pr[1] / 1e5, pr[2] / 1e5, pr[3] / 1e5, pr[8], pr[4] / 1e5, pr[9], pr[15], pr[5] / 1e5, pr[6] / 1e5, pr[7] / 1e5);
    } else if (std::getenv("DRFE_LSD_PROFILE")) {       /* LSD_PROFILE builds of k_lsd_grow: phase times of frame 0 (100 MHz ticks -> ms) and counts */
        const unsigned long long* pr = (const unsigned long long*)(A->h_out + 4);
        std::fprintf(stderr, "k_lsd_grow frame 0: total %.2f ms: bitmap %.2f, scan %.2f (%llu chunks), window loads %.2f (%llu groups), window growth %.2f (%llu regions, %llu member visits), "
                     "queue growth %.2f (%llu steps), region2rect %.2f (%llu), refine incl. its growth %.2f; alignment tests decided by the reference's arithmetic %llu; of the queue growth, waiting for the members' fields %.2f\n", pr[7] / 1e5, pr[0] / 1e5, pr[1] / 1e5, pr[8], pr[2] / 1e5, pr[9], pr[3] / 1e5, pr[10], pr[11],
                     pr[4] / 1e5, pr[12], pr[5] / 1e5, pr[13], pr[6] / 1e5, pr[14], pr[15] / 1e5);
    }
    for (int ch = 0; ch < nChunks; ch++) {
        (void)hipStreamSynchronize(J.chunkStream[ch]);
        (void)hipStreamDestroy(J.chunkStream[ch]);
        (void)hipEventDestroy(J.keysReady[ch]); (void)hipEventDestroy(J.growDone[ch]);
    }
    if (launchRc != DRFE_OK) return launchRc;
    if (J.firstRc != DRFE_OK) { c->err = J.firstErr; return J.firstRc; }
    return DRFE_OK;
}

extern "C" {

int drfe_lsd_extract(drfe_ctx* c, const uint8_t* gray, int w, int h, size_t stride, int max_lines, drfe_keyline* lines,
                     uint8_t* ldesc, double* line_f, int cap, int* n_lines, int* n_detected)
{
    if (!c || !gray || !n_lines || w < 16 || h < 16 || stride < (size_t)w || max_lines < 1) {
        if (c) c->err = "lsd_extract: invalid argument";
        return DRFE_ERR_INVALID;
    }
    LineWorker lw;
    lw.ls = c->ls;
    lw.stream = c->stream;
    lw.host = static_cast<LineHost*>(c->lineHost);
    lw.rectMode = c->lsdRectMode;
    const int rc = lsd_extract_core(&lw, c->device, gray, w, h, stride, max_lines, lines, ldesc, line_f, cap, n_lines, n_detected);
    c->ls = lw.ls;
    c->lineHost = lw.host;
    if (rc != DRFE_OK) c->err = lw.err;
    return rc;
}

/* where drfe_lsd_extract_batch grows its regions: 0 on the pool's host threads (the path of drfe_lsd_extract, frame by frame);
 * on the device: 1 (default) the kernel chosen by the size of the call, 2 one wavefront per frame (k_lsd_grow), 3 four wavefronts
 * per frame (k_lsd_grow_mw: speculation with in-order commit).  Results are identical. */
int drfe_lsd_configure(drfe_ctx* c, int device_grow)
{
    if (!c || device_grow < 0 || device_grow > 3) { if (c) c->err = "lsd_configure: invalid argument"; return DRFE_ERR_INVALID; }
    c->lsdDeviceGrow = device_grow;
    return DRFE_OK;
}

/* where drfe_lsd_extract_batch takes rect_improve's decisions: 1 (default) on the device (k_rect_improve: certified
 * comparisons, uncertain frames return to the host), 0 on the pool threads with the host's libm.  Results are identical. */
int drfe_lsd_configure_nfa(drfe_ctx* c, int device_nfa)
{
    if (!c || device_nfa < 0 || device_nfa > 1) { if (c) c->err = "lsd_configure_nfa: invalid argument"; return DRFE_ERR_INVALID; }
    c->lsdDeviceNfa = device_nfa;
    return DRFE_OK;
}

/* counters of this context's drfe_lsd_extract_batch calls since creation: [0] frames through the device path, [1] frames whose
 * region growing went back to the host (uncertified rounding, capacity), [2] frames whose NFA decisions went back to the host,
 * [3] frames whose key-line stage went back to the host (an atan2 / cos / sin rounding not certified, the sort's heap branch) */
/* 1: drfe_lsd_extract_batch and drfe_planes_ahc_post_batch bracket every kernel of their first chunk with HIP events on the stream
 * the kernels are launched on (two event records per kernel: nothing a throughput run notices, off by default) */
int drfe_long_kernel_clock(drfe_ctx* c, int on)
{
    if (!c) return DRFE_ERR_INVALID;
    c->longClock = on ? 1 : 0;
    return DRFE_OK;
}
/* ms of the last clocked call's first chunk.  Lines [0..6]: upload, image passes, k_lsd_keys (+ k_lsd_notdef), k_lsd_order, k_lsd_grow(_mw),
 * k_rect_improve, k_lsd_keylines + k_lbd.  Planes [8..14]: upload, k_ahc_blocks, k_ahc_cluster, k_ahc_refine, k_ahc_labels_* (three),
 * k_voxel_grid, k_plane_refit.  [7], [15]: 0.  An interval includes whatever its kernel waited for on the device: alone on the device
 * it is the kernel's duration, beside other work it is not */
int drfe_long_kernel_ms(drfe_ctx* c, float* out16)
{
    if (!c || !out16) return DRFE_ERR_INVALID;
    for (int i = 0; i < 16; i++) out16[i] = c->longMs[i];
    return DRFE_OK;
}

int drfe_lsd_stats(drfe_ctx* c, long long* out3 /* four entries */)
{
    if (!c || !out3) return DRFE_ERR_INVALID;
    for (int i = 0; i < 4; i++) out3[i] = c->lsdStats[i];
    return DRFE_OK;
}

/* Which reading of cv::LineSegmentDetectorImpl::rect_nfa (OpenCV 3.4 lsd.cpp, behind reference src/LSDextractor.cpp:14-17)
 * validates the rectangles: 0 (default) the literal source - `struct edge { cv::Point p; ... }`, so integer corners and
 * integer step quotients, and (y - tailp->p.x) in the second steps' guards AND denominators; 1 the real-valued reading
 * round 3 shipped (double quotients, (y - tailp->p.y) denominators).  SURVEY.md section 9: library bugs are preserved. */
int drfe_lsd_configure_rect(drfe_ctx* c, int rect_mode)
{
    if (!c || rect_mode < 0 || rect_mode > 2) { if (c) c->err = "lsd_configure_rect: invalid argument"; return DRFE_ERR_INVALID; }
    c->lsdRectMode = rect_mode;
    return DRFE_OK;
}

/* LineSegment::ExtractLineSegment for nframes host images at once.  Outputs are per frame: lines[f * cap ..],
 * ldesc[f * cap * 32 ..], line_f[f * cap * 3 ..], n_lines[f], n_detected[f].  n_threads <= 0: 1.25 threads per CPU. */
int drfe_lsd_extract_batch(drfe_ctx* c, const uint8_t* gray, size_t frame_stride, int w, int h, size_t stride, int nframes,
                           int max_lines, drfe_keyline* lines, uint8_t* ldesc, double* line_f, int cap, int* n_lines,
                           int* n_detected, int n_threads)
{
    if (!c || !gray || !n_lines || nframes < 0 || w < 16 || h < 16 || stride < (size_t)w || max_lines < 1 || cap < 1 ||
        frame_stride < stride * (size_t)h) {
        if (c) c->err = "lsd_extract_batch: invalid argument";
        return DRFE_ERR_INVALID;
    }
    if (nframes == 0) return DRFE_OK;
    /* default: 1.25 threads per CPU - a lane sleeps in stream synchronisations for about a fifth of a frame's time */
    int T = n_threads > 0 ? n_threads : std::max(1, drfe_default_host_threads() * 5 / 4);
    T = std::max(1, std::min(T, nframes));
    HIPCHK(c, hipSetDevice(c->device));
    auto* pool = static_cast<std::vector<LineWorker>*>(c->lineWorkers);
    if (!pool) { pool = new std::vector<LineWorker>(); c->lineWorkers = pool; }
    while ((int)pool->size() < T) {
        LineWorker lw;
        HIPCHK(c, hipStreamCreateWithFlags(&lw.stream, hipStreamNonBlocking));
        if (!std::getenv("DRFE_POOL_SPIN")) HIPCHK(c, hipEventCreateWithFlags(&lw.pollEv, hipEventDisableTiming));
        lw.ownsStream = true;
        pool->push_back(lw);
    }
    for (LineWorker& lw : *pool) lw.rectMode = c->lsdRectMode;
    static const bool envHost = std::getenv("DRFE_LSD_HOST_GROW") != nullptr;
    if (c->lsdDeviceGrow && !envHost && device_grow_fits(w, h))
        return lsd_extract_batch_device(c, pool, T, gray, frame_stride, w, h, stride, nframes, max_lines, lines, ldesc, line_f, cap, n_lines, n_detected);
    std::vector<int> rcs(T, DRFE_OK);
    std::vector<std::thread> th;
    th.reserve(T);
    std::atomic<int> next(0);
    for (int k = 0; k < T; k++)
        th.emplace_back([&, k]() {
            LineWorker* lw = &(*pool)[k];
            for (int f = next.fetch_add(1); f < nframes; f = next.fetch_add(1)) {
                int nd = 0;
                const int rc = lsd_extract_core(lw, c->device, gray + (size_t)f * frame_stride, w, h, stride, max_lines,
                                                lines ? lines + (size_t)f * cap : nullptr, ldesc ? ldesc + (size_t)f * cap * 32 : nullptr,
                                                line_f ? line_f + (size_t)f * cap * 3 : nullptr, cap, &n_lines[f], &nd);
                if (n_detected) n_detected[f] = nd;
                if (rc != DRFE_OK) { rcs[k] = rc; return; }
            }
        });
    for (std::thread& t : th) t.join();
    for (int k = 0; k < T; k++)
        if (rcs[k] != DRFE_OK) { c->err = (*pool)[k].err; return rcs[k]; }
    return DRFE_OK;
}

/* The sequential half of LSD on caller-supplied gradient fields, rectangle counting on the host as well: the host logic of
 * drfe_lsd_extract without a device (CPU tests and profiling of the ordering / region growing / fitting code).  modgrad,
 * angles: W x H doubles; cs: (cos, sin) of float(angle) per pixel; segs: up to cap x 4 floats (x1, y1, x2, y2 at input scale). */
int drfe_lsd_segments_host(const double* modgrad, const double* angles, const float* cs, int W, int H, double max_grad, float* segs,
                           int cap, int* n_segs)
{
    return drfe_lsd_segments_host_mode(modgrad, angles, cs, W, H, max_grad, 0, segs, cap, n_segs);
}

/* the same with rect_nfa's reading chosen by the caller (drfe_lsd_configure_rect: 0 literal OpenCV 3.4, 1 real-valued) */
int drfe_lsd_segments_host_mode(const double* modgrad, const double* angles, const float* cs, int W, int H, double max_grad,
                                int rect_mode, float* segs, int cap, int* n_segs)
{
    if (rect_mode < 0 || rect_mode > 2) return DRFE_ERR_INVALID;
    if (!modgrad || !angles || !cs || !n_segs || W < 4 || H < 4 || W > 2048 || H > 2048) return DRFE_ERR_INVALID;
    std::vector<uint8_t> used;
    std::vector<OPt> order, orderTmp;
    SegmentFinder finder(W, H, modgrad, angles, cs, max_grad, used, order, orderTmp);
    finder.setRectMode(rect_mode);
    std::vector<float> out;
    auto counts = [&](const std::vector<RectCand>& cands, std::vector<int2>& res) -> bool {
        res.resize(cands.size());
        for (size_t k = 0; k < cands.size(); k++) finder.countHost(cands[k], lsd_walk_mode(rect_mode), res[k].x, res[k].y);
        return true;
    };
    finder.timed_ = std::getenv("DRFE_TRACE_LINES") != nullptr;
    finder.run(out, counts);
    if (finder.timed_)
        std::fprintf(stderr, "drfe_lsd_segments_host: grow %.2f ms (%ld regions, %ld pixels); rect+refine %.2f ms (%ld); improve/NFA %.2f ms\n",
                     finder.tGrow_, finder.nGrow_, finder.nGrowPx_, finder.tRefine_, finder.nRect_, finder.tImprove_);
    *n_segs = (int)(out.size() / 4);
    if (*n_segs > cap) return DRFE_ERR_CAPACITY;
    if (segs && !out.empty()) std::memcpy(segs, out.data(), out.size() * sizeof(float));
    return DRFE_OK;
}

/* Test hook of lsd_order_kernels.hip: n LSD ordering keys (bin << 22 | y << 11 | x) sorted in place by k_lsd_order on the
 * device, to be compared with std::sort under compare_norm (drfe_debug_order_sort, mode 0).  *status = the kernel's status word
 * (0: done; 1: a range above 1024 keys ran out of the depth limit: heap sort of that length is the host's).  depth_limit >= 0
 * replaces introsort's 2 lg n, so that tests reach libstdc++'s heap-sort branch (std::__partial_sort) on ordinary data. */
int drfe_debug_device_order_sort(drfe_ctx* c, uint32_t* keys, size_t n, int* status) { return drfe_debug_device_order_sort_depth(c, keys, n, -1, status); }

int drfe_debug_device_order_sort_depth(drfe_ctx* c, uint32_t* keys, size_t n, int depth_limit, int* status)
{
    if (!c || !keys || !status || n < 1 || n > (1u << 22)) { if (c) c->err = "debug_device_order_sort: invalid argument"; return DRFE_ERR_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t *d = nullptr, *pl = nullptr, *pr = nullptr;
    int* ds = nullptr;
    hipError_t e = hipMalloc((void**)&d, n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&pl, n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&pr, n * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&ds, 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d, keys, n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = drfe_launch_lsd_order(d, n, (int)n, pl, pr, n, ds, 1, 1, c->stream, depth_limit);
    if (e == hipSuccess) e = hipMemcpyAsync(keys, d, n * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(status, ds, 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d); (void)hipFree(pl); (void)hipFree(pr); (void)hipFree(ds);
    if (e != hipSuccess) { c->err = std::string("debug_device_order_sort: ") + hipGetErrorString(e); return DRFE_ERR_HIP; }
    return DRFE_OK;
}

int drfe_debug_cr_sincos(const double* x, int n, double* s, double* c, int32_t* ok)
{
    if (!x || !s || !c || !ok || n < 0) return DRFE_ERR_INVALID;
    for (int i = 0; i < n; i++) ok[i] = drfe_cr_sincos(x[i], &s[i], &c[i]);
    return DRFE_OK;
}

/* Test hook of introsort_restated.h: records sorted in place.  kind 0: LSD keys (uint32: bin << 22 | y << 11 | x, larger bins
 * first); kind 1: VoxelGrid records (uint64: leaf << 32 | point, smaller leaves first).  mode 0: std::sort with the reference's
 * comparator; 1 / 2: the restatement with scalar / AVX2 stopper masks; 3: the plain transcription of libstdc++'s introsort.
 * depth_limit >= 0 replaces 2 lg n (modes 1-3).  skip_below (kind 0, modes 1 / 2): only the keys with bin >= skip_below are wanted
 * (lsd_order::sort).  DRFE_ERR_STATE for mode 2 on a CPU without AVX2. */
int drfe_debug_order_sort(void* recs, size_t n, int kind, int mode, int depth_limit, uint32_t skip_below)
{
    if (!recs || mode < 0 || mode > 3 || kind < 0 || kind > 1) return DRFE_ERR_INVALID;
    if (mode == 2 && !isr::have_avx2()) return DRFE_ERR_STATE;
    if (kind == 0) {
        uint32_t* keys = static_cast<uint32_t*>(recs);
        std::vector<uint32_t> tmp;
        if (mode == 0) std::sort(keys, keys + n, lsd_order::Before());
        else if (mode == 3) lsd_order::reference_sort(keys, n, depth_limit);
        else lsd_order::sort(keys, n, tmp, mode - 1, depth_limit, skip_below);
    } else {
        uint64_t* r = static_cast<uint64_t*>(recs);
        if (mode == 0) std::sort(r, r + n, voxel_order::Before());
        else if (mode == 3) voxel_order::reference_sort(r, n, depth_limit);
        else voxel_order::sort(r, n, mode - 1, depth_limit);
    }
    return DRFE_OK;
}

/* parity taps of the device passes (tests) */
int drfe_lsd_stages(drfe_ctx* c, uint8_t* scaled, double* modgrad, double* angles, int16_t* gx, int16_t* gy, int* sw, int* sh)
{
    if (!c || !c->ls) return c ? DRFE_ERR_STATE : DRFE_ERR_INVALID;
    LinesScratch* s = c->ls;
    if (sw) *sw = s->sw;
    if (sh) *sh = s->sh;
    const size_t ns = (size_t)s->sw * s->sh, n = (size_t)s->w * s->h;
    if (scaled) HIPCHK(c, hipMemcpy(scaled, s->d_scaled, ns, hipMemcpyDeviceToHost));
    if (modgrad) HIPCHK(c, hipMemcpy(modgrad, s->d_modgrad, ns * 8, hipMemcpyDeviceToHost));
    if (angles) HIPCHK(c, hipMemcpy(angles, s->d_angles, ns * 8, hipMemcpyDeviceToHost));
    if (gx) HIPCHK(c, hipMemcpy(gx, s->d_gx, n * 2, hipMemcpyDeviceToHost));
    if (gy) HIPCHK(c, hipMemcpy(gy, s->d_gy, n * 2, hipMemcpyDeviceToHost));
    return DRFE_OK;
}

} /* extern "C" */
