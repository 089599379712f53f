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

#include <algorithm>
#include <cfloat>
#include <functional>
#include <mutex>
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
struct RectD { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; };

/* sequential half of cv::LineSegmentDetectorImpl, fed with the device-computed gradient fields */
class SegmentFinder {
public:
    /* used / order: caller-owned buffers that survive between frames (a lane reuses them: no multi-megabyte
     * allocation, hence no mmap/page-fault traffic, per frame) */
    SegmentFinder(int W, int H, const double* modgrad, const double* angles, const float* cs, double maxGrad,
                  std::vector<uint8_t>& used, std::vector<OPt>& order, std::vector<OPt>& orderTmp)
        : W_(W), H_(H), mod_(modgrad), ang_(angles), cs_(cs), used_(used), order_(order), orderTmp_(orderTmp)
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
        logNT_ = 5 * (std::log10(double(W)) + std::log10(double(H))) / 2 + std::log10(11.0);
    }

    /* counts(cands, out): (pixels, aligned pixels) of every rectangle - the device kernel k_rect_counts */
    typedef std::function<bool(const std::vector<RectCand>&, std::vector<int2>&)> CountFn;

    bool run(std::vector<float>& lines, const CountFn& counts)
    {
        const double angTh = 22.5, scale = 0.8, densityTh = 0.7, logEps = 0;
        const double prec = M_PI * angTh / 180, p = angTh / 180;
        const size_t minReg = size_t(-logNT_ / std::log10(p));
        std::vector<RPt> reg;
        /* rect_improve only READS the angle field and decides whether the segment is kept: it is taken out of the seed loop
         * (whose `used` bookkeeping is the sequential part) and evaluated for all rectangles of the frame at once */
        std::vector<RectD> pending;
        for (const OPt& key : order_) {
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
        const auto t2 = timed_ ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        std::vector<double> logNfa;
        if (!improveAll(pending, logNfa, counts)) return false;
        if (timed_) tImprove_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count();
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

    double tGrow_ = 0, tRefine_ = 0, tImprove_ = 0; long nGrow_ = 0, nRect_ = 0, nGrowPx_ = 0; bool timed_ = false;
private:
    int W_, H_;
    const double *mod_, *ang_;
    const float* cs_;                /* device-computed (cos, sin) of float(angle) per pixel */
    std::vector<uint8_t>& used_;
    std::vector<OPt>& order_;
    std::vector<OPt>& orderTmp_;
    uint32_t minSeedBin_ = 0;
    double logNT_;

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
                if (!seeded) { sumdx = float(std::cos(seedAngle)); sumdy = float(std::sin(seedAngle)); seeded = true; }
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
        const double dx = std::cos(theta), dy = std::sin(theta);
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
        const double log1 = logGammaInt(n + 1) - logGammaInt(k + 1) - logGammaInt(n - k + 1) +
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
public:
    /* DRFE_LSD_CHECK=1: the same pixel loop on the host, to cross-check k_rect_counts (debug only) */
    void countHost(const RectCand& rec, int& total, int& alg) const
    {
        struct Corner { double x, y; bool taken; };
        const double hw = rec.width / 2.0, dyhw = rec.dy * hw, dxhw = rec.dx * hw;
        Corner c[4] = {{double(int(rec.x1 - dyhw)), double(int(rec.y1 + dxhw)), false},
                       {double(int(rec.x2 - dyhw)), double(int(rec.y2 + dxhw)), false},
                       {double(int(rec.x2 + dyhw)), double(int(rec.y2 - dxhw)), false},
                       {double(int(rec.x1 + dyhw)), double(int(rec.y1 - dxhw)), false}};
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
        const double fl = (lo->y != left->y) ? (lo->x - left->x) / (lo->y - left->y) : 0;
        double sl = (left->y != tail->x) ? (left->x - tail->x) / (left->y - tail->y) : 0;
        const double fr = (lo->y != right->y) ? (lo->x - right->x) / (lo->y - right->y) : 0;
        double sr = (right->y != tail->x) ? (right->x - tail->x) / (right->y - tail->y) : 0;
        if (!std::isfinite(sl)) sl = 0;
        if (!std::isfinite(sr)) sr = 0;
        double lstep = fl, rstep = fr, lx = lo->x, rx = lo->x;
        total = 0; alg = 0;
        for (int y = (int)lo->y; y <= (int)hi->y; ++y) {
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
private:
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

/* One line-extraction lane: device scratch + stream + its own error string.  The context owns one (the
 * single-frame entry) and, for drfe_lsd_extract_batch, a pool of them, one per host thread. */
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
};

static void scratch_free(LinesScratch*& s)
{
    if (!s) return;
    void* ptrs[] = {s->d_img, s->d_blur, s->d_scaled, s->d_tmp16, s->d_modgrad, s->d_angles, s->d_cs, s->d_maxGrad, s->d_gx, s->d_gy, s->d_cands, s->d_counts,
                    s->d_lbdLines, s->d_lbdOut};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete s;
    s = nullptr;
}

void drfe_lines_free(drfe_ctx* c)
{
    scratch_free(c->ls);
    delete static_cast<LineHost*>(c->lineHost);
    c->lineHost = nullptr;
    auto* pool = static_cast<std::vector<LineWorker>*>(c->lineWorkers);
    if (pool) {
        for (LineWorker& w : *pool) {
            scratch_free(w.ls);
            delete w.host;
            if (w.ownsStream && w.stream) (void)hipStreamDestroy(w.stream);
        }
        delete pool;
        c->lineWorkers = nullptr;
    }
}

static int ensure_lines(LineWorker* c, int w, int h)
{
    if (c->ls && c->ls->w == w && c->ls->h == h) return DRFE_OK;
    scratch_free(c->ls);
    LinesScratch* s = new (std::nothrow) LinesScratch();
    if (!s) return DRFE_ERR_INVALID;
    std::memset(s, 0, sizeof(*s));
    c->ls = s;
    s->w = w; s->h = h;
    s->sw = (int)std::rint(w * 0.8); s->sh = (int)std::rint(h * 0.8);   /* saturate_cast<int>(size * inv_scale) */
    const size_t n = (size_t)w * h, ns = (size_t)s->sw * s->sh;
    HIPCHK(c, hipMalloc((void**)&s->d_img, n));
    HIPCHK(c, hipMalloc((void**)&s->d_blur, n));
    HIPCHK(c, hipMalloc((void**)&s->d_scaled, ns));
    HIPCHK(c, hipMalloc((void**)&s->d_tmp16, n * 2));
    HIPCHK(c, hipMalloc((void**)&s->d_modgrad, ns * 8));
    HIPCHK(c, hipMalloc((void**)&s->d_angles, ns * 8));
    HIPCHK(c, hipMalloc((void**)&s->d_cs, ns * 8));
    HIPCHK(c, hipMalloc((void**)&s->d_maxGrad, 8));
    HIPCHK(c, hipMalloc((void**)&s->d_gx, n * 2));
    HIPCHK(c, hipMalloc((void**)&s->d_gy, n * 2));
    return DRFE_OK;
}

/* LineSegment::ExtractLineSegment for one frame on one lane */
static int lsd_extract_core(LineWorker* c, int device, const uint8_t* gray, int w, int h, size_t stride, int max_lines,
                            drfe_keyline* lines, uint8_t* ldesc, double* line_f, int cap, int* n_lines, int* n_detected)
{
    *n_lines = 0;
    HIPCHK(c, hipSetDevice(device));
    int rc = ensure_lines(c, w, h);
    if (rc != DRFE_OK) return rc;
    LinesScratch* s = c->ls;
    if (s->sw > 2048 || s->sh > 2048) { c->err = "lsd_extract: image larger than 2560 x 2560 (pixel-ordering keys)"; return DRFE_ERR_INVALID; }
    /* LineSegmentDetector defaults: scale 0.8, sigma_scale 0.6 -> sigma 0.75, 7x7 kernel; quant 2, ang_th 22.5 */
    const double sigma = 0.6 / 0.8;
    const int hk = (int)std::ceil(sigma * std::sqrt(2 * 3.0 * std::log(10.0)));
    const LineTaps lsdTaps = gaussTaps(1 + 2 * hk, sigma), lbdTaps = gaussTaps(5, 1.0);
    const double rho = 2.0 / std::sin(M_PI * 22.5 / 180);
    hipStream_t st = c->stream;
    const bool trace = std::getenv("DRFE_TRACE_LINES") != nullptr;
    const auto tStart = std::chrono::steady_clock::now();
    HIPCHK(c, hipMemcpy2DAsync(s->d_img, (size_t)w, gray, stride, (size_t)w, (size_t)h, hipMemcpyHostToDevice, st));
    HIPCHK(c, drfe_launch_lines_passes(s->d_img, w, h, lsdTaps, lbdTaps, s, rho, st));
    const size_t ns = (size_t)s->sw * s->sh;
    if (!c->host) c->host = new LineHost();
    LineHost& H = *c->host;
    std::vector<double>&modgrad = H.modgrad, &angles = H.angles;
    modgrad.resize(ns); angles.resize(ns); H.cs.resize(2 * ns);
    unsigned long long maxBits = 0;
    HIPCHK(c, hipMemcpyAsync(modgrad.data(), s->d_modgrad, ns * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(angles.data(), s->d_angles, ns * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(H.cs.data(), s->d_cs, ns * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(&maxBits, s->d_maxGrad, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    double maxGrad = -1;
    if (maxBits) std::memcpy(&maxGrad, &maxBits, 8);

    const auto tDev = std::chrono::steady_clock::now();
    std::vector<float> segs;
    SegmentFinder finder(s->sw, s->sh, modgrad.data(), angles.data(), H.cs.data(), maxGrad, H.used, H.order, H.orderTmp);
    const auto tSort = std::chrono::steady_clock::now();
    finder.timed_ = trace;
    int countRc = DRFE_OK;
    const bool checkCounts = std::getenv("DRFE_LSD_CHECK") != nullptr;
    auto counts = [&](const std::vector<RectCand>& cands, std::vector<int2>& out) -> bool {
        const size_t nc = cands.size();
        out.resize(nc);
        if (nc > s->candCap) {
            if (s->d_cands) (void)hipFree(s->d_cands);
            if (s->d_counts) (void)hipFree(s->d_counts);
            s->d_cands = nullptr; s->d_counts = nullptr;
            s->candCap = std::max<size_t>(nc * 2, 4096);
            if (hipMalloc((void**)&s->d_cands, s->candCap * sizeof(RectCand)) != hipSuccess ||
                hipMalloc((void**)&s->d_counts, s->candCap * sizeof(int2)) != hipSuccess) {
                s->candCap = 0; c->err = "lsd_extract: hipMalloc of the NFA scratch failed"; countRc = DRFE_ERR_HIP; return false;
            }
        }
        hipError_t e = hipMemcpyAsync(s->d_cands, cands.data(), nc * sizeof(RectCand), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = drfe_launch_rect_counts(s->d_cands, (int)nc, s->d_angles, s->sw, s->sh, s->d_counts, st);
        if (e == hipSuccess) e = hipMemcpyAsync(out.data(), s->d_counts, nc * sizeof(int2), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { c->err = std::string("lsd_extract: rectangle counting: ") + hipGetErrorString(e); countRc = DRFE_ERR_HIP; return false; }
        if (checkCounts)
            for (size_t k = 0; k < nc; k++) {
                int t = 0, a = 0;
                finder.countHost(cands[k], t, a);
                if (t != out[k].x || a != out[k].y)
                    std::fprintf(stderr, "k_rect_counts mismatch: cand %zu device (%d, %d) host (%d, %d)  x1 %.17g y1 %.17g x2 %.17g y2 %.17g w %.17g dx %.17g dy %.17g theta %.17g prec %.17g\n",
                                 k, out[k].x, out[k].y, t, a, cands[k].x1, cands[k].y1, cands[k].x2, cands[k].y2, cands[k].width, cands[k].dx, cands[k].dy, cands[k].theta, cands[k].prec);
            }
        return true;
    };
    if (!finder.run(segs, counts)) return countRc;
    const auto tSeg = std::chrono::steady_clock::now();
    if (trace) std::fprintf(stderr, "drfe_lsd_extract: pixel ordering (bins + std::sort) %.2f ms; grow %.2f ms (%ld regions, %ld pixels); rect+refine %.2f ms (%ld); improve/NFA %.2f ms\n",
                            std::chrono::duration<double, std::milli>(tSort - tDev).count(), finder.tGrow_, finder.nGrow_, finder.nGrowPx_, finder.tRefine_, finder.nRect_, finder.tImprove_);

    /* LSDDetector::detect: KeyLine fields for octave 0 (octaveScale = 1) */
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
    struct TraceAtExit {   /* DRFE_TRACE_LINES=1: where a call spends its time (device passes + copies | LSD host | LBD host) */
        bool on; std::chrono::steady_clock::time_point a, b, cc;
        ~TraceAtExit() {
            if (!on) return;
            const auto e = std::chrono::steady_clock::now();
            auto ms = [](auto x, auto y) { return std::chrono::duration<double, std::milli>(y - x).count(); };
            std::fprintf(stderr, "drfe_lsd_extract: device+copies %.2f ms, LSD host %.2f ms, keylines+LBD host %.2f ms\n", ms(a, b), ms(b, cc), ms(cc, e));
        }
    } traceAtExit{trace, tStart, tDev, tSeg};
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
        HIPCHK(c, drfe_launch_lbd(s->d_lbdLines, nl, s->d_gx, s->d_gy, w, h, lbdTables(), s->d_lbdOut, st));
        HIPCHK(c, hipMemcpyAsync(ldesc, s->d_lbdOut, (size_t)nl * 32, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
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
    const int rc = lsd_extract_core(&lw, c->device, gray, w, h, stride, max_lines, lines, ldesc, line_f, cap, n_lines, n_detected);
    c->ls = lw.ls;
    c->lineHost = lw.host;
    if (rc != DRFE_OK) c->err = lw.err;
    return rc;
}

/* LineSegment::ExtractLineSegment for nframes host images at once.  The device passes are microseconds; the
 * sequential host stages (region growing, rectangle refinement, NFA: ~25 ms per 640x480 frame) are what a frame
 * costs, and they are independent between frames — so the batch runs on a pool of host threads, one lane (device
 * scratch + stream) per thread, the way the reference spreads its four extractors over four threads
 * (src/Frame.cc:116-126).  Outputs are per frame: lines[f * cap ..], ldesc[f * cap * 32 ..], line_f[f * cap * 3 ..],
 * n_lines[f], n_detected[f].  n_threads <= 0: one thread per frame up to the hardware concurrency. */
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
        lw.ownsStream = true;
        pool->push_back(lw);
    }
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
    if (!modgrad || !angles || !cs || !n_segs || W < 4 || H < 4 || W > 2048 || H > 2048) return DRFE_ERR_INVALID;
    std::vector<uint8_t> used;
    std::vector<OPt> order, orderTmp;
    SegmentFinder finder(W, H, modgrad, angles, cs, max_grad, used, order, orderTmp);
    std::vector<float> out;
    auto counts = [&](const std::vector<RectCand>& cands, std::vector<int2>& res) -> bool {
        res.resize(cands.size());
        for (size_t k = 0; k < cands.size(); k++) finder.countHost(cands[k], res[k].x, res[k].y);
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
