/* oracle/lsd_oracle.cpp — TEST INFRASTRUCTURE (see oracle.h).  PARITY UNPINNED.
 *
 * CPU restatement of the line-feature extraction of the path:
 *   LineSegment::ExtractLineSegment                reference src/LSDextractor.cpp:12-43
 * whose arithmetic lives entirely in un-vendored libraries (opencv_contrib line_descriptor 3.4.x and
 * OpenCV imgproc 3.4.x, README.md:25), restated here from their published algorithm / source
 * structure — nothing in the reference pins it and it cannot be run here:
 *   cv::line_descriptor::LSDDetector::detect(img, keylines, scale, numOctaves)   (LSDDetector.cpp)
 *   cv::LineSegmentDetector (LSD_REFINE_ADV, scale 0.8, sigma_scale 0.6, quant 2, ang_th 22.5,
 *       log_eps 0, density_th 0.7, n_bins 1024)                                  (imgproc/lsd.cpp)
 *   cv::line_descriptor::BinaryDescriptor::compute -> computeLBD -> binaryConversion
 *                                                                         (binary_descriptor.cpp)
 *   cv::GaussianBlur CV_8U fixed-point path, cv::resize INTER_LINEAR_EXACT, cv::Sobel 3x3 CV_16S
 * Quirks of the OpenCV sources that are kept: integer division and the tailp->p.x/p.y mix in
 * rect_nfa's edge steps, the width test guarding the last "finer precision" loop of rect_improve,
 * std::sort (unstable, libstdc++) for the pseudo-ordering and for the reference's response sort.
 */
#include "lsd_oracle.h"
#include "oracle.h"
#include "oracle_math.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <type_traits>

namespace orc {

namespace {

const double NOTDEF = -1024.0;
const double M_3_2_PI_ = 3.0 * M_PI / 2.0, M_2__PI_ = 2.0 * M_PI;
const double DEG_TO_RADS = M_PI / 180.0;
const double RELATIVE_ERROR_FACTOR = 100.0;

/* 8-bit fixed-point Gaussian (getFixedpointGaussianKernel): taps = round(256 * normalised exp()) */
static std::vector<int> gauss_taps_q8(int n, double sigma)
{
    std::vector<double> v(n);
    double sum = 0;
    const double scale2X = -0.5 / (sigma * sigma);
    for (int i = 0; i < n; i++) {
        const double x = i - (n - 1) * 0.5;
        v[i] = std::exp(scale2X * x * x);
        sum += v[i];
    }
    sum = 1.0 / sum;
    std::vector<int> t(n);
    for (int i = 0; i < n; i++) t[i] = (int)std::rint(v[i] * sum * 256.0);
    return t;
}

} // namespace

void gaussian_blur_q8(const uint8_t* src, int w, int h, const std::vector<int>& taps, uint8_t* dst)
{
    const int n = (int)taps.size(), r = n / 2;
    std::vector<uint32_t> hb((size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t acc = 0;
            for (int k = 0; k < n; k++) acc += (uint32_t)taps[k] * src[(size_t)y * w + reflect101(x + k - r, w)];
            hb[(size_t)y * w + x] = std::min<uint32_t>(acc, 65535u);   /* ufixedpoint16 saturating sum */
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            uint32_t acc = 0;
            for (int k = 0; k < n; k++) acc += (uint32_t)taps[k] * hb[(size_t)reflect101(y + k - r, h) * w + x];
            dst[(size_t)y * w + x] = (uint8_t)std::min<uint32_t>(255u, (acc + 32768u) >> 16);
        }
}

/* cv::resize(..., Size(), 0.8, 0.8, INTER_LINEAR_EXACT) for CV_8UC1: 8.8 fixed-point weights from
 * the fractional part of (d + 0.5) * (1/0.8) - 0.5, 16.16 vertical accumulation, round to nearest. */
void resize_linear_exact_08(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh)
{
    const double scale_x = 1.0 / 0.8, scale_y = 1.0 / 0.8;
    std::vector<int> xo(dw), xc0(dw), xc1(dw);
    for (int dx = 0; dx < dw; dx++) {
        const double v = (dx + 0.5) * scale_x - 0.5;
        int o = (int)std::floor(v);
        int c1 = (int)std::rint((v - o) * 256.0);
        if (o < 0) { o = 0; c1 = 0; }
        if (o >= sw - 1) { o = sw - 1; c1 = 0; }
        xo[dx] = o; xc1[dx] = c1; xc0[dx] = 256 - c1;
    }
    std::vector<uint32_t> h0(dw), h1(dw);
    for (int dy = 0; dy < dh; dy++) {
        const double v = (dy + 0.5) * scale_y - 0.5;
        int o = (int)std::floor(v);
        int c1 = (int)std::rint((v - o) * 256.0);
        if (o < 0) { o = 0; c1 = 0; }
        if (o >= sh - 1) { o = sh - 1; c1 = 0; }
        const int o1 = std::min(o + 1, sh - 1), c0 = 256 - c1;
        for (int dx = 0; dx < dw; dx++) {
            const int a = xo[dx], b = std::min(a + 1, sw - 1);
            h0[dx] = (uint32_t)xc0[dx] * src[(size_t)o * sw + a] + (uint32_t)xc1[dx] * src[(size_t)o * sw + b];
            h1[dx] = (uint32_t)xc0[dx] * src[(size_t)o1 * sw + a] + (uint32_t)xc1[dx] * src[(size_t)o1 * sw + b];
        }
        for (int dx = 0; dx < dw; dx++) {
            const uint32_t acc = (uint32_t)c0 * h0[dx] + (uint32_t)c1 * h1[dx];
            dst[(size_t)dy * dw + dx] = (uint8_t)std::min<uint32_t>(255u, (acc + 32768u) >> 16);
        }
    }
}

/* cv::Sobel(src, dst, CV_16S, dx, dy, 3), BORDER_REFLECT_101 */
void sobel3_s16(const uint8_t* src, int w, int h, int16_t* gx, int16_t* gy)
{
    for (int y = 0; y < h; y++) {
        const int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
        for (int x = 0; x < w; x++) {
            const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
            const int a = src[(size_t)ym * w + xm], b = src[(size_t)ym * w + x], c = src[(size_t)ym * w + xp];
            const int d = src[(size_t)y * w + xm], f = src[(size_t)y * w + xp];
            const int g = src[(size_t)yp * w + xm], hh = src[(size_t)yp * w + x], i = src[(size_t)yp * w + xp];
            gx[(size_t)y * w + x] = (int16_t)((c + 2 * f + i) - (a + 2 * d + g));
            gy[(size_t)y * w + x] = (int16_t)((g + 2 * hh + i) - (a + 2 * b + c));
        }
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* cv::LineSegmentDetectorImpl (imgproc/lsd.cpp)                                                    */

namespace {

struct RegionPoint { int x, y; double angle, modgrad; };
struct NormPoint { int x, y, norm; };
struct Rect {
    double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p;
};

struct Lsd {
    int W = 0, H = 0;
    std::vector<double> angles, modgrad;
    std::vector<uint8_t> used;
    std::vector<NormPoint> ordered;
    double LOG_NT = 0;
    const double SCALE = 0.8, SIGMA_SCALE = 0.6, QUANT = 2.0, ANG_TH = 22.5, LOG_EPS = 0, DENSITY_TH = 0.7;
    const int N_BINS = 1024;
    /* 0 (default): the OpenCV 3.4 source text - rect_nfa's integer corners and nfa()'s `double(n) + 1` first term; 1: the LSD
     * paper's reading of both (real-valued corners, log_gamma(n + 1)) as rounds 2-3 had it; 2: integer corners with log_gamma(n + 1)
     * (round 4) */
    int rect_mode = 0;
    std::vector<int>* count_log = nullptr;     /* (total_pts, alg_pts) of every rect_nfa call, in call order */

    static double distSq(double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); }
    static double dist(double x1, double y1, double x2, double y2) { return std::sqrt(distSq(x1, y1, x2, y2)); }
    static double angle_diff_signed(double a, double b)
    {
        double diff = a - b;
        while (diff <= -M_PI) diff += M_2__PI_;
        while (diff > M_PI) diff -= M_2__PI_;
        return diff;
    }
    static double angle_diff(double a, double b) { return std::fabs(angle_diff_signed(a, b)); }
    static bool double_equal(double a, double b)
    {
        if (a == b) return true;
        const double abs_diff = std::fabs(a - b), aa = std::fabs(a), bb = std::fabs(b);
        double abs_max = (aa > bb) ? aa : bb;
        if (abs_max < DBL_MIN) abs_max = DBL_MIN;
        return (abs_diff / abs_max) <= (RELATIVE_ERROR_FACTOR * DBL_EPSILON);
    }
    static double log_gamma_windschitl(double x)
    {
        return 0.918938533204673 + (x - 0.5) * std::log(x) - x + 0.5 * x * std::log(x * std::sinh(1 / x) + 1 / (810.0 * std::pow(x, 6.0)));
    }
    static double log_gamma_lanczos(double x)
    {
        static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
        double a = (x + 0.5) * std::log(x + 5.5) - (x + 5.5);
        double b = 0;
        for (int n = 0; n < 7; ++n) { a -= std::log(x + double(n)); b += q[n] * std::pow(x, double(n)); }
        return a + std::log(b);
    }
    static double log_gamma(double x) { return x > 15.0 ? log_gamma_windschitl(x) : log_gamma_lanczos(x); }

    void ll_angle(const uint8_t* img, double threshold)
    {
        angles.assign((size_t)W * H, NOTDEF);
        modgrad.assign((size_t)W * H, 0.0);
        double max_grad = -1;
        for (int y = 0; y < H - 1; ++y)
            for (int x = 0; x < W - 1; ++x) {
                const int DA = img[(size_t)(y + 1) * W + x + 1] - img[(size_t)y * W + x];
                const int BC = img[(size_t)y * W + x + 1] - img[(size_t)(y + 1) * W + x];
                const int gx = DA + BC, gy = DA - BC;
                const double norm = std::sqrt((gx * gx + gy * gy) / 4.0);
                modgrad[(size_t)y * W + x] = norm;
                if (norm <= threshold) angles[(size_t)y * W + x] = NOTDEF;
                else {
                    angles[(size_t)y * W + x] = fast_atan2_deg(float(gx), float(-gy)) * DEG_TO_RADS;
                    if (norm > max_grad) max_grad = norm;
                }
            }
        const double bin_coef = (max_grad > 0) ? double(N_BINS - 1) / max_grad : 0;
        ordered.clear();
        ordered.reserve((size_t)(W - 1) * (H - 1));
        for (int y = 0; y < H - 1; ++y)
            for (int x = 0; x < W - 1; ++x) ordered.push_back({x, y, int(modgrad[(size_t)y * W + x] * bin_coef)});
        std::sort(ordered.begin(), ordered.end(), [](const NormPoint& a, const NormPoint& b) { return a.norm > b.norm; });
    }

    bool isAligned(int x, int y, double theta, double prec) const
    {
        if (x < 0 || y < 0 || x >= W || y >= H) return false;
        const double a = angles[(size_t)y * W + x];
        if (a == NOTDEF) return false;
        double n_theta = theta - a;
        if (n_theta < 0) n_theta = -n_theta;
        if (n_theta > M_3_2_PI_) {
            n_theta -= M_2__PI_;
            if (n_theta < 0) n_theta = -n_theta;
        }
        return n_theta <= prec;
    }

    void region_grow(int sx, int sy, std::vector<RegionPoint>& reg, double& reg_angle, double prec)
    {
        reg.clear();
        reg_angle = angles[(size_t)sy * W + sx];
        reg.push_back({sx, sy, reg_angle, modgrad[(size_t)sy * W + sx]});
        float sumdx = float(std::cos(reg_angle));
        float sumdy = float(std::sin(reg_angle));
        used[(size_t)sy * W + sx] = 1;
        for (size_t i = 0; i < reg.size(); i++) {
            const int px = reg[i].x, py = reg[i].y;
            const int xx_min = std::max(px - 1, 0), xx_max = std::min(px + 1, W - 1);
            const int yy_min = std::max(py - 1, 0), yy_max = std::min(py + 1, H - 1);
            for (int yy = yy_min; yy <= yy_max; ++yy)
                for (int xx = xx_min; xx <= xx_max; ++xx) {
                    uint8_t& is_used = used[(size_t)yy * W + xx];
                    if (is_used != 1 && isAligned(xx, yy, reg_angle, prec)) {
                        const double angle = angles[(size_t)yy * W + xx];
                        is_used = 1;
                        reg.push_back({xx, yy, angle, modgrad[(size_t)yy * W + xx]});
                        /* cos(float)/sin(float) of lsd.cpp resolve to cosf/sinf, whose last bit is libm dependent:
                         * canonicalised on the shared float routine, as for the ORB steering (oracle.h, §9.4) */
                        float sn, cn;
                        sincos_f(float(angle), &sn, &cn);
                        sumdx += cn;
                        sumdy += sn;
                        reg_angle = fast_atan2_deg(sumdy, sumdx) * DEG_TO_RADS;
                    }
                }
        }
    }

    double get_theta(const std::vector<RegionPoint>& reg, double x, double y, double reg_angle, double prec) const
    {
        double Ixx = 0.0, Iyy = 0.0, Ixy = 0.0;
        for (size_t i = 0; i < reg.size(); ++i) {
            const double dx = double(reg[i].x) - x, dy = double(reg[i].y) - y, w = reg[i].modgrad;
            Ixx += dy * dy * w;
            Iyy += dx * dx * w;
            Ixy -= dx * dy * w;
        }
        const double lambda = 0.5 * (Ixx + Iyy - std::sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
        double theta = (std::fabs(Ixx) > std::fabs(Iyy)) ? double(fast_atan2_deg(float(lambda - Ixx), float(Ixy)))
                                                         : double(fast_atan2_deg(float(Ixy), float(lambda - Iyy)));
        theta *= DEG_TO_RADS;
        if (angle_diff(theta, reg_angle) > prec) theta += M_PI;
        return theta;
    }

    void region2rect(const std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec) const
    {
        double x = 0, y = 0, sum = 0;
        for (size_t i = 0; i < reg.size(); ++i) {
            const double w = reg[i].modgrad;
            x += double(reg[i].x) * w;
            y += double(reg[i].y) * w;
            sum += w;
        }
        x /= sum; y /= sum;
        const double theta = get_theta(reg, x, y, reg_angle, prec);
        const double dx = std::cos(theta), dy = std::sin(theta);
        double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
        for (size_t i = 0; i < reg.size(); ++i) {
            const double regdx = double(reg[i].x) - x, regdy = double(reg[i].y) - y;
            const double l = regdx * dx + regdy * dy;
            const double w = -regdx * dy + regdy * dx;
            if (l > l_max) l_max = l; else if (l < l_min) l_min = l;
            if (w > w_max) w_max = w; else if (w < w_min) w_min = w;
        }
        rec.x1 = x + l_min * dx; rec.y1 = y + l_min * dy;
        rec.x2 = x + l_max * dx; rec.y2 = y + l_max * dy;
        rec.width = w_max - w_min;
        rec.x = x; rec.y = y; rec.theta = theta; rec.dx = dx; rec.dy = dy; rec.prec = prec; rec.p = p;
        if (rec.width < 1.0) rec.width = 1.0;
    }

    bool reduce_region_radius(std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec,
                              double density, double density_th)
    {
        const double xc = double(reg[0].x), yc = double(reg[0].y);
        const double radSq1 = distSq(xc, yc, rec.x1, rec.y1), radSq2 = distSq(xc, yc, rec.x2, rec.y2);
        double radSq = radSq1 > radSq2 ? radSq1 : radSq2;
        while (density < density_th) {
            radSq *= 0.75 * 0.75;
            for (size_t i = 0; i < reg.size(); ++i)
                if (distSq(xc, yc, double(reg[i].x), double(reg[i].y)) > radSq) {
                    used[(size_t)reg[i].y * W + reg[i].x] = 0;
                    std::swap(reg[i], reg[reg.size() - 1]);
                    reg.pop_back();
                    --i;
                }
            if (reg.size() < 2) return false;
            region2rect(reg, reg_angle, prec, p, rec);
            density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        }
        return true;
    }

    bool refine(std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec, double density_th)
    {
        double density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        if (density >= density_th) return true;
        const double xc = double(reg[0].x), yc = double(reg[0].y);
        const double ang_c = reg[0].angle;
        double sum = 0, s_sum = 0;
        int n = 0;
        for (size_t i = 0; i < reg.size(); ++i) {
            used[(size_t)reg[i].y * W + reg[i].x] = 0;
            if (dist(xc, yc, reg[i].x, reg[i].y) < rec.width) {
                const double ang_d = angle_diff_signed(reg[i].angle, ang_c);
                sum += ang_d;
                s_sum += ang_d * ang_d;
                ++n;
            }
        }
        const double mean_angle = sum / double(n);
        const double tau = 2.0 * std::sqrt((s_sum - 2.0 * mean_angle * sum) / double(n) + mean_angle * mean_angle);
        const int sx = reg[0].x, sy = reg[0].y;
        region_grow(sx, sy, reg, reg_angle, tau);
        if (reg.size() < 2) return false;
        region2rect(reg, reg_angle, prec, p, rec);
        density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        if (density < density_th) return reduce_region_radius(reg, reg_angle, prec, p, rec, density, density_th);
        return true;
    }

    double nfa(int n, int k, double p) const
    {
        if (n == 0 || k == 0) return -LOG_NT;
        if (n == k) return -LOG_NT - double(n) * std::log10(p);
        const double p_term = p / (1 - p);
        /* lsd.cpp, LineSegmentDetectorImpl::nfa: `double log1term = (double(n) + 1) - log_gamma(double(k) + 1) - log_gamma(double(n-k) + 1)
         * + double(k) * log(p) + (double(n-k)) * log(1.0 - p);` - the LSD paper (and lsd_1.6) has log_gamma(n + 1) for the first term; the
         * library's text lost the call.  With it gone log1term is lower by log_gamma(n + 1) - (n + 1) (262 at n = 100): the binomial
         * tail all but vanishes and nearly every rectangle with k > n p passes `log_nfa > LOG_EPS` at rect_improve's first test.
         * Restated from memory of the library source (not in /root/reference): unpinned, like the rest of this file; the paper's form
         * stays selectable (rect_mode 1 / 2) and tools/dump_opencv_reference.py dumps what settles it. */
        const double first = rect_mode == 0 ? (double(n) + 1) : log_gamma(double(n) + 1);
        const double log1term = first - log_gamma(double(k) + 1) - log_gamma(double(n - k) + 1) +
                                double(k) * std::log(p) + double(n - k) * std::log(1.0 - p);
        double term = std::exp(log1term);
        if (double_equal(term, 0)) {
            if (k > n * p) return -log1term / M_LN10 - LOG_NT;
            return -LOG_NT;
        }
        double bin_tail = term;
        const double tolerance = 0.1;
        for (int i = k + 1; i <= n; ++i) {
            const double bin_term = double(n - i + 1) / double(i);
            const double mult_term = bin_term * p_term;
            term *= mult_term;
            bin_tail += term;
            if (bin_term < 1) {
                const double err = term * ((1 - std::pow(mult_term, double(n - i + 1))) / (1 - mult_term) - 1);
                if (err < tolerance * std::fabs(-std::log10(bin_tail) - LOG_NT) * bin_tail) break;
            }
        }
        return -std::log10(bin_tail) - LOG_NT;
    }

    /* rect_nfa (OpenCV 3.4 imgproc/src/lsd.cpp), rect_mode 0 - THE LITERAL READING, default: `struct edge { cv::Point p; bool
     * taken; }` holds INTEGER corners, so the four edge steps are integer quotients (truncated towards zero), and the two
     * second steps divide by (y - tailp->p.x) - the same x / y mix their guards test, hence never by zero.  The scan lines
     * of an oblique rectangle are then not the rectangle's own (a step of magnitude below one is 0); that is the library's
     * behaviour in 3.4.x and a reference bug to preserve, not to repair.
     * rect_mode 1 - the round-3 reading kept for comparison: corners as doubles holding the truncated integers, real-valued
     * quotients, (y - tailp->p.y) in the second-step denominators, a step that would divide by zero taken as 0. */
    template <typename T> struct EdgeT { T x, y; bool taken; };
    template <typename T> void rect_counts(const Rect& rec, int& total_pts, int& alg_pts) const
    {
        typedef EdgeT<T> Edge;
        total_pts = 0; alg_pts = 0;
        const double half_width = rec.width / 2.0, dyhw = rec.dy * half_width, dxhw = rec.dx * half_width;
        Edge e[4];
        e[0] = {T(int(rec.x1 - dyhw)), T(int(rec.y1 + dxhw)), false};
        e[1] = {T(int(rec.x2 - dyhw)), T(int(rec.y2 + dxhw)), false};
        e[2] = {T(int(rec.x2 + dyhw)), T(int(rec.y2 - dxhw)), false};
        e[3] = {T(int(rec.x1 + dyhw)), T(int(rec.y1 - dxhw)), false};
        std::sort(e, e + 4, [](const Edge& a, const Edge& b) { return a.x == b.x ? a.y < b.y : a.x < b.x; });
        Edge *min_y = &e[0], *max_y = &e[0];
        for (unsigned i = 1; i < 4; ++i) {
            if (min_y->y > e[i].y) min_y = &e[i];
            if (max_y->y < e[i].y) max_y = &e[i];
        }
        min_y->taken = true;
        Edge* leftmost = 0;
        for (unsigned i = 0; i < 4; ++i)
            if (!e[i].taken) { if (!leftmost) leftmost = &e[i]; else if (leftmost->x > e[i].x) leftmost = &e[i]; }
        leftmost->taken = true;
        Edge* rightmost = 0;
        for (unsigned i = 0; i < 4; ++i)
            if (!e[i].taken) { if (!rightmost) rightmost = &e[i]; else if (rightmost->x < e[i].x) rightmost = &e[i]; }
        rightmost->taken = true;
        Edge* tailp = 0;
        for (unsigned i = 0; i < 4; ++i)
            if (!e[i].taken) { if (!tailp) tailp = &e[i]; else if (tailp->x > e[i].x) tailp = &e[i]; }
        tailp->taken = true;
        double flstep, slstep, frstep, srstep;
        if (std::is_integral<T>::value) {
            /* the expressions of lsd.cpp, evaluated in int and converted to double by the assignment */
            flstep = (min_y->y != leftmost->y) ? (min_y->x - leftmost->x) / (min_y->y - leftmost->y) : 0;
            slstep = (leftmost->y != tailp->x) ? (leftmost->x - tailp->x) / (leftmost->y - tailp->x) : 0;
            frstep = (min_y->y != rightmost->y) ? (min_y->x - rightmost->x) / (min_y->y - rightmost->y) : 0;
            srstep = (rightmost->y != tailp->x) ? (rightmost->x - tailp->x) / (rightmost->y - tailp->x) : 0;
        } else {
            flstep = (min_y->y != leftmost->y) ? (min_y->x - leftmost->x) / (min_y->y - leftmost->y) : 0;
            slstep = (leftmost->y != tailp->x) ? (leftmost->x - tailp->x) / (leftmost->y - tailp->y) : 0;
            frstep = (min_y->y != rightmost->y) ? (min_y->x - rightmost->x) / (min_y->y - rightmost->y) : 0;
            srstep = (rightmost->y != tailp->x) ? (rightmost->x - tailp->x) / (rightmost->y - tailp->y) : 0;
            /* a guard that misses (equal y, different x) would divide by zero; such a step is taken as 0 */
            if (!std::isfinite(slstep)) slstep = 0;
            if (!std::isfinite(srstep)) srstep = 0;
        }
        double lstep = flstep, rstep = frstep;
        double left_x = min_y->x, right_x = min_y->x;
        const int min_iter = (int)min_y->y, max_iter = (int)max_y->y;
        for (int y = min_iter; y <= max_iter; ++y) {
            if (y < 0 || y >= H) continue;
            for (int x = int(left_x); x <= int(right_x); ++x) {
                if (x < 0 || x >= W) continue;
                ++total_pts;
                if (isAligned(x, y, rec.theta, rec.prec)) ++alg_pts;
            }
            if (y >= leftmost->y) lstep = slstep;
            if (y >= rightmost->y) rstep = srstep;
            left_x += lstep;
            right_x += rstep;
        }
    }

    double rect_nfa(const Rect& rec) const
    {
        int total_pts = 0, alg_pts = 0;
        if (rect_mode != 1) rect_counts<int>(rec, total_pts, alg_pts);
        else rect_counts<double>(rec, total_pts, alg_pts);
        if (count_log) { count_log->push_back(total_pts); count_log->push_back(alg_pts); }
        return nfa(total_pts, alg_pts, rec.p);
    }

    double rect_improve(Rect& rec) const
    {
        const double delta = 0.5, delta_2 = delta / 2.0;
        double log_nfa = rect_nfa(rec);
        if (log_nfa > LOG_EPS) return log_nfa;
        Rect r = rec;
        for (int n = 0; n < 5; ++n) {
            r.p /= 2; r.prec = r.p * M_PI;
            const double v = rect_nfa(r);
            if (v > log_nfa) { log_nfa = v; rec = r; }
        }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (unsigned n = 0; n < 5; ++n)
            if ((r.width - delta) >= 0.5) {
                r.width -= delta;
                const double v = rect_nfa(r);
                if (v > log_nfa) { rec = r; log_nfa = v; }
            }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (unsigned n = 0; n < 5; ++n)
            if ((r.width - delta) >= 0.5) {
                r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2; r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2;
                r.width -= delta;
                const double v = rect_nfa(r);
                if (v > log_nfa) { rec = r; log_nfa = v; }
            }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (unsigned n = 0; n < 5; ++n)
            if ((r.width - delta) >= 0.5) {
                r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2; r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2;
                r.width -= delta;
                const double v = rect_nfa(r);
                if (v > log_nfa) { rec = r; log_nfa = v; }
            }
        if (log_nfa > LOG_EPS) return log_nfa;
        r = rec;
        for (unsigned n = 0; n < 5; ++n)
            if ((r.width - delta) >= 0.5) {
                r.p /= 2; r.prec = r.p * M_PI;
                const double v = rect_nfa(r);
                if (v > log_nfa) { rec = r; log_nfa = v; }
            }
        return log_nfa;
    }

    /* flsd */
    void detect(const uint8_t* image, int w, int h, std::vector<float>& lines, LsdStages* st)
    {
        const double prec = M_PI * ANG_TH / 180, p = ANG_TH / 180, rho = QUANT / std::sin(prec);
        const double sigma = SIGMA_SCALE / SCALE, sprec = 3;
        const unsigned hk = (unsigned)(std::ceil(sigma * std::sqrt(2 * sprec * std::log(10.0))));
        const int ksize = 1 + 2 * (int)hk;
        std::vector<uint8_t> blurred((size_t)w * h);
        gaussian_blur_q8(image, w, h, gauss_taps_q8(ksize, sigma), blurred.data());
        W = (int)std::rint(w * SCALE); H = (int)std::rint(h * SCALE);
        std::vector<uint8_t> scaled((size_t)W * H);
        resize_linear_exact_08(blurred.data(), w, h, scaled.data(), W, H);
        ll_angle(scaled.data(), rho);
        if (st) { st->sw = W; st->sh = H; st->scaled = scaled; st->modgrad = modgrad; st->angles = angles; }
        LOG_NT = 5 * (std::log10(double(W)) + std::log10(double(H))) / 2 + std::log10(11.0);
        const size_t min_reg_size = size_t(-LOG_NT / std::log10(p));
        used.assign((size_t)W * H, 0);
        std::vector<RegionPoint> reg;
        for (size_t i = 0; i < ordered.size(); ++i) {
            const int px = ordered[i].x, py = ordered[i].y;
            if (used[(size_t)py * W + px] == 0 && angles[(size_t)py * W + px] != NOTDEF) {
                double reg_angle;
                region_grow(px, py, reg, reg_angle, prec);
                if (reg.size() < min_reg_size) continue;
                Rect rec;
                region2rect(reg, reg_angle, prec, p, rec);
                if (!refine(reg, reg_angle, prec, p, rec, DENSITY_TH)) continue;
                const double log_nfa = rect_improve(rec);
                if (log_nfa <= LOG_EPS) continue;
                rec.x1 += 0.5; rec.y1 += 0.5; rec.x2 += 0.5; rec.y2 += 0.5;
                rec.x1 /= SCALE; rec.y1 /= SCALE; rec.x2 /= SCALE; rec.y2 /= SCALE; rec.width /= SCALE;
                lines.push_back(float(rec.x1)); lines.push_back(float(rec.y1));
                lines.push_back(float(rec.x2)); lines.push_back(float(rec.y2));
                /* what LineSegmentDetector::detect also reports per segment (LSD_REFINE_ADV): width, precision, -log10(NFA) */
                if (st) { st->segInfo.push_back(rec.width); st->segInfo.push_back(rec.p); st->segInfo.push_back(log_nfa); }
            }
        }
    }
};

static const int kComb[32][2] = {{0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5}, {0, 6}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {1, 6},
                                 {2, 3}, {2, 4}, {2, 5}, {2, 6}, {2, 7}, {2, 8}, {3, 4}, {3, 5}, {3, 6}, {3, 7}, {3, 8},
                                 {4, 5}, {4, 6}, {4, 7}, {4, 8}, {5, 6}, {5, 7}, {5, 8}, {6, 7}, {6, 8}, {7, 8}};

} // namespace

/* BinaryDescriptor::computeLBD for one line on octave 0 + binaryConversion */
void lbd_descriptor(const int16_t* dxImg, const int16_t* dyImg, int realWidth, int realHeight, const KeyLine& kl,
                    float* desVec72, uint8_t* desc32)
{
    const int NUM_OF_BANDS = 9, widthOfBand = 7;
    static double gaussCoefL[21], gaussCoefG[63];
    static bool init = false;
    if (!init) {
        double u = (widthOfBand * 3 - 1) / 2;
        double sigma = (widthOfBand * 2 + 1) / 2;
        double invsigma2 = -1 / (2 * sigma * sigma);
        for (int i = 0; i < widthOfBand * 3; i++) { const double dis = i - u; gaussCoefL[i] = std::exp(dis * dis * invsigma2); }
        u = (NUM_OF_BANDS * widthOfBand - 1) / 2;
        sigma = u;
        invsigma2 = -1 / (2 * sigma * sigma);
        for (int i = 0; i < NUM_OF_BANDS * widthOfBand; i++) { const double dis = i - u; gaussCoefG[i] = std::exp(dis * dis * invsigma2); }
        init = true;
    }
    const short heightOfLSP = (short)(widthOfBand * NUM_OF_BANDS);
    const short descriptor_size = NUM_OF_BANDS * 8;
    float band[8][9];
    std::memset(band, 0, sizeof(band));   /* pgdL ngdL pgdL2 ngdL2 pgdO ngdO pgdO2 ngdO2 */
    const short imageWidth = (short)(realWidth - 1), imageHeight = (short)(realHeight - 1);
    const short lengthOfLSP = (short)kl.numOfPixels;
    const short halfHeight = (heightOfLSP - 1) / 2;
    const short halfWidth = (lengthOfLSP - 1) / 2;
    const float lineMiddlePointX = (float)(0.5 * (kl.sPointInOctaveX + kl.ePointInOctaveX));
    const float lineMiddlePointY = (float)(0.5 * (kl.sPointInOctaveY + kl.ePointInOctaveY));
    float dL[2], dO[2];
    dL[0] = (float)std::cos((double)kl.angle);   /* unqualified C cos()/sin() of the contrib source: double versions */
    dL[1] = (float)std::sin((double)kl.angle);
    dO[0] = -dL[1];
    dO[1] = dL[0];
    float sCorX0 = -dL[0] * halfWidth + dL[1] * halfHeight + lineMiddlePointX;
    float sCorY0 = -dL[1] * halfWidth - dL[0] * halfHeight + lineMiddlePointY;
    for (short hID = 0; hID < heightOfLSP; hID++) {
        float sCorX = sCorX0, sCorY = sCorY0;
        float pgdLRowSum = 0, ngdLRowSum = 0, pgdORowSum = 0, ngdORowSum = 0;
        for (short wID = 0; wID < lengthOfLSP; wID++) {
            short tempCor = (short)std::round(sCorX);
            const short xCor = (tempCor < 0) ? 0 : (tempCor > imageWidth) ? imageWidth : tempCor;
            tempCor = (short)std::round(sCorY);
            const short yCor = (tempCor < 0) ? 0 : (tempCor > imageHeight) ? imageHeight : tempCor;
            const short dx = dxImg[yCor * realWidth + xCor], dy = dyImg[yCor * realWidth + xCor];
            const float gDL = dx * dL[0] + dy * dL[1];
            const float gDO = dx * dO[0] + dy * dO[1];
            if (gDL > 0) pgdLRowSum += gDL; else ngdLRowSum -= gDL;
            if (gDO > 0) pgdORowSum += gDO; else ngdORowSum -= gDO;
            sCorX += dL[0];
            sCorY += dL[1];
        }
        sCorX0 -= dL[1];
        sCorY0 += dL[0];
        float coef = (float)gaussCoefG[hID];
        pgdLRowSum = coef * pgdLRowSum; ngdLRowSum = coef * ngdLRowSum;
        const float pgdL2RowSum = pgdLRowSum * pgdLRowSum, ngdL2RowSum = ngdLRowSum * ngdLRowSum;
        pgdORowSum = coef * pgdORowSum; ngdORowSum = coef * ngdORowSum;
        const float pgdO2RowSum = pgdORowSum * pgdORowSum, ngdO2RowSum = ngdORowSum * ngdORowSum;
        const float rows[8] = {pgdLRowSum, ngdLRowSum, pgdL2RowSum, ngdL2RowSum, pgdORowSum, ngdORowSum, pgdO2RowSum, ngdO2RowSum};
        auto add = [&](short bandID, float c) {
            for (int q = 0; q < 8; q++) band[q][bandID] += ((q & 2) ? c * c : c) * rows[q];
        };
        short bandID = (short)(hID / widthOfBand);
        add(bandID, (float)gaussCoefL[hID % widthOfBand + widthOfBand]);
        bandID--;
        if (bandID >= 0) add(bandID, (float)gaussCoefL[hID % widthOfBand + 2 * widthOfBand]);
        bandID = bandID + 2;
        if (bandID < NUM_OF_BANDS) add(bandID, (float)gaussCoefL[hID % widthOfBand]);
    }
    const float invN2 = (float)(1.0 / (widthOfBand * 2.0)), invN3 = (float)(1.0 / (widthOfBand * 3.0));
    for (short b = 0; b < NUM_OF_BANDS; b++) {
        const float invN = (b == 0 || b == NUM_OF_BANDS - 1) ? invN2 : invN3;
        const short desID = b * 8;
        float temp = band[0][b] * invN;
        desVec72[desID] = temp;
        desVec72[desID + 4] = std::sqrt(band[2][b] * invN - temp * temp);
        temp = band[1][b] * invN;
        desVec72[desID + 1] = temp;
        desVec72[desID + 5] = std::sqrt(band[3][b] * invN - temp * temp);
        temp = band[4][b] * invN;
        desVec72[desID + 2] = temp;
        desVec72[desID + 6] = std::sqrt(band[6][b] * invN - temp * temp);
        temp = band[5][b] * invN;
        desVec72[desID + 3] = temp;
        desVec72[desID + 7] = std::sqrt(band[7][b] * invN - temp * temp);
    }
    float tempM = 0, tempS = 0;
    for (int b = 0; b < NUM_OF_BANDS; b++) {
        const float* d = desVec72 + 8 * b;
        tempM += d[0] * d[0]; tempM += d[1] * d[1]; tempM += d[2] * d[2]; tempM += d[3] * d[3];
        tempS += d[4] * d[4]; tempS += d[5] * d[5]; tempS += d[6] * d[6]; tempS += d[7] * d[7];
    }
    tempM = 1 / std::sqrt(tempM);
    tempS = 1 / std::sqrt(tempS);
    for (int b = 0; b < NUM_OF_BANDS; b++) {
        float* d = desVec72 + 8 * b;
        d[0] *= tempM; d[1] *= tempM; d[2] *= tempM; d[3] *= tempM;
        d[4] *= tempS; d[5] *= tempS; d[6] *= tempS; d[7] *= tempS;
    }
    for (short i = 0; i < descriptor_size; i++)
        if (desVec72[i] > 0.4) desVec72[i] = (float)0.4;
    float temp = 0;
    for (short i = 0; i < descriptor_size; i++) temp += desVec72[i] * desVec72[i];
    temp = 1 / std::sqrt(temp);
    for (short i = 0; i < descriptor_size; i++) desVec72[i] = desVec72[i] * temp;
    for (int c = 0; c < 32; c++) {
        const float* f1 = &desVec72[8 * kComb[c][0]];
        const float* f2 = &desVec72[8 * kComb[c][1]];
        uint8_t r = 0;
        for (int i = 0; i < 8; i++)
            if (f1[i] > f2[i]) r += (uint8_t)(1 << i);
        desc32[c] = r;
    }
}

/* LineSegment::ExtractLineSegment, reference src/LSDextractor.cpp:12-43 */
LineResult extract_lines(const uint8_t* img, int w, int h, int maxLines, LsdStages* stages, int rectMode)
{
    LineResult out;
    /* LSDDetector::detect(img, keylines, scale = 1 (int from 1.2f), numOctaves = 1): octave 0 = the image */
    Lsd lsd;
    lsd.rect_mode = rectMode;
    if (stages) lsd.count_log = &stages->rectCounts;
    std::vector<float> segs;
    lsd.detect(img, w, h, segs, stages);
    if (stages) stages->segments = segs;
    int class_counter = -1;
    for (size_t k = 0; k + 3 < segs.size(); k += 4) {
        float e[4] = {segs[k], segs[k + 1], segs[k + 2], segs[k + 3]};
        /* checkLineExtremes */
        if (e[0] < 0) e[0] = 0;
        if (e[0] >= w) e[0] = (float)w - 1.0f;
        if (e[2] < 0) e[2] = 0;
        if (e[2] >= w) e[2] = (float)w - 1.0f;
        if (e[1] < 0) e[1] = 0;
        if (e[1] >= h) e[1] = (float)h - 1.0f;
        if (e[3] < 0) e[3] = 0;
        if (e[3] >= h) e[3] = (float)h - 1.0f;
        KeyLine kl;
        const float octaveScale = 1.0f;
        kl.startPointX = e[0] * octaveScale; kl.startPointY = e[1] * octaveScale;
        kl.endPointX = e[2] * octaveScale; kl.endPointY = e[3] * octaveScale;
        kl.sPointInOctaveX = e[0]; kl.sPointInOctaveY = e[1]; kl.ePointInOctaveX = e[2]; kl.ePointInOctaveY = e[3];
        kl.lineLength = (float)std::sqrt(std::pow(e[0] - e[2], 2) + std::pow(e[1] - e[3], 2));
        /* LineIterator(img, Point(e0,e1), Point(e2,e3)).count, 8-connected */
        const int x0 = round_he(e[0]), y0 = round_he(e[1]);
        const int x1 = round_he(e[2]), y1 = round_he(e[3]);
        kl.numOfPixels = std::max(std::abs(x1 - x0), std::abs(y1 - y0)) + 1;
        kl.angle = (float)std::atan2((double)(kl.endPointY - kl.startPointY), (double)(kl.endPointX - kl.startPointX));
        kl.class_id = ++class_counter;
        kl.octave = 0;
        kl.size = (kl.endPointX - kl.startPointX) * (kl.endPointY - kl.startPointY);
        kl.response = kl.lineLength / std::max(w, h);
        kl.ptX = (kl.endPointX + kl.startPointX) / 2; kl.ptY = (kl.endPointY + kl.startPointY) / 2;
        out.lines.push_back(kl);
    }
    out.detected = (int)out.lines.size();
    if ((int)out.lines.size() > maxLines) {
        std::sort(out.lines.begin(), out.lines.end(), [](const KeyLine& a, const KeyLine& b) { return a.response > b.response; });
        out.lines.resize(maxLines);
        for (int i = 0; i < maxLines; i++) out.lines[i].class_id = i;
    }
    /* BinaryDescriptor::compute: octave image = GaussianBlur(5x5, sigma 1), Sobel 3x3 -> CV_16S */
    if (!out.lines.empty()) {
        std::vector<uint8_t> blur((size_t)w * h);
        gaussian_blur_q8(img, w, h, gauss_taps_q8(5, 1.0), blur.data());
        std::vector<int16_t> gx((size_t)w * h), gy((size_t)w * h);
        sobel3_s16(blur.data(), w, h, gx.data(), gy.data());
        if (stages) { stages->gx = gx; stages->gy = gy; }
        out.desc.assign(out.lines.size() * 32, 0);
        out.descf.assign(out.lines.size() * 72, 0.f);
        for (size_t i = 0; i < out.lines.size(); i++)
            lbd_descriptor(gx.data(), gy.data(), w, h, out.lines[i], &out.descf[i * 72], &out.desc[i * 32]);
    }
    /* keylineFunctions: normalised cross product of the homogeneous end points (:32-42) */
    for (const KeyLine& kl : out.lines) {
        const double sx = kl.startPointX, sy = kl.startPointY, ex = kl.endPointX, ey = kl.endPointY;
        double l0 = sy * 1.0 - 1.0 * ey, l1 = 1.0 * ex - sx * 1.0, l2 = sx * ey - sy * ex;
        const double nrm = std::sqrt(l0 * l0 + l1 * l1 + l2 * l2);
        out.lineF.push_back(l0 / nrm); out.lineF.push_back(l1 / nrm); out.lineF.push_back(l2 / nrm);
    }
    return out;
}

} // namespace orc
