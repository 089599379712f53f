/* lines_3d.cpp — Frame::isLineGood (reference src/Frame.cc:481-558) and the 3-D line lifting helpers it calls
 * (src/LineExtractor.cpp: depthStdDev :1180, compPt3dCov :1196, extract3dline_mahdist :1266, verify3dLine :1362,
 * mah_dist3d_pt_line :1419, computeLine3d_svd :1157, projectPt3d2Ln3d :278).  SURVEY.md row a-8: <= 40 lines x
 * <= 51 samples of double arithmetic with a rand()-driven RANSAC — host code, no device work.
 *
 * What the reference really computes.  isLineGood receives mK, a CV_32F matrix, and compPt3dCov reads it with
 * K.at<double>(0,0): the eight bytes of (fx, 0.0f) reinterpreted as a double are a subnormal ~5.6e-315, so
 * J0(0,0) = z / f overflows to +inf for every valid depth, J0 * diag(1,1,sigma^2) * J0^T contains inf * 0 = NaN in
 * every row and column, cv::SVD of it has NaN singular values, every Mahalanobis distance is NaN, `dist < 1.5` is
 * never true, no sample is an inlier and the line is rejected: the shipped function leaves mvDepthLine = -1 and
 * mvLines3D = 0 for every line.  k_as_f64 = 0 reproduces exactly that by doing the same arithmetic on the same
 * bytes (nothing is hard-coded); k_as_f64 = 1 runs the algorithm as it was evidently meant, with f = fx.  In that
 * mode the two cv::SVD calls are replaced by a symmetric 3x3 eigen-solve (the Mahalanobis distance is invariant
 * to the sign and order of the singular vectors), so results agree with an OpenCV build to rounding, not to the
 * bit, and the endpoint order (A, B) may be swapped; rand() is glibc's TYPE_3 generator restated, seeded per call
 * because the reference shares the process-wide state between its extraction threads. */
#include "../../include/drfe.h"
#include "ahc_math.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace {

/* glibc random_r TYPE_3 (x^31 + x^3 + 1), what rand() runs: r[i] = r[i-31] + r[i-3], output >> 1; the seed
 * expands through the 16807 Lehmer step and the first 310 outputs are discarded. */
struct GlibcRand {
    uint32_t r[34];
    int pos;
    explicit GlibcRand(uint32_t seed)
    {
        std::vector<uint32_t> t(344);
        int32_t w = seed ? (int32_t)seed : 1;
        t[0] = (uint32_t)w;
        for (int i = 1; i < 31; i++) {
            const int32_t hi = w / 127773, lo = w % 127773;
            w = 16807 * lo - 2836 * hi;
            if (w < 0) w += 2147483647;
            t[i] = (uint32_t)w;
        }
        for (int i = 31; i < 34; i++) t[i] = t[i - 31];
        for (int i = 34; i < 344; i++) t[i] = t[i - 31] + t[i - 3];
        for (int i = 0; i < 34; i++) r[i] = t[310 + i];   /* the last 34 values: enough history for i-31 */
        pos = 0;
    }
    int next()
    {
        /* ring of 34: newest at (pos+33)%34; r[i-31] is 31 back from the new element, r[i-3] three back */
        const uint32_t v = r[(pos + 34 - 31) % 34] + r[(pos + 34 - 3) % 34];
        r[pos] = v;
        pos = (pos + 1) % 34;
        return (int)(v >> 1);
    }
};

struct P3 { double x, y, z; };
inline P3 operator-(const P3& a, const P3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline P3 operator+(const P3& a, const P3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline P3 operator*(const P3& a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(const P3& a, const P3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline double norm(const P3& a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }

struct RandomPoint3d {
    P3 pos;
    double DU[9];
};

double depth_std_dev(double d) { return 0.00273 * d * d + 0.00074 * d + (-0.00058); }

/* compPt3dCov: cov0 = J0 * diag(1, 1, sigma^2) * J0^T as two plain 3x3 products (every term kept, so inf * 0 is
 * NaN exactly where OpenCV's gemm makes it), then DU = diag(1/sqrt(w)) * U^T from the decomposition of cov0 */
RandomPoint3d comp_pt3d_cov(const P3& pt, double f)
{
    RandomPoint3d rp;
    rp.pos = pt;
    const double J[3][3] = {{pt.z / f, 0, pt.x / pt.z}, {0, pt.z / f, pt.y / pt.z}, {0, 0, 1}};
    const double s = depth_std_dev(pt.z) * depth_std_dev(pt.z);
    const double C[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, s}};
    double M[3][3], cov[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) M[i][j] = J[i][0] * C[0][j] + J[i][1] * C[1][j] + J[i][2] * C[2][j];
    bool finite = true;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            cov[i][j] = M[i][0] * J[j][0] + M[i][1] * J[j][1] + M[i][2] * J[j][2];
            finite = finite && std::isfinite(cov[i][j]);
        }
    if (!finite) {   /* Jacobi SVD of a matrix with a NaN in every column: all singular values NaN -> DU all NaN */
        for (double& v : rp.DU) v = std::numeric_limits<double>::quiet_NaN();
        return rp;
    }
    double ev[3], Q[9];
    ahc_eig3(cov[0][0], cov[1][0], cov[2][0], cov[1][1], cov[2][1], cov[2][2], ev, Q);
    for (int i = 0; i < 3; i++) {            /* singular values descending = eigenvalues descending */
        const int e = 2 - i;
        const double ws = std::sqrt(ev[e]);
        for (int c = 0; c < 3; c++) rp.DU[i * 3 + c] = (1 / ws) * Q[e * 3 + c];
    }
    return rp;
}

/* mah_dist3d_pt_line, src/LineExtractor.cpp:1419-1470, term by term */
double mah_dist3d_pt_line(const RandomPoint3d& pt, const P3& q1, const P3& q2)
{
    const double xa = q1.x, ya = q1.y, za = q1.z, xb = q2.x, yb = q2.y, zb = q2.z;
    const double c1 = pt.DU[0], c2 = pt.DU[1], c3 = pt.DU[2], c4 = pt.DU[3], c5 = pt.DU[4], c6 = pt.DU[5], c7 = pt.DU[6],
                 c8 = pt.DU[7], c9 = pt.DU[8];
    const double x1 = pt.pos.x, x2 = pt.pos.y, x3 = pt.pos.z;
    const double a1 = c1 * (x1 - xa) + c2 * (x2 - ya) + c3 * (x3 - za), b1 = c1 * (x1 - xb) + c2 * (x2 - yb) + c3 * (x3 - zb);
    const double a2 = c4 * (x1 - xa) + c5 * (x2 - ya) + c6 * (x3 - za), b2 = c4 * (x1 - xb) + c5 * (x2 - yb) + c6 * (x3 - zb);
    const double a3 = c7 * (x1 - xa) + c8 * (x2 - ya) + c9 * (x3 - za), b3 = c7 * (x1 - xb) + c8 * (x2 - yb) + c9 * (x3 - zb);
    const double term1 = a1 * b2 - a2 * b1, term2 = a1 * b3 - a3 * b1, term3 = a2 * b3 - a3 * b2;
    const double term4 = c1 * (x1 - xa) - c1 * (x1 - xb) + c2 * (x2 - ya) - c2 * (x2 - yb) + c3 * (x3 - za) - c3 * (x3 - zb);
    const double term5 = c4 * (x1 - xa) - c4 * (x1 - xb) + c5 * (x2 - ya) - c5 * (x2 - yb) + c6 * (x3 - za) - c6 * (x3 - zb);
    const double term6 = c7 * (x1 - xa) - c7 * (x1 - xb) + c8 * (x2 - ya) - c8 * (x2 - yb) + c9 * (x3 - za) - c9 * (x3 - zb);
    return std::sqrt((term1 * term1 + term2 * term2 + term3 * term3) / (term4 * term4 + term5 * term5 + term6 * term6));
}

P3 project_pt_to_line(const P3& P, const P3& mid, const P3& drct)
{
    const P3 A = mid, B = mid + drct, AB = B - A, AP = P - A;
    return A + AB * (dot(AB, AP) / dot(AB, AB));
}

/* verify3dLine, :1362-1417: the inliers must populate more than 7 of the 10 cells between their extremities */
bool verify_3d_line(const std::vector<RandomPoint3d>& pts, const P3& A, const P3& B)
{
    const double EPS = 1e-10;
    int cells[10] = {0};
    double minv = 100, maxv = -100;
    int idx1 = 0, idx2 = 0;
    for (int i = 0; i < (int)pts.size(); i++) {
        const double v = dot(pts[i].pos - A, B - A);
        if (v < minv) { minv = v; idx1 = i; }
        if (v > maxv) { maxv = v; idx2 = i; }
    }
    const P3 C = project_pt_to_line(pts[idx1].pos, (A + B) * 0.5, B - A);
    const P3 D = project_pt_to_line(pts[idx2].pos, (A + B) * 0.5, B - A);
    const double cd = norm(D - C);
    if (cd < EPS) return false;
    for (const RandomPoint3d& p : pts) {
        const double lambda = std::fabs(dot(p.pos - C, D - C) / cd / cd);
        if (lambda >= 1) cells[9] += 1;
        else cells[(unsigned)std::floor(lambda * 10)] += 1;
    }
    double sum = 0;
    for (int c : cells)
        if (c > 0) sum = sum + 1;
    return sum / 10 > 0.7;
}

/* computeLine3d_svd, :1157-1178: mean and principal direction of the selected points */
void compute_line3d(const std::vector<RandomPoint3d>& pts, const std::vector<int>& idx, P3& mean, P3& drct)
{
    const int n = (int)idx.size();
    mean = {0, 0, 0};
    for (int i : idx) mean = mean + pts[i].pos;
    mean = mean * (1.0 / n);
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int i : idx) {
        const P3 d = pts[i].pos - mean;
        s[0] += d.x * d.x; s[1] += d.y * d.x; s[2] += d.z * d.x; s[3] += d.y * d.y; s[4] += d.z * d.y; s[5] += d.z * d.z;
    }
    double ev[3], Q[9];
    ahc_eig3(s[0], s[1], s[2], s[3], s[4], s[5], ev, Q);
    drct = {Q[6], Q[7], Q[8]};      /* eigenvector of the largest eigenvalue = first right singular vector */
}

struct Line3d { P3 A{0, 0, 0}, B{0, 0, 0}; int nInliers = 0; };

/* extract3dline_mahdist, :1266-1360 */
Line3d extract_3d_line(const std::vector<RandomPoint3d>& pts, GlibcRand& rng)
{
    const double EPS = 1e-10, distThresh = 1.5;
    const int n = (int)pts.size();
    const int maxIterNo = std::min(10, (int)(n * (n - 1) * 0.5));
    std::vector<int> indexes(n);
    for (int i = 0; i < n; i++) indexes[i] = i;
    std::vector<int> maxInlierSet;
    P3 bestA{0, 0, 0}, bestB{0, 0, 0};
    for (int iter = 0; iter < maxIterNo; iter++) {
        /* random_unique(begin, end, 2), include/LSDextractor.h:241-251: two Fisher-Yates steps */
        size_t left = indexes.size();
        for (int k = 0; k < 2; k++) {
            const size_t r = (size_t)rng.next() % left;
            std::swap(indexes[k], indexes[k + r]);
            --left;
        }
        const RandomPoint3d& A = pts[indexes[0]];
        const RandomPoint3d& B = pts[indexes[1]];
        if (norm(B.pos - A.pos) < EPS) continue;
        std::vector<int> inlierSet;
        for (int i = 0; i < n; i++)
            if (mah_dist3d_pt_line(pts[i], A.pos, B.pos) < distThresh) inlierSet.push_back(i);
        if (inlierSet.size() > maxInlierSet.size()) {
            std::vector<RandomPoint3d> inlierPts(inlierSet.size());
            for (size_t ii = 0; ii < inlierSet.size(); ii++) inlierPts[ii] = pts[inlierSet[ii]];
            if (verify_3d_line(inlierPts, A.pos, B.pos)) {
                maxInlierSet = inlierSet;
                bestA = A.pos; bestB = B.pos;
            }
        }
        if (maxInlierSet.size() > n * 0.6) break;
    }
    Line3d rl;
    if (maxInlierSet.size() >= 2) {
        P3 m = (bestA + bestB) * 0.5, d = bestB - bestA;
        while (true) {   /* refit on the inliers and reselect while the set grows */
            std::vector<int> tmp;
            P3 tm, td;
            compute_line3d(pts, maxInlierSet, tm, td);
            for (int i = 0; i < n; i++)
                if (mah_dist3d_pt_line(pts[i], tm, tm + td) < distThresh) tmp.push_back(i);
            if (tmp.size() > maxInlierSet.size()) { maxInlierSet = tmp; m = tm; d = td; }
            else break;
        }
        double minv = 100, maxv = -100;
        int e1 = 0, e2 = 0;
        for (size_t i = 0; i < maxInlierSet.size(); i++) {
            const double dp = dot(pts[maxInlierSet[i]].pos - m, d);
            if (dp < minv) { minv = dp; e1 = (int)i; }
            if (dp > maxv) { maxv = dp; e2 = (int)i; }
        }
        rl.A = pts[maxInlierSet[e1]].pos;
        rl.B = pts[maxInlierSet[e2]].pos;
    }
    rl.nInliers = (int)maxInlierSet.size();
    return rl;
}

} // namespace

extern "C" int drfe_lines_is_good(const drfe_keyline* lines, int n, const float* depth, int w, int h, size_t stride,
                                  const float* K, int k_as_f64, float cx, float cy, float invfx, float invfy,
                                  uint32_t seed, float* depth_line, double* lines3d, int32_t* n_inliers, int* n_good)
{
    if (n < 0 || (n && (!lines || !depth_line || !lines3d)) || !depth || !K || w < 1 || h < 1 || stride < (size_t)w)
        return DRFE_ERR_INVALID;
    double f;
    if (k_as_f64) f = (double)K[0];
    else std::memcpy(&f, K, sizeof(double));           /* K.at<double>(0,0) on CV_32F storage: bytes of (K[0], K[1]) */
    GlibcRand rng(seed);
    int good = 0;
    for (int i = 0; i < n; i++) {
        depth_line[i] = -1.0f;
        for (int k = 0; k < 6; k++) lines3d[6 * i + k] = 0.0;
        if (n_inliers) n_inliers[i] = 0;
        const drfe_keyline& kl = lines[i];
        const float dxs = kl.start_point_x - kl.end_point_x, dys = kl.start_point_y - kl.end_point_y;
        const double len = std::sqrt((double)dxs * dxs + (double)dys * dys);      /* cv::norm(Point2f) */
        const double numSmp = (double)std::min((int)len, 50);
        if (!(numSmp >= 1)) continue;                  /* a sub-pixel line: the reference divides 0 by 0 here */
        std::vector<P3> pts3d;
        for (int j = 0; j <= numSmp; ++j) {
            /* Point2f * double -> Point2f (saturate_cast<float> of the double product), Point2f + Point2f */
            const double t = j / numSmp;
            const float px = (float)((double)kl.start_point_x * (1 - t)) + (float)((double)kl.end_point_x * t);
            const float py = (float)((double)kl.start_point_y * (1 - t)) + (float)((double)kl.end_point_y * t);
            const double x = px, y = py;
            if (x < 0 || y < 0 || x >= w || y >= h) continue;
            int row, col;
            if (std::floor(x) == x && std::floor(y) == y) { col = std::max((int)(x - 1), 0); row = std::max((int)(y - 1), 0); }
            else { col = (int)x; row = (int)y; }
            const float d = depth[(size_t)row * stride + col];
            if ((double)d <= 0.01) continue;
            P3 p;
            p.z = d;
            p.x = (double)((float)col - cx) * p.z * (double)invfx;
            p.y = (double)((float)row - cy) * p.z * (double)invfy;
            pts3d.push_back(p);
        }
        if (pts3d.size() < 10) continue;
        std::vector<RandomPoint3d> rnd;
        rnd.reserve(pts3d.size());
        for (const P3& p : pts3d) rnd.push_back(comp_pt3d_cov(p, f));
        const Line3d ln = extract_3d_line(rnd, rng);
        if (n_inliers) n_inliers[i] = ln.nInliers;
        if (ln.nInliers / len > 0.4 && norm(ln.A - ln.B) > 0.02) {
            const int ex = (int)kl.end_point_x, ey = (int)kl.end_point_y, sx = (int)kl.start_point_x, sy = (int)kl.start_point_y;
            const bool in = ex >= 0 && ex < w && ey >= 0 && ey < h && sx >= 0 && sx < w && sy >= 0 && sy < h;
            depth_line[i] = in ? std::min(depth[(size_t)ey * stride + ex], depth[(size_t)sy * stride + sx]) : -1.0f;
            lines3d[6 * i + 0] = ln.A.x; lines3d[6 * i + 1] = ln.A.y; lines3d[6 * i + 2] = ln.A.z;
            lines3d[6 * i + 3] = ln.B.x; lines3d[6 * i + 4] = ln.B.y; lines3d[6 * i + 5] = ln.B.z;
            good++;
        }
    }
    if (n_good) *n_good = good;
    return DRFE_OK;
}
