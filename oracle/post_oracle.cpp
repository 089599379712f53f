/* oracle/post_oracle.cpp — TEST INFRASTRUCTURE.  CPU restatement of the plane post-processing and the surface normals of
 * Frame::ComputePlanes (reference src/Frame.cc:949-1094) and Frame::MaxPointDistanceFromPlane (src/Frame.cc:1222-1307).
 *
 * The arithmetic of this path lives in PCL 1.9 (CMakeLists.txt:59; not vendored, not in this image): pcl::VoxelGrid,
 * pcl::SACSegmentation (SACMODEL_PLANE / SAC_RANSAC, optimize on) and pcl::IntegralImageNormalEstimation
 * (AVERAGE_3D_GRADIENT, MaxDepthChangeFactor 0.05, NormalSmoothingSize 10).  They are restated here from the published
 * PCL 1.9.1 sources (filters/impl/voxel_grid.hpp, sample_consensus/impl/{ransac,sac_model_plane}.hpp,
 * sample_consensus/sac_model.h, common/impl/centroid.hpp, common/impl/eigen.hpp, features/impl/
 * integral_image_normal.hpp, features/impl/integral_image2D.hpp) — PARITY UNPINNED: nothing in the reference pins
 * these results and PCL cannot be built here.
 *
 * Canonical choices where PCL's result depends on the host it was compiled for (SURVEY.md section 9 style):
 *  - Eigen float reductions of 4 elements (VectorXf::normalize, dot) are evaluated left to right, no FMA;
 *  - pcl::computeRoots' float sqrt/atan2/cos/sin are evaluated in double and rounded once to float;
 *  - pcl::VoxelGrid orders the points of a leaf with std::sort on the leaf index alone (unstable): the order libstdc++'s
 *    introsort leaves IS the reference's behaviour on a libstdc++ host and is kept (std::sort on the same records);
 *  - the sample consensus RNG is boost::mt19937 seeded 12345 behind boost::uniform_int<>(0, INT_MAX), i.e. the 32-bit
 *    Mersenne twister output shifted right by one.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct P3 { float x, y, z; };

/* ---- pcl::VoxelGrid<PointXYZRGB>::applyFilter, leaf (l, l, l), downsample_all_data (xyz part) ------------------ */
struct LeafRec {
    unsigned idx; unsigned pt;
    bool operator<(const LeafRec& o) const { return idx < o.idx; }
};

int voxel_grid(const P3* in, int n, float leaf, std::vector<P3>& out)
{
    out.clear();
    if (n <= 0) return 0;
    const float inv = 1.0f / leaf;                       /* inverse_leaf_size_ = Array4f::Ones() / leaf_size_ */
    float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    float mx[3] = {-mn[0], -mn[0], -mn[0]};
    for (int i = 0; i < n; i++) {                        /* getMinMax3D (dense cloud) */
        const float v[3] = {in[i].x, in[i].y, in[i].z};
        for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], v[k]); mx[k] = std::max(mx[k], v[k]); }
    }
    const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv) + 1,
                  dz = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (dx * dy * dz > (int64_t)std::numeric_limits<int32_t>::max()) {   /* "leaf size too small": output = input */
        out.assign(in, in + n);
        return n;
    }
    int minb[3], maxb[3], divb[3];
    for (int k = 0; k < 3; k++) {
        minb[k] = (int)std::floor(mn[k] * inv);
        maxb[k] = (int)std::floor(mx[k] * inv);
        divb[k] = maxb[k] - minb[k] + 1;
    }
    const int mul[3] = {1, divb[0], divb[0] * divb[1]};
    std::vector<LeafRec> iv;
    iv.reserve(n);
    for (int i = 0; i < n; i++) {
        const int i0 = (int)(std::floor(in[i].x * inv) - (float)minb[0]);
        const int i1 = (int)(std::floor(in[i].y * inv) - (float)minb[1]);
        const int i2 = (int)(std::floor(in[i].z * inv) - (float)minb[2]);
        iv.push_back(LeafRec{(unsigned)(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]), (unsigned)i});
    }
    std::sort(iv.begin(), iv.end(), std::less<LeafRec>());
    size_t a = 0;
    while (a < iv.size()) {
        size_t b = a + 1;
        while (b < iv.size() && iv[b].idx == iv[a].idx) b++;
        /* CentroidPoint<PointXYZRGB>: AccumulatorXYZ adds getVector3fMap() in float, get() divides by the count */
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (size_t k = a; k < b; k++) { sx += in[iv[k].pt].x; sy += in[iv[k].pt].y; sz += in[iv[k].pt].z; }
        const float cnt = (float)(b - a);
        out.push_back(P3{sx / cnt, sy / cnt, sz / cnt});
        a = b;
    }
    return (int)out.size();
}

/* ---- boost::mt19937 + uniform_int<>(0, INT_MAX) ------------------------------------------------------------------ */
struct Mt19937 {
    uint32_t mt[624]; int pos;
    explicit Mt19937(uint32_t seed)
    {
        mt[0] = seed;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        pos = 624;
    }
    uint32_t next()
    {
        if (pos >= 624) {
            for (int i = 0; i < 624; i++) {
                const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7FFFFFFFu);
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
            }
            pos = 0;
        }
        uint32_t y = mt[pos++];
        y ^= y >> 11; y ^= (y << 7) & 0x9D2C5680u; y ^= (y << 15) & 0xEFC60000u; y ^= y >> 18;
        return y;
    }
    int rnd() { return (int)(next() >> 1); }             /* generate_uniform_int: bucket size 2 over [0, 2^32) */
};

inline float dot4(const float c[4], const P3& p) { return ((c[0] * p.x + c[1] * p.y) + c[2] * p.z) + c[3] * 1.0f; }

/* SampleConsensusModelPlane::computeModelCoefficients */
bool plane_from_samples(const P3* pts, const int s[3], float c[4])
{
    const P3 &p0 = pts[s[0]], &p1 = pts[s[1]], &p2 = pts[s[2]];
    const float a[4] = {p1.x - p0.x, p1.y - p0.y, p1.z - p0.z, 0.f};   /* data[3] = 1 on both sides */
    const float b[4] = {p2.x - p0.x, p2.y - p0.y, p2.z - p0.z, 0.f};
    const float r0 = a[0] / b[0], r1 = a[1] / b[1], r2 = a[2] / b[2];
    if (r0 == r1 && r2 == r1) return false;
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
    c[3] = 0.f;
    const float z = ((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]) + c[3] * c[3];
    if (z > 0.f) { const float nrm = std::sqrt(z); for (int k = 0; k < 4; k++) c[k] /= nrm; }
    c[3] = -1.f * (((c[0] * p0.x + c[1] * p0.y) + c[2] * p0.z) + c[3] * 1.0f);
    return true;
}

bool sample_good(const P3* pts, const int s[3])
{
    const P3 &p0 = pts[s[0]], &p1 = pts[s[1]], &p2 = pts[s[2]];
    const float r0 = (p1.x - p0.x) / (p2.x - p0.x), r1 = (p1.y - p0.y) / (p2.y - p0.y), r2 = (p1.z - p0.z) / (p2.z - p0.z);
    return (r0 != r1) || (r2 != r1);
}

/* pcl::computeRoots (common/impl/eigen.hpp), Scalar = float; the transcendental steps in double, rounded once */
void compute_roots2(float b, float c, float roots[3])
{
    roots[0] = 0.f;
    float d = (float)((double)(b * b) - 4.0 * (double)c);       /* Scalar (b * b - 4.0 * c): the subtraction is a double one */
    if (d < 0.0f) d = 0.0f;
    const float sd = std::sqrt(d);
    roots[2] = 0.5f * (b + sd);
    roots[1] = 0.5f * (b - sd);
}
void compute_roots(const float m[9], float roots[3])
{
    const float c0 = m[0] * m[4] * m[8] + 2.0f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] - m[8] * m[1] * m[1];
    const float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
    const float c2 = m[0] + m[4] + m[8];
    if (std::fabs(c0) < std::numeric_limits<float>::epsilon()) { compute_roots2(c2, c1, roots); return; }
    const float s_inv3 = 1.0f / 3.0f;
    const float s_sqrt3 = (float)std::sqrt(3.0);
    const float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.0f) a_over_3 = 0.0f;
    const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.0f) q = 0.0f;
    const float rho = (float)std::sqrt((double)-a_over_3);
    const float theta = (float)std::atan2((double)(float)std::sqrt((double)-q), (double)half_b) * s_inv3;
    const float cos_theta = (float)std::cos((double)theta), sin_theta = (float)std::sin((double)theta);
    roots[0] = c2_over_3 + 2.0f * rho * cos_theta;
    roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]);
    if (roots[1] >= roots[2]) {
        std::swap(roots[1], roots[2]);
        if (roots[0] >= roots[1]) std::swap(roots[0], roots[1]);
    }
    if (roots[0] <= 0.0f) compute_roots2(c2, c1, roots);
}

inline void cross3(const float a[3], const float b[3], float o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

/* SampleConsensusModelPlane::optimizeModelCoefficients: computeMeanAndCovarianceMatrix + pcl::eigen33 (smallest) */
void optimize_plane(const P3* pts, const std::vector<int>& inl, const float cin[4], float cout[4])
{
    if (inl.size() < 4) { std::memcpy(cout, cin, 16); return; }
    float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i : inl) {
        const P3& p = pts[i];
        acc[0] += p.x * p.x; acc[1] += p.x * p.y; acc[2] += p.x * p.z;
        acc[3] += p.y * p.y; acc[4] += p.y * p.z; acc[5] += p.z * p.z;
        acc[6] += p.x; acc[7] += p.y; acc[8] += p.z;
    }
    const float cnt = (float)inl.size();
    for (int k = 0; k < 9; k++) acc[k] /= cnt;
    float m[9];
    m[0] = acc[0] - acc[6] * acc[6]; m[1] = acc[1] - acc[6] * acc[7]; m[2] = acc[2] - acc[6] * acc[8];
    m[4] = acc[3] - acc[7] * acc[7]; m[5] = acc[4] - acc[7] * acc[8]; m[8] = acc[5] - acc[8] * acc[8];
    m[3] = m[1]; m[6] = m[2]; m[7] = m[5];
    float scale = 0.f;
    for (int k = 0; k < 9; k++) scale = std::max(scale, std::fabs(m[k]));
    if (scale <= std::numeric_limits<float>::min()) scale = 1.0f;
    float sm[9];
    for (int k = 0; k < 9; k++) sm[k] = m[k] / scale;
    float ev[3];
    compute_roots(sm, ev);
    sm[0] -= ev[0]; sm[4] -= ev[0]; sm[8] -= ev[0];
    float v1[3], v2[3], v3[3];
    cross3(&sm[0], &sm[3], v1); cross3(&sm[0], &sm[6], v2); cross3(&sm[3], &sm[6], v3);
    const float l1 = (v1[0] * v1[0] + v1[1] * v1[1]) + v1[2] * v1[2], l2 = (v2[0] * v2[0] + v2[1] * v2[1]) + v2[2] * v2[2],
                l3 = (v3[0] * v3[0] + v3[1] * v3[1]) + v3[2] * v3[2];
    const float* v; float l;
    if (l1 >= l2 && l1 >= l3) { v = v1; l = l1; } else if (l2 >= l1 && l2 >= l3) { v = v2; l = l2; } else { v = v3; l = l3; }
    const float s = std::sqrt(l);
    cout[0] = v[0] / s; cout[1] = v[1] / s; cout[2] = v[2] / s; cout[3] = 0.f;
    cout[3] = -1.f * (((cout[0] * acc[6] + cout[1] * acc[7]) + cout[2] * acc[8]) + cout[3] * 0.0f);   /* xyz_centroid[3] = 0 */
    for (int k = 0; k < 4; k++)
        if (!std::isfinite(cout[k])) { std::memcpy(cout, cin, 16); return; }       /* isModelValid: size 4 (finite here) */
}

/* Frame::MaxPointDistanceFromPlane: 0 = rejected, 1 = accepted (coef overwritten by the refit) */
int refit(float coef[4], const P3* pts, int n, double disTh)
{
    for (int i = 0; i < n; i++) {
        const double a = std::fabs((double)(((coef[0] * pts[i].x + coef[1] * pts[i].y) + coef[2] * pts[i].z) + coef[3]));
        if (a > disTh) return 0;
    }
    if (n < 3) return 0;                                 /* getSamples: "Can not select 3 unique points" -> no inliers */
    std::vector<int> shuffled(n);
    for (int i = 0; i < n; i++) shuffled[i] = i;
    Mt19937 rng(12345u);
    const int maxIter = 50;
    int iterations = 0, best = -std::numeric_limits<int>::max();
    double k = 1.0;
    const double logp = std::log(1.0 - 0.99), oneOver = 1.0 / (double)n;
    unsigned skipped = 0;
    const unsigned maxSkip = maxIter * 10;
    float bestC[4] = {0, 0, 0, 0};
    bool have = false;
    while (iterations < k && skipped < maxSkip) {
        int s[3];
        bool got = false;
        for (unsigned it = 0; it < 1000; it++) {         /* max_sample_checks_ */
            for (int i = 0; i < 3; i++) std::swap(shuffled[i], shuffled[i + (rng.rnd() % (n - i))]);
            s[0] = shuffled[0]; s[1] = shuffled[1]; s[2] = shuffled[2];
            if (sample_good(pts, s)) { got = true; break; }
        }
        if (!got) break;
        float c[4];
        if (!plane_from_samples(pts, s, c)) { skipped++; continue; }
        int cnt = 0;
        for (int i = 0; i < n; i++)
            if (std::fabs((double)dot4(c, pts[i])) < disTh) cnt++;
        if (cnt > best) {
            best = cnt; have = true;
            std::memcpy(bestC, c, 16);
            const double w = (double)best * oneOver;
            double pno = 1.0 - std::pow(w, 3.0);
            pno = std::max(std::numeric_limits<double>::epsilon(), pno);
            pno = std::min(1.0 - std::numeric_limits<double>::epsilon(), pno);
            k = logp / std::log(pno);
        }
        iterations++;
        if (iterations > maxIter) break;
    }
    if (!have) return 0;
    std::vector<int> inl;
    for (int i = 0; i < n; i++)
        if (std::fabs((double)dot4(bestC, pts[i])) < disTh) inl.push_back(i);
    if (inl.empty()) return 0;
    float opt[4];
    optimize_plane(pts, inl, bestC, opt);
    /* the refined inlier set is recomputed by PCL but only its emptiness is read by the caller */
    int refined = 0;
    for (int i = 0; i < n; i++)
        if (std::fabs((double)dot4(opt, pts[i])) < disTh) refined++;
    if (refined == 0) return 0;
    const float oldVal = coef[3], newVal = opt[3];
    std::memcpy(coef, opt, 16);
    if ((newVal < 0 && oldVal > 0) || (newVal > 0 && oldVal < 0))
        for (int k2 = 0; k2 < 4; k2++) coef[k2] = -coef[k2];
    return 1;
}

/* ---- pcl::IntegralImageNormalEstimation, AVERAGE_3D_GRADIENT, BORDER_POLICY_IGNORE, fixed smoothing -------------- */
void surface_normals(const float* depth, int w, int h, size_t stride, float fx, float fy, float cx, float cy, float maxDist,
                     float maxDepthChange, float smoothingSize, std::vector<P3>& cloud, std::vector<P3>& normals, int& W, int& H)
{
    W = (int)std::ceil(w / 3.0); H = (int)std::ceil(h / 3.0);
    cloud.assign((size_t)W * H, P3{0, 0, 0});
    for (int m = 0, r = 0; m < h; m += 3, r++)
        for (int n = 0, c = 0; n < w; n += 3, c++) {
            const float d = depth[(size_t)m * stride + n];
            P3 p;
            p.z = d > maxDist ? 0.f : d;
            p.x = ((float)n - cx) * p.z / fx;
            p.y = ((float)m - cy) * p.z / fy;
            cloud[(size_t)r * W + c] = p;
        }
    const float nanv = std::numeric_limits<float>::quiet_NaN();
    normals.assign((size_t)W * H, P3{nanv, nanv, nanv});
    const size_t N = (size_t)W * H;
    /* initAverage3DGradientMethod: central differences, zero on the one-pixel frame */
    std::vector<float> dxm(N * 3, 0.f), dym(N * 3, 0.f);
    for (int r = 1; r < H - 1; r++)
        for (int c = 1; c < W - 1; c++) {
            const P3 &rg = cloud[(size_t)r * W + c + 1], &lf = cloud[(size_t)r * W + c - 1];
            const P3 &dn = cloud[(size_t)(r + 1) * W + c], &up = cloud[(size_t)(r - 1) * W + c];
            float* a = &dxm[((size_t)r * W + c) * 3];
            float* b = &dym[((size_t)r * W + c) * 3];
            a[0] = rg.x - lf.x; a[1] = rg.y - lf.y; a[2] = rg.z - lf.z;
            b[0] = dn.x - up.x; b[1] = dn.y - up.y; b[2] = dn.z - up.z;
        }
    /* IntegralImage2D<float, 3>::computeIntegralImages: double, row running sum + the row above; finite counts */
    const int IW = W + 1;
    std::vector<double> ix((size_t)IW * (H + 1) * 3, 0.0), iy((size_t)IW * (H + 1) * 3, 0.0);
    std::vector<unsigned> cntx((size_t)IW * (H + 1), 0u), cnty((size_t)IW * (H + 1), 0u);
    auto integrate = [&](const std::vector<float>& src, std::vector<double>& dst, std::vector<unsigned>& cnt) {
        for (int r = 0; r < H; r++)
            for (int c = 0; c < W; c++) {
                const float* e = &src[((size_t)r * W + c) * 3];
                const size_t cur = (size_t)(r + 1) * IW + c + 1, pre = (size_t)r * IW + c + 1;
                for (int k = 0; k < 3; k++) dst[cur * 3 + k] = dst[pre * 3 + k] + dst[(cur - 1) * 3 + k] - dst[(pre - 1) * 3 + k];
                cnt[cur] = cnt[pre] + cnt[cur - 1] - cnt[pre - 1];
                if (std::isfinite((e[0] + e[1]) + e[2])) {                      /* pcl_isfinite (element->sum ()) */
                    for (int k = 0; k < 3; k++) dst[cur * 3 + k] += (double)e[k];
                    cnt[cur]++;
                }
            }
    };
    integrate(dxm, ix, cntx);
    integrate(dym, iy, cnty);
    /* depth change map */
    std::vector<unsigned char> chg(N, 255);
    for (int r = 0; r < H - 1; r++)
        for (int c = 0; c < W - 1; c++) {
            const size_t i = (size_t)r * W + c;
            const float d = cloud[i].z, dR = cloud[i + 1].z, dD = cloud[i + W].z;
            const float lim = (maxDepthChange * (std::fabs(d) + 1.0f) * 2.0f);
            if (std::fabs(d - dR) > lim || !std::isfinite(d) || !std::isfinite(dR)) { chg[i] = 0; chg[i + 1] = 0; }
            if (std::fabs(d - dD) > lim || !std::isfinite(d) || !std::isfinite(dD)) { chg[i] = 0; chg[i + W] = 0; }
        }
    /* chamfer distance to the nearest depth change: two raster passes */
    std::vector<float> dist(N);
    for (size_t i = 0; i < N; i++) dist[i] = chg[i] == 0 ? 0.0f : (float)(W + H);
    for (int r = 1; r < H; r++) {
        float* prev = &dist[(size_t)(r - 1) * W];
        float* cur = &dist[(size_t)r * W];
        for (int c = 1; c < W; c++) {
            const float ul = prev[c - 1] + 1.4f, u = prev[c] + 1.0f;
            /* previous_row[ci + 1] at the last column is the first element of the current row (contiguous buffer) */
            const float ur = (c + 1 < W ? prev[c + 1] : cur[0]) + 1.4f;
            const float l = cur[c - 1] + 1.0f, ce = cur[c];
            const float mv = std::min(std::min(ul, u), std::min(l, ur));
            if (mv < ce) cur[c] = mv;
        }
    }
    for (int r = H - 2; r >= 0; r--) {
        float* nxt = &dist[(size_t)(r + 1) * W];
        float* cur = &dist[(size_t)r * W];
        for (int c = W - 2; c >= 0; c--) {
            /* next_row[ci - 1] at column 0 is the last element of the current row */
            const float ll = (c >= 1 ? nxt[c - 1] : cur[W - 1]) + 1.4f, lo = nxt[c] + 1.0f, lr = nxt[c + 1] + 1.4f;
            const float rt = cur[c + 1] + 1.0f, ce = cur[c];
            const float mv = std::min(std::min(ll, lo), std::min(rt, lr));
            if (mv < ce) cur[c] = mv;
        }
    }
    const int border = (int)smoothingSize;
    for (int r = border; r < H - border; r++)
        for (int c = border; c < W - border; c++) {
            const size_t i = (size_t)r * W + c;
            if (!std::isfinite(cloud[i].z)) continue;
            const float sm = std::min(dist[i], smoothingSize);
            if (!(sm > 2.0f)) continue;
            const int rw = (int)sm, rh = (int)sm, rw2 = rw >> 1, rh2 = rh >> 1;
            const int x0 = c - rw2, y0 = r - rh2;
            const size_t ul = (size_t)y0 * IW + x0, ur = ul + rw, ll = (size_t)(y0 + rh) * IW + x0, lr = ll + rw;
            const unsigned cx_ = cntx[ul] + cntx[lr] - cntx[ur] - cntx[ll], cy_ = cnty[ul] + cnty[lr] - cnty[ur] - cnty[ll];
            if (cx_ == 0 || cy_ == 0) continue;
            double gx[3], gy[3];
            for (int k = 0; k < 3; k++) {
                gx[k] = ix[lr * 3 + k] + ix[ul * 3 + k] - ix[ur * 3 + k] - ix[ll * 3 + k];
                gy[k] = iy[lr * 3 + k] + iy[ul * 3 + k] - iy[ur * 3 + k] - iy[ll * 3 + k];
            }
            double nv[3] = {gy[1] * gx[2] - gy[2] * gx[1], gy[2] * gx[0] - gy[0] * gx[2], gy[0] * gx[1] - gy[1] * gx[0]};
            const double len = (nv[0] * nv[0] + nv[1] * nv[1]) + nv[2] * nv[2];
            if (len == 0.0) continue;
            const double sl = std::sqrt(len);
            float nx = (float)(nv[0] / sl), ny = (float)(nv[1] / sl), nz = (float)(nv[2] / sl);
            /* flipNormalTowardsViewpoint, viewpoint (0, 0, 0) */
            const float vx = 0.f - cloud[i].x, vy = 0.f - cloud[i].y, vz = 0.f - cloud[i].z;
            const float ct = (vx * nx + vy * ny) + vz * nz;
            if (ct < 0) { nx *= -1; ny *= -1; nz *= -1; }
            normals[i] = P3{nx, ny, nz};
        }
}

}  // namespace

extern "C" {

/* xyz: n x 3 floats in inputCloud order; out: capacity n x 3; returns the number of leaves */
int orc_post_voxel_grid(const float* xyz, int n, float leaf, float* out)
{
    std::vector<P3> o;
    voxel_grid(reinterpret_cast<const P3*>(xyz), n, leaf, o);
    if (!o.empty()) std::memcpy(out, o.data(), o.size() * sizeof(P3));
    return (int)o.size();
}

int orc_post_refit(float* coef4, const float* xyz, int n, double dis_th)
{
    return refit(coef4, reinterpret_cast<const P3*>(xyz), n, dis_th);
}

/* depth: float metres (h x stride); cloud / normals: ceil(w/3) x ceil(h/3) x 3 floats */
void orc_post_surface_normals(const float* depth, int w, int h, long stride, float fx, float fy, float cx, float cy,
                              float max_dist, float* cloud, float* normals, float* dist_dbg)
{
    std::vector<P3> c, nrm;
    int W, H;
    surface_normals(depth, w, h, (size_t)stride, fx, fy, cx, cy, max_dist, 0.05f, 10.0f, c, nrm, W, H);
    std::memcpy(cloud, c.data(), c.size() * sizeof(P3));
    std::memcpy(normals, nrm.data(), nrm.size() * sizeof(P3));
    (void)dist_dbg;
}

uint32_t orc_post_mt19937(uint32_t seed, int skip)
{
    Mt19937 r(seed);
    uint32_t v = 0;
    for (int i = 0; i <= skip; i++) v = r.next();
    return v;
}
}
