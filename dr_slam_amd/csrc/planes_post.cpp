/* planes_post.cpp — what Frame::ComputePlanes / ComputePlanes_CAPE do AFTER the extractor (reference src/Frame.cc:949-1094,
 * 1096-1213) and Frame::MaxPointDistanceFromPlane (:1222-1307), behind the C-ABI of include/drfe.h:
 *   per plane: member points -> pcl::VoxelGrid(0.05) -> distance / size gates -> RANSAC + least-squares refit that
 *   OVERWRITES the coefficients the tracker consumes (host: a few thousand voxel points, order-defined by std::sort and
 *   by the sample consensus RNG);
 *   per frame: surface normals of the 3x-subsampled cloud (device: normals_kernels.hip).
 * PCL 1.9.1 semantics and the canonical choices are listed in DESIGN.md section 9. */
#include "post_internal.h"
#include "introsort_restated.h"

#include <time.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <limits>
#include <random>
#include <string>
#include <vector>

#define HIPCHK(c, call)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (c)->err = std::string(#call) + ": " + hipGetErrorString(e__);                      \
            return DRFE_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

namespace {

struct Pt { float x, y, z; };

/* std::floor(float) without the libm call (the build targets baseline x86-64, where floorf is a function): exact for every
 * finite value - |v| >= 2^23 is integral already, below that the truncation is corrected downwards */
inline float floor_f(float v)
{
    if (!(std::fabs(v) < 8388608.0f)) return v;
    const float t = (float)(int)v;
    return t > v ? t - 1.0f : t;
}

/* pcl::VoxelGrid<PointXYZRGB>::applyFilter (filters/impl/voxel_grid.hpp), xyz part of the all-fields centroid.  Its
 * std::sort(index_vector) compares leaf indices only (cloud_point_index_idx::operator<), and the unstable sort's order inside
 * a leaf is the order of the float centroid sums: voxel_order::sort makes libstdc++'s moves (introsort_restated.h) on
 * leaf << 32 | point records; DRFE_VOXEL_STD_SORT=1 calls std::sort itself. */
void voxel_downsample(const std::vector<Pt>& src, float leafSize, std::vector<Pt>* dst)
{
    static thread_local std::vector<uint64_t> keys;      /* scratch kept per host thread */
    dst->clear();
    if (src.empty()) return;
    const float inv = 1.0f / leafSize;
    Pt lo{std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    Pt hi{-lo.x, -lo.x, -lo.x};
    for (const Pt& p : src) {
        lo.x = std::min(lo.x, p.x); lo.y = std::min(lo.y, p.y); lo.z = std::min(lo.z, p.z);
        hi.x = std::max(hi.x, p.x); hi.y = std::max(hi.y, p.y); hi.z = std::max(hi.z, p.z);
    }
    const int64_t nx = (int64_t)((hi.x - lo.x) * inv) + 1, ny = (int64_t)((hi.y - lo.y) * inv) + 1, nz = (int64_t)((hi.z - lo.z) * inv) + 1;
    if (nx * ny * nz > (int64_t)std::numeric_limits<int32_t>::max()) { *dst = src; return; }
    const int bx = (int)floor_f(lo.x * inv), by = (int)floor_f(lo.y * inv), bz = (int)floor_f(lo.z * inv);
    const int ex = (int)floor_f(hi.x * inv), ey = (int)floor_f(hi.y * inv);
    const int sx = ex - bx + 1, sxy = sx * (ey - by + 1);
    keys.resize(src.size());
    for (size_t i = 0; i < src.size(); i++) {
        const int a = (int)(floor_f(src[i].x * inv) - (float)bx);
        const int b = (int)(floor_f(src[i].y * inv) - (float)by);
        const int c = (int)(floor_f(src[i].z * inv) - (float)bz);
        keys[i] = voxel_order::record((unsigned)(a + b * sx + c * sxy), (unsigned)i);
    }
    static const bool stdSort = std::getenv("DRFE_VOXEL_STD_SORT") != nullptr;
    if (stdSort) std::sort(keys.begin(), keys.end(), voxel_order::Before());
    else voxel_order::sort(keys.data(), keys.size());
    for (size_t first = 0; first < keys.size();) {
        size_t last = first;
        const uint32_t leaf = voxel_order::leaf_of(keys[first]);
        Pt acc{0.f, 0.f, 0.f};
        while (last < keys.size() && voxel_order::leaf_of(keys[last]) == leaf) {
            const Pt& p = src[voxel_order::point_of(keys[last])];
            acc.x += p.x; acc.y += p.y; acc.z += p.z;
            last++;
        }
        const float n = (float)(last - first);
        dst->push_back(Pt{acc.x / n, acc.y / n, acc.z / n});
        first = last;
    }
}

inline float plane_eval(const float c[4], const Pt& p) { return ((c[0] * p.x + c[1] * p.y) + c[2] * p.z) + c[3] * 1.0f; }

/* SampleConsensusModelPlane: isSampleGood / computeModelCoefficients (sample_consensus/impl/sac_model_plane.hpp) */
bool three_point_plane(const Pt& p0, const Pt& p1, const Pt& p2, float c[4])
{
    const float ux = p1.x - p0.x, uy = p1.y - p0.y, uz = p1.z - p0.z;
    const float vx = p2.x - p0.x, vy = p2.y - p0.y, vz = p2.z - p0.z;
    const float qx = ux / vx, qy = uy / vy, qz = uz / vz;
    if (qx == qy && qz == qy) return false;               /* collinear */
    float n[4] = {uy * vz - uz * vy, uz * vx - ux * vz, ux * vy - uy * vx, 0.f};
    const float sq = ((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]) + n[3] * n[3];
    if (sq > 0.f) {
        const float len = std::sqrt(sq);
        n[0] /= len; n[1] /= len; n[2] /= len; n[3] /= len;
    }
    c[0] = n[0]; c[1] = n[1]; c[2] = n[2];
    c[3] = -1.f * (((n[0] * p0.x + n[1] * p0.y) + n[2] * p0.z) + n[3] * 1.0f);
    return true;
}

/* pcl::computeRoots / computeRoots2 (common/impl/eigen.hpp) for float; sqrt / atan2 / cos / sin in double, rounded once
 * (the float libm results are host dependent) */
void quadratic_roots(float b, float c, float r[3])
{
    r[0] = 0.f;
    float d = (float)((double)(b * b) - 4.0 * (double)c);
    if (d < 0.0f) d = 0.0f;
    const float sd = std::sqrt(d);
    r[2] = 0.5f * (b + sd);
    r[1] = 0.5f * (b - sd);
}

void symmetric_roots(const float M[3][3], float r[3])
{
    const float c0 = M[0][0] * M[1][1] * M[2][2] + 2.0f * M[0][1] * M[0][2] * M[1][2] - M[0][0] * M[1][2] * M[1][2] -
                     M[1][1] * M[0][2] * M[0][2] - M[2][2] * M[0][1] * M[0][1];
    const float c1 = M[0][0] * M[1][1] - M[0][1] * M[0][1] + M[0][0] * M[2][2] - M[0][2] * M[0][2] + M[1][1] * M[2][2] -
                     M[1][2] * M[1][2];
    const float c2 = M[0][0] + M[1][1] + M[2][2];
    if (std::fabs(c0) < std::numeric_limits<float>::epsilon()) { quadratic_roots(c2, c1, r); return; }
    const float inv3 = (float)(1.0 / 3.0), sqrt3 = (float)std::sqrt(3.0);
    const float c2o3 = c2 * inv3;
    float ao3 = (c1 - c2 * c2o3) * inv3;
    if (ao3 > 0.f) ao3 = 0.f;
    const float hb = 0.5f * (c0 + c2o3 * (2.0f * c2o3 * c2o3 - c1));
    float q = hb * hb + ao3 * ao3 * ao3;
    if (q > 0.f) q = 0.f;
    const float rho = (float)std::sqrt((double)-ao3);
    const float sq = (float)std::sqrt((double)-q);
    const float theta = (float)std::atan2((double)sq, (double)hb) * inv3;
    const float ct = (float)std::cos((double)theta), st = (float)std::sin((double)theta);
    r[0] = c2o3 + 2.0f * rho * ct;
    r[1] = c2o3 - rho * (ct + sqrt3 * st);
    r[2] = c2o3 - rho * (ct - sqrt3 * st);
    if (r[0] >= r[1]) std::swap(r[0], r[1]);
    if (r[1] >= r[2]) {
        std::swap(r[1], r[2]);
        if (r[0] >= r[1]) std::swap(r[0], r[1]);
    }
    if (r[0] <= 0.f) quadratic_roots(c2, c1, r);
}

/* optimizeModelCoefficients: computeMeanAndCovarianceMatrix (float accumulators) + pcl::eigen33 smallest eigenvector */
void least_squares_plane(const std::vector<Pt>& pts, const std::vector<int>& inl, const float start[4], float out[4])
{
    if (inl.size() < 4) { std::memcpy(out, start, 16); return; }
    float xx = 0, xy = 0, xz = 0, yy = 0, yz = 0, zz = 0, mx = 0, my = 0, mz = 0;
    for (int i : inl) {
        const Pt& p = pts[i];
        xx += p.x * p.x; xy += p.x * p.y; xz += p.x * p.z; yy += p.y * p.y; yz += p.y * p.z; zz += p.z * p.z;
        mx += p.x; my += p.y; mz += p.z;
    }
    const float n = (float)inl.size();
    xx /= n; xy /= n; xz /= n; yy /= n; yz /= n; zz /= n; mx /= n; my /= n; mz /= n;
    float C[3][3];
    C[0][0] = xx - mx * mx; C[0][1] = xy - mx * my; C[0][2] = xz - mx * mz;
    C[1][1] = yy - my * my; C[1][2] = yz - my * mz; C[2][2] = zz - mz * mz;
    C[1][0] = C[0][1]; C[2][0] = C[0][2]; C[2][1] = C[1][2];
    float scale = 0.f;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) scale = std::max(scale, std::fabs(C[i][j]));
    if (scale <= std::numeric_limits<float>::min()) scale = 1.0f;
    float S[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) S[i][j] = C[i][j] / scale;
    float ev[3];
    symmetric_roots(S, ev);
    for (int i = 0; i < 3; i++) S[i][i] -= ev[0];
    auto cross = [](const float a[3], const float b[3], float o[3]) {
        o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
    };
    float v[3][3];
    cross(S[0], S[1], v[0]); cross(S[0], S[2], v[1]); cross(S[1], S[2], v[2]);
    float len[3];
    for (int k = 0; k < 3; k++) len[k] = (v[k][0] * v[k][0] + v[k][1] * v[k][1]) + v[k][2] * v[k][2];
    const int pick = (len[0] >= len[1] && len[0] >= len[2]) ? 0 : (len[1] >= len[0] && len[1] >= len[2]) ? 1 : 2;
    const float l = std::sqrt(len[pick]);
    out[0] = v[pick][0] / l; out[1] = v[pick][1] / l; out[2] = v[pick][2] / l;
    out[3] = -1.f * (((out[0] * mx + out[1] * my) + out[2] * mz) + 0.f * 1.0f);
    if (!(std::isfinite(out[0]) && std::isfinite(out[1]) && std::isfinite(out[2]) && std::isfinite(out[3]))) std::memcpy(out, start, 16);
}

/* Frame::MaxPointDistanceFromPlane (src/Frame.cc:1222-1307): pcl::SACSegmentation, SACMODEL_PLANE, SAC_RANSAC, 50
 * iterations, probability 0.99, optimize on; sample draws as SampleConsensusModel::drawIndexSample with boost::mt19937
 * (seed 12345) behind uniform_int<>(0, INT_MAX) == std::mt19937 output >> 1 */
/* The inlier count of a plane over a cloud kept as three float arrays: |c . (p, 1)| < disTh with the reference's float
 * expression and its double comparison.  A float widens to double exactly, so `fabs((double)e) < disTh` is `|e| <= tLess` with
 * tLess the largest float whose double lies below disTh: the loop is pure float arithmetic and runs eight points at a time
 * under AVX2 (same IEEE operations per lane, no contraction), which is what RANSAC's 5-10 counting passes per plane cost. */
template <int W>
static inline int count_within_w(const float* X, const float* Y, const float* Z, int n, const float c[4], float tLess)
{
    typedef float vf __attribute__((vector_size(W * 4)));
    typedef int vi __attribute__((vector_size(W * 4)));
    int k = 0, i = 0;
    if (W > 1) {
        vi acc = {};
        const vf c0 = c[0] - (vf){}, c1 = c[1] - (vf){}, c2 = c[2] - (vf){}, c3 = c[3] * 1.0f - (vf){}, t = tLess - (vf){};
        for (; i + W <= n; i += W) {
            vf x, y, z;
            std::memcpy(&x, X + i, sizeof(vf)); std::memcpy(&y, Y + i, sizeof(vf)); std::memcpy(&z, Z + i, sizeof(vf));
            vf e = ((c0 * x + c1 * y) + c2 * z) + c3;
            const vf ne = -e;
            const vf a = e > ne ? e : ne;                       /* |e|; a NaN stays NaN and fails the comparison below */
            acc -= (vi)(a <= t);                                /* a true lane is -1 */
        }
        for (int q = 0; q < W; q++) k += acc[q];
    }
    for (; i < n; i++) {
        const float e = ((c[0] * X[i] + c[1] * Y[i]) + c[2] * Z[i]) + c[3] * 1.0f;
        k += std::fabs(e) <= tLess ? 1 : 0;
    }
    return k;
}
__attribute__((target("avx2"))) static int count_within_avx2(const float* X, const float* Y, const float* Z, int n, const float c[4], float t) { return count_within_w<8>(X, Y, Z, n, c, t); }
static int count_within_sse(const float* X, const float* Y, const float* Z, int n, const float c[4], float t) { return count_within_w<4>(X, Y, Z, n, c, t); }
static int count_within(const float* X, const float* Y, const float* Z, int n, const float c[4], float t)
{
    static const bool avx2 = __builtin_cpu_supports("avx2");
    return avx2 ? count_within_avx2(X, Y, Z, n, c, t) : count_within_sse(X, Y, Z, n, c, t);
}

bool refit_plane(float coef[4], const std::vector<Pt>& pts, double disTh)
{
    const int n = (int)pts.size();
    /* largest float strictly below disTh (as doubles), and the cloud as three arrays */
    float tLess = (float)disTh;
    if (!((double)tLess < disTh)) tLess = std::nextafterf(tLess, -std::numeric_limits<float>::infinity());
    static thread_local std::vector<float> soa;
    soa.resize((size_t)3 * n);
    float *X = soa.data(), *Y = X + n, *Z = Y + n;
    for (int i = 0; i < n; i++) { X[i] = pts[i].x; Y[i] = pts[i].y; Z[i] = pts[i].z; }
    /* every point within disTh of the extractor's plane: `fabs((double)e) > disTh` for none <=> `|e| <= tMost` for all, tMost the
     * largest float whose double does not exceed disTh */
    float tMost = (float)disTh;
    if ((double)tMost > disTh) tMost = std::nextafterf(tMost, -std::numeric_limits<float>::infinity());
    {
        const float cc[4] = {coef[0], coef[1], coef[2], coef[3]};
        /* the gate's expression ends in `+ coef[3]`, plane_eval's in `+ c[3] * 1.0f`: the same float */
        int within = count_within(X, Y, Z, n, cc, tMost);
        if (within != n) {
            /* a NaN distance does not trip the reference's `>` test: count those as passing, as it does */
            int nans = 0;
            for (int i = 0; i < n; i++) { const float e = ((coef[0] * X[i] + coef[1] * Y[i]) + coef[2] * Z[i]) + coef[3]; nans += e != e ? 1 : 0; }
            if (within + nans != n) return false;
        }
    }
    if (n < 3) return false;
    static thread_local std::vector<int> order;
    order.resize(n);
    for (int i = 0; i < n; i++) order[i] = i;
    std::mt19937 gen(12345u);
    auto inliers_of = [&](const float c[4]) { return count_within(X, Y, Z, n, c, tLess); };
    float best[4] = {0, 0, 0, 0};
    int bestCount = -std::numeric_limits<int>::max(), iterations = 0;
    unsigned skipped = 0;
    double k = 1.0;
    const double logP = std::log(1.0 - 0.99);
    while (iterations < k && skipped < 500u) {
        bool drawn = false;
        for (int attempt = 0; attempt < 1000 && !drawn; attempt++) {
            for (int i = 0; i < 3; i++) {
                const int r = (int)(gen() >> 1);
                std::swap(order[i], order[i + r % (n - i)]);
            }
            const Pt &a = pts[order[0]], &b = pts[order[1]], &c = pts[order[2]];
            const float qx = (b.x - a.x) / (c.x - a.x), qy = (b.y - a.y) / (c.y - a.y), qz = (b.z - a.z) / (c.z - a.z);
            drawn = (qx != qy) || (qz != qy);
        }
        if (!drawn) break;
        float c[4];
        if (!three_point_plane(pts[order[0]], pts[order[1]], pts[order[2]], c)) { skipped++; continue; }
        const int cnt = inliers_of(c);
        if (cnt > bestCount) {
            bestCount = cnt;
            std::memcpy(best, c, 16);
            const double w = (double)cnt * (1.0 / (double)n);
            double pNo = 1.0 - std::pow(w, 3.0);
            pNo = std::max(std::numeric_limits<double>::epsilon(), pNo);
            pNo = std::min(1.0 - std::numeric_limits<double>::epsilon(), pNo);
            k = logP / std::log(pNo);
        }
        if (++iterations > 50) break;
    }
    if (bestCount < 0) return false;
    std::vector<int> inl;
    for (int i = 0; i < n; i++)
        if (std::fabs((double)plane_eval(best, pts[i])) < disTh) inl.push_back(i);
    if (inl.empty()) return false;
    float fit[4];
    least_squares_plane(pts, inl, best, fit);
    if (inliers_of(fit) == 0) return false;
    const float oldD = coef[3], newD = fit[3];
    std::memcpy(coef, fit, 16);
    if ((newD < 0 && oldD > 0) || (newD > 0 && oldD < 0))
        for (int i = 0; i < 4; i++) coef[i] = -coef[i];
    return true;
}

thread_local double g_tVoxel = 0;       /* DRFE_TRACE_PLANES accounting */

/* one plane of the per-plane loop: gates + refit; appends the voxel cloud of an accepted plane */
struct PostOut {
    drfe_plane_post* post; float* vox; int32_t* voxOff; int capVox; int used; int nAccepted; int failPlanes; bool overflow;
};

/* gates + refit of one plane on its voxel cloud */
void post_one_coarse(const std::vector<Pt>& coarse, const float coefIn[4], bool gateD, double disTh, bool invalidCountsAsFail, int i,
                     PostOut* o)
{
    drfe_plane_post& P = o->post[i];
    std::memcpy(P.coef, coefIn, 16);
    P.n_voxels = (int32_t)coarse.size();
    P.accepted = 0;
    o->voxOff[i] = o->used;
    if (gateD || coarse.size() < 100) { o->failPlanes++; return; }
    float coef[4];
    std::memcpy(coef, coefIn, 16);
    if (!refit_plane(coef, coarse, disTh)) { if (invalidCountsAsFail) o->failPlanes++; return; }
    std::memcpy(P.coef, coef, 16);
    P.accepted = 1;
    o->nAccepted++;
    if (o->vox) {
        if (o->used + (int)coarse.size() > o->capVox) { o->overflow = true; return; }
        std::memcpy(o->vox + (size_t)o->used * 3, coarse.data(), coarse.size() * sizeof(Pt));
        o->used += (int)coarse.size();
    }
}

void post_one_plane(const std::vector<Pt>& input, const float coefIn[4], bool gateD, double disTh, bool invalidCountsAsFail, int i,
                    PostOut* o)
{
    std::vector<Pt> coarse;
    const auto tv = std::chrono::steady_clock::now();
    voxel_downsample(input, 0.05f, &coarse);
    g_tVoxel += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tv).count();
    post_one_coarse(coarse, coefIn, gateD, disTh, invalidCountsAsFail, i, o);
}

}  // namespace

/* ---- a lane's device voxel grid ------------------------------------------------------------------------------------------- */
struct VoxelDevice {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;
    size_t cap = 0;               /* points the buffers hold */
    float* d_pts = nullptr; unsigned long long* d_recs = nullptr; unsigned long long* d_tmp = nullptr;
    uint32_t* d_posL = nullptr; uint32_t* d_posR = nullptr; float* d_out = nullptr;
    int2* d_jobs = nullptr; int* d_counts = nullptr; int* d_list = nullptr;
    float* h_pts = nullptr; float* h_out = nullptr; int2* h_jobs = nullptr; int* h_counts = nullptr;
};
#define VOX_MAX_JOBS 256

static void voxel_buffers_free(VoxelDevice* v)
{
    void* d[] = {v->d_pts, v->d_recs, v->d_tmp, v->d_posL, v->d_posR, v->d_out};
    for (void* p : d) if (p) (void)hipFree(p);
    if (v->h_pts) (void)hipHostFree(v->h_pts);
    if (v->h_out) (void)hipHostFree(v->h_out);
    v->d_pts = nullptr; v->d_recs = nullptr; v->d_tmp = nullptr; v->d_posL = nullptr; v->d_posR = nullptr; v->d_out = nullptr;
    v->h_pts = nullptr; v->h_out = nullptr; v->cap = 0;
}

VoxelDevice* drfe_voxel_device_create(int device, std::string* err)
{
    VoxelDevice* v = new (std::nothrow) VoxelDevice();
    if (!v) return nullptr;
    v->device = device;
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&v->ev, hipEventDisableTiming) == hipSuccess &&
              hipMalloc((void**)&v->d_jobs, VOX_MAX_JOBS * sizeof(int2)) == hipSuccess && hipMalloc((void**)&v->d_list, (VOX_MAX_JOBS + 2) * sizeof(int)) == hipSuccess && hipMalloc((void**)&v->d_counts, VOX_MAX_JOBS * sizeof(int)) == hipSuccess &&
              hipHostMalloc((void**)&v->h_jobs, VOX_MAX_JOBS * sizeof(int2), hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc((void**)&v->h_counts, VOX_MAX_JOBS * sizeof(int), hipHostMallocDefault) == hipSuccess;
    if (!ok) { if (err) *err = "voxel grid lane: allocation failed"; drfe_voxel_device_free(v); return nullptr; }
    return v;
}

void drfe_voxel_device_free(VoxelDevice* v)
{
    if (!v) return;
    voxel_buffers_free(v);
    if (v->d_jobs) (void)hipFree(v->d_jobs);
    if (v->d_counts) (void)hipFree(v->d_counts);
    if (v->d_list) (void)hipFree(v->d_list);
    if (v->h_jobs) (void)hipHostFree(v->h_jobs);
    if (v->h_counts) (void)hipHostFree(v->h_counts);
    if (v->ev) (void)hipEventDestroy(v->ev);
    if (v->stream) (void)hipStreamDestroy(v->stream);
    delete v;
}

/* pcl::VoxelGrid(0.05) of every plane's cloud in one launch; coarse[i] filled for the planes the device finished, done[i] = 0
 * for those it handed back (grid overflow, heap-sort branch) or that did not fit */
static bool voxel_downsample_device(VoxelDevice* v, const std::vector<Pt>* inputs, int np, std::vector<Pt>* coarse,
                                    std::vector<char>& done, std::string* err)
{
    done.assign(np, 0);
    if (np == 0) return true;
    if (np > VOX_MAX_JOBS) return true;                   /* host path for all */
    size_t total = 0;
    for (int i = 0; i < np; i++) total += inputs[i].size();
    if (total == 0) return true;
    auto fail = [&](const char* what, hipError_t e) { if (err) *err = std::string("voxel grid lane: ") + what + ": " + hipGetErrorString(e); return false; };
    hipError_t e = hipSetDevice(v->device);
    if (e != hipSuccess) return fail("hipSetDevice", e);
    if (v->cap < total) {
        voxel_buffers_free(v);
        const size_t cap = std::max<size_t>(total + total / 4, 1 << 16);
        if ((e = hipMalloc((void**)&v->d_pts, cap * 12)) != hipSuccess || (e = hipMalloc((void**)&v->d_recs, cap * 8)) != hipSuccess ||
            (e = hipMalloc((void**)&v->d_tmp, cap * 8)) != hipSuccess || (e = hipMalloc((void**)&v->d_posL, cap * 4)) != hipSuccess ||
            (e = hipMalloc((void**)&v->d_posR, cap * 4)) != hipSuccess || (e = hipMalloc((void**)&v->d_out, cap * 12)) != hipSuccess ||
            (e = hipHostMalloc((void**)&v->h_pts, cap * 12, hipHostMallocDefault)) != hipSuccess ||
            (e = hipHostMalloc((void**)&v->h_out, cap * 12, hipHostMallocDefault)) != hipSuccess)
            return fail("buffer allocation", e);
        v->cap = cap;
    }
    size_t off = 0;
    for (int i = 0; i < np; i++) {
        v->h_jobs[i] = make_int2((int)off, (int)inputs[i].size());
        if (!inputs[i].empty()) std::memcpy(v->h_pts + 3 * off, inputs[i].data(), inputs[i].size() * sizeof(Pt));
        off += inputs[i].size();
    }
    e = hipMemcpyAsync(v->d_pts, v->h_pts, total * 12, hipMemcpyHostToDevice, v->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(v->d_jobs, v->h_jobs, np * sizeof(int2), hipMemcpyHostToDevice, v->stream);
    if (e == hipSuccess) e = drfe_launch_voxel_grid(v->d_pts, v->d_jobs, np, v->d_list, v->d_recs, v->d_tmp, v->d_posL, v->d_posR, v->d_out, v->d_counts, 0.05f, v->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(v->h_counts, v->d_counts, np * sizeof(int), hipMemcpyDeviceToHost, v->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(v->h_out, v->d_out, total * 12, hipMemcpyDeviceToHost, v->stream);
    if (e == hipSuccess) e = drfe_pool_sync(v->stream, v->ev);
    if (e != hipSuccess) return fail("launch", e);
    for (int i = 0; i < np; i++) {
        const int cnt = v->h_counts[i];
        if (cnt < 0) continue;                            /* handed back */
        coarse[i].resize((size_t)cnt);
        if (cnt) std::memcpy(coarse[i].data(), v->h_out + 3 * (size_t)v->h_jobs[i].x, (size_t)cnt * sizeof(Pt));
        done[i] = 1;
    }
    return true;
}


/* the per-plane loop of Frame::ComputePlanes without a context (error text to *err): shared by drfe_planes_ahc_postprocess
 * and the worker threads of drfe_planes_ahc_post_batch */
int drfe_ahc_post_core(std::string* err, const uint16_t* depth, int w, int h, size_t stride, const float* K4, float depth_factor,
                       const drfe_plane* planes, int n_planes, const int32_t* member_offsets, const int32_t* member_idx,
                       float max_point_dist, double dist_threshold, drfe_plane_post* post, float* voxel_xyz, int32_t* voxel_offsets,
                       int cap_voxels, int* n_accepted, int* plane_num, VoxelDevice* vox)
{
    if (!depth || !K4 || (!planes && n_planes) || n_planes < 0 || !member_offsets || (!member_idx && n_planes) || (!post && n_planes) ||
        !voxel_offsets || !n_accepted) {
        if (err) *err = "planes_ahc_postprocess: invalid argument";
        return DRFE_ERR_INVALID;
    }
    PostOut o{post, voxel_xyz, voxel_offsets, cap_voxels, 0, 0, 0, false};
    static thread_local std::vector<std::vector<Pt>> inputs, coarse;
    inputs.resize(std::max<size_t>(inputs.size(), (size_t)n_planes));
    coarse.resize(std::max<size_t>(coarse.size(), (size_t)n_planes));
    const double invW = 1.0 / (double)w;
    static const bool trace = std::getenv("DRFE_TRACE_PLANES") != nullptr;      /* wall time per stage on stderr */
    double tGather = 0, tPlane = 0;
    size_t nPts = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n_planes; i++) {
        std::vector<Pt>& input = inputs[i];
        input.clear();
        for (int k = member_offsets[i]; k < member_offsets[i + 1]; k++) {
            const int j = member_idx[k];
            if (j < 0 || j >= w * h) { if (err) *err = "planes_ahc_postprocess: member index outside the image"; return DRFE_ERR_INVALID; }
            /* j / w without the division instruction: (j + 0.5) / w is at least 0.5 / w away from an integer, far more than the
             * rounding error of the double product */
            const int row = (int)(((double)j + 0.5) * invW), col = j - row * w;
            /* PlaneDetection::readDepthImage (src/PlaneExtractor.cpp:39-52): doubles, K floats promoted */
            const double z = (double)depth[(size_t)row * stride + col] * depth_factor;
            double X = 0, Y = 0, Z = 0;
            if (!(z > 5.0)) {
                X = ((double)col - K4[2]) * z / K4[0];
                Y = ((double)row - K4[3]) * z / K4[1];
                Z = z;
            }
            const Pt p{(float)X, (float)Y, (float)Z};
            if (p.z > max_point_dist) continue;
            input.push_back(p);
        }
        nPts += input.size();
    }
    const auto t1 = std::chrono::steady_clock::now();
    struct timespec cpu0, cpu1;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &cpu0);
    /* pcl::VoxelGrid(0.05) of every plane: on the lane's device grid (one launch for the frame's planes) or on the host */
    std::vector<char> done((size_t)n_planes, 0);
    if (vox && !voxel_downsample_device(vox, inputs.data(), n_planes, coarse.data(), done, err)) return DRFE_ERR_HIP;
    for (int i = 0; i < n_planes; i++)
        if (!done[i]) voxel_downsample(inputs[i], 0.05f, &coarse[i]);
    const auto t2 = std::chrono::steady_clock::now();
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &cpu1);
    const double voxCpuMs = (cpu1.tv_sec - cpu0.tv_sec) * 1e3 + (cpu1.tv_nsec - cpu0.tv_nsec) * 1e-6;
    g_tVoxel = std::chrono::duration<double, std::milli>(t2 - t1).count();
    for (int i = 0; i < n_planes; i++) {
        const drfe_plane& e = planes[i];
        const float d = (float)-(e.normal[0] * e.center[0] + e.normal[1] * e.center[1] + e.normal[2] * e.center[2]);
        const float coef[4] = {(float)e.normal[0], (float)e.normal[1], (float)e.normal[2], d};
        post_one_coarse(coarse[i], coef, d > max_point_dist, dist_threshold, false, i, &o);
    }
    tGather = std::chrono::duration<double, std::milli>(t1 - t0).count();
    tPlane = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    if (trace) std::fprintf(stderr, "drfe_ahc_post_core: %d planes, %zu member points; gather %.2f ms; voxel grid %.2f ms (%.2f ms of this thread's CPU; %s); refit %.2f ms\n",
                            n_planes, nPts, tGather, g_tVoxel, voxCpuMs, vox ? "device" : "host", tPlane - g_tVoxel), g_tVoxel = 0;
    voxel_offsets[n_planes] = o.used;
    *n_accepted = o.nAccepted;
    if (plane_num) *plane_num = n_planes - o.failPlanes;     /* planeDetector.plane_num_ -= fail_planes (:1023) */
    if (o.overflow) { if (err) *err = "planes_ahc_postprocess: voxel buffer too small"; return DRFE_ERR_CAPACITY; }
    return DRFE_OK;
}

/* the same loop when every plane's voxel cloud is already there (k_voxel_grid ran behind k_ahc_refine): gates + refit only */
int drfe_ahc_post_from_coarse(std::string* err, const drfe_plane* planes, int n_planes, const float* const* coarse_xyz, const int* coarse_n,
                              float max_point_dist, double dist_threshold, drfe_plane_post* post, float* voxel_xyz, int32_t* voxel_offsets,
                              int cap_voxels, int* n_accepted, int* plane_num)
{
    PostOut o{post, voxel_xyz, voxel_offsets, cap_voxels, 0, 0, 0, false};
    static thread_local std::vector<Pt> coarse;
    for (int i = 0; i < n_planes; i++) {
        const drfe_plane& e = planes[i];
        const float d = (float)-(e.normal[0] * e.center[0] + e.normal[1] * e.center[1] + e.normal[2] * e.center[2]);
        const float coef[4] = {(float)e.normal[0], (float)e.normal[1], (float)e.normal[2], d};
        coarse.assign((const Pt*)coarse_xyz[i], (const Pt*)coarse_xyz[i] + coarse_n[i]);
        post_one_coarse(coarse, coef, d > max_point_dist, dist_threshold, false, i, &o);
    }
    voxel_offsets[n_planes] = o.used;
    *n_accepted = o.nAccepted;
    if (plane_num) *plane_num = n_planes - o.failPlanes;
    if (o.overflow) { if (err) *err = "planes_ahc_postprocess: voxel buffer too small"; return DRFE_ERR_CAPACITY; }
    return DRFE_OK;
}

extern "C" {

int drfe_plane_voxel_grid(const float* xyz, int n, float leaf, float* out_xyz, int cap, int* n_out)
{
    if (!xyz || n < 0 || !n_out || leaf <= 0.f) return DRFE_ERR_INVALID;
    std::vector<Pt> src((const Pt*)xyz, (const Pt*)xyz + n), dst;
    voxel_downsample(src, leaf, &dst);
    *n_out = (int)dst.size();
    if ((int)dst.size() > cap) return DRFE_ERR_CAPACITY;
    if (out_xyz && !dst.empty()) std::memcpy(out_xyz, dst.data(), dst.size() * sizeof(Pt));
    return DRFE_OK;
}

int drfe_plane_refit(float* coef4, const float* xyz, int n, double dist_threshold, int* valid)
{
    if (!coef4 || !xyz || n < 0 || !valid) return DRFE_ERR_INVALID;
    std::vector<Pt> pts((const Pt*)xyz, (const Pt*)xyz + n);
    *valid = refit_plane(coef4, pts, dist_threshold) ? 1 : 0;
    return DRFE_OK;
}

int drfe_planes_ahc_postprocess(drfe_ctx* c, const uint16_t* depth, int w, int h, size_t stride, const float* K4, float depth_factor,
                                const drfe_plane* planes, int n_planes, const int32_t* member_offsets, const int32_t* member_idx,
                                float max_point_dist, double dist_threshold, drfe_plane_post* post, float* voxel_xyz,
                                int32_t* voxel_offsets, int cap_voxels, int* n_accepted, int* plane_num)
{
    std::string local;                    /* ctx == NULL: host code, usable without a device (no error text then) */
    return drfe_ahc_post_core(c ? &c->err : &local, depth, w, h, stride, K4, depth_factor, planes, n_planes, member_offsets, member_idx, max_point_dist,
                              dist_threshold, post, voxel_xyz, voxel_offsets, cap_voxels, n_accepted, plane_num);
}

int drfe_planes_cape_postprocess(drfe_ctx* c, const float* depth_m, int w, int h, size_t stride, const float* K4, const uint8_t* seg,
                                 const drfe_cape_plane* planes, int n_planes, float max_point_dist, double dist_threshold,
                                 drfe_plane_post* post, float* voxel_xyz, int32_t* voxel_offsets, int cap_voxels, int* n_accepted,
                                 int* plane_num)
{
    if (!c) return DRFE_ERR_INVALID;
    if (!depth_m || !K4 || !seg || !planes || n_planes < 0 || n_planes > 255 || !post || !voxel_offsets || !n_accepted) {
        c->err = "planes_cape_postprocess: invalid argument";
        return DRFE_ERR_INVALID;
    }
    /* plane_cloud[code - 1] in raster order (src/PlaneExtractor.cpp:171-188); cloud_array is float, filled from doubles */
    std::vector<std::vector<Pt>> clouds(n_planes);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            const int code = seg[(size_t)i * w + j];
            if (code <= 0 || code > n_planes) continue;
            const double z = (double)depth_m[(size_t)i * stride + j];
            const double x = ((double)j - K4[2]) * z / K4[0];
            const double y = ((double)i - K4[3]) * z / K4[1];
            clouds[code - 1].push_back(Pt{(float)x, (float)y, (float)z});
        }
    PostOut o{post, voxel_xyz, voxel_offsets, cap_voxels, 0, 0, 0, false};
    for (int i = 0; i < n_planes; i++) {
        const drfe_cape_plane& e = planes[i];
        const float coef[4] = {(float)e.normal[0], (float)e.normal[1], (float)e.normal[2], (float)e.d};
        post_one_plane(clouds[i], coef, e.d > (double)max_point_dist, dist_threshold, true, i, &o);
    }
    voxel_offsets[n_planes] = o.used;
    *n_accepted = o.nAccepted;
    if (plane_num) *plane_num = n_planes - o.failPlanes;
    if (o.overflow) { c->err = "planes_cape_postprocess: voxel buffer too small"; return DRFE_ERR_CAPACITY; }
    return DRFE_OK;
}

/* ---- surface normals ---------------------------------------------------------------------------------------------- */

static int sn_ensure(drfe_ctx* c, int w, int h, int frames, bool needStage)
{
    SnBuffers* b = static_cast<SnBuffers*>(c->sn);
    if (!b) { b = new SnBuffers(); std::memset(b, 0, sizeof(*b)); c->sn = b; }
    if (b->frames >= (size_t)frames && b->w == (size_t)w && b->h == (size_t)h && (!needStage || b->d_depth)) return DRFE_OK;
    drfe_post_free(c);
    b = new SnBuffers(); std::memset(b, 0, sizeof(*b)); c->sn = b;
    const size_t W = drfe_sn_w(w), H = drfe_sn_h(h), N = W * H, NI = (W + 1) * (H + 1), F = (size_t)frames;
    HIPCHK(c, hipMalloc((void**)&b->d_cloud, F * N * 3 * sizeof(float)));
    HIPCHK(c, hipMalloc((void**)&b->d_dist, F * N * sizeof(float)));
    HIPCHK(c, hipMalloc((void**)&b->d_integ, F * NI * 6 * sizeof(double)));
    HIPCHK(c, hipMalloc((void**)&b->d_cnt, F * NI * 2 * sizeof(unsigned)));
    HIPCHK(c, hipMalloc((void**)&b->d_normals, F * N * 3 * sizeof(float)));
    HIPCHK(c, hipMalloc((void**)&b->d_recs, F * (W / 2) * (H / 2) * sizeof(drfe_surface_normal)));
    if (needStage) HIPCHK(c, hipMalloc(&b->d_depth, (size_t)w * h * sizeof(float)));
    b->frames = F; b->w = (size_t)w; b->h = (size_t)h;
    return DRFE_OK;
}

int drfe_surface_normals(drfe_ctx* c, const float* depth_m, int w, int h, size_t stride, const float* K4, float max_point_dist,
                         drfe_surface_normal* out, int cap, int* n_out, float* cloud_tap, float* normals_tap, float* dist_tap)
{
    if (!c) return DRFE_ERR_INVALID;
    if (!depth_m || !K4 || !n_out || w < 3 || h < 3 || stride < (size_t)w) { c->err = "surface_normals: invalid argument"; return DRFE_ERR_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    int rc = sn_ensure(c, w, h, 1, true);
    if (rc != DRFE_OK) return rc;
    SnBuffers* b = static_cast<SnBuffers*>(c->sn);
    const size_t W = drfe_sn_w(w), H = drfe_sn_h(h), nrec = (W / 2) * (H / 2);
    *n_out = (int)nrec;
    if (out && (size_t)cap < nrec) { c->err = "surface_normals: output buffer too small"; return DRFE_ERR_CAPACITY; }
    HIPCHK(c, hipMemcpy2DAsync(b->d_depth, (size_t)w * 4, depth_m, stride * 4, (size_t)w * 4, h, hipMemcpyHostToDevice, c->stream));
    hipError_t e = drfe_launch_surface_normals(b->d_depth, 0, 1.0f, (size_t)w * h, w, w, h, K4, max_point_dist, 1, *b, c->stream);
    if (e != hipSuccess) { c->err = std::string("surface_normals: ") + hipGetErrorString(e); return DRFE_ERR_HIP; }
    if (out) HIPCHK(c, hipMemcpyAsync(out, b->d_recs, nrec * sizeof(drfe_surface_normal), hipMemcpyDeviceToHost, c->stream));
    if (cloud_tap) HIPCHK(c, hipMemcpyAsync(cloud_tap, b->d_cloud, W * H * 12, hipMemcpyDeviceToHost, c->stream));
    if (normals_tap) HIPCHK(c, hipMemcpyAsync(normals_tap, b->d_normals, W * H * 12, hipMemcpyDeviceToHost, c->stream));
    if (dist_tap) HIPCHK(c, hipMemcpyAsync(dist_tap, b->d_dist, W * H * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DRFE_OK;
}

int drfe_surface_normals_batch(drfe_ctx* c, const uint16_t* d_depth, size_t frame_stride, size_t row_stride, int w, int h,
                               const float* K4, float depth_factor, float max_point_dist, int nframes, void* stream)
{
    if (!c) return DRFE_ERR_INVALID;
    if (!d_depth || !K4 || nframes < 1 || w < 3 || h < 3) { c->err = "surface_normals_batch: invalid argument"; return DRFE_ERR_INVALID; }
    HIPCHK(c, hipSetDevice(c->device));
    int rc = sn_ensure(c, w, h, nframes, false);
    if (rc != DRFE_OK) return rc;
    SnBuffers* b = static_cast<SnBuffers*>(c->sn);
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    hipError_t e = drfe_launch_surface_normals(d_depth, 1, depth_factor, frame_stride, row_stride, w, h, K4, max_point_dist, nframes, *b, s);
    if (e != hipSuccess) { c->err = std::string("surface_normals_batch: ") + hipGetErrorString(e); return DRFE_ERR_HIP; }
    return DRFE_OK;
}

int drfe_surface_normals_download(drfe_ctx* c, int slot, drfe_surface_normal* out, int cap, int* n_out)
{
    if (!c) return DRFE_ERR_INVALID;
    SnBuffers* b = static_cast<SnBuffers*>(c->sn);
    if (!b || slot < 0 || (size_t)slot >= b->frames || !n_out) { c->err = "surface_normals_download: no such slot"; return DRFE_ERR_INVALID; }
    const size_t W = drfe_sn_w((int)b->w), H = drfe_sn_h((int)b->h), nrec = (W / 2) * (H / 2);
    *n_out = (int)nrec;
    if (!out) return DRFE_OK;
    if ((size_t)cap < nrec) { c->err = "surface_normals_download: output buffer too small"; return DRFE_ERR_CAPACITY; }
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(out, b->d_recs + (size_t)slot * nrec, nrec * sizeof(drfe_surface_normal), hipMemcpyDeviceToHost));
    return DRFE_OK;
}

}  // extern "C"

void drfe_post_free(drfe_ctx* c)
{
    SnBuffers* b = static_cast<SnBuffers*>(c->sn);
    if (!b) return;
    void* ptrs[] = {b->d_cloud, b->d_dist, b->d_integ, b->d_cnt, b->d_normals, b->d_recs, b->d_depth};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete b;
    c->sn = nullptr;
}
