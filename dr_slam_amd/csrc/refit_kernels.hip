/* refit_kernels.hip - the last host stage of Frame::ComputePlanes' per-plane loop on the device (round 5): the gates and
 * Frame::MaxPointDistanceFromPlane (reference src/Frame.cc:1003-1011, 1222-1307) on the voxel clouds k_voxel_grid left:
 *   gate 1  d > Point.MaxDistance, gate 2  fewer than 100 voxels, gate 3  a voxel farther than Plane.DistanceThreshold from the
 *   extractor's plane; then pcl::SACSegmentation (SACMODEL_PLANE, SAC_RANSAC, 50 iterations, probability 0.99, optimize on):
 *   drawIndexSample with boost::mt19937 (seed 12345) behind uniform_int<>(0, INT_MAX), computeModelCoefficients, countWithinDistance,
 *   the adaptive iteration bound, selectWithinDistance, optimizeModelCoefficients (float covariance sums in inlier order,
 *   pcl::eigen33), the sign kept on the side of the extractor's d.  Restated for the host in planes_post.cpp (refit_plane), which
 *   stays the checker of this kernel and the path of the single-frame entries.
 *
 * ONE WAVEFRONT PER PLANE.  What the reference does one point at a time - the inlier counts of the ~5-10 hypotheses, the gate, the
 * final inlier list - runs across the lanes (the same float expression per point, ballots for the counts); what is a sequence -
 * the generator, the partial shuffle of the index vector, the three-point model, the iteration bound - is wave-uniform scalar work
 * every lane repeats; the nine covariance sums, float additions in inlier order, are nine lanes walking LDS columns the other
 * lanes filled (a point outside the inlier set contributes +0.0f, which never changes a sum that started at +0.0f).
 * The shuffled index vector is sparse: the reference swaps three entries per draw of an identity permutation, so the kernel keeps
 * positions 0..2 in registers and the few displaced others in a small LDS table.
 * Three places call the host's libm in the reference - log / pow for the iteration bound, atan2 / cos / sin (in double, rounded
 * to float once) inside pcl::computeRoots.  Here: cos / sin correctly rounded (cr_sincos.h), atan2 / log / pow the device's, each
 * result used only when it is certain to round or compare as the host's does (margins below); otherwise the plane's status says
 * "uncertain" and the pool thread runs refit_plane on the centroids - as it does for a plane whose voxel grid came back from the
 * device. */
#include "drfe_internal.h"
#include "planes_internal.h"
#include "post_internal.h"
#include "cr_sincos.h"

#include <cfloat>
#include <climits>

#define RF_TAB 512                     /* displaced entries of the index vector this kernel keeps (3 per draw; ~30 typical) */

namespace {

__device__ __forceinline__ int rf_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float rf_unif(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

struct Mt {                            /* std::mt19937 */
    uint32_t* s; int i;
    __device__ void twist(int lane)
    {
        if (lane == 0) {
            for (int k = 0; k < 624; k++) {
                const uint32_t y = (s[k] & 0x80000000u) | (s[(k + 1) % 624] & 0x7fffffffu);
                s[k] = s[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        i = 0;
    }
    __device__ uint32_t next(int lane)
    {
        if (i >= 624) twist(lane);
        uint32_t y = s[i++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return (uint32_t)rf_uni((int)y);
    }
};

/* the index vector's entry at position p (>= 3): the table, else p itself */
__device__ __forceinline__ int ord_get(const int* key, const int* val, int p)
{
    uint32_t s = ((uint32_t)p * 2654435761u) >> 23;
    for (int k = 0; k < RF_TAB; k++) {
        const int q = key[s];
        if (q == p) return val[s];
        if (q < 0) return p;
        s = (s + 1) & (RF_TAB - 1);
    }
    return p;
}
__device__ __forceinline__ bool ord_set(int* key, int* val, int p, int v, int lane)
{
    uint32_t s = ((uint32_t)p * 2654435761u) >> 23;
    bool ok = false;
    for (int k = 0; k < RF_TAB; k++) {
        const int q = key[s];
        if (q == p || q < 0) { if (lane == 0) { key[s] = p; val[s] = v; } ok = true; break; }
        s = (s + 1) & (RF_TAB - 1);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    return ok;
}

/* the points with |c . (p, 1)| <= t: SampleConsensusModelPlane::countWithinDistance's float expression */
__device__ __forceinline__ int count_within(const float* __restrict__ pts, int n, float c0, float c1, float c2, float c3, float t, int lane)
{
    int k = 0;
    for (int base = 0; base < n; base += 64) {
        const int p = base + lane;
        bool in = false;
        if (p < n) {
            const float x = pts[3 * (size_t)p], y = pts[3 * (size_t)p + 1], z = pts[3 * (size_t)p + 2];
            const float e = ((c0 * x + c1 * y) + c2 * z) + c3 * 1.0f;
            in = fabsf(e) <= t;
        }
        k += __popcll(__ballot(in));
    }
    return k;
}

/* (float)v as the host's libm value would round, when v is within `rel` of it: false if a float rounding boundary is that close */
__device__ __forceinline__ bool rounds_surely(double v, double rel, float* out)
{
    const float f = (float)v;
    const double m = fabs(v) * rel + 1e-300;
    *out = f;
    return (float)(v + m) == f && (float)(v - m) == f;
}

__device__ __forceinline__ void quadratic_roots(float b, float c, float r[3])
{
    r[0] = 0.f;
    float d = (float)((double)(b * b) - 4.0 * (double)c);
    if (d < 0.0f) d = 0.0f;
    const float sd = sqrtf(d);
    r[2] = 0.5f * (b + sd);
    r[1] = 0.5f * (b - sd);
}

/* pcl::computeRoots (common/impl/eigen.hpp) for float; returns false when a libm result could not be certified */
__device__ __forceinline__ bool symmetric_roots(const float M[3][3], float r[3])
{
    const float c0 = M[0][0] * M[1][1] * M[2][2] + 2.0f * M[0][1] * M[0][2] * M[1][2] - M[0][0] * M[1][2] * M[1][2] -
                     M[1][1] * M[0][2] * M[0][2] - M[2][2] * M[0][1] * M[0][1];
    const float c1 = M[0][0] * M[1][1] - M[0][1] * M[0][1] + M[0][0] * M[2][2] - M[0][2] * M[0][2] + M[1][1] * M[2][2] -
                     M[1][2] * M[1][2];
    const float c2 = M[0][0] + M[1][1] + M[2][2];
    if (fabsf(c0) < FLT_EPSILON) { quadratic_roots(c2, c1, r); return true; }
    const float inv3 = (float)(1.0 / 3.0), sqrt3 = 1.7320508075688772f;      /* (float)sqrt(3.0) */
    const float c2o3 = c2 * inv3;
    float ao3 = (c1 - c2 * c2o3) * inv3;
    if (ao3 > 0.f) ao3 = 0.f;
    const float hb = 0.5f * (c0 + c2o3 * (2.0f * c2o3 * c2o3 - c1));
    float q = hb * hb + ao3 * ao3 * ao3;
    if (q > 0.f) q = 0.f;
    const float rho = (float)sqrt((double)-ao3);                             /* double sqrt is correctly rounded on both sides */
    const float sq = (float)sqrt((double)-q);
    bool sure = true;
    float at;
    sure = rounds_surely(atan2((double)sq, (double)hb), 4e-16, &at) && sure;  /* glibc's and the device's atan2: < 1 ulp each */
    const float theta = at * inv3;
    double sd, cd;
    if (!drfe_cr_sincos((double)theta, &sd, &cd)) sure = false;               /* theta in [0, pi / 3] */
    float ct, st;
    sure = rounds_surely(cd, 2.3e-16, &ct) && sure;                           /* correctly rounded here, within an ulp on the host */
    sure = rounds_surely(sd, 2.3e-16, &st) && sure;
    r[0] = c2o3 + 2.0f * rho * ct;
    r[1] = c2o3 - rho * (ct + sqrt3 * st);
    r[2] = c2o3 - rho * (ct - sqrt3 * st);
    if (r[0] >= r[1]) { const float t = r[0]; r[0] = r[1]; r[1] = t; }
    if (r[1] >= r[2]) {
        const float t = r[1]; r[1] = r[2]; r[2] = t;
        if (r[0] >= r[1]) { const float u = r[0]; r[0] = r[1]; r[1] = u; }
    }
    if (r[0] <= 0.f) quadratic_roots(c2, c1, r);
    return sure;
}

}  // namespace

struct RefitParams { float tLess, tMost, maxPointDist; int planeCap; double logP; };

/* status[job]: 0 = post[job] is final; 1 = a libm-dependent step could not be certified, or the index table overflowed: the host
 * refits this plane; 2 = its voxel grid came back from the device (counts < 0): the host does grid + refit; -1 = no such plane */
extern "C" __global__ __launch_bounds__(64) void k_plane_refit(const AhcDevFrame* __restrict__ frames, const int2* __restrict__ jobs, const int* __restrict__ vcounts,
                                                              const float* __restrict__ vout, const uint32_t* __restrict__ mtState, RefitParams P,
                                                              drfe_plane_post* __restrict__ post, int* __restrict__ status)
{
    __shared__ uint32_t mtS[624];
    __shared__ int ordKey[RF_TAB], ordVal[RF_TAB];
    __shared__ float col[9][64];
    const int job = blockIdx.x, lane = threadIdx.x;
    const int f = job / P.planeCap, pi = job - f * P.planeCap;
    const int nP = frames[f].out[0], fst = frames[f].out[1];
    if (fst != 0 || pi >= nP) { if (lane == 0) status[job] = -1; return; }
    const drfe_plane e = frames[f].planes[pi];
    const float dd = (float)-(e.normal[0] * e.center[0] + e.normal[1] * e.center[1] + e.normal[2] * e.center[2]);
    float coef[4] = {(float)e.normal[0], (float)e.normal[1], (float)e.normal[2], dd};
    const int n = vcounts[job];
    drfe_plane_post R;
    R.coef[0] = coef[0]; R.coef[1] = coef[1]; R.coef[2] = coef[2]; R.coef[3] = coef[3]; R.accepted = 0; R.n_voxels = n < 0 ? 0 : n;
    auto finish = [&](int st) { if (lane == 0) { post[job] = R; status[job] = st; } };
    if (n < 0) { finish(2); return; }
    if (dd > P.maxPointDist || n < 100) { finish(0); return; }
    const float* pts = vout + 3 * (size_t)jobs[job].x;

    /* gate 3: every voxel within the threshold of the extractor's plane (a NaN distance does not trip the reference's `>`) */
    {
        int within = 0, nans = 0;
        for (int base = 0; base < n; base += 64) {
            const int p = base + lane;
            bool in = false, nan = false;
            if (p < n) {
                const float x = pts[3 * (size_t)p], y = pts[3 * (size_t)p + 1], z = pts[3 * (size_t)p + 2];
                const float ev = ((coef[0] * x + coef[1] * y) + coef[2] * z) + coef[3];
                in = fabsf(ev) <= P.tMost; nan = ev != ev;
            }
            within += __popcll(__ballot(in)); nans += __popcll(__ballot(nan));
        }
        if (within + nans != n) { finish(0); return; }
    }

    /* ---- pcl::RandomSampleConsensus::computeModel ---- */
    for (int k = lane; k < 624; k += 64) mtS[k] = mtState[k];
    for (int k = lane; k < RF_TAB; k += 64) ordKey[k] = -1;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    Mt gen; gen.s = mtS; gen.i = 624;                        /* the seeded state: the first draw twists it */
    int o3[3] = {0, 1, 2};
    float best[4] = {0, 0, 0, 0};
    int bestCount = -INT_MAX, iterations = 0;
    unsigned skipped = 0;
    double k = 1.0;
    bool unsure = false;
    for (;;) {
        /* `iterations < k`: k = log(0.01) / log(1 - w^3) from the device's log / pow; their few ulps only matter when k sits on an integer */
        if (fabs(k - (double)iterations) <= 1e-9 * fmax(1.0, fabs(k))) unsure = true;
        if (!((double)iterations < k) || !(skipped < 500u)) break;
        bool drawn = false;
        float ax = 0, ay = 0, az = 0, bx = 0, by = 0, bz = 0, cx = 0, cy = 0, cz = 0;
        for (int attempt = 0; attempt < 1000 && !drawn; attempt++) {
            for (int i = 0; i < 3; i++) {
                const int r = (int)(gen.next(lane) >> 1);
                const int j = i + r % (n - i);
                /* std::swap(order[i], order[j]), i in 0..2 */
                if (j < 3) { const int t = o3[i]; o3[i] = o3[j]; o3[j] = t; }
                else {
                    const int vj = rf_uni(ord_get(ordKey, ordVal, j));
                    if (!ord_set(ordKey, ordVal, j, o3[i], lane)) unsure = true;
                    o3[i] = vj;
                }
            }
            ax = rf_unif(pts[3 * (size_t)o3[0]]); ay = rf_unif(pts[3 * (size_t)o3[0] + 1]); az = rf_unif(pts[3 * (size_t)o3[0] + 2]);
            bx = rf_unif(pts[3 * (size_t)o3[1]]); by = rf_unif(pts[3 * (size_t)o3[1] + 1]); bz = rf_unif(pts[3 * (size_t)o3[1] + 2]);
            cx = rf_unif(pts[3 * (size_t)o3[2]]); cy = rf_unif(pts[3 * (size_t)o3[2] + 1]); cz = rf_unif(pts[3 * (size_t)o3[2] + 2]);
            const float qx = (bx - ax) / (cx - ax), qy = (by - ay) / (cy - ay), qz = (bz - az) / (cz - az);
            drawn = (qx != qy) || (qz != qy);
        }
        if (unsure || !drawn) break;
        /* SampleConsensusModelPlane::computeModelCoefficients */
        float c[4];
        {
            const float ux = bx - ax, uy = by - ay, uz = bz - az;
            const float vx = cx - ax, vy = cy - ay, vz = cz - az;
            const float qx = ux / vx, qy = uy / vy, qz = uz / vz;
            if (qx == qy && qz == qy) { skipped++; continue; }               /* collinear */
            float nn[4] = {uy * vz - uz * vy, uz * vx - ux * vz, ux * vy - uy * vx, 0.f};
            const float sq = ((nn[0] * nn[0] + nn[1] * nn[1]) + nn[2] * nn[2]) + nn[3] * nn[3];
            if (sq > 0.f) {
                const float len = sqrtf(sq);
                nn[0] /= len; nn[1] /= len; nn[2] /= len; nn[3] /= len;
            }
            c[0] = nn[0]; c[1] = nn[1]; c[2] = nn[2];
            c[3] = -1.f * (((nn[0] * ax + nn[1] * ay) + nn[2] * az) + nn[3] * 1.0f);
        }
        const int cnt = count_within(pts, n, c[0], c[1], c[2], c[3], P.tLess, lane);
        if (cnt > bestCount) {
            bestCount = cnt;
            best[0] = c[0]; best[1] = c[1]; best[2] = c[2]; best[3] = c[3];
            const double w = (double)cnt * (1.0 / (double)n);
            double pNo = 1.0 - pow(w, 3.0);
            pNo = fmax(DBL_EPSILON, pNo);
            pNo = fmin(1.0 - DBL_EPSILON, pNo);
            k = P.logP / log(pNo);
        }
        if (++iterations > 50) break;
    }
    if (unsure) { finish(1); return; }
    if (bestCount < 0) { finish(0); return; }

    /* selectWithinDistance + optimizeModelCoefficients: the nine float sums over the inliers, in index order */
    int nInl = 0;
    float acc = 0.f;                                         /* lanes 0..8: xx xy xz yy yz zz mx my mz */
    for (int base = 0; base < n; base += 64) {
        const int p = base + lane;
        bool in = false;
        float x = 0.f, y = 0.f, z = 0.f;
        if (p < n) {
            x = pts[3 * (size_t)p]; y = pts[3 * (size_t)p + 1]; z = pts[3 * (size_t)p + 2];
            const float ev = ((best[0] * x + best[1] * y) + best[2] * z) + best[3] * 1.0f;
            in = fabsf(ev) <= P.tLess;
        }
        nInl += __popcll(__ballot(in));
        col[0][lane] = in ? x * x : 0.f; col[1][lane] = in ? x * y : 0.f; col[2][lane] = in ? x * z : 0.f;
        col[3][lane] = in ? y * y : 0.f; col[4][lane] = in ? y * z : 0.f; col[5][lane] = in ? z * z : 0.f;
        col[6][lane] = in ? x : 0.f; col[7][lane] = in ? y : 0.f; col[8][lane] = in ? z : 0.f;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if (lane < 9) {
            const float* cc = col[lane];
#pragma unroll 8
            for (int t = 0; t < 64; t++) acc += cc[t];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    if (nInl == 0) { finish(0); return; }
    float fit[4] = {best[0], best[1], best[2], best[3]};
    bool sure = true;
    if (nInl >= 4) {
        float s9[9];
#pragma unroll
        for (int t = 0; t < 9; t++) s9[t] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), t));
        const float nf = (float)nInl;
        float xx = s9[0] / nf, xy = s9[1] / nf, xz = s9[2] / nf, yy = s9[3] / nf, yz = s9[4] / nf, zz = s9[5] / nf, mx = s9[6] / nf, my = s9[7] / nf, mz = s9[8] / nf;
        float C[3][3];
        C[0][0] = xx - mx * mx; C[0][1] = xy - mx * my; C[0][2] = xz - mx * mz;
        C[1][1] = yy - my * my; C[1][2] = yz - my * mz; C[2][2] = zz - mz * mz;
        C[1][0] = C[0][1]; C[2][0] = C[0][2]; C[2][1] = C[1][2];
        float scale = 0.f;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) scale = fmaxf(scale, fabsf(C[i][j]));
        if (scale <= FLT_MIN) scale = 1.0f;
        float S[3][3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) S[i][j] = C[i][j] / scale;
        float ev[3];
        sure = symmetric_roots(S, ev);
        for (int i = 0; i < 3; i++) S[i][i] -= ev[0];
        float v[3][3];
        v[0][0] = S[0][1] * S[1][2] - S[0][2] * S[1][1]; v[0][1] = S[0][2] * S[1][0] - S[0][0] * S[1][2]; v[0][2] = S[0][0] * S[1][1] - S[0][1] * S[1][0];
        v[1][0] = S[0][1] * S[2][2] - S[0][2] * S[2][1]; v[1][1] = S[0][2] * S[2][0] - S[0][0] * S[2][2]; v[1][2] = S[0][0] * S[2][1] - S[0][1] * S[2][0];
        v[2][0] = S[1][1] * S[2][2] - S[1][2] * S[2][1]; v[2][1] = S[1][2] * S[2][0] - S[1][0] * S[2][2]; v[2][2] = S[1][0] * S[2][1] - S[1][1] * S[2][0];
        float len[3];
        for (int t = 0; t < 3; t++) len[t] = (v[t][0] * v[t][0] + v[t][1] * v[t][1]) + v[t][2] * v[t][2];
        const int pick = (len[0] >= len[1] && len[0] >= len[2]) ? 0 : (len[1] >= len[0] && len[1] >= len[2]) ? 1 : 2;
        const float l = sqrtf(len[pick]);
        fit[0] = v[pick][0] / l; fit[1] = v[pick][1] / l; fit[2] = v[pick][2] / l;
        fit[3] = -1.f * (((fit[0] * mx + fit[1] * my) + fit[2] * mz) + 0.f * 1.0f);
        if (!(isfinite(fit[0]) && isfinite(fit[1]) && isfinite(fit[2]) && isfinite(fit[3]))) { fit[0] = best[0]; fit[1] = best[1]; fit[2] = best[2]; fit[3] = best[3]; }
    }
    if (!sure) { finish(1); return; }
    if (count_within(pts, n, fit[0], fit[1], fit[2], fit[3], P.tLess, lane) == 0) { finish(0); return; }
    const float oldD = coef[3], newD = fit[3];
    const bool flip = (newD < 0 && oldD > 0) || (newD > 0 && oldD < 0);
    for (int t = 0; t < 4; t++) R.coef[t] = flip ? -fit[t] : fit[t];
    R.accepted = 1;
    finish(0);
}

hipError_t drfe_launch_plane_refit(const AhcDevFrame* d_frames, const int2* d_jobs, const int* d_vcounts, const float* d_vout, const uint32_t* d_mtState,
                                   int njobs, int planeCap, float maxPointDist, double distThreshold, double logP, drfe_plane_post* d_post, int* d_status,
                                   hipStream_t s)
{
    if (njobs <= 0) return hipSuccess;
    RefitParams P;
    /* `fabs((double)e) < disTh` <=> |e| <= tLess, the largest float whose double lies below disTh; `> disTh` for none <=> |e| <= tMost for all */
    float tLess = (float)distThreshold;
    if (!((double)tLess < distThreshold)) tLess = nextafterf(tLess, -INFINITY);
    float tMost = (float)distThreshold;
    if ((double)tMost > distThreshold) tMost = nextafterf(tMost, -INFINITY);
    P.tLess = tLess; P.tMost = tMost; P.maxPointDist = maxPointDist; P.planeCap = planeCap; P.logP = logP;
    hipLaunchKernelGGL(k_plane_refit, dim3(njobs), dim3(64), 0, s, d_frames, d_jobs, d_vcounts, d_vout, d_mtState, P, d_post, d_status);
    return hipGetLastError();
}
