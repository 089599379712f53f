/* oracle/ahc_oracle.cpp — TEST INFRASTRUCTURE (see oracle.h; parity unpinned vs Eigen 3.3.7).
 *
 * CPU restatement of the live plane extractor of the reference:
 *   PlaneDetection::readDepthImage / runPlaneDetection       src/PlaneExtractor.cpp:28-63
 *   ahc::PlaneFitter::run and everything it calls             include/peac/AHCPlaneFitter.hpp
 *   ahc::PlaneSeg, PlaneSeg::Stats                            include/peac/AHCPlaneSeg.hpp
 *   ahc::ParamSet (defaults, never overridden: SURVEY.md §9.8) include/peac/AHCParamSet.hpp
 *   DisjointSet                                               include/peac/DisjointSet.hpp
 *   LA::eig33sym -> Eigen::SelfAdjointEigenSolver<Matrix3d>   include/peac/eig33sym.hpp:70-74
 *
 * Canonicalised where the reference is address/implementation dependent: neighbour
 * sets iterate in node creation order, the min-MSE queue breaks exact MSE ties by creation order, the
 * final sort by N is stable.  None of these matter unless two doubles tie exactly.
 */
#include "ahc_oracle.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <map>
#include <memory>
#include <queue>
#include <set>

namespace orc {

/* ---------------------------------------------------------------------------------------------- */
/* Eigen 3.3.7 SelfAdjointEigenSolver<Matrix3d>::compute (iterative path), SURVEY.md §10.9          */

namespace {
struct Givens { double c, s; };
static Givens make_givens(double p, double q)   /* JacobiRotation::makeGivens, real case */
{
    Givens g;
    if (q == 0.0) { g.c = p < 0.0 ? -1.0 : 1.0; g.s = 0.0; }
    else if (p == 0.0) { g.c = 0.0; g.s = q < 0.0 ? 1.0 : -1.0; }
    else if (std::fabs(p) > std::fabs(q)) {
        const double t = q / p;
        double u = std::sqrt(1.0 + t * t);
        if (p < 0.0) u = -u;
        g.c = 1.0 / u;
        g.s = -t * g.c;
    } else {
        const double t = p / q;
        double u = std::sqrt(1.0 + t * t);
        if (q < 0.0) u = -u;
        g.s = -1.0 / u;
        g.c = -t * g.s;
    }
    return g;
}
static double eigen_hypot(double x, double y)   /* numext::hypot of 3.3.x */
{
    const double ax = std::fabs(x), ay = std::fabs(y);
    double p, qp;
    if (ax > ay) { p = ax; qp = ay / p; } else { p = ay; qp = ax / p; }
    if (p == 0.0) return 0.0;
    return p * std::sqrt(1.0 + qp * qp);
}
/* tridiagonal_qr_step; Q is column-major 3x3 (q[col*3+row]) */
static void qr_step(double* diag, double* subdiag, int start, int end, double* Q)
{
    const double td = (diag[end - 1] - diag[end]) * 0.5;
    const double e = subdiag[end - 1];
    double mu = diag[end];
    if (td == 0.0) mu -= std::fabs(e);
    else {
        const double e2 = e * e;
        const double h = eigen_hypot(td, e);
        if (e2 == 0.0) mu -= (e / (td + (td > 0.0 ? 1.0 : -1.0))) * (e / h);
        else mu -= e2 / (td + (td > 0.0 ? h : -h));
    }
    double x = diag[start] - mu;
    double z = subdiag[start];
    for (int k = start; k < end; ++k) {
        const Givens r = make_givens(x, z);
        const double sdk = r.s * diag[k] + r.c * subdiag[k];
        const double dkp1 = r.s * subdiag[k] + r.c * diag[k + 1];
        diag[k] = r.c * (r.c * diag[k] - r.s * subdiag[k]) - r.s * (r.c * subdiag[k] - r.s * diag[k + 1]);
        diag[k + 1] = r.s * sdk + r.c * dkp1;
        subdiag[k] = r.c * sdk - r.s * dkp1;
        if (k > start) subdiag[k - 1] = r.c * subdiag[k - 1] - r.s * z;
        x = subdiag[k];
        if (k < end - 1) {
            z = -r.s * subdiag[k + 1];
            subdiag[k + 1] = r.c * subdiag[k + 1];
        }
        /* Q = Q * G: applyOnTheRight(k, k+1, rot) == rotation (c, -s) on columns k, k+1 */
        for (int i = 0; i < 3; i++) {
            const double xi = Q[k * 3 + i], yi = Q[(k + 1) * 3 + i];
            Q[k * 3 + i] = r.c * xi - r.s * yi;
            Q[(k + 1) * 3 + i] = r.s * xi + r.c * yi;
        }
    }
}
} // namespace

void eig33sym(const double K[3][3], double s[3], double V[3][3])
{
    /* lower triangle of the (symmetric) input, scaled to [-1,1] */
    double m00 = K[0][0], m10 = K[0][1], m20 = K[0][2], m11 = K[1][1], m21 = K[1][2], m22 = K[2][2];
    double scale = std::max({std::fabs(m00), std::fabs(m10), std::fabs(m20), std::fabs(m11), std::fabs(m21), std::fabs(m22)});
    if (scale == 0.0) scale = 1.0;
    m00 /= scale; m10 /= scale; m20 /= scale; m11 /= scale; m21 /= scale; m22 /= scale;
    /* tridiagonalization_inplace_selector<Matrix3d,3,false>::run */
    double diag[3], sub[2], Q[9];
    const double tol = std::numeric_limits<double>::min();
    diag[0] = m00;
    const double v1norm2 = m20 * m20;
    if (v1norm2 <= tol) {
        diag[1] = m11; diag[2] = m22; sub[0] = m10; sub[1] = m21;
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        std::copy(I, I + 9, Q);
    } else {
        const double beta = std::sqrt(m10 * m10 + v1norm2);
        const double invBeta = 1.0 / beta;
        const double m01 = m10 * invBeta, m02 = m20 * invBeta;
        const double q = 2.0 * m01 * m21 + m02 * (m22 - m11);
        diag[1] = m11 + m02 * q;
        diag[2] = m22 - m02 * q;
        sub[0] = beta;
        sub[1] = m21 - m01 * q;
        /* mat << 1,0,0, 0,m01,m02, 0,m02,-m01 (row-wise fill) stored column-major */
        const double Qm[9] = {1, 0, 0, 0, m01, m02, 0, m02, -m01};
        std::copy(Qm, Qm + 9, Q);
    }
    /* computeFromTridiagonal_impl */
    const int n = 3, maxIterations = 30;
    int end = n - 1, start = 0, iter = 0;
    const double considerAsZero = std::numeric_limits<double>::min();
    const double precision = 2.0 * std::numeric_limits<double>::epsilon();
    while (end > 0) {
        for (int i = start; i < end; ++i)
            if (std::fabs(sub[i]) <= (std::fabs(diag[i]) + std::fabs(diag[i + 1])) * precision ||
                std::fabs(sub[i]) <= considerAsZero)
                sub[i] = 0;
        while (end > 0 && sub[end - 1] == 0.0) end--;
        if (end <= 0) break;
        iter++;
        if (iter > maxIterations * n) break;
        start = end - 1;
        while (start > 0 && sub[start - 1] != 0) start--;
        qr_step(diag, sub, start, end, Q);
    }
    if (iter <= maxIterations * n) {
        for (int i = 0; i < n - 1; ++i) {
            int k = 0;
            for (int j = 1; j < n - i; j++)
                if (diag[i + j] < diag[i + k]) k = j;
            if (k > 0) {
                std::swap(diag[i], diag[k + i]);
                for (int r = 0; r < 3; r++) std::swap(Q[i * 3 + r], Q[(k + i) * 3 + r]);
            }
        }
    }
    for (int i = 0; i < 3; i++) s[i] = diag[i] * scale;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) V[r][c] = Q[c * 3 + r];
}

/* ---------------------------------------------------------------------------------------------- */

namespace {

struct Params {   /* ahc::ParamSet defaults, AHCParamSet.hpp:68-75 */
    double depthSigma = 1.6e-6, stdTol_init = 5, stdTol_merge = 8, z_near = 500, z_far = 4000;
    double angle_near = 15.0 * M_PI / 180.0, angle_far = 90.0 * M_PI / 180.0;
    double similarityTh_merge = std::cos(60.0 * M_PI / 180.0), similarityTh_refine = std::cos(30.0 * M_PI / 180.0);
    double depthAlpha = 0.04, depthChangeTol = 0.02;
    enum Phase { P_INIT, P_MERGING, P_REFINE };
    double T_mse(Phase ph, double z) const
    {
        return ph == P_INIT ? std::pow(depthSigma * z * z + stdTol_init, 2) : std::pow(depthSigma * z * z + stdTol_merge, 2);
    }
    double T_ang(Phase ph, double z) const
    {
        if (ph == P_INIT) {
            double cz = z;
            cz = std::max(cz, z_near);
            cz = std::min(cz, z_far);
            const double factor = (angle_far - angle_near) / (z_far - z_near);
            return std::cos(factor * cz + angle_near - factor * z_near);
        }
        return ph == P_MERGING ? similarityTh_merge : similarityTh_refine;
    }
    double T_dz(double z) const { return depthAlpha * std::fabs(z) + depthChangeTol; }
};

struct Cloud {
    int w, h;
    std::vector<double> xyz;
    bool get(int row, int col, double& x, double& y, double& z) const
    {
        const size_t p = ((size_t)row * w + col) * 3;
        z = xyz[p + 2];
        if (z == 0 || std::isnan(z)) return false;
        x = xyz[p];
        y = xyz[p + 1];
        return true;
    }
};

struct Stats {
    double sx = 0, sy = 0, sz = 0, sxx = 0, syy = 0, szz = 0, sxy = 0, syz = 0, sxz = 0;
    int N = 0;
    void push(double x, double y, double z)
    {
        sx += x; sy += y; sz += z;
        sxx += x * x; syy += y * y; szz += z * z;
        sxy += x * y; syz += y * z; sxz += x * z;
        ++N;
    }
    void clear() { *this = Stats(); }
    void compute(double center[3], double normal[3], double& mse, double& curvature) const
    {
        const double sc = 1.0 / N;
        center[0] = sx * sc; center[1] = sy * sc; center[2] = sz * sc;
        double K[3][3] = {{sxx - sx * sx * sc, sxy - sx * sy * sc, sxz - sx * sz * sc},
                          {0, syy - sy * sy * sc, syz - sy * sz * sc},
                          {0, 0, szz - sz * sz * sc}};
        K[1][0] = K[0][1]; K[2][0] = K[0][2]; K[2][1] = K[1][2];
        double sv[3] = {0, 0, 0}, V[3][3];
        eig33sym(K, sv, V);
        if (V[0][0] * center[0] + V[1][0] * center[1] + V[2][0] * center[2] <= 0) {
            normal[0] = V[0][0]; normal[1] = V[1][0]; normal[2] = V[2][0];
        } else {
            normal[0] = -V[0][0]; normal[1] = -V[1][0]; normal[2] = -V[2][0];
        }
        mse = sv[0] * sc;
        curvature = sv[0] / (sv[0] + sv[1] + sv[2]);
    }
};

struct DisjointSet {
    std::vector<int> parent, size;
    explicit DisjointSet(int n) : parent(n), size(n, 1) { for (int i = 0; i < n; i++) parent[i] = i; }
    int Find(int x) { if (parent[x] != x) parent[x] = Find(parent[x]); return parent[x]; }
    int getSetSize(int x) { return size[Find(x)]; }
    int Union(int x, int y)
    {
        const int xr = Find(x), yr = Find(y);
        if (xr == yr) return xr;
        if (size[xr] < size[yr]) { parent[xr] = yr; size[yr] += size[xr]; return yr; }
        parent[yr] = xr; size[xr] += size[yr]; return xr;
    }
};

struct Seg;
struct BySeq { bool operator()(const Seg* a, const Seg* b) const; };
struct Seg {
    Stats stats;
    int rid = 0, N = 0;
    long seq = 0;
    double mse = 0, center[3] = {0, 0, 0}, normal[3] = {0, 0, 0}, curvature = 0;
    bool nouse = false;
    std::set<Seg*, BySeq> nbs;
    double normalSimilarity(const Seg& p) const
    {
        return std::fabs(normal[0] * p.normal[0] + normal[1] * p.normal[1] + normal[2] * p.normal[2]);
    }
    double signedDist(const double pt[3]) const
    {
        return normal[0] * (pt[0] - center[0]) + normal[1] * (pt[1] - center[1]) + normal[2] * (pt[2] - center[2]);
    }
    void connect(Seg* p) { if (p) { nbs.insert(p); p->nbs.insert(this); } }
    void disconnectAllNbs()
    {
        for (Seg* nb : nbs) nb->nbs.erase(this);
        nbs.clear();
    }
    void mergeNbsFrom(Seg& pa, Seg& pb, DisjointSet& ds)
    {
        ds.Union(pa.rid, pb.rid);
        nbs.insert(pa.nbs.begin(), pa.nbs.end());
        nbs.insert(pb.nbs.begin(), pb.nbs.end());
        nbs.erase(&pa);
        nbs.erase(&pb);
        pa.disconnectAllNbs();
        pb.disconnectAllNbs();
        for (Seg* nb : nbs) nb->nbs.insert(this);
        pa.nouse = pb.nouse = true;
    }
};
bool BySeq::operator()(const Seg* a, const Seg* b) const { return a->seq < b->seq; }

struct MinMse {   /* PlaneSegMinMSECmp + canonical tie-break */
    bool operator()(const Seg* a, const Seg* b) const
    {
        if (b->mse < a->mse) return true;
        if (a->mse < b->mse) return false;
        return b->seq < a->seq;
    }
};
typedef std::priority_queue<Seg*, std::vector<Seg*>, MinMse> MinQ;

struct Fitter {
    const Cloud* points = nullptr;
    int width = 0, height = 0;
    int maxStep = 100000, minSupport = 3000, windowWidth = 10, windowHeight = 10;
    Params params;
    std::unique_ptr<DisjointSet> ds;
    std::vector<std::unique_ptr<Seg>> pool;
    std::vector<Seg*> extractedPlanes;
    std::vector<int> membershipImg;
    std::map<int, int> rid2plid;
    std::vector<int> blkMap;
    std::vector<std::pair<int, int>> rfQueue;
    long seq = 0;

    Seg* newSeg() { pool.emplace_back(new Seg()); pool.back()->seq = seq++; return pool.back().get(); }

    /* PlaneSeg(points, rid, seed_row, seed_col, ...), AHCPlaneSeg.hpp:210-285 (INIT_STRICT) */
    Seg* initSeg(int rid, int seed_row, int seed_col)
    {
        Seg* s = newSeg();
        s->rid = rid;
        bool windowValid = true;
        for (int i = seed_row, icnt = 0; icnt < windowHeight && i < height; ++i, ++icnt) {
            for (int j = seed_col, jcnt = 0; jcnt < windowWidth && j < width; ++j, ++jcnt) {
                double x = 0, y = 0, z = 10000;
                if (!points->get(i, j, x, y, z)) { windowValid = false; break; }
                double xn = 0, yn = 0, zn = 10000;
                if (j + 1 < width && (points->get(i, j + 1, xn, yn, zn) && std::fabs(z - zn) > params.T_dz(z))) {
                    windowValid = false; break;
                }
                if (i + 1 < height && (points->get(i + 1, j, xn, yn, zn) && std::fabs(z - zn) > params.T_dz(z))) {
                    windowValid = false; break;
                }
                s->stats.push(x, y, z);
            }
            if (!windowValid) break;
        }
        if (windowValid) { s->nouse = false; s->N = s->stats.N; }
        else { s->N = 0; s->stats.clear(); s->nouse = true; }
        if (s->N < 4) s->mse = s->curvature = std::numeric_limits<double>::quiet_NaN();
        else s->stats.compute(s->center, s->normal, s->mse, s->curvature);
        return s;
    }
    /* PlaneSeg(pa, pb), AHCPlaneSeg.hpp:291-308 */
    Seg* mergeSeg(const Seg& pa, const Seg& pb)
    {
        Seg* s = newSeg();
        const Stats &a = pa.stats, &b = pb.stats;
        s->stats.sx = a.sx + b.sx; s->stats.sy = a.sy + b.sy; s->stats.sz = a.sz + b.sz;
        s->stats.sxx = a.sxx + b.sxx; s->stats.syy = a.syy + b.syy; s->stats.szz = a.szz + b.szz;
        s->stats.sxy = a.sxy + b.sxy; s->stats.syz = a.syz + b.syz; s->stats.sxz = a.sxz + b.sxz;
        s->stats.N = a.N + b.N;
        s->nouse = false;
        s->rid = pa.N >= pb.N ? pa.rid : pb.rid;
        s->N = s->stats.N;
        s->stats.compute(s->center, s->normal, s->mse, s->curvature);
        return s;
    }

    /* initGraph, AHCPlaneFitter.hpp:789-975 */
    void initGraph(MinQ& minQ, std::vector<AhcBlock>* blocksOut)
    {
        const int Nh = height / windowHeight, Nw = width / windowWidth;
        std::vector<Seg*> G(Nh * Nw, nullptr);
        if (blocksOut) blocksOut->assign(Nh * Nw, AhcBlock());
        for (int i = 0; i < Nh; ++i)
            for (int j = 0; j < Nw; ++j) {
                Seg* p = initSeg(i * Nw + j, i * windowHeight, j * windowWidth);
                const bool ok = p->mse < params.T_mse(Params::P_INIT, p->center[2]) && !p->nouse;
                if (ok) { G[i * Nw + j] = p; minQ.push(p); }
                if (blocksOut) {
                    AhcBlock& b = (*blocksOut)[i * Nw + j];
                    b.valid = ok; b.N = p->N; b.mse = p->mse; b.curvature = p->curvature;
                    for (int k = 0; k < 3; k++) { b.center[k] = p->center[k]; b.normal[k] = p->normal[k]; }
                    const Stats& st = p->stats;
                    const double v[9] = {st.sx, st.sy, st.sz, st.sxx, st.syy, st.szz, st.sxy, st.syz, st.sxz};
                    std::copy(v, v + 9, b.sums);
                }
            }
        for (int i = 0; i < Nh; ++i)
            for (int j = 1; j < Nw; j += 2) {
                const int cidx = i * Nw + j;
                if (G[cidx - 1] == 0) { --j; continue; }
                if (G[cidx] == 0) continue;
                if (j < Nw - 1 && G[cidx + 1] == 0) { ++j; continue; }
                const double th = params.T_ang(Params::P_INIT, G[cidx]->center[2]);
                if ((j < Nw - 1 && G[cidx - 1]->normalSimilarity(*G[cidx + 1]) >= th) ||
                    (j == Nw - 1 && G[cidx]->normalSimilarity(*G[cidx - 1]) >= th)) {
                    G[cidx]->connect(G[cidx - 1]);
                    if (j < Nw - 1) G[cidx]->connect(G[cidx + 1]);
                } else --j;
            }
        for (int j = 0; j < Nw; ++j)
            for (int i = 1; i < Nh; i += 2) {
                const int cidx = i * Nw + j;
                if (G[cidx - Nw] == 0) { --i; continue; }
                if (G[cidx] == 0) continue;
                if (i < Nh - 1 && G[cidx + Nw] == 0) { ++i; continue; }
                const double th = params.T_ang(Params::P_INIT, G[cidx]->center[2]);
                if ((i < Nh - 1 && G[cidx - Nw]->normalSimilarity(*G[cidx + Nw]) >= th) ||
                    (i == Nh - 1 && G[cidx]->normalSimilarity(*G[cidx - Nw]) >= th)) {
                    G[cidx]->connect(G[cidx - Nw]);
                    if (i < Nh - 1) G[cidx]->connect(G[cidx + Nw]);
                } else --i;
            }
    }

    /* ahCluster, AHCPlaneFitter.hpp:986-1192 */
    int ahCluster(MinQ& minQ)
    {
        int step = 0;
        while (!minQ.empty() && step <= maxStep) {
            Seg* p = minQ.top();
            minQ.pop();
            if (p->nouse) continue;
            Seg* cand_merge = nullptr;
            Seg* cand_nb = nullptr;
            for (Seg* nb : p->nbs) {
                if (p->normalSimilarity(*nb) < params.T_ang(Params::P_MERGING, p->center[2])) continue;
                Seg* merge = mergeSeg(*p, *nb);
                if (cand_merge == nullptr || cand_merge->mse > merge->mse ||
                    (cand_merge->mse == merge->mse && cand_merge->N < merge->mse)) {
                    cand_merge = merge;
                    cand_nb = nb;
                }
            }
            if (cand_merge != nullptr && cand_merge->mse < params.T_mse(Params::P_MERGING, cand_merge->center[2])) {
                minQ.push(cand_merge);
                cand_merge->mergeNbsFrom(*p, *cand_nb, *ds);
            } else {
                if (p->N >= minSupport) extractedPlanes.push_back(p);
                p->disconnectAllNbs();
            }
            ++step;
        }
        while (!minQ.empty()) {
            Seg* p = minQ.top();
            minQ.pop();
            if (p->N >= minSupport) extractedPlanes.push_back(p);
            p->disconnectAllNbs();
        }
        std::stable_sort(extractedPlanes.begin(), extractedPlanes.end(), [](const Seg* a, const Seg* b) { return b->N < a->N; });
        return step;
    }

    static int valid4(int i, int j, int H, int W, int nbs[4])
    {
        const int id = i * W + j;
        int cnt = 0;
        if (j > 0) nbs[cnt++] = id - 1;
        if (j < W - 1) nbs[cnt++] = id + 1;
        if (i > 0) nbs[cnt++] = id - W;
        if (i < H - 1) nbs[cnt++] = id + W;
        return cnt;
    }
    int getBlockIdx(int pixX, int pixY) const
    {
        const int Nw = width / windowWidth, Nh = height / windowHeight;
        const int by = pixY / windowHeight, bx = pixX / windowWidth;
        return (by < Nh && bx < Nw) ? (by * Nw + bx) : -1;
    }

    /* findBlockMembership(isValidExtractedPlane), AHCPlaneFitter.hpp:488-590 (ERODE_ALL_BORDER) */
    void findBlockMembership(std::vector<bool>& isValid)
    {
        rid2plid.clear();
        for (int plid = 0; plid < (int)extractedPlanes.size(); ++plid) rid2plid.insert({extractedPlanes[plid]->rid, plid});
        const int Nh = height / windowHeight, Nw = width / windowWidth, NptsPerBlk = windowHeight * windowWidth;
        membershipImg.assign((size_t)width * height, -1);
        blkMap.assign(Nh * Nw, 0);
        isValid.assign(extractedPlanes.size(), false);
        for (int i = 0, blkid = 0; i < Nh; ++i)
            for (int j = 0; j < Nw; ++j, ++blkid) {
                const int setid = ds->Find(blkid);
                const int setSize = ds->getSetSize(setid) * NptsPerBlk;
                if (setSize >= minSupport) {
                    int nbs[4] = {-1};
                    const int nNbs = valid4(i, j, Nh, Nw, nbs);
                    bool same = true;
                    for (int k = 0; k < nNbs; ++k)
                        if (ds->Find(nbs[k]) != setid) { same = false; break; }   /* ERODE_ALL_BORDER */
                    const int plid = rid2plid[setid];
                    if (same) {
                        blkMap[blkid] = plid;
                        for (int y = i * windowHeight; y < (i + 1) * windowHeight; y++)
                            for (int x = j * windowWidth; x < (j + 1) * windowWidth; x++) membershipImg[(size_t)y * width + x] = plid;
                        isValid[plid] = true;
                    } else blkMap[blkid] = -1;
                } else blkMap[blkid] = -1;
                if (blkMap[blkid] < 0) {
                    if (i > 0) {
                        const int u = blkid - Nw;
                        if (blkMap[u] >= 0) {
                            const int spix = (i * windowHeight - 1) * width + j * windowWidth;
                            for (int k = 1; k < windowWidth; ++k) rfQueue.push_back({spix + k, blkMap[u]});
                        }
                    }
                    if (j > 0) {
                        const int l = blkid - 1;
                        if (blkMap[l] >= 0) {
                            const int spix = (i * windowHeight) * width + j * windowWidth - 1;
                            for (int k = 0; k < windowHeight - 1; ++k) rfQueue.push_back({spix + k * width, blkMap[l]});
                        }
                    }
                } else {
                    const int plid = blkMap[blkid];
                    if (i > 0) {
                        const int u = blkid - Nw;
                        if (blkMap[u] != plid) {
                            const int spix = (i * windowHeight) * width + j * windowWidth;
                            for (int k = 0; k < windowWidth - 1; ++k) rfQueue.push_back({spix + k, plid});
                        }
                    }
                    if (j > 0) {
                        const int l = blkid - 1;
                        if (blkMap[l] != plid) {
                            const int spix = (i * windowHeight) * width + j * windowWidth;
                            for (int k = 1; k < windowHeight; ++k) rfQueue.push_back({spix + k * width, plid});
                        }
                    }
                }
            }
    }

    /* floodFill, AHCPlaneFitter.hpp:431-479 */
    void floodFill()
    {
        std::vector<float> distMap((size_t)height * width, std::numeric_limits<float>::max());
        for (int k = 0; k < (int)rfQueue.size(); ++k) {
            const int sIdx = rfQueue[k].first;
            const int seedy = sIdx / width, seedx = sIdx - seedy * width;
            const int plid = rfQueue[k].second;
            Seg& pl = *extractedPlanes[plid];
            int nbs[4] = {-1};
            const int Nnbs = valid4(seedy, seedx, height, width, nbs);
            for (int itr = 0; itr < Nnbs; ++itr) {
                const int cIdx = nbs[itr];
                int& trail = membershipImg[cIdx];
                if (trail <= -6) continue;
                if (trail >= 0 && trail == plid) continue;
                const int cy = cIdx / width, cx = cIdx - cy * width;
                const int blkid = getBlockIdx(cx, cy);
                if (blkid >= 0 && blkMap[blkid] >= 0) continue;
                double pt[3] = {0};
                float cdist = -1;
                if (points->get(cy, cx, pt[0], pt[1], pt[2]) &&
                    std::pow(cdist = (float)std::fabs(pl.signedDist(pt)), 2) < 9 * pl.mse + 1e-5) {
                    if (trail >= 0) {
                        Seg& n_pl = *extractedPlanes[trail];
                        if (pl.normalSimilarity(n_pl) >= params.T_ang(Params::P_REFINE, pl.center[2])) n_pl.connect(extractedPlanes[plid]);
                    }
                    float& old_dist = distMap[cIdx];
                    if (cdist < old_dist) {
                        trail = plid;
                        old_dist = cdist;
                        rfQueue.push_back({cIdx, plid});
                    } else if (trail < 0) trail -= 1;
                } else {
                    if (trail < 0) trail -= 1;
                }
            }
        }
    }

    /* refineDetails, AHCPlaneFitter.hpp:299-382 */
    void refineDetails(AhcResult& out)
    {
        std::vector<bool> isValid;
        findBlockMembership(isValid);
        floodFill();
        std::vector<Seg*> old;
        extractedPlanes.swap(old);
        MinQ minQ;
        for (int i = 0; i < (int)old.size(); ++i)
            if (isValid[i]) minQ.push(old[i]);
        ahCluster(minQ);
        std::vector<int> plidmap(old.size(), -1);
        const size_t nFinal = extractedPlanes.size();
        for (int i = 0; i < (int)old.size(); ++i) {
            if (!isValid[i]) { plidmap[i] = -1; continue; }
            const int np_rid = ds->Find(old[i]->rid);
            for (size_t j = 0; j < extractedPlanes.size(); ++j)
                if (np_rid == extractedPlanes[j]->rid) { plidmap[i] = (int)j; break; }
        }
        out.membership.assign(nFinal, {});
        out.seg.assign((size_t)width * height, 0);
        const int nPixels = width * height;
        for (int i = 0; i < nPixels; ++i) {
            int& plid = membershipImg[i];
            if (plid >= 0 && plidmap[plid] >= 0) {
                plid = plidmap[plid];
                out.seg[i] = (uint8_t)(plid + 1);
                out.membership[plid].push_back(i);
            }
        }
    }
};

} // namespace

/* PlaneDetection::readDepthImage (src/PlaneExtractor.cpp:28-55) + runPlaneDetection (:57-63) */
AhcResult ahc_run(const uint16_t* depth, int w, int h, const float K4[4], float depthfactor, std::vector<AhcBlock>* blocksOut)
{
    Cloud cloud;
    cloud.w = w; cloud.h = h;
    cloud.xyz.assign((size_t)w * h * 3, 0.0);
    const float fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++) {
            const double z = (double)depth[(size_t)i * w + j] * depthfactor;
            double* p = &cloud.xyz[((size_t)i * w + j) * 3];
            if (z > 5.0) { p[0] = p[1] = p[2] = 0; continue; }
            p[0] = ((double)j - cx) * z / fx;
            p[1] = ((double)i - cy) * z / fy;
            p[2] = z;
        }
    Fitter f;
    f.points = &cloud;
    f.width = w; f.height = h;
    f.ds.reset(new DisjointSet((h / f.windowHeight) * (w / f.windowWidth)));
    MinQ minQ;
    f.initGraph(minQ, blocksOut);
    f.ahCluster(minQ);
    AhcResult out;
    f.refineDetails(out);
    for (Seg* s : f.extractedPlanes) {
        AhcPlane pl;
        for (int k = 0; k < 3; k++) { pl.normal[k] = s->normal[k]; pl.center[k] = s->center[k]; }
        pl.mse = s->mse; pl.curvature = s->curvature; pl.N = s->N; pl.rid = s->rid;
        out.planes.push_back(pl);
    }
    return out;
}

/* taps for tests/test_ref_pins.py: the thresholds and the disjoint set of this restatement, to be compared with the
 * reference's own include/peac/AHCParamSet.hpp and DisjointSet.hpp compiled into oracle/_ref */
void ahc_thresholds(int phase, double z, double out3[3])
{
    const Params p;
    out3[0] = p.T_mse((Params::Phase)phase, z);
    out3[1] = p.T_ang((Params::Phase)phase, z);
    out3[2] = p.T_dz(z);
}
void ahc_disjoint_set(int n, const int32_t* pairs, int npairs, int32_t* unionRet, int32_t* findOut, int32_t* sizeOut)
{
    DisjointSet ds(n);
    for (int i = 0; i < npairs; i++) unionRet[i] = ds.Union(pairs[2 * i], pairs[2 * i + 1]);
    for (int i = 0; i < n; i++) { findOut[i] = ds.Find(i); sizeOut[i] = ds.getSetSize(i); }
}

} // namespace orc
