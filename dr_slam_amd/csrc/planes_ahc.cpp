/* planes_ahc.cpp — AHC depth-plane extraction behind drfe_planes_ahc (include/drfe.h).
 *
 * Split of the reference's ahc::PlaneFitter::run (include/peac/AHCPlaneFitter.hpp:211-260):
 *   device  k_ahc_blocks: cloud + 3072 init-block plane fits (the only image-sized pass)
 *   host    graph edges (initGraph :896-975), agglomerative clustering (ahCluster :986-1192),
 *           block erosion + seed collection (findBlockMembership :488-590), region growing
 *           (floodFill :431-479), final re-merge and relabel (refineDetails :299-382)
 * The host part is the round-1 placement ("clustering on host first"): it is a
 * sequential min-heap / FIFO walk over <= 3072 nodes.  It is written index-based (node ids = creation
 * order, sorted-vector adjacency) so it can move into a one-workgroup-per-frame kernel later.
 *
 * Canonical tie rules where the reference depends on heap addresses: neighbours
 * iterate in creation order, equal-MSE queue entries pop in creation order, the final sort by N is
 * stable.
 */
#include "drfe_internal.h"
#include <dlfcn.h>
#include "planes_internal.h"
#include "ahc_math.h"
#include "ahc_math_simd.h"
#include "post_internal.h"
#include "cr_sincos.h"
#include <condition_variable>
#include <deque>
#include <mutex>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <new>
#include <queue>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <thread>
#include <unordered_map>
#include <vector>
#include <string>

#define HIPCHK(c, call)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (c)->err = std::string(#call) + ": " + hipGetErrorString(e__);                      \
            return DRFE_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

namespace {

struct Node {
    double S[9];
    int N, rid;
    AhcFit fit;
    bool nouse;
    std::vector<int> nbs;   /* sorted node ids */
};

struct Graph {
    std::vector<Node> nodes;
    std::vector<int> dsParent, dsSize;

    int dsFind(int x) { while (dsParent[x] != x) { dsParent[x] = dsParent[dsParent[x]]; x = dsParent[x]; } return x; }
    void dsUnion(int x, int y)
    {
        const int xr = dsFind(x), yr = dsFind(y);
        if (xr == yr) return;
        if (dsSize[xr] < dsSize[yr]) { dsParent[xr] = yr; dsSize[yr] += dsSize[xr]; }
        else { dsParent[yr] = xr; dsSize[xr] += dsSize[yr]; }
    }
    static void insertSorted(std::vector<int>& v, int id)
    {
        auto it = std::lower_bound(v.begin(), v.end(), id);
        if (it == v.end() || *it != id) v.insert(it, id);
    }
    static void eraseSorted(std::vector<int>& v, int id)
    {
        auto it = std::lower_bound(v.begin(), v.end(), id);
        if (it != v.end() && *it == id) v.erase(it);
    }
    void connect(int a, int b) { insertSorted(nodes[a].nbs, b); insertSorted(nodes[b].nbs, a); }
    void disconnectAll(int a)
    {
        for (int nb : nodes[a].nbs) eraseSorted(nodes[nb].nbs, a);
        nodes[a].nbs.clear();
    }
    double similarity(int a, int b) const
    {
        const double* n = nodes[a].fit.normal;
        const double* m = nodes[b].fit.normal;
        return std::fabs(n[0] * m[0] + n[1] * m[1] + n[2] * m[2]);
    }
};

struct QEntry { double mse; int id; };
struct QCmp {
    bool operator()(const QEntry& a, const QEntry& b) const
    {
        if (b.mse < a.mse) return true;
        if (a.mse < b.mse) return false;
        return b.id < a.id;
    }
};
typedef std::priority_queue<QEntry, std::vector<QEntry>, QCmp> MinQ;

const double kCos60 = std::cos(60.0 * M_PI / 180.0);   /* similarityTh_merge */
const double kCos30 = std::cos(30.0 * M_PI / 180.0);   /* similarityTh_refine */

static double tMseMerge(double z) { const double t = 1.6e-6 * z * z + 8.0; return t * t; }
static double tAngInit(double z)
{   /* ParamSet::T_ang(P_INIT, z) with the never-overridden millimetre defaults */
    const double z_near = 500, z_far = 4000, a_near = 15.0 * M_PI / 180.0, a_far = 90.0 * M_PI / 180.0;
    double cz = std::max(z, z_near);
    cz = std::min(cz, z_far);
    const double factor = (a_far - a_near) / (z_far - z_near);
    /* the cosine correctly rounded (cr_sincos.h), as the device path computes it (ahc_frame_kernels.hip); the host's libm
     * agrees except where it is not correctly rounded (glibc >= 2.28: one argument in ~10^4, by one ulp) */
    const double a = factor * cz + a_near - factor * z_near;
    double sn, cs;
    if (drfe_cr_sincos(a, &sn, &cs)) return cs;
    return std::cos(a);
}

/* merged statistics of two nodes (ahc::PlaneSeg(pa, pb): sums added, plane refitted) without the neighbour list */
struct Merged { double S[9]; int N, rid; AhcFit fit; };

/* The plane fits of n trial merges (their sums in m[i].S / m[i].N): W at a time in the lanes of the host's vector unit
 * (ahc_math_simd.h: bit-identical to the scalar routine), scalar where the CPU has neither AVX2 nor AVX-512F. */
template <int W>
static inline __attribute__((always_inline)) void solve_trials_w(Merged* m, int n)
{
    for (int i = 0; i < n; i += W) {
        double S[9][W];
        int N[W];
        AhcFit fit[W];
        for (int l = 0; l < W; l++) {
            const Merged& t = m[std::min(i + l, n - 1)];         /* idle lanes repeat the last trial */
            for (int k = 0; k < 9; k++) S[k][l] = t.S[k];
            N[l] = t.N;
        }
        ahc_simd::plane_from_sums<W>(S, N, fit);
        for (int l = 0; l < W && i + l < n; l++) m[i + l].fit = fit[l];
    }
}
/* AVX2: 8 lanes as two 4-wide registers per value - the two halves' division / square-root chains overlap (1.53 ms against
 * 1.74 ms with 4 lanes for the 32 000 trials of the room frame on the EPYC 9575F; AVX-512F: 1.10 ms; scalar 4.47 ms) */
__attribute__((target("avx2"))) static void solve_trials_avx2(Merged* m, int n) { solve_trials_w<8>(m, n); }
__attribute__((target("avx512f"))) static void solve_trials_avx512(Merged* m, int n) { solve_trials_w<8>(m, n); }

static void solve_trials_scalar(Merged* m, int n)
{
    for (int i = 0; i < n; i++) ahc_plane_from_sums(m[i].S, m[i].N, &m[i].fit);
}
/* mode: 0 scalar, 1 AVX2 (8 lanes in two registers), 2 AVX-512F (8 lanes); -1: the widest this CPU has (DRFE_AHC_SIMD=0/1/2 overrides) */
static void solve_trials(Merged* m, int n, int mode = -1)
{
    static const int best = [] {
        int b = 0;
        if (__builtin_cpu_supports("avx2")) b = 1;
        if (__builtin_cpu_supports("avx512f")) b = 2;
        if (const char* e = std::getenv("DRFE_AHC_SIMD")) b = std::min(b, std::max(0, std::atoi(e)));
        return b;
    }();
    if (mode < 0) mode = best;
    if (mode == 2 && n > 4) solve_trials_avx512(m, n);
    else if (mode >= 1 && n > 1) solve_trials_avx2(m, n);
    else solve_trials_scalar(m, n);
}

/* ahCluster: pops min-MSE nodes, merges with the neighbour giving the least merged MSE.
 * Measured on the host harness (drfe_planes_ahc_from_blocks, room frame: 1525 steps): 31 600 trial solves, ~21 per step - a
 * grown plane tries every block of its boundary - and they are the cost (9.2 of 9.9 ms there).  An edge dies with the first
 * of its end points to be popped (merged or extracted), so no trial is ever asked for twice: a per-pair cache cannot hit
 * (tried: 7.0 ms against 5.1 ms on the box).  Skipping trials by the eigenvalue bound 4 det / tr^2 pruned 1.4 % of them: the
 * candidates of a step are blocks of the same plane and their merged MSEs differ by percents, not factors.  What does help:
 * the trials of one step are independent, so they are solved side by side in vector lanes (solve_trials): 4.47 -> 1.10 ms for
 * the solves, 5.2 -> 1.8 ms for ahCluster on the EPYC 9575F of the GPU box (profiles/r02_ahc_simd_modes.txt). */
static void cluster(Graph& g, MinQ& q, std::vector<int>& extracted)
{
    const int maxStep = 100000;
    int step = 0;
    std::vector<int> nbs, u, trialNb;
    std::vector<Merged> trials;
    while (!q.empty() && step <= maxStep) {
        const int p = q.top().id;
        q.pop();
        if (g.nodes[p].nouse) continue;
        int candNb = -1;
        const Merged* cand = nullptr;
        nbs = g.nodes[p].nbs;
        trials.clear();
        trialNb.clear();
        for (int nb : nbs) {
            if (g.similarity(p, nb) < kCos60) continue;
            Merged m;
            const Node &a = g.nodes[p], &b = g.nodes[nb];
            for (int k = 0; k < 9; k++) m.S[k] = a.S[k] + b.S[k];
            m.N = a.N + b.N;
            m.rid = a.N >= b.N ? a.rid : b.rid;
            trials.push_back(m);
            trialNb.push_back(nb);
        }
        /* the ~21 trial fits of a step are independent: solved side by side, then folded in neighbour order */
        solve_trials(trials.data(), (int)trials.size());
        for (size_t t = 0; t < trials.size(); t++) {
            const Merged& m = trials[t];
            if (!cand || cand->fit.mse > m.fit.mse || (cand->fit.mse == m.fit.mse && cand->N < m.fit.mse)) {
                cand = &m;
                candNb = trialNb[t];
            }
        }
        if (cand && cand->fit.mse < tMseMerge(cand->fit.center[2])) {
            const int id = (int)g.nodes.size();
            g.nodes.emplace_back();
            Node& nn = g.nodes.back();
            std::memcpy(nn.S, cand->S, sizeof(nn.S));
            nn.N = cand->N; nn.rid = cand->rid; nn.fit = cand->fit; nn.nouse = false;
            q.push({nn.fit.mse, id});
            /* mergeNbsFrom */
            g.dsUnion(g.nodes[p].rid, g.nodes[candNb].rid);
            u.clear();
            std::set_union(g.nodes[p].nbs.begin(), g.nodes[p].nbs.end(), g.nodes[candNb].nbs.begin(),
                           g.nodes[candNb].nbs.end(), std::back_inserter(u));
            Graph::eraseSorted(u, p);
            Graph::eraseSorted(u, candNb);
            g.disconnectAll(p);
            g.disconnectAll(candNb);
            g.nodes[id].nbs = u;
            for (int nb : u) Graph::insertSorted(g.nodes[nb].nbs, id);
            g.nodes[p].nouse = g.nodes[candNb].nouse = true;
        } else {
            if (g.nodes[p].N >= AHC_MIN_SUPPORT) extracted.push_back(p);
            g.disconnectAll(p);
        }
        ++step;
    }
    while (!q.empty()) {
        const int p = q.top().id;
        q.pop();
        if (g.nodes[p].N >= AHC_MIN_SUPPORT) extracted.push_back(p);
        g.disconnectAll(p);
    }
    std::stable_sort(extracted.begin(), extracted.end(), [&](int a, int b) { return g.nodes[b].N < g.nodes[a].N; });
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

struct DepthView {
    const uint16_t* d; size_t stride; int w, h; double factor, fx, fy, cx, cy;
    bool get(int row, int col, double pt[3]) const
    {
        double z = (double)d[(size_t)row * stride + col] * factor;
        if (z > 5.0) z = 0.0;
        if (z == 0.0) return false;
        pt[0] = ((double)col - cx) * z / fx;
        pt[1] = ((double)row - cy) * z / fy;
        pt[2] = z;
        return true;
    }
};

} // namespace

/* One plane-extraction lane of drfe_planes_ahc_batch: device scratch + stream + its own error string (the
 * templates below only touch these members, which drfe_ctx has under the same names). */
struct PlaneLane {
    std::string err;
    int device = 0;
    PlanesScratch* ps = nullptr;
    hipStream_t stream = nullptr;
    struct VoxelDevice* vox = nullptr;      /* drfe_planes_ahc_post_batch: the lane's device voxel grid (post_internal.h) */
    float* h_coarse = nullptr; size_t coarseCap = 0;   /* pinned: a frame's voxel clouds as the device extractor's pipeline left them */
};

static void scratch_free(PlanesScratch*& p)
{
    if (!p) return;
    if (p->d_blocks) (void)hipFree(p->d_blocks);
    if (p->d_depth) (void)hipFree(p->d_depth);
    delete p;
    p = nullptr;
}

void drfe_planes_free(drfe_ctx* c)
{
    scratch_free(c->ps);
    if (CapeScratch* cs = static_cast<CapeScratch*>(c->cape)) {
        void* ptrs[] = {cs->d_depth, cs->d_cells, cs->d_seg, cs->d_tab};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
        delete cs;
        c->cape = nullptr;
    }
    drfe_ahc_arena_free(c);
    auto* pool = static_cast<std::vector<PlaneLane>*>(c->planeLanes);
    if (pool) {
        for (PlaneLane& l : *pool) {
            scratch_free(l.ps);
            if (l.stream) (void)hipStreamDestroy(l.stream);
            drfe_voxel_device_free(l.vox);
            if (l.h_coarse) (void)hipHostFree(l.h_coarse);
        }
        delete pool;
        c->planeLanes = nullptr;
    }
}

template <class Ctx> static int ensure_scratch(Ctx* c, int w, int h)
{
    if (!c->ps) {
        c->ps = new (std::nothrow) PlanesScratch();
        if (!c->ps) return DRFE_ERR_INVALID;
        std::memset(c->ps, 0, sizeof(PlanesScratch));
    }
    PlanesScratch* p = c->ps;
    const size_t nb = (size_t)(w / AHC_WIN) * (h / AHC_WIN), px = (size_t)w * h;
    if (p->blocksCap < nb) {
        if (p->d_blocks) (void)hipFree(p->d_blocks);
        p->d_blocks = nullptr; p->blocksCap = 0;
        HIPCHK(c, hipMalloc((void**)&p->d_blocks, nb * sizeof(AhcBlockRec)));
        p->blocksCap = nb;
    }
    if (p->depthCap < px) {
        if (p->d_depth) (void)hipFree(p->d_depth);
        p->d_depth = nullptr; p->depthCap = 0;
        HIPCHK(c, hipMalloc((void**)&p->d_depth, px * sizeof(uint16_t)));
        p->depthCap = px;
    }
    return DRFE_OK;
}

template <class Ctx> static int run_blocks(Ctx* c, const uint16_t* depth, int w, int h, size_t stride, const float K4[4],
                      float depth_factor, std::vector<AhcBlockRec>& blocks)
{
    if (!c || !depth || !K4 || w < AHC_WIN || h < AHC_WIN || stride < (size_t)w) {
        if (c) c->err = "planes_ahc: invalid argument";
        return DRFE_ERR_INVALID;
    }
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_scratch(c, w, h);
    if (rc != DRFE_OK) return rc;
    PlanesScratch* p = c->ps;
    HIPCHK(c, hipMemcpy2DAsync(p->d_depth, (size_t)w * 2, depth, stride * 2, (size_t)w * 2, (size_t)h, hipMemcpyHostToDevice,
                               c->stream));
    HIPCHK(c, drfe_launch_ahc_blocks(p->d_depth, (size_t)w * h, (size_t)w, w, h, K4, depth_factor, 1, p->d_blocks, c->stream));
    const size_t nb = (size_t)(w / AHC_WIN) * (h / AHC_WIN);
    blocks.resize(nb);
    HIPCHK(c, hipMemcpyAsync(blocks.data(), p->d_blocks, nb * sizeof(AhcBlockRec), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DRFE_OK;
}

/* the host half alone: block fits supplied by the caller (drfe_planes_ahc_from_blocks: CPU tests and profiling) */
struct HostBlocks { std::string err; const AhcBlockRec* blocks; size_t n; };
static int run_blocks(HostBlocks* c, const uint16_t* depth, int w, int h, size_t stride, const float K4[4], float, std::vector<AhcBlockRec>& blocks)
{
    const size_t nb = (size_t)(w / AHC_WIN) * (h / AHC_WIN);
    if (!c || !depth || !K4 || w < AHC_WIN || h < AHC_WIN || stride < (size_t)w || c->n != nb) {
        if (c) c->err = "planes_ahc_from_blocks: invalid argument";
        return DRFE_ERR_INVALID;
    }
    blocks.assign(c->blocks, c->blocks + nb);
    return DRFE_OK;
}

struct AhcTrace {
    bool on = std::getenv("DRFE_TRACE_PLANES") != nullptr;
    std::chrono::steady_clock::time_point t[8];
    int n = 0;
    AhcTrace() { if (on) t[n++] = std::chrono::steady_clock::now(); }
    void mark(int) { if (on && n < 8) t[n++] = std::chrono::steady_clock::now(); }
    ~AhcTrace()
    {
        if (!on) return;
        t[n++] = std::chrono::steady_clock::now();
        static const char* names[] = {"device blocks + copies", "initGraph", "ahCluster", "findBlockMembership", "floodFill", "re-merge + relabel"};
        std::fprintf(stderr, "drfe_planes_ahc:");
        for (int i = 0; i + 1 < n && i < 6; i++)
            std::fprintf(stderr, " %s %.2f ms;", names[i], std::chrono::duration<double, std::milli>(t[i + 1] - t[i]).count());
        std::fprintf(stderr, "\n");
    }
};

/* PlaneDetection::readDepthImage + runPlaneDetection for one frame on one lane (Ctx: drfe_ctx or PlaneLane) */
template <class Ctx>
static int planes_ahc_core(Ctx* c, const uint16_t* depth, int w, int h, size_t stride, const float* K4, float depth_factor,
                           drfe_plane* planes, int cap, int* n_planes, uint8_t* seg, int32_t* member_offsets,
                           int32_t* member_idx)
{
    if (!n_planes) return DRFE_ERR_INVALID;
    *n_planes = 0;
    std::vector<AhcBlockRec> blocks;
    AhcTrace tr;               /* DRFE_TRACE_PLANES=1: wall time per stage on stderr */
    int rc = run_blocks(c, depth, w, h, stride, K4, depth_factor, blocks);
    if (rc != DRFE_OK) return rc;
    const int Nw = w / AHC_WIN, Nh = h / AHC_WIN, NB = Nw * Nh;

    tr.mark(0);
    /* --- initGraph: nodes + 4-neighbour edges with the reference's skip pattern ------------------- */
    Graph g;
    g.dsParent.resize(NB); g.dsSize.assign(NB, 1);
    for (int i = 0; i < NB; i++) g.dsParent[i] = i;
    g.nodes.reserve(2 * NB);
    std::vector<int> G(NB, -1);
    MinQ q;
    for (int b = 0; b < NB; b++) {
        if (!blocks[b].valid) continue;
        Node n;
        std::memcpy(n.S, blocks[b].sums, sizeof(n.S));
        n.N = blocks[b].N; n.rid = b; n.nouse = false;
        std::memcpy(n.fit.center, blocks[b].center, 24);
        std::memcpy(n.fit.normal, blocks[b].normal, 24);
        n.fit.mse = blocks[b].mse; n.fit.curvature = blocks[b].curvature;
        G[b] = (int)g.nodes.size();
        g.nodes.push_back(n);
        q.push({n.fit.mse, G[b]});
    }
    for (int i = 0; i < Nh; ++i)
        for (int j = 1; j < Nw; j += 2) {
            const int cidx = i * Nw + j;
            if (G[cidx - 1] < 0) { --j; continue; }
            if (G[cidx] < 0) continue;
            if (j < Nw - 1 && G[cidx + 1] < 0) { ++j; continue; }
            const double th = tAngInit(g.nodes[G[cidx]].fit.center[2]);
            if ((j < Nw - 1 && g.similarity(G[cidx - 1], G[cidx + 1]) >= th) ||
                (j == Nw - 1 && g.similarity(G[cidx], G[cidx - 1]) >= th)) {
                g.connect(G[cidx], G[cidx - 1]);
                if (j < Nw - 1) g.connect(G[cidx], G[cidx + 1]);
            } else --j;
        }
    for (int j = 0; j < Nw; ++j)
        for (int i = 1; i < Nh; i += 2) {
            const int cidx = i * Nw + j;
            if (G[cidx - Nw] < 0) { --i; continue; }
            if (G[cidx] < 0) continue;
            if (i < Nh - 1 && G[cidx + Nw] < 0) { ++i; continue; }
            const double th = tAngInit(g.nodes[G[cidx]].fit.center[2]);
            if ((i < Nh - 1 && g.similarity(G[cidx - Nw], G[cidx + Nw]) >= th) ||
                (i == Nh - 1 && g.similarity(G[cidx], G[cidx - Nw]) >= th)) {
                g.connect(G[cidx], G[cidx - Nw]);
                if (i < Nh - 1) g.connect(G[cidx], G[cidx + Nw]);
            } else --i;
        }

    tr.mark(1);
    /* --- ahCluster ------------------------------------------------------------------------------- */
    std::vector<int> extracted;
    cluster(g, q, extracted);

    tr.mark(2);
    /* --- refineDetails: findBlockMembership ------------------------------------------------------- */
    std::map<int, int> rid2plid;
    for (int plid = 0; plid < (int)extracted.size(); ++plid) rid2plid.insert({g.nodes[extracted[plid]].rid, plid});
    std::vector<int> membership((size_t)w * h, -1), blkMap(NB, 0);
    std::vector<char> isValid(extracted.size(), 0);
    std::vector<std::pair<int, int>> rf;
    const int NptsPerBlk = AHC_WIN * AHC_WIN;
    for (int i = 0, blkid = 0; i < Nh; ++i)
        for (int j = 0; j < Nw; ++j, ++blkid) {
            const int setid = g.dsFind(blkid);
            const int setSize = g.dsSize[setid] * NptsPerBlk;
            if (setSize >= AHC_MIN_SUPPORT) {
                int nbs[4] = {-1};
                const int nNbs = valid4(i, j, Nh, Nw, nbs);
                bool same = true;
                for (int k = 0; k < nNbs; ++k)
                    if (g.dsFind(nbs[k]) != setid) { same = false; break; }      /* ERODE_ALL_BORDER */
                const int plid = rid2plid[setid];      /* std::map::operator[] as in the reference */
                if (same) {
                    blkMap[blkid] = plid;
                    for (int y = i * AHC_WIN; y < (i + 1) * AHC_WIN; y++)
                        for (int x = j * AHC_WIN; x < (j + 1) * AHC_WIN; x++) membership[(size_t)y * w + x] = plid;
                    isValid[plid] = 1;
                } else blkMap[blkid] = -1;
            } else blkMap[blkid] = -1;
            if (blkMap[blkid] < 0) {
                if (i > 0 && blkMap[blkid - Nw] >= 0) {
                    const int spix = (i * AHC_WIN - 1) * w + j * AHC_WIN;
                    for (int k = 1; k < AHC_WIN; ++k) rf.push_back({spix + k, blkMap[blkid - Nw]});
                }
                if (j > 0 && blkMap[blkid - 1] >= 0) {
                    const int spix = (i * AHC_WIN) * w + j * AHC_WIN - 1;
                    for (int k = 0; k < AHC_WIN - 1; ++k) rf.push_back({spix + k * w, blkMap[blkid - 1]});
                }
            } else {
                const int plid = blkMap[blkid];
                if (i > 0 && blkMap[blkid - Nw] != plid) {
                    const int spix = (i * AHC_WIN) * w + j * AHC_WIN;
                    for (int k = 0; k < AHC_WIN - 1; ++k) rf.push_back({spix + k, plid});
                }
                if (j > 0 && blkMap[blkid - 1] != plid) {
                    const int spix = (i * AHC_WIN) * w + j * AHC_WIN;
                    for (int k = 1; k < AHC_WIN; ++k) rf.push_back({spix + k * w, plid});
                }
            }
        }

    tr.mark(3);
    /* --- floodFill -------------------------------------------------------------------------------- */
    DepthView dv = {depth, stride, w, h, (double)depth_factor, (double)K4[0], (double)K4[1], (double)K4[2], (double)K4[3]};
    {
        std::vector<float> distMap((size_t)w * h, std::numeric_limits<float>::max());
        const double invW = 1.0 / (double)w;
        for (size_t k = 0; k < rf.size(); ++k) {
            const int sIdx = rf[k].first, plid = rf[k].second;
            /* sIdx / w without the division instruction: (sIdx + 0.5) / w is at least 0.5 / w away from an integer */
            const int seedy = (int)(((double)sIdx + 0.5) * invW), seedx = sIdx - seedy * w;
            const Node& pl = g.nodes[extracted[plid]];
            /* valid4's order - left, right, up, down - with the neighbour's row / column carried along */
            int nbs[4], nby[4], nbx[4], Nn = 0;
            if (seedx > 0) { nbs[Nn] = sIdx - 1; nby[Nn] = seedy; nbx[Nn++] = seedx - 1; }
            if (seedx < w - 1) { nbs[Nn] = sIdx + 1; nby[Nn] = seedy; nbx[Nn++] = seedx + 1; }
            if (seedy > 0) { nbs[Nn] = sIdx - w; nby[Nn] = seedy - 1; nbx[Nn++] = seedx; }
            if (seedy < h - 1) { nbs[Nn] = sIdx + w; nby[Nn] = seedy + 1; nbx[Nn++] = seedx; }
            for (int t = 0; t < Nn; ++t) {
                const int cIdx = nbs[t];
                int& trail = membership[cIdx];
                if (trail <= -6) continue;
                if (trail >= 0 && trail == plid) continue;
                const int cy = nby[t], cx = nbx[t];
                const int by = cy / AHC_WIN, bx = cx / AHC_WIN;
                const int blkid = (by < Nh && bx < Nw) ? by * Nw + bx : -1;
                if (blkid >= 0 && blkMap[blkid] >= 0) continue;
                double pt[3] = {0, 0, 0};
                float cdist = -1;
                bool in = false;
                if (dv.get(cy, cx, pt)) {
                    const double sd = pl.fit.normal[0] * (pt[0] - pl.fit.center[0]) + pl.fit.normal[1] * (pt[1] - pl.fit.center[1]) +
                                      pl.fit.normal[2] * (pt[2] - pl.fit.center[2]);
                    cdist = (float)std::fabs(sd);
                    const double cd = (double)cdist;
                    in = cd * cd < 9 * pl.fit.mse + 1e-5;       /* std::pow(float -> double, 2) */
                }
                if (in) {
                    if (trail >= 0) {
                        const int other = extracted[trail];
                        if (g.similarity(extracted[plid], other) >= kCos30) g.connect(other, extracted[plid]);
                    }
                    float& old = distMap[cIdx];
                    if (cdist < old) {
                        trail = plid;
                        old = cdist;
                        rf.push_back({cIdx, plid});
                    } else if (trail < 0) trail -= 1;
                } else if (trail < 0) trail -= 1;
            }
        }
    }

    tr.mark(4);
    /* --- re-merge the grown planes and relabel ----------------------------------------------------- */
    std::vector<int> old;
    old.swap(extracted);
    MinQ q2;
    for (size_t i = 0; i < old.size(); ++i)
        if (isValid[i]) q2.push({g.nodes[old[i]].fit.mse, old[i]});
    cluster(g, q2, extracted);
    std::vector<int> plidmap(old.size(), -1);
    for (size_t i = 0; i < old.size(); ++i) {
        if (!isValid[i]) continue;
        const int np_rid = g.dsFind(g.nodes[old[i]].rid);
        for (size_t j = 0; j < extracted.size(); ++j)
            if (np_rid == g.nodes[extracted[j]].rid) { plidmap[i] = (int)j; break; }
    }
    const int nFinal = (int)extracted.size();
    *n_planes = nFinal;
    if (nFinal > cap) { c->err = "planes_ahc: plane buffer too small"; return DRFE_ERR_CAPACITY; }
    if (nFinal > 254) { c->err = "planes_ahc: more planes than a CV_8U label image holds"; return DRFE_ERR_CAPACITY; }
    for (int i = 0; i < nFinal && planes; i++) {
        const Node& n = g.nodes[extracted[i]];
        std::memcpy(planes[i].normal, n.fit.normal, 24);
        std::memcpy(planes[i].center, n.fit.center, 24);
        planes[i].mse = n.fit.mse; planes[i].curvature = n.fit.curvature;
        planes[i].n_points = n.N; planes[i].rid = n.rid;
    }
    std::vector<int> counts(nFinal + 1, 0);
    {   /* final plane number per pixel, the label image and the member counts in one pass over RUNS of equal membership (a
         * per-pixel `counts[plid]++` serialises on the counter's store-to-load chain: 2.7 ns per pixel against 0.4) */
        const size_t np = membership.size();
        int* mem = membership.data();
        for (size_t i = 0; i < np;) {
            const int raw = mem[i];
            size_t j = i + 1;
            while (j < np && mem[j] == raw) j++;
            const int plid = raw >= 0 ? plidmap[raw] : -1;
            if (plid != raw) std::fill(mem + i, mem + j, plid);
            if (plid >= 0) counts[plid + 1] += (int)(j - i);
            if (seg) std::memset(seg + i, plid + 1, j - i);
            i = j;
        }
    }
    if (member_offsets) {
        for (int i = 0; i < nFinal; i++) counts[i + 1] += counts[i];
        for (int i = 0; i <= nFinal; i++) member_offsets[i] = counts[i];
        if (member_idx) {
            std::vector<int> fill(counts.begin(), counts.end() - 1);
            const size_t np = membership.size();
            const int* mem = membership.data();
            for (size_t i = 0; i < np;) {                    /* run by run, as above */
                const int plid = mem[i];
                size_t j = i + 1;
                while (j < np && mem[j] == plid) j++;
                if (plid >= 0) {
                    int32_t* dst = member_idx + fill[plid];
                    for (size_t k = i; k < j; k++) *dst++ = (int32_t)k;
                    fill[plid] += (int)(j - i);
                }
                i = j;
            }
        }
    }
    return DRFE_OK;
}

extern "C" {

int drfe_planes_ahc_blocks(drfe_ctx* c, const uint16_t* depth, int w, int h, size_t stride, const float* K4,
                           float depth_factor, double* blocks17, int32_t* valid_n, int cap)
{
    std::vector<AhcBlockRec> blocks;
    int rc = run_blocks(c, depth, w, h, stride, K4, depth_factor, blocks);
    if (rc != DRFE_OK) return rc;
    if ((int)blocks.size() > cap) return DRFE_ERR_CAPACITY;
    for (size_t i = 0; i < blocks.size(); i++) {
        const AhcBlockRec& b = blocks[i];
        for (int k = 0; k < 9; k++) blocks17[17 * i + k] = b.sums[k];
        for (int k = 0; k < 3; k++) { blocks17[17 * i + 9 + k] = b.center[k]; blocks17[17 * i + 12 + k] = b.normal[k]; }
        blocks17[17 * i + 15] = b.mse; blocks17[17 * i + 16] = b.curvature;
        valid_n[2 * i] = b.valid; valid_n[2 * i + 1] = b.N;
    }
    return DRFE_OK;
}

int drfe_planes_ahc(drfe_ctx* c, const uint16_t* depth, int w, int h, size_t stride, const float* K4, float depth_factor,
                    drfe_plane* planes, int cap, int* n_planes, uint8_t* seg, int32_t* member_offsets,
                    int32_t* member_idx)
{
    return planes_ahc_core(c, depth, w, h, stride, K4, depth_factor, planes, cap, n_planes, seg, member_offsets, member_idx);
}

/* The same for nframes host depth images (depth + f * frame_stride elements).  The block fits are microseconds on
 * the device; the clustering / erosion / flood fill of a frame (~7 ms) is sequential host code, independent between
 * frames, so the batch runs on n_threads host threads with one device lane each.  Outputs per frame f:
 * planes[f * cap], n_planes[f], seg + f * w * h, member_offsets[f * (cap + 1)], member_idx + f * w * h (any of the
 * last three may be NULL).  Results equal nframes calls of drfe_planes_ahc. */
} /* extern "C" */
static int planes_ahc_post_batch_device(drfe_ctx* c, std::vector<PlaneLane>* pool, int T, const uint16_t* depth, size_t frame_stride, int w, int h,
                                        size_t stride, int nframes, const float* K4, float depth_factor, float max_point_dist, double dist_threshold,
                                        drfe_plane* planes, int cap, int* n_planes, uint8_t* seg, drfe_plane_post* post, int* n_accepted, int* plane_num,
                                        int32_t* member_offsets, int32_t* member_idx);
extern "C" {

int drfe_planes_ahc_batch(drfe_ctx* c, const uint16_t* depth, size_t frame_stride, int w, int h, size_t stride, int nframes,
                          const float* K4, float depth_factor, drfe_plane* planes, int cap, int* n_planes, uint8_t* seg,
                          int32_t* member_offsets, int32_t* member_idx, int n_threads)
{
    if (!c || !depth || !K4 || !planes || !n_planes || nframes < 0 || cap < 1 || frame_stride < stride * (size_t)h) {
        if (c) c->err = "planes_ahc_batch: invalid argument";
        return DRFE_ERR_INVALID;
    }
    if (nframes == 0) return DRFE_OK;
    /* default: 1.25 threads per CPU - a lane sleeps in stream synchronisations for about a fifth of a frame's time */
    int T = n_threads > 0 ? n_threads : std::max(1, drfe_default_host_threads() * 5 / 4);
    T = std::max(1, std::min(T, nframes));
    HIPCHK(c, hipSetDevice(c->device));
    auto* pool = static_cast<std::vector<PlaneLane>*>(c->planeLanes);
    if (!pool) { pool = new std::vector<PlaneLane>(); c->planeLanes = pool; }
    while ((int)pool->size() < T) {
        PlaneLane l;
        l.device = c->device;
        HIPCHK(c, hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        pool->push_back(l);
    }
    /* the extractor on the device, one wavefront per frame (drfe_planes_configure_extractor; same limits as the post batch) */
    if (c->planesDeviceAhc && nframes > 1 && drfe_ahc_device_fits(w, h) && !std::getenv("DRFE_AHC_HOST"))
        return planes_ahc_post_batch_device(c, pool, T, depth, frame_stride, w, h, stride, nframes, K4, depth_factor, 0.f, 0.0, planes, cap, n_planes, seg,
                                            nullptr, nullptr, nullptr, member_offsets, member_idx);
    std::vector<int> rcs(T, DRFE_OK);
    std::vector<std::thread> th;
    th.reserve(T);
    std::atomic<int> next(0);
    const size_t px = (size_t)w * h;
    for (int k = 0; k < T; k++)
        th.emplace_back([&, k]() {
            PlaneLane* l = &(*pool)[k];
            for (int f = next.fetch_add(1); f < nframes; f = next.fetch_add(1)) {
                const int rc = planes_ahc_core(l, depth + (size_t)f * frame_stride, w, h, stride, K4, depth_factor,
                                               planes + (size_t)f * cap, cap, &n_planes[f], seg ? seg + f * px : nullptr,
                                               member_offsets ? member_offsets + (size_t)f * (cap + 1) : nullptr,
                                               member_idx ? member_idx + f * px : nullptr);
                if (rc != DRFE_OK) { rcs[k] = rc; return; }
            }
        });
    for (std::thread& t : th) t.join();
    for (int k = 0; k < T; k++)
        if (rcs[k] != DRFE_OK) { c->err = (*pool)[k].err; return rcs[k]; }
    return DRFE_OK;
}

/* Test hook: the plane fits of n (sums, N) records by the scalar routine (mode 0), the 4-lane (1) or the 8-lane (2) vector
 * instantiation; out8 = center, normal, mse, curvature per record.  Returns DRFE_ERR_STATE if the CPU lacks the mode. */
int drfe_debug_ahc_trials(const double* sums9, const int32_t* N, int n, int mode, double* out8)
{
    if (!sums9 || !N || !out8 || n < 0 || mode < 0 || mode > 2) return DRFE_ERR_INVALID;
    if ((mode == 1 && !__builtin_cpu_supports("avx2")) || (mode == 2 && !__builtin_cpu_supports("avx512f"))) return DRFE_ERR_STATE;
    std::vector<Merged> m((size_t)n);
    for (int i = 0; i < n; i++) { std::memcpy(m[i].S, sums9 + 9 * (size_t)i, 72); m[i].N = N[i]; m[i].rid = 0; }
    if (mode == 0) solve_trials_scalar(m.data(), n);
    else if (mode == 1) solve_trials_avx2(m.data(), n);
    else solve_trials_avx512(m.data(), n);
    for (int i = 0; i < n; i++) {
        double* o = out8 + 8 * (size_t)i;
        std::memcpy(o, m[i].fit.center, 24); std::memcpy(o + 3, m[i].fit.normal, 24);
        o[6] = m[i].fit.mse; o[7] = m[i].fit.curvature;
    }
    return DRFE_OK;
}

/* The host half of drfe_planes_ahc (graph, agglomerative clustering, block membership, flood fill, re-merge, labels) on
 * caller-supplied block fits, without a device: blocks17 = per 10x10 block 9 sums, center, normal, mse, curvature (the layout
 * drfe_planes_ahc_blocks returns), valid_n = (enters-graph flag, N) per block.  Host code: CPU tests and profiling. */
int drfe_planes_ahc_from_blocks(const double* blocks17, const int32_t* valid_n, const uint16_t* depth, int w, int h, size_t stride,
                                const float* K4, float depth_factor, drfe_plane* planes, int cap, int* n_planes, uint8_t* seg,
                                int32_t* member_offsets, int32_t* member_idx)
{
    if (!blocks17 || !valid_n || !n_planes || w < AHC_WIN || h < AHC_WIN) return DRFE_ERR_INVALID;
    const size_t nb = (size_t)(w / AHC_WIN) * (h / AHC_WIN);
    std::vector<AhcBlockRec> rec(nb);
    for (size_t i = 0; i < nb; i++) {
        const double* b = blocks17 + 17 * i;
        std::memcpy(rec[i].sums, b, 72);
        std::memcpy(rec[i].center, b + 9, 24);
        std::memcpy(rec[i].normal, b + 12, 24);
        rec[i].mse = b[15]; rec[i].curvature = b[16];
        rec[i].valid = valid_n[2 * i]; rec[i].N = valid_n[2 * i + 1];
    }
    HostBlocks hb{std::string(), rec.data(), nb};
    return planes_ahc_core(&hb, depth, w, h, stride, K4, depth_factor, planes, cap, n_planes, seg, member_offsets, member_idx);
}

/* drfe_planes_ahc_batch followed, on the same worker thread and frame, by the per-plane loop of Frame::ComputePlanes
 * (drfe_planes_ahc_postprocess): what a frame of the plane path costs end to end, nframes at a time.  post: [nframes][cap];
 * n_accepted / plane_num: [nframes]; the voxel clouds are not returned (use the single-frame call for mvPlanePoints). */
} /* extern "C" */

/* ---- the extractor on the device for a batch (ahc_frame_kernels.hip): frame slots in HBM ------------------------------------ */
struct AhcArena {
    int w = 0, h = 0, frames = 0, cap = 0;
    bool ready = false;       /* every allocation succeeded: a half-built arena is freed, not reused */
    AhcDevParams P;
    uint16_t* d_depth = nullptr; AhcBlockRec* d_blocks = nullptr;
    uint8_t* d_scratch = nullptr; size_t slotBytes = 0;      /* per-frame scratch + outputs, one block per slot */
    AhcDevFrame* d_frames = nullptr; AhcDevFrame* h_frames = nullptr;
    int* h_out = nullptr; drfe_plane* h_planes = nullptr; int* h_memberOff = nullptr;
    uint16_t* h_depth = nullptr;                              /* pinned staging of the caller's depth images */
    /* pcl::VoxelGrid behind the extractor (voxel_kernels.hip): every frame's plane clouds, the sort's scratch, centroids, jobs */
    float* d_vpts = nullptr; unsigned long long* d_vrecs = nullptr; unsigned long long* d_vtmp = nullptr; uint32_t* d_vposL = nullptr;
    uint32_t* d_vposR = nullptr; float* d_vout = nullptr; int2* d_jobs = nullptr; int* d_vcounts = nullptr; int* d_vlist = nullptr;
    int2* h_jobs = nullptr; int* h_vcounts = nullptr;
    /* gates + RANSAC refit behind the voxel grids (refit_kernels.hip): per plane slot the post record and its status */
    drfe_plane_post* d_post = nullptr; int* d_postStatus = nullptr; uint32_t* d_mtState = nullptr;
    drfe_plane_post* h_post = nullptr; int* h_postStatus = nullptr;
    /* offsets of the outputs inside a slot */
    size_t offPlanes = 0, offSeg = 0, offMemberOff = 0, offMemberIdx = 0, offOut = 0;
};

static void arena_free(AhcArena*& a)
{
    if (!a) return;
    void* d[] = {a->d_depth, a->d_blocks, a->d_scratch, a->d_frames, a->d_vpts, a->d_vrecs, a->d_vtmp, a->d_vposL, a->d_vposR, a->d_vout, a->d_jobs, a->d_vcounts, a->d_vlist,
                 a->d_post, a->d_postStatus, a->d_mtState};
    for (void* p : d) if (p) (void)hipFree(p);
    void* hp[] = {a->h_frames, a->h_out, a->h_planes, a->h_memberOff, a->h_depth, a->h_jobs, a->h_vcounts, a->h_post, a->h_postStatus};
    for (void* p : hp) if (p) (void)hipHostFree(p);
    delete a;
    a = nullptr;
}

void drfe_ahc_arena_free(drfe_ctx* c)
{
    AhcArena* a = static_cast<AhcArena*>(c->ahcArena);
    arena_free(a);
    c->ahcArena = nullptr;
}

#define AHC_DEV_PLANE_CAP 64

static int ensure_arena(drfe_ctx* c, int w, int h, int frames, const float* K4, float depth_factor, float max_point_dist)
{
    AhcArena* a = static_cast<AhcArena*>(c->ahcArena);
    if (a && (!a->ready || a->w != w || a->h != h || a->frames < frames)) { arena_free(a); c->ahcArena = nullptr; }
    if (!a) {
        a = new (std::nothrow) AhcArena();
        if (!a) return DRFE_ERR_INVALID;
        c->ahcArena = a;
        a->w = w; a->h = h; a->frames = frames;
        const int Nw = w / AHC_WIN, Nh = h / AHC_WIN, NB = Nw * Nh;
        const size_t npx = (size_t)w * h;
        AhcDevParams& P = a->P;
        P.w = w; P.h = h; P.Nw = Nw; P.Nh = Nh; P.NB = NB;
        /* capacities scale with the frame: 2^19 neighbour-pool words and 2^20 flood-fill queue entries for the 3072 blocks /
         * 307 200 pixels of a 640 x 480 frame (3.4 entries per pixel) */
        P.maxNodes = 2 * NB + 256; P.planeCap = AHC_DEV_PLANE_CAP;
        P.poolCap = 1 << 19; P.rfCap = 1 << 20;
        while (P.poolCap < 170 * NB) P.poolCap <<= 1;
        while ((size_t)P.rfCap < npx * 7 / 2) P.rfCap <<= 1;
        if (P.maxNodes > 65535) { c->ahcArena = nullptr; delete a; c->err = "planes: the device extractor keeps node ids in 16 bits (frame too large)"; return DRFE_ERR_INVALID; }
        /* per-slot layout, every array 256-byte aligned */
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
        const size_t oS = take((size_t)P.maxNodes * 72), oFit = take((size_t)P.maxNodes * 64), oN = take((size_t)P.maxNodes * 4), oRid = take((size_t)P.maxNodes * 4),
                     oNouse = take((size_t)P.maxNodes), oNbOff = take((size_t)P.maxNodes * 4), oNbLen = take((size_t)P.maxNodes * 4), oPool = take((size_t)P.poolCap * 4),
                     oDsP = take((size_t)NB * 4), oDsS = take((size_t)NB * 4), oG = take((size_t)NB * 4), oBlk = take((size_t)NB * 4), oR2P = take((size_t)NB * 4),
                     oMem = take(npx * 2), oDist = take(npx * 4), oRf = take((size_t)P.rfCap * 4), oHo = take(AHC_HANDOFF_INTS * 4);
        a->offPlanes = take((size_t)P.planeCap * sizeof(drfe_plane)); a->offSeg = take(npx); a->offMemberOff = take(((size_t)P.planeCap + 1) * 4);
        a->offMemberIdx = take(npx * 4); a->offOut = take(16);
        a->slotBytes = off;
        const size_t F = (size_t)frames;
        HIPCHK(c, hipMalloc((void**)&a->d_depth, npx * 2 * F));
        HIPCHK(c, hipMalloc((void**)&a->d_blocks, (size_t)NB * sizeof(AhcBlockRec) * F));
        HIPCHK(c, hipMalloc((void**)&a->d_scratch, a->slotBytes * F));
        HIPCHK(c, hipMalloc((void**)&a->d_frames, sizeof(AhcDevFrame) * F));
        HIPCHK(c, hipHostMalloc((void**)&a->h_frames, sizeof(AhcDevFrame) * F, hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void**)&a->h_out, 16 * F, hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void**)&a->h_planes, (size_t)P.planeCap * sizeof(drfe_plane) * F, hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void**)&a->h_memberOff, ((size_t)P.planeCap + 1) * 4 * F, hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void**)&a->h_depth, npx * 2 * F, hipHostMallocDefault));
        if (npx * F > (size_t)0x7fffffff) { c->err = "planes_ahc_post_batch: batch too large"; return DRFE_ERR_CAPACITY; }
        HIPCHK(c, hipMalloc((void**)&a->d_vpts, npx * 12 * F));
        HIPCHK(c, hipMalloc((void**)&a->d_vrecs, npx * 8 * F));
        HIPCHK(c, hipMalloc((void**)&a->d_vtmp, npx * 8 * F));
        HIPCHK(c, hipMalloc((void**)&a->d_vposL, npx * 4 * F));
        HIPCHK(c, hipMalloc((void**)&a->d_vposR, npx * 4 * F));
        HIPCHK(c, hipMalloc((void**)&a->d_vout, npx * 12 * F));
        HIPCHK(c, hipMalloc((void**)&a->d_jobs, sizeof(int2) * P.planeCap * F));
        HIPCHK(c, hipMalloc((void**)&a->d_vcounts, sizeof(int) * P.planeCap * F));
        HIPCHK(c, hipMalloc((void**)&a->d_vlist, sizeof(int) * (P.planeCap * F + 2 * 16)));      /* job order of each chunk (<= 16 chunks) */
        HIPCHK(c, hipHostMalloc((void**)&a->h_jobs, sizeof(int2) * P.planeCap * F, hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void**)&a->h_vcounts, sizeof(int) * P.planeCap * F, hipHostMallocDefault));
        HIPCHK(c, hipMalloc((void**)&a->d_post, sizeof(drfe_plane_post) * P.planeCap * F));
        HIPCHK(c, hipMalloc((void**)&a->d_postStatus, sizeof(int) * P.planeCap * F));
        HIPCHK(c, hipHostMalloc((void**)&a->h_post, sizeof(drfe_plane_post) * P.planeCap * F, hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void**)&a->h_postStatus, sizeof(int) * P.planeCap * F, hipHostMallocDefault));
        {
            /* std::mt19937(12345): the state after seeding (its first draw twists it) */
            uint32_t mt[624];
            mt[0] = 12345u;
            for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
            HIPCHK(c, hipMalloc((void**)&a->d_mtState, sizeof(mt)));
            HIPCHK(c, hipMemcpy(a->d_mtState, mt, sizeof(mt), hipMemcpyHostToDevice));
        }
        for (size_t f = 0; f < F; f++) {
            uint8_t* s = a->d_scratch + a->slotBytes * f;
            AhcDevFrame& g = a->h_frames[f];
            g.depth = a->d_depth + npx * f; g.rowStride = (size_t)w;
            g.blocks = a->d_blocks + (size_t)NB * f;
            g.nodeS = (double*)(s + oS); g.nodeFit = (double*)(s + oFit); g.nodeN = (int*)(s + oN); g.nodeRid = (int*)(s + oRid);
            g.nodeNouse = s + oNouse; g.nbOff = (int*)(s + oNbOff); g.nbLen = (int*)(s + oNbLen); g.nbPool = (int*)(s + oPool);
            g.dsParent = (int*)(s + oDsP); g.dsSize = (int*)(s + oDsS); g.G = (int*)(s + oG); g.blkMap = (int*)(s + oBlk); g.ridToPlid = (int*)(s + oR2P);
            g.membership = (int16_t*)(s + oMem); g.distMap = (float*)(s + oDist); g.rf = (uint32_t*)(s + oRf); g.handoff = (int*)(s + oHo);
            g.planes = (drfe_plane*)(s + a->offPlanes); g.seg = s + a->offSeg; g.memberOff = (int*)(s + a->offMemberOff);
            g.memberIdx = (int*)(s + a->offMemberIdx); g.out = (int*)(s + a->offOut);
            g.ptsBase = (int)(npx * f); g.pts = a->d_vpts + 3 * npx * f; g.jobs = a->d_jobs + (size_t)P.planeCap * f;
        }
        HIPCHK(c, hipMemcpy(a->d_frames, a->h_frames, sizeof(AhcDevFrame) * F, hipMemcpyHostToDevice));
        a->ready = true;
    }
    AhcDevParams& P = a->P;
    P.fx = (double)K4[0]; P.fy = (double)K4[1]; P.cx = (double)K4[2]; P.cy = (double)K4[3]; P.factor = (double)depth_factor;
    P.cos60 = kCos60; P.cos30 = kCos30; P.maxPointDist = max_point_dist;
    return DRFE_OK;
}

namespace {
struct AhcBatchJob {
    drfe_ctx* c; AhcArena* A; std::vector<PlaneLane>* pool;
    const uint16_t* depth; size_t frameStride, stride; int w, h, nframes, cap;
    const float* K4; float depthFactor, maxPointDist; double distThreshold;
    drfe_plane* planes; int* nPlanes; uint8_t* seg; drfe_plane_post* post; int* nAccepted; int* planeNum;
    int32_t* memberOffsets; int32_t* memberIdx;      /* drfe_planes_ahc_batch (post == null): the member lists go to the caller */
    int chunk, nChunks;
    bool voxDevice;          /* k_voxel_grid ran behind the extractor: the workers fetch centroids instead of member lists */
    bool refitDevice;        /* k_plane_refit ran behind the voxel grids: the workers fetch 24-byte post records instead of centroids */
    std::atomic<int> refitFallbacks{0};
    std::vector<hipStream_t> chunkStream; std::vector<hipEvent_t> chunkDone; std::vector<int> chunkState;   /* 1 = on the device, 3 = a worker fetches its results, 2 = released */
    std::mutex mu; std::condition_variable cv; std::deque<int> finishQ; int pending = 0;
    int firstRc = DRFE_OK; std::string firstErr; bool abort = false;
    std::atomic<int> fallbacks{0}, voxFallbacks{0};
};
}

static void ahc_batch_fail(AhcBatchJob& J, int rc, const std::string& err)
{
    std::lock_guard<std::mutex> lk(J.mu);
    if (J.firstRc == DRFE_OK) { J.firstRc = rc; J.firstErr = err; }
    J.abort = true;
    J.cv.notify_all();
}

/* a worker: takes frames whose chunk has left the device, fetches the frame's member lists and runs the per-plane loop */
static std::atomic<long long> g_planeCpuNs[4];      /* DRFE_TRACE_PLANES: thread CPU time of the workers by section: fetch | frame download + wait | grids redone | gates + refit */
static inline long long plane_thread_cpu_ns() { struct timespec t; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t); return t.tv_sec * 1000000000LL + t.tv_nsec; }

static void ahc_batch_worker(AhcBatchJob& J, PlaneLane* l)
{
    AhcArena* A = J.A;
    const size_t px = (size_t)J.w * J.h;
    (void)hipSetDevice(J.c->device);
    std::vector<int32_t> off((size_t)J.cap + 1), idx(px), voff((size_t)J.cap + 1);
    std::vector<const float*> cptr((size_t)A->P.planeCap);
    std::vector<int> vcl((size_t)A->P.planeCap);
    std::vector<std::vector<float>> redo;
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ahc_batch_fail(J, DRFE_ERR_HIP, "planes batch: event"); return; }
    for (;;) {
        int f = -1, waitCh = -1, fetchCh = -1;
        {
            std::unique_lock<std::mutex> lk(J.mu);
            for (;;) {
                if (J.abort) { (void)hipEventDestroy(ev); return; }
                for (int ch = 0; ch < J.nChunks && fetchCh < 0; ch++)
                    if (J.chunkState[ch] == 1 && hipEventQuery(J.chunkDone[ch]) == hipSuccess) { J.chunkState[ch] = 3; fetchCh = ch; }
                if (fetchCh >= 0) break;
                if (!J.finishQ.empty()) { f = J.finishQ.front(); J.finishQ.pop_front(); break; }
                if (J.pending == 0) { (void)hipEventDestroy(ev); return; }
                for (int ch = 0; ch < J.nChunks && waitCh < 0; ch++) if (J.chunkState[ch] == 1) waitCh = ch;
                if (waitCh >= 0) break;
                J.cv.wait(lk);
            }
        }
        if (fetchCh >= 0) {
            const long long tF = plane_thread_cpu_ns();
            struct AccF { long long t; ~AccF() { g_planeCpuNs[0] += plane_thread_cpu_ns() - t; } } accF{tF};
            /* the small results of every frame of the chunk; member lists, centroids and label images are fetched per frame */
            const int f0 = fetchCh * J.chunk, nf = std::min(J.chunk, J.nframes - f0);
            const size_t pc = (size_t)A->P.planeCap;
            hipStream_t st = l->stream;
            hipError_t e = hipMemcpy2DAsync(A->h_out + 4 * (size_t)f0, 16, A->d_scratch + A->slotBytes * f0 + A->offOut, A->slotBytes, 16, nf, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess)
                e = hipMemcpy2DAsync(A->h_planes + pc * f0, sizeof(drfe_plane) * pc, A->d_scratch + A->slotBytes * f0 + A->offPlanes, A->slotBytes, sizeof(drfe_plane) * pc, nf,
                                     hipMemcpyDeviceToHost, st);
            if (e == hipSuccess)
                e = hipMemcpy2DAsync(A->h_memberOff + (pc + 1) * f0, 4 * (pc + 1), A->d_scratch + A->slotBytes * f0 + A->offMemberOff, A->slotBytes, 4 * (pc + 1), nf,
                                     hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && J.voxDevice) e = hipMemcpyAsync(A->h_jobs + pc * f0, A->d_jobs + pc * f0, sizeof(int2) * pc * nf, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && J.voxDevice) e = hipMemcpyAsync(A->h_vcounts + pc * f0, A->d_vcounts + pc * f0, sizeof(int) * pc * nf, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && J.refitDevice) e = hipMemcpyAsync(A->h_post + pc * f0, A->d_post + pc * f0, sizeof(drfe_plane_post) * pc * nf, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess && J.refitDevice) e = hipMemcpyAsync(A->h_postStatus + pc * f0, A->d_postStatus + pc * f0, sizeof(int) * pc * nf, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = drfe_pool_sync(st, ev);
            if (e != hipSuccess) { ahc_batch_fail(J, DRFE_ERR_HIP, std::string("planes batch: results of a chunk: ") + hipGetErrorString(e)); (void)hipEventDestroy(ev); return; }
            std::lock_guard<std::mutex> lk(J.mu);
            J.chunkState[fetchCh] = 2;
            for (int k = 0; k < nf; k++) J.finishQ.push_back(f0 + k);
            J.cv.notify_all();
            continue;
        }
        if (waitCh >= 0) {
            if (drfe_event_wait_sleeping(J.chunkDone[waitCh]) != hipSuccess) { ahc_batch_fail(J, DRFE_ERR_HIP, "planes batch: device"); (void)hipEventDestroy(ev); return; }
            continue;
        }
        const uint16_t* d = J.depth + (size_t)f * J.frameStride;
        drfe_plane* pl = J.planes + (size_t)f * J.cap;
        int rc = DRFE_OK;
        bool viaCoarse = false;
        const int nP = A->h_out[4 * (size_t)f], status = A->h_out[4 * (size_t)f + 1];
        if (status != 0 || nP > J.cap) {
            /* a capacity of the device path ran out (or a cosine could not be certified): this frame on the host */
            J.fallbacks++;
            rc = planes_ahc_core(l, d, J.w, J.h, J.stride, J.K4, J.depthFactor, pl, J.cap, &J.nPlanes[f], J.seg ? J.seg + f * px : nullptr, off.data(), idx.data());
        } else {
            J.nPlanes[f] = nP;
            if (nP) std::memcpy(pl, A->h_planes + (size_t)A->P.planeCap * f, sizeof(drfe_plane) * nP);
            const int* mo = A->h_memberOff + ((size_t)A->P.planeCap + 1) * f;
            for (int i = 0; i <= nP; i++) off[i] = mo[i];
            const uint8_t* slot = A->d_scratch + A->slotBytes * f;
            const int2* jobs = A->h_jobs + (size_t)A->P.planeCap * f;
            const int* vc = A->h_vcounts + (size_t)A->P.planeCap * f;
            /* planes the device handed back (vc < 0: the sort's heap-sort branch, a grid beyond int32): their gathered clouds come
             * back instead and this thread runs their grids */
            /* gates + refit ran on the device too (k_plane_refit): the frame's post records are here already; only a plane whose
             * decision the device could not certify, or whose grid came back, sends the frame down the centroid path below */
            bool postReady = false;
            if (J.refitDevice) {
                const int* ps = A->h_postStatus + (size_t)A->P.planeCap * f;
                postReady = true;
                for (int i = 0; i < nP; i++) if (ps[i] != 0) postReady = false;
                if (postReady) {
                    const drfe_plane_post* hp = A->h_post + (size_t)A->P.planeCap * f;
                    int nAcc = 0, fail = 0;
                    for (int i = 0; i < nP; i++) {
                        J.post[(size_t)f * J.cap + i] = hp[i];
                        nAcc += hp[i].accepted;
                        const drfe_plane& e = pl[i];
                        const float d = (float)-(e.normal[0] * e.center[0] + e.normal[1] * e.center[1] + e.normal[2] * e.center[2]);
                        if (d > J.maxPointDist || hp[i].n_voxels < 100) fail++;
                    }
                    J.nAccepted[f] = nAcc;
                    if (J.planeNum) J.planeNum[f] = nP - fail;
                } else J.refitFallbacks++;
            }
            const bool coarseReady = J.voxDevice && !postReady;
            size_t nCoarse = 0;
            int handedBack = 0;
            for (int i = 0; i < nP && coarseReady; i++) { vcl[i] = vc[i]; if (vc[i] < 0) handedBack++; nCoarse += (size_t)(vc[i] < 0 ? jobs[i].y : vc[i]); }
            hipError_t e = hipSuccess;
            const long long tD = plane_thread_cpu_ns();
            if (coarseReady) {
                /* ONE download for the frame: the planes' centroid ranges lie in job order inside the frame's part of d_vout, so the
                 * span from the first plane's range to the end of the last one's centroids is fetched whole (the gaps - a plane's
                 * unused tail - cost PCIe bytes, a copy per plane cost ten stream operations per frame, and the rate of those is what
                 * saturates first with hundreds of frames in flight).  A plane the device handed back sends its gathered cloud. */
                size_t at = 0;
                const size_t first = nP ? (size_t)jobs[0].x : 0;
                size_t spanEnd = first;
                for (int i = 0; i < nP; i++) if (vc[i] > 0) spanEnd = std::max(spanEnd, (size_t)jobs[i].x + (size_t)vc[i]);
                const size_t span = spanEnd - first;
                size_t extra = 0;
                for (int i = 0; i < nP; i++) if (vc[i] < 0) extra += (size_t)jobs[i].y;
                if (e == hipSuccess && l->coarseCap < span + extra) {
                    if (l->h_coarse) (void)hipHostFree(l->h_coarse);
                    l->h_coarse = nullptr; l->coarseCap = 0;
                    const size_t capPts = std::max<size_t>(span + extra + (span + extra) / 2, 1 << 15);
                    e = hipHostMalloc((void**)&l->h_coarse, capPts * 12, hipHostMallocDefault);
                    if (e == hipSuccess) l->coarseCap = capPts;
                }
                if (e == hipSuccess && span > 0) e = hipMemcpyAsync(l->h_coarse, A->d_vout + 3 * first, span * 12, hipMemcpyDeviceToHost, l->stream);
                at = span;
                for (int i = 0; i < nP && e == hipSuccess; i++) {
                    if (vc[i] >= 0) { cptr[i] = l->h_coarse + 3 * ((size_t)jobs[i].x - first); continue; }
                    cptr[i] = l->h_coarse + 3 * at;
                    const size_t cnt = (size_t)jobs[i].y;
                    if (cnt > 0) e = hipMemcpyAsync(l->h_coarse + 3 * at, A->d_vpts + 3 * (size_t)jobs[i].x, cnt * 12, hipMemcpyDeviceToHost, l->stream);
                    at += cnt;
                }
            } else if (!postReady && off[nP] > 0 && (J.post || J.memberIdx))
                e = hipMemcpyAsync(idx.data(), slot + A->offMemberIdx, sizeof(int) * (size_t)off[nP], hipMemcpyDeviceToHost, l->stream);
            if (e == hipSuccess && J.seg) e = hipMemcpyAsync(J.seg + f * px, slot + A->offSeg, px, hipMemcpyDeviceToHost, l->stream);
            if (e == hipSuccess && (nCoarse > 0 || (!coarseReady && !postReady && (J.post || J.memberIdx)) || J.seg)) e = drfe_pool_sync(l->stream, ev);
            if (e != hipSuccess) { l->err = std::string("planes batch: results of a frame: ") + hipGetErrorString(e); rc = DRFE_ERR_HIP; }
            const long long tR = plane_thread_cpu_ns();
            g_planeCpuNs[1] += tR - tD;
            if (rc == DRFE_OK && coarseReady && handedBack) {
                J.voxFallbacks += handedBack;
                static const bool traceBack = std::getenv("DRFE_TRACE_PLANES") != nullptr;
                if (traceBack)
                    for (int i = 0; i < nP; i++)
                        if (vc[i] < 0) std::fprintf(stderr, "frame %d plane %d (%d points): voxel grid handed back by the device, code %d (-1 grid beyond int32, -2 heap-sort branch, <= -9 loop bound)\n", f, i, jobs[i].y, vc[i]);
                redo.resize((size_t)handedBack);
                int r = 0;
                for (int i = 0; i < nP; i++)
                    if (vc[i] < 0) {
                        int cnt = 0;
                        redo[r].resize(3 * (size_t)jobs[i].y);
                        (void)drfe_plane_voxel_grid(cptr[i], jobs[i].y, 0.05f, redo[r].data(), jobs[i].y, &cnt);
                        cptr[i] = redo[r].data(); vcl[i] = cnt; r++;
                    }
            }
            const long long tP = plane_thread_cpu_ns();
            g_planeCpuNs[2] += tP - tR;
            if (rc == DRFE_OK && postReady) viaCoarse = true;
            if (rc == DRFE_OK && coarseReady) {
                struct AccP { long long t; ~AccP() { g_planeCpuNs[3] += plane_thread_cpu_ns() - t; } } accP{tP};
                rc = drfe_ahc_post_from_coarse(&l->err, pl, nP, cptr.data(), vcl.data(), J.maxPointDist, J.distThreshold, J.post + (size_t)f * J.cap, nullptr, voff.data(),
                                               0, &J.nAccepted[f], J.planeNum ? &J.planeNum[f] : nullptr);
                viaCoarse = true;
            }
        }
        if (rc == DRFE_OK && !J.post) {
            /* drfe_planes_ahc_batch: the extractor's outputs only */
            const int np = J.nPlanes[f];
            if (J.memberOffsets) std::memcpy(J.memberOffsets + (size_t)f * (J.cap + 1), off.data(), sizeof(int32_t) * (size_t)(np + 1));
            if (J.memberIdx && off[np] > 0) std::memcpy(J.memberIdx + f * px, idx.data(), sizeof(int32_t) * (size_t)off[np]);
        } else if (rc == DRFE_OK && !viaCoarse)
            rc = drfe_ahc_post_core(&l->err, d, J.w, J.h, J.stride, J.K4, J.depthFactor, pl, J.nPlanes[f], off.data(), idx.data(), J.maxPointDist,
                                    J.distThreshold, J.post + (size_t)f * J.cap, nullptr, voff.data(), 0, &J.nAccepted[f],
                                    J.planeNum ? &J.planeNum[f] : nullptr, nullptr);
        if (rc != DRFE_OK) { ahc_batch_fail(J, rc, l->err); (void)hipEventDestroy(ev); return; }
        {
            std::lock_guard<std::mutex> lk(J.mu);
            if (--J.pending == 0) J.cv.notify_all();
        }
    }
}

static int planes_ahc_post_batch_device(drfe_ctx* c, std::vector<PlaneLane>* pool, int T, const uint16_t* depth, size_t frame_stride, int w, int h,
                                        size_t stride, int nframes, const float* K4, float depth_factor, float max_point_dist, double dist_threshold,
                                        drfe_plane* planes, int cap, int* n_planes, uint8_t* seg, drfe_plane_post* post, int* n_accepted, int* plane_num,
                                        int32_t* member_offsets, int32_t* member_idx)
{
    DrfeRange range("drfe:planes batch (upload, block fits, clustering, flood fill, clouds, voxel grids; gates + refit on the pool)");
    int rc = ensure_arena(c, w, h, nframes, K4, depth_factor, max_point_dist);
    if (rc != DRFE_OK) return rc;
    AhcArena* A = static_cast<AhcArena*>(c->ahcArena);
    AhcBatchJob J;
    J.c = c; J.A = A; J.pool = pool; J.depth = depth; J.frameStride = frame_stride; J.stride = stride; J.w = w; J.h = h; J.nframes = nframes; J.cap = cap;
    J.K4 = K4; J.depthFactor = depth_factor; J.maxPointDist = max_point_dist; J.distThreshold = dist_threshold;
    J.planes = planes; J.nPlanes = n_planes; J.seg = seg; J.post = post; J.nAccepted = n_accepted; J.planeNum = plane_num;
    J.memberOffsets = member_offsets; J.memberIdx = member_idx;
    /* chunks (= low-priority streams = hardware queues) of this call: the runtime has four queues per priority, and the line and the
     * plane batch of a front-end step run side by side - two each (DRFE_BATCH_CHUNKS overrides) */
    static const int nch = [] { const char* e = std::getenv("DRFE_BATCH_CHUNKS"); const int v = e ? std::atoi(e) : 1; return v < 1 ? 1 : v > 16 ? 16 : v; }();
    J.chunk = std::max(1, std::min(nframes, std::max(16, (nframes + nch - 1) / nch)));
    J.nChunks = (nframes + J.chunk - 1) / J.chunk;
    J.pending = nframes;
    J.voxDevice = post != nullptr && c->planesDeviceVoxel != 0 && !std::getenv("DRFE_VOXEL_HOST");
    J.refitDevice = J.voxDevice && c->planesDeviceRefit != 0 && !std::getenv("DRFE_REFIT_HOST");
    J.chunkStream.resize(J.nChunks); J.chunkDone.resize(J.nChunks); J.chunkState.assign(J.nChunks, 0);
    int prLow = 0, prHigh = 0;
    HIPCHK(c, hipDeviceGetStreamPriorityRange(&prLow, &prHigh));
    const size_t px = (size_t)w * h;
    const int NB = A->P.NB;
    int launchRc = DRFE_OK;
    static const bool traceStages = std::getenv("DRFE_TRACE_PLANES") != nullptr;
    hipEvent_t stageEv[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};       /* chunk 0: start, depth up, blocks, extractor, voxel grids */
    if (traceStages) for (hipEvent_t& e : stageEv) HIPCHK(c, hipEventCreate(&e));
    /* drfe_long_kernel_clock: start | upload | k_ahc_blocks | k_ahc_cluster | k_ahc_refine | k_ahc_labels_* | k_voxel_grid | k_plane_refit (chunk 0) */
    hipEvent_t clk[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (c->longClock) for (hipEvent_t& e : clk) HIPCHK(c, hipEventCreate(&e));
    const auto tCall = std::chrono::steady_clock::now();
    for (int ch = 0; ch < J.nChunks; ch++) {
        /* low priority, like the line path's growth: the runtime keeps separate hardware queues per priority, so the pools'
         * short kernels never queue behind a chunk that runs for a hundred milliseconds */
        HIPCHK(c, drfe_long_kernel_stream(&J.chunkStream[ch], 1));
        HIPCHK(c, hipEventCreateWithFlags(&J.chunkDone[ch], hipEventDisableTiming | hipEventBlockingSync));
    }
    for (int ch = 0; ch < J.nChunks && launchRc == DRFE_OK; ch++) {
        const int f0 = ch * J.chunk, nf = std::min(J.chunk, nframes - f0);
        hipStream_t st = J.chunkStream[ch];
        /* pinned, dense caller frames are uploaded where they lie; anything else through the arena's pinned mirror */
        const bool direct = stride == (size_t)w && frame_stride == px && drfe_host_is_pinned(depth + px * f0, px * 2 * nf);
        if (!direct)
            for (int f = f0; f < f0 + nf; f++)
                for (int y = 0; y < h; y++) std::memcpy(A->h_depth + px * f + (size_t)y * w, depth + (size_t)f * frame_stride + (size_t)y * stride, (size_t)w * 2);
        const bool tr = traceStages && ch == 0;
        const bool ck = ch == 0 && clk[0];
        if (tr) (void)hipEventRecord(stageEv[0], st);
        if (ck) (void)hipEventRecord(clk[0], st);
        hipError_t e = hipMemcpyAsync(A->d_depth + px * f0, direct ? depth + px * f0 : A->h_depth + px * f0, px * 2 * nf, hipMemcpyHostToDevice, st);
        if (tr) (void)hipEventRecord(stageEv[1], st);
        if (ck) (void)hipEventRecord(clk[1], st);
        if (e == hipSuccess) e = drfe_launch_ahc_blocks(A->d_depth + px * f0, px, (size_t)w, w, h, K4, depth_factor, nf, A->d_blocks + (size_t)NB * f0, st);
        if (tr) (void)hipEventRecord(stageEv[2], st);
        if (ck) (void)hipEventRecord(clk[2], st);
        if (e == hipSuccess) e = drfe_launch_ahc_frames(A->d_frames + f0, nf, A->P, st, ck ? clk + 3 : nullptr);
        if (tr) (void)hipEventRecord(stageEv[3], st);
        if (e == hipSuccess && J.voxDevice) {
            const size_t pc = (size_t)A->P.planeCap;
            e = drfe_launch_voxel_grid(A->d_vpts, A->d_jobs + pc * f0, (int)(pc * nf), A->d_vlist + pc * f0 + 2 * (size_t)ch, A->d_vrecs, A->d_vtmp, A->d_vposL, A->d_vposR, A->d_vout, A->d_vcounts + pc * f0,
                                       0.05f, st);
            if (ck) (void)hipEventRecord(clk[6], st);
            if (e == hipSuccess && J.refitDevice)
                e = drfe_launch_plane_refit(A->d_frames + f0, A->d_jobs + pc * f0, A->d_vcounts + pc * f0, A->d_vout, A->d_mtState, (int)(pc * nf), (int)pc, max_point_dist,
                                            dist_threshold, std::log(1.0 - 0.99), A->d_post + pc * f0, A->d_postStatus + pc * f0, st);
            if (tr) (void)hipEventRecord(stageEv[4], st);
            if (ck) (void)hipEventRecord(clk[7], st);
        }
        /* no download behind the kernels: a copy queued on a DMA ring waits there for its kernel and holds up the copies of every
         * other stream behind it (the line path's, CAPE's); the worker that sees the event fetches the chunk's small results */
        if (e == hipSuccess) e = hipEventRecord(J.chunkDone[ch], st);
        if (e != hipSuccess) { c->err = std::string("planes_ahc_post_batch: device path: ") + hipGetErrorString(e); launchRc = DRFE_ERR_HIP; }
        else J.chunkState[ch] = 1;
    }
    const auto tLaunched = std::chrono::steady_clock::now();
    if (launchRc == DRFE_OK) {
        std::vector<std::thread> th;
        for (int k = 0; k < T; k++) th.emplace_back([&J, pool, k]() { DrfePoolCpuScope cpu(1); ahc_batch_worker(J, &(*pool)[k]); });
        for (std::thread& t : th) t.join();
    }
    if (traceStages) {
        float ms[4] = {0, 0, 0, 0};
        for (int k = 0; k < 4; k++) if (k < 3 || J.voxDevice) (void)hipEventElapsedTime(&ms[k], stageEv[k], stageEv[k + 1]);
        const double tl = std::chrono::duration<double, std::milli>(tLaunched - tCall).count(), ta = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tCall).count();
        std::fprintf(stderr, "drfe_planes_ahc_post_batch stages (chunk 0): staging + enqueue %.1f ms of host time; depth upload %.1f ms, k_ahc_blocks %.1f, k_ahc_cluster + k_ahc_refine %.1f, k_voxel_grid %.1f; call %.1f ms\n",
                     tl, ms[0], ms[1], ms[2], ms[3], ta);
        for (hipEvent_t& e : stageEv) (void)hipEventDestroy(e);
    }
    if (clk[0]) {
        if (launchRc == DRFE_OK) {
            (void)hipStreamSynchronize(J.chunkStream[0]);
            const int last = J.voxDevice ? (J.refitDevice ? 7 : 6) : 5;
            for (int k = 0; k < 7; k++) { float ms = 0; if (k < last) (void)hipEventElapsedTime(&ms, clk[k], clk[k + 1]); c->longMs[8 + k] = ms; }
        }
        for (hipEvent_t& e : clk) (void)hipEventDestroy(e);
    }
    for (int ch = 0; ch < J.nChunks; ch++) {
        (void)hipStreamSynchronize(J.chunkStream[ch]);
        (void)hipStreamDestroy(J.chunkStream[ch]);
        (void)hipEventDestroy(J.chunkDone[ch]);
    }
    if (std::getenv("DRFE_AHC_PROFILE")) {      /* AHC_PROFILE builds of k_ahc_cluster + k_ahc_refine: phase times of frame 0 */
        uint32_t t[8] = {0};
        (void)hipMemcpy(t, A->h_frames[0].rf, sizeof(t), hipMemcpyDeviceToHost);
        std::fprintf(stderr, "k_ahc_cluster + k_ahc_refine, frame 0: initGraph %.2f ms, ahCluster %.2f, membership + seeds %.2f, floodFill %.2f (%d queue entries), re-merge %.2f, labels + member lists %.2f; %d nodes\n",
                     t[0] / 1e5, t[1] / 1e5, t[2] / 1e5, t[3] / 1e5, A->h_out[2], t[4] / 1e5, t[5] / 1e5, A->h_out[3]);
        int ff[4] = {0, 0, 0, 0};
        (void)hipMemcpy(ff, A->h_frames[0].handoff + 4 + 2 * 128 + 4, sizeof(ff), hipMemcpyDeviceToHost);
        std::fprintf(stderr, "  flood fill: %d steps, %d with two visits of one pixel (chain depth summed: %d), %d live visits\n", ff[0], ff[1], ff[2], ff[3]);
        typedef int (*prof_fn)(unsigned long long*);
        if (prof_fn fn = (prof_fn)dlsym(RTLD_DEFAULT, "drfe_debug_ahc_cluster_profile")) {
            unsigned long long v[8] = {0};
            if (fn(v) == 0 && v[0])
                std::fprintf(stderr, "  ahCluster, frame 0 of every call since the last print: %llu pops (%llu merges); ms per call: heap pop %.2f, node + list loads %.2f, trial merges %.2f, merges %.2f, extract / disconnect %.2f; longest neighbour list of any frame %llu\n",
                             v[0], v[6], v[1] / 1e5, v[2] / 1e5, v[3] / 1e5, v[4] / 1e5, v[5] / 1e5, v[7]);
        }
    }
    if (std::getenv("DRFE_TRACE_PLANES"))
        std::fprintf(stderr, "drfe_planes_ahc_post_batch workers, CPU ms per frame: chunk fetch %.3f, frame download + wait %.3f, grids redone on the host %.3f, gates + refit %.3f\n",
                     g_planeCpuNs[0].exchange(0) / 1e6 / nframes, g_planeCpuNs[1].exchange(0) / 1e6 / nframes, g_planeCpuNs[2].exchange(0) / 1e6 / nframes, g_planeCpuNs[3].exchange(0) / 1e6 / nframes);
    if (std::getenv("DRFE_TRACE_PLANES"))
        std::fprintf(stderr, "drfe_planes_ahc_post_batch (device extractor): %d frames, %d chunks, %d frames redone on the host; voxel grids on the %s (%d planes' grids redone on the host); gates + refit on the %s (%d frames sent to the host's refit)\n",
                     nframes, J.nChunks, J.fallbacks.load(), J.voxDevice ? "device" : "host", J.voxFallbacks.load(), J.refitDevice ? "device" : "host", J.refitFallbacks.load());
    if (launchRc != DRFE_OK) return launchRc;
    if (J.firstRc != DRFE_OK) { c->err = J.firstErr; return J.firstRc; }
    c->ahcStats[0] += nframes; c->ahcStats[1] += J.fallbacks.load();
    if (J.voxDevice) {
        long long grids = 0;
        for (int f = 0; f < nframes; f++) if (A->h_out[4 * (size_t)f + 1] == 0 && n_planes[f] <= cap) grids += n_planes[f];
        c->ahcStats[2] += grids; c->ahcStats[3] += J.voxFallbacks.load();
    }
    if (J.refitDevice) { c->ahcRefitStats[0] += nframes - J.fallbacks.load(); c->ahcRefitStats[1] += J.refitFallbacks.load(); }
    return DRFE_OK;
}

extern "C" {

/* out4[0] = frames through the device extractor (drfe_planes_ahc_batch / drfe_planes_ahc_post_batch) since drfe_create, out4[1] = of
 * those, redone on the host (a capacity of the device path, an uncertified cosine), out4[2] = plane voxel grids run on the device,
 * out4[3] = of those, redone on the host */
int drfe_planes_ahc_stats(drfe_ctx* c, long long* out4)
{
    if (!c || !out4) return DRFE_ERR_INVALID;
    for (int i = 0; i < 4; i++) out4[i] = c->ahcStats[i];
    return DRFE_OK;
}

/* pcl::VoxelGrid of every plane in drfe_planes_ahc_post_batch.  1 (default): on the device (voxel_kernels.hip) behind the device
 * extractor - one launch for all planes of a chunk of frames, the workers fetch centroids; the host-extractor mode keeps the host
 * grid.  2: on the device in the host-extractor mode too (a launch per frame from each worker).  0: always on the pool's host
 * threads.  Results are identical. */
int drfe_planes_configure(drfe_ctx* c, int device_voxel_grid)
{
    if (!c || device_voxel_grid < 0 || device_voxel_grid > 2) { if (c) c->err = "planes_configure: invalid argument"; return DRFE_ERR_INVALID; }
    c->planesDeviceVoxel = device_voxel_grid;
    return DRFE_OK;
}

/* 1 (default): drfe_planes_ahc_post_batch runs PEAC's extractor (graph, clustering, flood fill, re-merge, labels) on the device,
 * one wavefront per frame (ahc_frame_kernels.hip); 0: on the pool's host threads.  Results are identical. */
/* 1 (default): gates + Frame::MaxPointDistanceFromPlane (RANSAC + least-squares refit) of drfe_planes_ahc_post_batch on the device
 * behind the device voxel grids (refit_kernels.hip, one wavefront per plane); 0: on the pool's host threads.  Results are identical. */
int drfe_planes_configure_refit(drfe_ctx* c, int on_device)
{
    if (!c || on_device < 0 || on_device > 1) { if (c) c->err = "planes_configure_refit: invalid argument"; return DRFE_ERR_INVALID; }
    c->planesDeviceRefit = on_device;
    return DRFE_OK;
}

/* out2[0] = frames whose gates + refit ran on the device since drfe_create, out2[1] = of those, sent to the host's refit (a decision
 * that could not be certified, a voxel grid that came back) */
int drfe_planes_refit_stats(drfe_ctx* c, long long* out2)
{
    if (!c || !out2) return DRFE_ERR_INVALID;
    out2[0] = c->ahcRefitStats[0]; out2[1] = c->ahcRefitStats[1];
    return DRFE_OK;
}

int drfe_planes_configure_extractor(drfe_ctx* c, int on_device)
{
    if (!c || on_device < 0 || on_device > 1) { if (c) c->err = "planes_configure_extractor: invalid argument"; return DRFE_ERR_INVALID; }
    c->planesDeviceAhc = on_device;
    return DRFE_OK;
}

int drfe_planes_ahc_post_batch(drfe_ctx* c, const uint16_t* depth, size_t frame_stride, int w, int h, size_t stride, int nframes,
                               const float* K4, float depth_factor, float max_point_dist, double dist_threshold, drfe_plane* planes,
                               int cap, int* n_planes, uint8_t* seg, drfe_plane_post* post, int* n_accepted, int* plane_num, int n_threads)
{
    if (!c || !depth || !K4 || !planes || !n_planes || !post || !n_accepted || nframes < 0 || cap < 1 || frame_stride < stride * (size_t)h) {
        if (c) c->err = "planes_ahc_post_batch: invalid argument";
        return DRFE_ERR_INVALID;
    }
    if (nframes == 0) return DRFE_OK;
    /* default: 1.25 threads per CPU - a lane sleeps in stream synchronisations for about a fifth of a frame's time */
    int T = n_threads > 0 ? n_threads : std::max(1, drfe_default_host_threads() * 5 / 4);
    T = std::max(1, std::min(T, nframes));
    HIPCHK(c, hipSetDevice(c->device));
    auto* pool = static_cast<std::vector<PlaneLane>*>(c->planeLanes);
    if (!pool) { pool = new std::vector<PlaneLane>(); c->planeLanes = pool; }
    while ((int)pool->size() < T) {
        PlaneLane l;
        l.device = c->device;
        HIPCHK(c, hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        pool->push_back(l);
    }
    /* the extractor itself on the device (drfe_planes_configure_extractor; frames of up to 12 800 init blocks and 2^21 pixels:
     * 1280 x 960 included - the kernels' queue capacity and pixel index) */
    if ((c->planesDeviceAhc) && drfe_ahc_device_fits(w, h) && !std::getenv("DRFE_AHC_HOST"))
        return planes_ahc_post_batch_device(c, pool, T, depth, frame_stride, w, h, stride, nframes, K4, depth_factor, max_point_dist, dist_threshold, planes,
                                            cap, n_planes, seg, post, n_accepted, plane_num, nullptr, nullptr);
    std::vector<int> rcs(T, DRFE_OK);
    std::vector<std::thread> th;
    th.reserve(T);
    std::atomic<int> next(0);
    const size_t px = (size_t)w * h;
    for (int k = 0; k < T; k++)
        th.emplace_back([&, k]() {
            PlaneLane* l = &(*pool)[k];
            std::vector<int32_t> off((size_t)cap + 1), idx(px), voff((size_t)cap + 1);
            /* pcl::VoxelGrid of the frame's planes on this thread (default) or on the device (drfe_planes_configure(ctx, 1):
             * voxel_kernels.hip - identical results; it frees ~2.5 ms of CPU per frame, but the thread then sleeps ~3-5 ms per
             * frame waiting for its ten 256-thread sorts, and beside the line path's long-running kernels the hardware queues
             * serialise: measured 360-590 frames/s for the whole front-end against 1040-1080 with the host grid) */
            const bool voxHost = c->planesDeviceVoxel != 2;
            if (!voxHost && !l->vox) {
                l->vox = drfe_voxel_device_create(l->device, &l->err);
                if (!l->vox) { rcs[k] = DRFE_ERR_HIP; return; }
            }
            for (int f = next.fetch_add(1); f < nframes; f = next.fetch_add(1)) {
                const uint16_t* d = depth + (size_t)f * frame_stride;
                int rc = planes_ahc_core(l, d, w, h, stride, K4, depth_factor, planes + (size_t)f * cap, cap, &n_planes[f],
                                         seg ? seg + f * px : nullptr, off.data(), idx.data());
                if (rc == DRFE_OK)
                    rc = drfe_ahc_post_core(&l->err, d, w, h, stride, K4, depth_factor, planes + (size_t)f * cap, n_planes[f], off.data(),
                                            idx.data(), max_point_dist, dist_threshold, post + (size_t)f * cap, nullptr, voff.data(), 0,
                                            &n_accepted[f], plane_num ? &plane_num[f] : nullptr, voxHost ? nullptr : l->vox);
                if (rc != DRFE_OK) { rcs[k] = rc; return; }
            }
        });
    for (std::thread& t : th) t.join();
    for (int k = 0; k < T; k++)
        if (rcs[k] != DRFE_OK) { c->err = (*pool)[k].err; return rcs[k]; }
    return DRFE_OK;
}

} /* extern "C" */
