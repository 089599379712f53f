/* planes_cape.cpp — CAPE depth-plane extraction behind drfe_planes_cape (include/drfe.h).
 *
 * Split of PlaneDetection_CAPE::runPlaneDetection + CAPE::process (reference
 * src/PlaneExtractor.cpp:111-191, src/CAPE/CAPE.cpp:47-457):
 *   device  k_cape_cells: cloud, cell-major gather, per-cell validity / jump tests, the nine float32
 *           sums, PlaneSeg::fitPlane (f64 Eigen 3x3 solve) and the cell's merge tolerance
 *   host    20x20 normal histogram + seeding, recursive 4-neighbour cell growing, plane merging,
 *           3x3 erode/dilate on the <= 64x48 cell mask, per-pixel refinement in boundary cells and the
 *           label image — all on a grid of at most 3072 cells
 * Reference bugs are kept and the two canonicalisations of oracle/cape_oracle.cpp apply here as well
 * (sequential float32 cell sums; fit fields of non-planar cells read as 0).
 */
#include "drfe_internal.h"
#include "planes_internal.h"
#include <thread>
#include <atomic>
#include "ahc_math.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>

#define HIPCHK(c, call)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (c)->err = std::string(#call) + ": " + hipGetErrorString(e__);                      \
            return DRFE_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

namespace {

struct PSeg {
    double acc[9];        /* x y z xx yy zz xy xz yz */
    int nr_pts;
    double mean[3], normal[3], d;
    float MSE, score;
    void expand(const double* a, int n) { for (int k = 0; k < 9; k++) acc[k] += a[k]; nr_pts += n; }
    void fit()
    {
        const double n = nr_pts;
        mean[0] = acc[0] / n; mean[1] = acc[1] / n; mean[2] = acc[2] / n;
        double ev[3], Q[9];
        ahc_eig3(acc[3] - acc[0] * acc[0] / n, acc[6] - acc[0] * acc[1] / n, acc[7] - acc[0] * acc[2] / n,
                 acc[4] - acc[1] * acc[1] / n, acc[8] - acc[1] * acc[2] / n, acc[5] - acc[2] * acc[2] / n, ev, Q);
        d = -(Q[0] * mean[0] + Q[1] * mean[1] + Q[2] * mean[2]);
        if (d > 0) { normal[0] = Q[0]; normal[1] = Q[1]; normal[2] = Q[2]; }
        else { normal[0] = -Q[0]; normal[1] = -Q[1]; normal[2] = -Q[2]; d = -d; }
        MSE = (float)(ev[0] / n);
        score = (float)(ev[1] / ev[0]);
    }
};

struct Grow {
    int W, H;
    const std::vector<char>* in;
    std::vector<char>* out;
    const CapeCellRec* cells;
    float minCos;
    void run(int x, int y, const double* n1, double d)
    {
        const int idx = x + W * y;
        if (!(*in)[idx] || (*out)[idx]) return;
        const CapeCellRec& c = cells[idx];
        const double v = n1[0] * c.mean[0] + n1[1] * c.mean[1] + n1[2] * c.mean[2] + d;
        if (n1[0] * c.normal[0] + n1[1] * c.normal[1] + n1[2] * c.normal[2] < minCos || v * v > c.tol) return;
        (*out)[idx] = 1;
        if (x > 0) run(x - 1, y, c.normal, c.d);
        if (x < W - 1) run(x + 1, y, c.normal, c.d);
        if (y > 0) run(x, y - 1, c.normal, c.d);
        if (y < H - 1) run(x, y + 1, c.normal, c.d);
    }
};

} // namespace

hipError_t drfe_launch_cape_cells(const float* d_depth, size_t rowStride, int w, int h, const float K4[4], int patch,
                                  float sinCos, float maxMergeDist, CapeCellRec* d_out, hipStream_t s);

/* one CAPE lane of drfe_planes_cape_batch: what the core below touches of a context (same member names) */
struct CapeLane {
    std::string err;
    int device = 0;
    hipStream_t stream = nullptr;
    void* cape = nullptr;
    hipEvent_t pollEv = nullptr;      /* drfe_pool_sync: the batch's threads sleep between polls instead of spinning */
};

/* the single-frame entry spins (latency is the point there), a batch lane sleeps */
static inline hipError_t cape_sync(drfe_ctx* c) { return hipStreamSynchronize(c->stream); }
static inline hipError_t cape_sync(CapeLane* l) { return l->pollEv ? drfe_pool_sync(l->stream, l->pollEv) : hipStreamSynchronize(l->stream); }

template <class Ctx>
static int planes_cape_core(Ctx* c, const float* depth_m, int w, int h, size_t stride, const float* K4, int patch,
                            float cos_angle_max, float max_merge_dist, drfe_cape_plane* planes, int cap, int* n_planes,
                            uint8_t* seg, double* cells16, float* cells_mst, int32_t* cells_pn)
{
    if (!c || !depth_m || !K4 || !n_planes || !seg) return DRFE_ERR_INVALID;
    *n_planes = 0;
    if (patch < 4 || patch > 64 || w % patch || h % patch || stride < (size_t)w) {
        c->err = "planes_cape: width/height must be multiples of PATCH_SIZE (4..64)";
        return DRFE_ERR_INVALID;
    }
    HIPCHK(c, hipSetDevice(c->device));
    const int nh = w / patch, nv = h / patch, ncell = nh * nv, npx = w * h;
    /* device stage: buffers owned by the context (no allocation per frame) */
    CapeScratch* cs = static_cast<CapeScratch*>(c->cape);
    if (!cs) { cs = new (std::nothrow) CapeScratch(); if (!cs) return DRFE_ERR_INVALID; std::memset(cs, 0, sizeof(*cs)); c->cape = cs; }
    if (cs->depthCap < (size_t)npx) {
        if (cs->d_depth) (void)hipFree(cs->d_depth);
        if (cs->d_seg) (void)hipFree(cs->d_seg);
        if (cs->h_depth) (void)hipHostFree(cs->h_depth);
        if (cs->h_seg) (void)hipHostFree(cs->h_seg);
        cs->d_depth = nullptr; cs->d_seg = nullptr; cs->h_depth = nullptr; cs->h_seg = nullptr; cs->depthCap = cs->segCap = 0;
        HIPCHK(c, hipMalloc((void**)&cs->d_depth, (size_t)npx * sizeof(float)));
        HIPCHK(c, hipMalloc((void**)&cs->d_seg, (size_t)npx));
        HIPCHK(c, hipHostMalloc((void**)&cs->h_depth, (size_t)npx * sizeof(float), hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void**)&cs->h_seg, (size_t)npx, hipHostMallocDefault));
        cs->depthCap = cs->segCap = (size_t)npx;
    }
    if (cs->cellCap < (size_t)ncell) {
        if (cs->d_cells) (void)hipFree(cs->d_cells);
        if (cs->h_cells) (void)hipHostFree(cs->h_cells);
        cs->d_cells = nullptr; cs->h_cells = nullptr; cs->cellCap = 0;
        HIPCHK(c, hipMalloc((void**)&cs->d_cells, (size_t)ncell * sizeof(CapeCellRec)));
        HIPCHK(c, hipHostMalloc((void**)&cs->h_cells, (size_t)ncell * sizeof(CapeCellRec), hipHostMallocDefault));
        cs->cellCap = (size_t)ncell;
    }
    float* d_depth = cs->d_depth;
    CapeCellRec* d_cells = cs->d_cells;
    const float sinCos = (float)std::sqrt(1 - (double)cos_angle_max * (double)cos_angle_max);
    /* through the pinned mirrors: a copy call on pageable memory stages or pins inside the call, on this thread */
    for (int y = 0; y < h; y++) std::memcpy(cs->h_depth + (size_t)y * w, depth_m + (size_t)y * stride, (size_t)w * 4);
    const CapeCellRec* cells = cs->h_cells;
    hipError_t e = hipMemcpyAsync(d_depth, cs->h_depth, (size_t)npx * 4, hipMemcpyHostToDevice, c->stream);
    /* downloads are issued only when their kernel has finished: queued behind it, a copy sits in a DMA ring until then and holds
     * up the copies of every other stream behind it */
    if (e == hipSuccess) e = drfe_launch_cape_cells(d_depth, (size_t)w, w, h, K4, patch, sinCos, max_merge_dist, d_cells, c->stream);
    if (e == hipSuccess) e = cape_sync(c);
    if (e == hipSuccess) e = hipMemcpyAsync(cs->h_cells, d_cells, (size_t)ncell * sizeof(CapeCellRec), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = cape_sync(c);
    if (e != hipSuccess) { c->err = std::string("planes_cape: ") + hipGetErrorString(e); return DRFE_ERR_HIP; }
    for (int i = 0; i < ncell && cells16; i++) {
        const CapeCellRec& r = cells[i];
        for (int k = 0; k < 9; k++) cells16[16 * i + k] = r.acc[k];
        for (int k = 0; k < 3; k++) { cells16[16 * i + 9 + k] = r.mean[k]; cells16[16 * i + 12 + k] = r.normal[k]; }
        cells16[16 * i + 15] = r.d;
        if (cells_mst) { cells_mst[3 * i] = r.MSE; cells_mst[3 * i + 1] = r.score; cells_mst[3 * i + 2] = r.tol; }
        if (cells_pn) { cells_pn[2 * i] = r.planar; cells_pn[2 * i + 1] = r.nr_pts; }
    }

    /* histogram of cell normals in spherical coordinates, src/CAPE/CAPE.cpp:81-104, Histogram.cpp */
    const int NB = 20;
    std::vector<int> Hh(NB * NB, 0), Bq(ncell, -1);
    std::vector<char> unassigned(ncell, 0), act(ncell, 0);
    int remaining = 0;
    for (int i = 0; i < ncell; i++) {
        if (!cells[i].planar) continue;
        const double nx = cells[i].normal[0], ny = cells[i].normal[1], nz = cells[i].normal[2];
        const double npn = std::sqrt(nx * nx + ny * ny);
        const double p0 = std::acos(-nz), p1 = std::atan2(nx / npn, ny / npn);
        const int Xq = (int)((NB - 1) * (p0 - 0.0) / (3.14 - 0.0));
        int Yq = 0;
        if (Xq > 0) Yq = (int)((NB - 1) * (p1 - (-3.14)) / (3.14 - (-3.14)));
        Bq[i] = Yq * NB + Xq;
        Hh[Bq[i]]++;
        unassigned[i] = 1;
        remaining++;
    }
    std::vector<PSeg> segs;
    std::vector<int> gridMap(ncell, 0);
    while (remaining > 0) {
        int best = -1, mx = 0;
        for (int b = 0; b < NB * NB; b++)
            if (Hh[b] > mx) { best = b; mx = Hh[b]; }
        std::vector<int> cand;
        if (mx > 0)
            for (int i = 0; i < ncell; i++)
                if (Bq[i] == best) cand.push_back(i);
        if (cand.size() < 5) break;
        int seed = cand[0];
        float minMSE = (float)INT_MAX;
        for (size_t i = 0; i < cand.size(); i++)
            if (cells[cand[i]].MSE < minMSE) { seed = cand[i]; minMSE = cells[i].MSE; }   /* sic: Grid[i], CAPE.cpp:130 */
        PSeg ps;
        std::memcpy(ps.acc, cells[seed].acc, sizeof(ps.acc));
        ps.nr_pts = cells[seed].nr_pts;
        std::memcpy(ps.mean, cells[seed].mean, 24);
        std::memcpy(ps.normal, cells[seed].normal, 24);
        ps.d = cells[seed].d; ps.MSE = cells[seed].MSE; ps.score = cells[seed].score;
        std::fill(act.begin(), act.end(), 0);
        Grow g{nh, nv, &unassigned, &act, cells, cos_angle_max};
        const double sn[3] = {ps.normal[0], ps.normal[1], ps.normal[2]};
        g.run(seed % nh, seed / nh, sn, ps.d);
        int nact = 0;
        for (int i = 0; i < ncell; i++)
            if (act[i]) {
                ps.expand(cells[i].acc, cells[i].nr_pts);
                nact++;
                Hh[Bq[i]]--; Bq[i] = -1;
                unassigned[i] = 0;
                remaining--;
            }
        if (nact < 4) continue;
        ps.fit();
        if (ps.score > 100) {
            segs.push_back(ps);
            const int nr = (int)segs.size();
            for (int i = 0; i < ncell; i++)
                if (act[i]) gridMap[i] = nr;
        }
    }

    /* plane merging, CAPE.cpp:208-245 */
    const int np = (int)segs.size();
    std::vector<char> assoc((size_t)np * np, 0);
    for (int r = 0; r < nv - 1; r++)
        for (int cc = 0; cc < nh - 1; cc++) {
            const int px = gridMap[r * nh + cc];
            if (px <= 0) continue;
            const int right = gridMap[r * nh + cc + 1], below = gridMap[(r + 1) * nh + cc];
            if (right > 0 && px != right) assoc[(size_t)(px - 1) * np + right - 1] = 1;
            if (below > 0 && px != below) assoc[(size_t)(px - 1) * np + below - 1] = 1;
        }
    for (int r = 0; r < np; r++)
        for (int k = r + 1; k < np; k++) assoc[(size_t)r * np + k] = assoc[(size_t)r * np + k] || assoc[(size_t)k * np + r];
    std::vector<int> label(np);
    for (int i = 0; i < np; i++) label[i] = i;
    for (int r = 0; r < np; r++) {
        const int pid = label[r];
        bool expanded = false;
        for (int k = r + 1; k < np; k++) {
            if (!assoc[(size_t)r * np + k]) continue;
            const PSeg &P = segs[pid], &Q = segs[k];
            const double cosA = P.normal[0] * Q.normal[0] + P.normal[1] * Q.normal[1] + P.normal[2] * Q.normal[2];
            const double dv = segs[r].normal[0] * Q.mean[0] + P.normal[1] * Q.mean[1] + P.normal[2] * Q.mean[2] + P.d;   /* sic */
            if (cosA > cos_angle_max && dv * dv < max_merge_dist) {
                segs[pid].expand(Q.acc, Q.nr_pts);
                label[k] = pid;
                expanded = true;
            } else assoc[(size_t)r * np + k] = 0;
        }
        if (expanded) segs[pid].fit();
    }

    /* boundary refinement + label image, CAPE.cpp:247-319, 395-431: the cell masks (<= 64 x 48 cells) are erode / dilate work
     * on the host; the per-pixel assignment runs on the device (k_cape_refine) from the depth image already there */
    std::vector<uint8_t> mask(ncell), er(ncell), di(ncell), gridEroded(ncell, 0), boundary;
    std::vector<CapeRefinePlane> rp;
    int nFinal = 0;
    for (int i = 0; i < np; i++) {
        if (i != label[i]) continue;
        std::fill(mask.begin(), mask.end(), 0);
        for (int j = i; j < np; j++)
            if (label[j] == label[i])
                for (int k = 0; k < ncell; k++)
                    if (gridMap[k] == j + 1) mask[k] = 1;
        int mx = 0;
        for (int r = 0; r < nv; r++)
            for (int k = 0; k < nh; k++) {
                int e2 = mask[r * nh + k];
                if (k > 0) e2 = std::min<int>(e2, mask[r * nh + k - 1]);
                if (k < nh - 1) e2 = std::min<int>(e2, mask[r * nh + k + 1]);
                if (r > 0) e2 = std::min<int>(e2, mask[(r - 1) * nh + k]);
                if (r < nv - 1) e2 = std::min<int>(e2, mask[(r + 1) * nh + k]);
                er[r * nh + k] = (uint8_t)e2;
                mx = std::max(mx, e2);
                int dl = 0;
                for (int dr = -1; dr <= 1; dr++)
                    for (int dc = -1; dc <= 1; dc++) {
                        const int rr = r + dr, kk = k + dc;
                        if (rr >= 0 && rr < nv && kk >= 0 && kk < nh) dl = std::max<int>(dl, mask[rr * nh + kk]);
                    }
                di[r * nh + k] = (uint8_t)dl;
            }
        if (mx == 0) continue;
        if (nFinal >= cap || nFinal >= 254) { c->err = "planes_cape: plane buffer too small"; *n_planes = nFinal; return DRFE_ERR_CAPACITY; }
        if (planes) {
            drfe_cape_plane& o = planes[nFinal];
            std::memcpy(o.normal, segs[i].normal, 24);
            std::memcpy(o.mean, segs[i].mean, 24);
            o.d = segs[i].d; o.mse = segs[i].MSE; o.score = segs[i].score; o.n_points = segs[i].nr_pts;
        }
        nFinal++;
        const uint8_t nr = (uint8_t)nFinal;
        rp.push_back(CapeRefinePlane{(float)segs[i].normal[0], (float)segs[i].normal[1], (float)segs[i].normal[2], (float)segs[i].d,
                                     9 * segs[i].MSE});
        boundary.resize((size_t)nFinal * ncell);
        uint8_t* bnd = boundary.data() + (size_t)(nFinal - 1) * ncell;
        for (int cell = 0; cell < ncell; cell++) {
            if (er[cell] > 0) gridEroded[cell] = nr;
            bnd[cell] = ((int)di[cell] - (int)er[cell] > 0) ? 1 : 0;
        }
    }
    *n_planes = nFinal;
    const size_t tabBytes = ((rp.size() * sizeof(CapeRefinePlane) + 15) & ~(size_t)15) + (size_t)ncell + boundary.size() + 16;
    if (cs->tabCap < tabBytes) {
        if (cs->d_tab) (void)hipFree(cs->d_tab);
        cs->d_tab = nullptr; cs->tabCap = 0;
        HIPCHK(c, hipMalloc((void**)&cs->d_tab, tabBytes * 2));
        cs->tabCap = tabBytes * 2;
    }
    const size_t offGrid = (rp.size() * sizeof(CapeRefinePlane) + 15) & ~(size_t)15, offBnd = offGrid + (size_t)ncell;
    if (!rp.empty()) HIPCHK(c, hipMemcpyAsync(cs->d_tab, rp.data(), rp.size() * sizeof(CapeRefinePlane), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(cs->d_tab + offGrid, gridEroded.data(), (size_t)ncell, hipMemcpyHostToDevice, c->stream));
    if (!boundary.empty()) HIPCHK(c, hipMemcpyAsync(cs->d_tab + offBnd, boundary.data(), boundary.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, drfe_launch_cape_refine(cs->d_depth, (size_t)w, w, h, K4, patch, reinterpret_cast<const CapeRefinePlane*>(cs->d_tab), nFinal,
                                      cs->d_tab + offGrid, cs->d_tab + offBnd, cs->d_seg, c->stream));
    HIPCHK(c, cape_sync(c));
    HIPCHK(c, hipMemcpyAsync(cs->h_seg, cs->d_seg, (size_t)npx, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, cape_sync(c));
    std::memcpy(seg, cs->h_seg, (size_t)npx);
    return DRFE_OK;
}

static void cape_scratch_free(void*& p)
{
    CapeScratch* cs = static_cast<CapeScratch*>(p);
    if (!cs) return;
    void* ptrs[] = {cs->d_depth, cs->d_cells, cs->d_seg, cs->d_tab};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    void* hptrs[] = {cs->h_depth, cs->h_cells, cs->h_seg};
    for (void* q : hptrs) if (q) (void)hipHostFree(q);
    delete cs;
    p = nullptr;
}

/* device arena of drfe_planes_cape_batch's device path (grow-only, kept by the context) */
struct CapeBatchArena {
    int frames = 0, w = 0, h = 0, ncell = 0;
    bool withSeg = false;
    float* d_depth = nullptr; CapeCellRec* d_cells = nullptr; drfe_cape_plane* d_planes = nullptr; uint8_t* d_tabs = nullptr;
    CapeFrameOut* d_out = nullptr; uint8_t* d_seg = nullptr;
    float* h_stage[2] = {nullptr, nullptr}; hipEvent_t stageFree[2] = {nullptr, nullptr};      /* pinned upload staging, CAPE_STAGE_FRAMES frames each */
    drfe_cape_plane* h_planes = nullptr; CapeFrameOut* h_out = nullptr; uint8_t* h_seg = nullptr;
    hipStream_t stream = nullptr, copyStream = nullptr;
    hipEvent_t kernelsDone = nullptr;
    bool ready = false;       /* every allocation below succeeded: a half-built arena (an allocation failed) is never reused */
};
#define CAPE_STAGE_FRAMES 32

static void cape_batch_free(void*& p)
{
    CapeBatchArena* A = static_cast<CapeBatchArena*>(p);
    if (!A) return;
    void* dp[] = {A->d_depth, A->d_cells, A->d_planes, A->d_tabs, A->d_out, A->d_seg};
    for (void* q : dp) if (q) (void)hipFree(q);
    void* hp[] = {A->h_stage[0], A->h_stage[1], A->h_planes, A->h_out, A->h_seg};
    for (void* q : hp) if (q) (void)hipHostFree(q);
    for (hipEvent_t e : A->stageFree) if (e) (void)hipEventDestroy(e);
    if (A->stream) (void)hipStreamDestroy(A->stream);
    if (A->copyStream) (void)hipStreamDestroy(A->copyStream);
    if (A->kernelsDone) (void)hipEventDestroy(A->kernelsDone);
    delete A;
    p = nullptr;
}

void drfe_cape_lanes_free(drfe_ctx* c)
{
    cape_batch_free(c->capeBatch);
    auto* pool = static_cast<std::vector<CapeLane>*>(c->capeLanes);
    if (!pool) return;
    for (CapeLane& l : *pool) { cape_scratch_free(l.cape); if (l.stream) (void)hipStreamDestroy(l.stream); if (l.pollEv) (void)hipEventDestroy(l.pollEv); }
    delete pool;
    c->capeLanes = nullptr;
}

/* drfe_planes_cape_batch with CAPE::process on the device: k_cape_cells (batch) -> k_cape_frame (one wavefront per frame:
 * histogram seeding, cell growing, segment fits, merging, masks) -> k_cape_refine (batch), frames uploaded through two pinned
 * staging buffers while the previous ones compute.  hostFrames: the frames whose status word asks for the host path. */
static int planes_cape_batch_device(drfe_ctx* c, const float* depth_m, size_t frame_stride, int w, int h, size_t stride, int nframes,
                                    const float* K4, int patch, float cos_angle_max, float max_merge_dist, drfe_cape_plane* planes, int cap,
                                    int* n_planes, uint8_t* seg, std::vector<int>& hostFrames)
{
    const int nh = w / patch, nv = h / patch, ncell = nh * nv;
    const size_t npx = (size_t)w * h, tabStride = drfe_cape_tab_bytes(ncell);
    CapeBatchArena* A = static_cast<CapeBatchArena*>(c->capeBatch);
    if (!A || !A->ready || A->frames < nframes || A->w != w || A->h != h || A->ncell != ncell || (seg && !A->withSeg)) {
        cape_batch_free(c->capeBatch);
        A = new (std::nothrow) CapeBatchArena();
        if (!A) return DRFE_ERR_INVALID;
        c->capeBatch = A;
        A->frames = nframes; A->w = w; A->h = h; A->ncell = ncell; A->withSeg = seg != nullptr;
        const size_t F = (size_t)nframes;
        HIPCHK(c, hipMalloc((void**)&A->d_depth, F * npx * sizeof(float)));
        HIPCHK(c, hipMalloc((void**)&A->d_cells, F * ncell * sizeof(CapeCellRec)));
        HIPCHK(c, hipMalloc((void**)&A->d_planes, F * CAPE_DEV_MAXP * sizeof(drfe_cape_plane)));
        HIPCHK(c, hipMalloc((void**)&A->d_tabs, F * tabStride));
        HIPCHK(c, hipMalloc((void**)&A->d_out, F * sizeof(CapeFrameOut)));
        HIPCHK(c, hipMalloc((void**)&A->d_seg, F * npx));
        for (int k = 0; k < 2; k++) {
            HIPCHK(c, hipHostMalloc((void**)&A->h_stage[k], (size_t)CAPE_STAGE_FRAMES * npx * sizeof(float), hipHostMallocDefault));
            HIPCHK(c, hipEventCreateWithFlags(&A->stageFree[k], hipEventDisableTiming));
        }
        HIPCHK(c, hipHostMalloc((void**)&A->h_planes, F * CAPE_DEV_MAXP * sizeof(drfe_cape_plane), hipHostMallocDefault));
        HIPCHK(c, hipHostMalloc((void**)&A->h_out, F * sizeof(CapeFrameOut), hipHostMallocDefault));
        if (seg) HIPCHK(c, hipHostMalloc((void**)&A->h_seg, F * npx, hipHostMallocDefault));
        HIPCHK(c, hipStreamCreateWithFlags(&A->stream, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&A->copyStream, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&A->kernelsDone, hipEventDisableTiming));
        A->ready = true;
    }
    hipStream_t st = A->stream, cs = A->copyStream;
    DrfeRange range("drfe:cape batch (upload, cell fits, frame stage, refinement)");
    DrfePoolCpuScope cpu(2);                      /* accounted to the CAPE pool: the calling thread is its only worker here */
    const float sinCos = (float)std::sqrt(1 - (double)cos_angle_max * (double)cos_angle_max);
    /* Copies never sit behind kernels in a queue: a copy whose stream predecessor is a kernel occupies a DMA ring until that kernel
     * has run - in the full front-end, where these kernels wait for CUs behind 0.1 s wavefronts, that held up every other path's
     * transfers (this path's 512 frames took 500 ms instead of 25, the line path's upload 317 ms instead of 15).  Uploads go to
     * their own stream and depend on nothing but their staging buffer; the kernels wait for them through an event; the results
     * are fetched after the last kernel has finished. */
    const bool direct = stride == (size_t)w && frame_stride == npx && drfe_host_is_pinned(depth_m, (size_t)nframes * npx * sizeof(float));
    int chunkNo = 0;
    for (int f0 = 0; f0 < nframes; f0 += CAPE_STAGE_FRAMES, chunkNo++) {
        const int nf = std::min(CAPE_STAGE_FRAMES, nframes - f0), b = chunkNo & 1;
        if (direct) {
            /* the caller's frames are pinned and dense: the DMA engine reads them where they lie */
            HIPCHK(c, hipMemcpyAsync(A->d_depth + (size_t)f0 * npx, depth_m + (size_t)f0 * frame_stride, (size_t)nf * npx * sizeof(float), hipMemcpyHostToDevice, cs));
        } else {
            if (chunkNo >= 2) HIPCHK(c, drfe_event_wait_sleeping(A->stageFree[b]));      /* the upload that last read this staging buffer is done */
            float* hs = A->h_stage[b];
            for (int k = 0; k < nf; k++) {
                const float* src = depth_m + (size_t)(f0 + k) * frame_stride;
                if (stride == (size_t)w) std::memcpy(hs + (size_t)k * npx, src, npx * sizeof(float));
                else for (int y = 0; y < h; y++) std::memcpy(hs + (size_t)k * npx + (size_t)y * w, src + (size_t)y * stride, (size_t)w * 4);
            }
            HIPCHK(c, hipMemcpyAsync(A->d_depth + (size_t)f0 * npx, hs, (size_t)nf * npx * sizeof(float), hipMemcpyHostToDevice, cs));
        }
        HIPCHK(c, hipEventRecord(A->stageFree[b], cs));
        HIPCHK(c, hipStreamWaitEvent(st, A->stageFree[b], 0));
        HIPCHK(c, drfe_launch_cape_cells_batch(A->d_depth + (size_t)f0 * npx, npx, (size_t)w, w, h, K4, patch, sinCos, max_merge_dist, nf,
                                               A->d_cells + (size_t)f0 * ncell, st));
        HIPCHK(c, drfe_launch_cape_frames(A->d_cells + (size_t)f0 * ncell, nh, nv, cos_angle_max, max_merge_dist, nf,
                                          A->d_planes + (size_t)f0 * CAPE_DEV_MAXP, A->d_tabs + (size_t)f0 * tabStride, tabStride, A->d_out + f0, st));
        HIPCHK(c, drfe_launch_cape_refine_batch(A->d_depth + (size_t)f0 * npx, npx, (size_t)w, w, h, K4, patch, A->d_tabs + (size_t)f0 * tabStride,
                                                tabStride, A->d_out + f0, nf, A->d_seg + (size_t)f0 * npx, st));
    }
    HIPCHK(c, drfe_pool_sync(st, A->kernelsDone));                                   /* sleeps between polls */
    HIPCHK(c, hipMemcpyAsync(A->h_out, A->d_out, (size_t)nframes * sizeof(CapeFrameOut), hipMemcpyDeviceToHost, cs));
    HIPCHK(c, hipMemcpyAsync(A->h_planes, A->d_planes, (size_t)nframes * CAPE_DEV_MAXP * sizeof(drfe_cape_plane), hipMemcpyDeviceToHost, cs));
    if (seg) HIPCHK(c, hipMemcpyAsync(A->h_seg, A->d_seg, (size_t)nframes * npx, hipMemcpyDeviceToHost, cs));
    HIPCHK(c, drfe_pool_sync(cs, A->kernelsDone));
    for (int f = 0; f < nframes; f++) {
        if (A->h_out[f].status != 0) { hostFrames.push_back(f); continue; }
        const int np = A->h_out[f].nPlanes;
        n_planes[f] = np;
        if (np > cap) { c->err = "planes_cape: plane buffer too small"; return DRFE_ERR_CAPACITY; }
        std::memcpy(planes + (size_t)f * cap, A->h_planes + (size_t)f * CAPE_DEV_MAXP, (size_t)np * sizeof(drfe_cape_plane));
        if (seg) std::memcpy(seg + (size_t)f * npx, A->h_seg + (size_t)f * npx, npx);
    }
    c->capeStats[0] += nframes; c->capeStats[1] += (long long)hostFrames.size();
    return DRFE_OK;
}

extern "C" {

/* 1 (default): drfe_planes_cape_batch runs CAPE::process on the device (cape_frame_kernels.hip), 0: on the pool's host threads
 * between the device's cell fits and per-pixel refinement (rounds 1-3).  Results are identical. */
int drfe_planes_configure_cape(drfe_ctx* c, int on_device)
{
    if (!c || on_device < 0 || on_device > 1) { if (c) c->err = "planes_configure_cape: invalid argument"; return DRFE_ERR_INVALID; }
    c->planesDeviceCape = on_device;
    return DRFE_OK;
}

/* out2[0] = frames through drfe_planes_cape_batch's device path since drfe_create, out2[1] = of those, finished by the host */
int drfe_planes_cape_stats(drfe_ctx* c, long long* out2)
{
    if (!c || !out2) return DRFE_ERR_INVALID;
    out2[0] = c->capeStats[0]; out2[1] = c->capeStats[1];
    return DRFE_OK;
}

int drfe_planes_cape(drfe_ctx* c, const float* depth_m, int w, int h, size_t stride, const float* K4, int patch,
                     float cos_angle_max, float max_merge_dist, drfe_cape_plane* planes, int cap, int* n_planes,
                     uint8_t* seg, double* cells16, float* cells_mst, int32_t* cells_pn)
{
    return planes_cape_core(c, depth_m, w, h, stride, K4, patch, cos_angle_max, max_merge_dist, planes, cap, n_planes, seg, cells16,
                            cells_mst, cells_pn);
}

/* PlaneDetection_CAPE for nframes depth images (metres, frame_stride floats apart): planes[f * cap ..], n_planes[f],
 * seg[f * w * h ..] (may be NULL: the label images are then not returned).  Default: the device path above; the pool of
 * n_threads host threads (one device lane - stream + scratch - each) takes the frames the device hands back, or all of them
 * in the host mode.  Results are identical to nframes calls of drfe_planes_cape.  n_threads <= 0: up to 4. */
int drfe_planes_cape_batch(drfe_ctx* c, const float* depth_m, size_t frame_stride, int w, int h, size_t stride, int nframes,
                           const float* K4, int patch, float cos_angle_max, float max_merge_dist, drfe_cape_plane* planes, int cap,
                           int* n_planes, uint8_t* seg, int n_threads)
{
    if (!c || !depth_m || !K4 || !planes || !n_planes || nframes < 0 || cap < 1 || frame_stride < stride * (size_t)h) {
        if (c) c->err = "planes_cape_batch: invalid argument";
        return DRFE_ERR_INVALID;
    }
    if (nframes == 0) return DRFE_OK;
    const int T = std::max(1, std::min(n_threads > 0 ? n_threads : 4, nframes));
    HIPCHK(c, hipSetDevice(c->device));
    /* the whole extractor on the device (drfe_planes_configure_cape; cape_frame_kernels.hip): the calling thread uploads, launches
     * and copies the results out; a frame the device could not finish (status word) goes through the host path below */
    std::vector<int> hostFrames;
    bool deviceDone = false;
    if (c->planesDeviceCape && nframes > 1 && !std::getenv("DRFE_CAPE_HOST") && patch >= 4 && patch <= 64 && w % patch == 0 && h % patch == 0 &&
        (w / patch) * (h / patch) <= CAPE_DEV_MAXCELLS && stride >= (size_t)w && cap >= 1) {
        const int rc = planes_cape_batch_device(c, depth_m, frame_stride, w, h, stride, nframes, K4, patch, cos_angle_max, max_merge_dist, planes, cap,
                                                n_planes, seg, hostFrames);
        if (rc != DRFE_OK) return rc;
        deviceDone = true;
        if (hostFrames.empty()) return DRFE_OK;
    }
    auto* pool = static_cast<std::vector<CapeLane>*>(c->capeLanes);
    if (!pool) { pool = new std::vector<CapeLane>(); c->capeLanes = pool; }
    while ((int)pool->size() < T) {
        CapeLane l;
        l.device = c->device;
        HIPCHK(c, hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&l.pollEv, hipEventDisableTiming));
        pool->push_back(l);
    }
    std::vector<int> rcs(T, DRFE_OK);
    std::vector<std::thread> th;
    std::atomic<int> next(0);
    const size_t px = (size_t)w * h;
    for (int k = 0; k < T; k++)
        th.emplace_back([&, k]() {
            DrfePoolCpuScope cpu(2);
            CapeLane* l = &(*pool)[k];
            std::vector<uint8_t> segTmp(seg ? 0 : px);
            const int todo = deviceDone ? (int)hostFrames.size() : nframes;
            for (int q = next.fetch_add(1); q < todo; q = next.fetch_add(1)) {
                const int f = deviceDone ? hostFrames[q] : q;
                const int rc = planes_cape_core(l, depth_m + (size_t)f * frame_stride, w, h, stride, K4, patch, cos_angle_max, max_merge_dist,
                                                planes + (size_t)f * cap, cap, &n_planes[f], seg ? seg + (size_t)f * px : segTmp.data(),
                                                nullptr, nullptr, nullptr);
                if (rc != DRFE_OK) { rcs[k] = rc; return; }
            }
        });
    for (std::thread& t : th) t.join();
    for (int k = 0; k < T; k++)
        if (rcs[k] != DRFE_OK) { c->err = (*pool)[k].err; return rcs[k]; }
    return DRFE_OK;
}

} /* extern "C" */
