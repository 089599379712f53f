/* cape_frame_kernels.hip - CAPE::process between the per-cell plane fits and the per-pixel boundary refinement, on the device:
 * the histogram of cell normals and its seeding, the 4-neighbour cell growing, the segment fits, plane merging and the
 * erode / dilate masks (reference src/CAPE/CAPE.cpp:81-293 with Histogram.cpp and PlaneSeg.cpp; restated for the host in
 * planes_cape.cpp and for the oracle in oracle/cape_oracle.cpp).  One wavefront per frame, frames of a batch side by side.
 *
 * What is order-defined is kept in order, what is not runs across the lanes:
 *   - a cell joins a growing region iff some chain of accepted steps leads to it from the seed (the reference's recursion marks
 *     nothing on a rejected visit), and a step's test reads only its two cells: the region is a reachability closure, grown
 *     here breadth first, 16 frontier cells x 4 neighbours per step;
 *   - a segment's nine sums are added in ascending cell index starting from the seed's own sums (the seed is counted twice, as
 *     in the reference): nine lanes, one sum each, walk the activated cells in index order;
 *   - the seed choice carries the reference's index slip (CAPE.cpp:130: the running minimum is refreshed from Grid[i], i the
 *     position in the candidate list) and the merge loop its r / plane_id mix (:238-240): both replayed literally by one lane;
 *   - the histogram bins come from acos / atan2 of the cell normals - the host libm's in the reference.  The device computes
 *     them itself and certifies the quantisation (the scaled angle further than 1e-9 from an integer); a frame with a bin too
 *     close to call is finished by the host (planes_cape.cpp). */
#include <hip/hip_runtime.h>

#include <climits>

#include "drfe_internal.h"
#include "planes_internal.h"
#include "ahc_math.h"
#include "../../include/drfe.h"

namespace {

struct DevSeg {
    double acc[9];
    double mean[3], normal[3], d;
    float MSE, score;
    int nr_pts, pad;
};

__device__ void seg_fit(DevSeg& s)
{
    const double n = s.nr_pts;
    s.mean[0] = s.acc[0] / n; s.mean[1] = s.acc[1] / n; s.mean[2] = s.acc[2] / n;
    double ev[3], Q[9];
    ahc_eig3(s.acc[3] - s.acc[0] * s.acc[0] / n, s.acc[6] - s.acc[0] * s.acc[1] / n, s.acc[7] - s.acc[0] * s.acc[2] / n,
             s.acc[4] - s.acc[1] * s.acc[1] / n, s.acc[8] - s.acc[1] * s.acc[2] / n, s.acc[5] - s.acc[2] * s.acc[2] / n, ev, Q);
    double d = -(Q[0] * s.mean[0] + Q[1] * s.mean[1] + Q[2] * s.mean[2]);
    if (d > 0) { s.normal[0] = Q[0]; s.normal[1] = Q[1]; s.normal[2] = Q[2]; }
    else { s.normal[0] = -Q[0]; s.normal[1] = -Q[1]; s.normal[2] = -Q[2]; d = -d; }
    s.d = d;
    s.MSE = (float)(ev[0] / n);
    s.score = (float)(ev[1] / ev[0]);
}

struct CapeShared {
    int16_t Bq[CAPE_DEV_MAXCELLS];
    uint8_t unassigned[CAPE_DEV_MAXCELLS], act[CAPE_DEV_MAXCELLS], gridMap[CAPE_DEV_MAXCELLS];
    uint8_t mask[CAPE_DEV_MAXCELLS], er[CAPE_DEV_MAXCELLS], di[CAPE_DEV_MAXCELLS];
    uint16_t frontier[2][CAPE_DEV_MAXCELLS];
    int Hh[400];
    DevSeg segs[CAPE_DEV_MAXP];
    uint8_t assoc[CAPE_DEV_MAXP * CAPE_DEV_MAXP];
    int label[CAPE_DEV_MAXP];
    int remaining, nFront[2], seed, np, unc, nact, anyEroded;
};

} // namespace

__global__ __launch_bounds__(64) void k_cape_frame(const CapeCellRec* __restrict__ cellsBase, int nh, int nv, float cosAngleMax, float maxMergeDist,
                                                   drfe_cape_plane* __restrict__ planesBase, uint8_t* __restrict__ tabsBase, size_t tabStride,
                                                   CapeFrameOut* __restrict__ outBase)
{
    __shared__ CapeShared S;
    const int f = blockIdx.x, lane = threadIdx.x, ncell = nh * nv;
    const CapeCellRec* cells = cellsBase + (size_t)f * ncell;
    drfe_cape_plane* planes = planesBase + (size_t)f * CAPE_DEV_MAXP;
    uint8_t* tab = tabsBase + tabStride * f;
    CapeRefinePlane* rp = reinterpret_cast<CapeRefinePlane*>(tab);
    uint8_t* gridEroded = tab + CAPE_DEV_MAXP * sizeof(CapeRefinePlane);
    uint8_t* boundary = gridEroded + ncell;
    const int NB = 20;
    for (int b = lane; b < NB * NB; b += 64) S.Hh[b] = 0;
    if (lane == 0) { S.remaining = 0; S.np = 0; S.unc = 0; }
    __syncthreads();
    /* histogram of the cell normals in spherical coordinates (CAPE.cpp:81-104, Histogram.cpp) */
    for (int i = lane; i < ncell; i += 64) {
        S.Bq[i] = -1; S.unassigned[i] = 0; S.gridMap[i] = 0; gridEroded[i] = 0;
        if (!cells[i].planar) continue;
        const double nx = cells[i].normal[0], ny = cells[i].normal[1], nz = cells[i].normal[2];
        const double npn = sqrt(nx * nx + ny * ny);
        const double p0 = acos(-nz), p1 = atan2(nx / npn, ny / npn);
        const double vx = (NB - 1) * (p0 - 0.0) / (3.14 - 0.0);
        const int Xq = (int)vx;
        int Yq = 0;
        bool unc = !(fabs(vx - rint(vx)) > 1e-9);                  /* also catches NaN */
        if (Xq > 0) {
            const double vy = (NB - 1) * (p1 - (-3.14)) / (3.14 - (-3.14));
            Yq = (int)vy;
            if (!(fabs(vy - rint(vy)) > 1e-9)) unc = true;
        } else if (!(vx < 1.0 - 1e-9)) unc = true;
        const int bq = Yq * NB + Xq;
        if (unc || bq < 0 || bq >= NB * NB) { atomicOr(&S.unc, 1); continue; }
        S.Bq[i] = (int16_t)bq;
        atomicAdd(&S.Hh[bq], 1);
        S.unassigned[i] = 1;
        atomicAdd(&S.remaining, 1);
    }
    __syncthreads();
    int status = S.unc ? CAPE_STATUS_UNCERTAIN : 0;
    int guard = 0;
    while (status == 0 && S.remaining > 0) {
        if (++guard > 4 * CAPE_DEV_MAXCELLS) { status = CAPE_STATUS_CAPACITY; break; }
        /* the fullest bin, the first of equals */
        int mx = 0, best = -1;
        for (int b = lane; b < NB * NB; b += 64) { const int v = S.Hh[b]; if (v > mx) { mx = v; best = b; } }
        for (int o = 32; o > 0; o >>= 1) {
            const int omx = __shfl_xor(mx, o), ob = __shfl_xor(best, o);
            if (omx > mx || (omx == mx && omx > 0 && ob < best)) { mx = omx; best = ob; }
        }
        /* its cells in index order: fewer than five ends the seeding; the seed with the reference's slip: minMSE is refreshed from
         * Grid[position in the list], not from the candidate */
        int ncand = 0;
        for (int base = 0; base < ncell; base += 64) {
            const int i = base + lane;
            ncand += __popcll(__ballot(i < ncell && mx > 0 && S.Bq[i] == best));
        }
        if (ncand < 5) break;
        if (lane == 0) {
            int seed = -1, pos = 0;
            float minMSE = (float)INT_MAX;
            for (int i = 0; i < ncell; i++)
                if (S.Bq[i] == best) {
                    if (seed < 0) seed = i;
                    if (cells[i].MSE < minMSE) { seed = i; minMSE = cells[pos].MSE; }
                    pos++;
                }
            S.seed = seed;
            DevSeg& ps = S.segs[S.np < CAPE_DEV_MAXP ? S.np : CAPE_DEV_MAXP - 1];      /* the slot the next segment would take: scratch until accepted */
            const CapeCellRec& c = cells[seed];
            for (int k = 0; k < 9; k++) ps.acc[k] = c.acc[k];
            ps.nr_pts = c.nr_pts;
            for (int k = 0; k < 3; k++) { ps.mean[k] = c.mean[k]; ps.normal[k] = c.normal[k]; }
            ps.d = c.d; ps.MSE = c.MSE; ps.score = c.score;
            S.nFront[0] = 0; S.nFront[1] = 0; S.nact = 0;
        }
        for (int i = lane; i < ncell; i += 64) S.act[i] = 0;
        __syncthreads();
        const int seed = S.seed;
        /* the seed is visited with its own plane (RegionGrowing(seed, ..., Grid[seed]->normal, Grid[seed]->d)) */
        if (lane == 0) {
            const CapeCellRec& c = cells[seed];
            const double v = c.normal[0] * c.mean[0] + c.normal[1] * c.mean[1] + c.normal[2] * c.mean[2] + c.d;
            const double dot = c.normal[0] * c.normal[0] + c.normal[1] * c.normal[1] + c.normal[2] * c.normal[2];
            if (S.unassigned[seed] && !(dot < cosAngleMax || v * v > c.tol)) { S.act[seed] = 1; S.frontier[0][0] = (uint16_t)seed; S.nFront[0] = 1; }
        }
        __syncthreads();
        int cur = 0;
        while (S.nFront[cur] > 0) {
            const int nf = S.nFront[cur];
            if (lane == 0) S.nFront[cur ^ 1] = 0;
            __syncthreads();
            for (int base = 0; base < nf; base += 16) {
                const int e = base + (lane >> 2), dir = lane & 3;
                if (e < nf) {
                    const int from = S.frontier[cur][e];
                    const int x = from % nh, y = from / nh;
                    int nx = x, ny = y;
                    if (dir == 0) nx = x - 1; else if (dir == 1) nx = x + 1; else if (dir == 2) ny = y - 1; else ny = y + 1;
                    if (nx >= 0 && nx < nh && ny >= 0 && ny < nv) {
                        const int idx = nx + nh * ny;
                        if (S.unassigned[idx] && !S.act[idx]) {
                            const CapeCellRec& p = cells[from];
                            const CapeCellRec& c = cells[idx];
                            const double v = p.normal[0] * c.mean[0] + p.normal[1] * c.mean[1] + p.normal[2] * c.mean[2] + p.d;
                            const double dot = p.normal[0] * c.normal[0] + p.normal[1] * c.normal[1] + p.normal[2] * c.normal[2];
                            if (!(dot < cosAngleMax || v * v > c.tol)) {
                                /* claimed once: two frontier cells may reach it in the same step */
                                const unsigned sh = 8u * (idx & 3);
                                const unsigned old = atomicOr(reinterpret_cast<unsigned*>(S.act) + (idx >> 2), 1u << sh);
                                if (!((old >> sh) & 1u)) { const int q = atomicAdd(&S.nFront[cur ^ 1], 1); S.frontier[cur ^ 1][q] = (uint16_t)idx; }
                            }
                        }
                    }
                }
            }
            __syncthreads();
            cur ^= 1;
        }
        /* the activated cells leave the histogram; their sums join the segment in index order (nine lanes, one sum each) */
        DevSeg& ps = S.segs[S.np < CAPE_DEV_MAXP ? S.np : CAPE_DEV_MAXP - 1];
        double a = lane < 9 ? ps.acc[lane] : 0.0;
        int npts = ps.nr_pts, nact = 0;
        for (int base = 0; base < ncell; base += 64) {
            const int i = base + lane;
            const bool on = i < ncell && S.act[i];
            unsigned long long m = __ballot(on);
            if (on) { atomicSub(&S.Hh[S.Bq[i]], 1); S.Bq[i] = -1; S.unassigned[i] = 0; }
            nact += __popcll(m);
            while (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const CapeCellRec& c = cells[base + l];
                if (lane < 9) a += c.acc[lane];
                npts += c.nr_pts;
            }
        }
        __syncthreads();
        if (lane < 9) ps.acc[lane] = a;
        if (lane == 0) { ps.nr_pts = npts; S.remaining -= nact; }
        __syncthreads();
        if (nact < 4) continue;
        if (lane == 0) seg_fit(ps);
        __syncthreads();
        if (ps.score > 100) {
            if (S.np >= CAPE_DEV_MAXP - 1) { status = CAPE_STATUS_CAPACITY; break; }
            const int nr = S.np + 1;
            for (int i = lane; i < ncell; i += 64) if (S.act[i]) S.gridMap[i] = (uint8_t)nr;
            __syncthreads();
            if (lane == 0) S.np = nr;
        }
        __syncthreads();
    }
    __syncthreads();
    const int np = S.np;
    int nFinal = 0;
    if (status == 0) {
        /* plane merging, CAPE.cpp:208-245: association of neighbouring segments, then the sequential merge with its index slip */
        for (int k = lane; k < np * np; k += 64) S.assoc[k] = 0;
        __syncthreads();
        for (int i = lane; i < ncell; i += 64) {
            const int r = i / nh, cc = i - r * nh;
            if (r >= nv - 1 || cc >= nh - 1) continue;
            const int px = S.gridMap[i];
            if (px <= 0) continue;
            const int right = S.gridMap[i + 1], below = S.gridMap[i + nh];
            if (right > 0 && px != right) S.assoc[(px - 1) * np + right - 1] = 1;
            if (below > 0 && px != below) S.assoc[(px - 1) * np + below - 1] = 1;
        }
        __syncthreads();
        if (lane == 0) {
            for (int r = 0; r < np; r++)
                for (int k = r + 1; k < np; k++) S.assoc[r * np + k] = S.assoc[r * np + k] || S.assoc[k * np + r];
            for (int i = 0; i < np; i++) S.label[i] = i;
            for (int r = 0; r < np; r++) {
                const int pid = S.label[r];
                bool expanded = false;
                for (int k = r + 1; k < np; k++) {
                    if (!S.assoc[r * np + k]) continue;
                    const DevSeg &P = S.segs[pid], &Q = S.segs[k];
                    const double cosA = P.normal[0] * Q.normal[0] + P.normal[1] * Q.normal[1] + P.normal[2] * Q.normal[2];
                    const double dv = S.segs[r].normal[0] * Q.mean[0] + P.normal[1] * Q.mean[1] + P.normal[2] * Q.mean[2] + P.d;   /* sic */
                    if (cosA > cosAngleMax && dv * dv < maxMergeDist) {
                        for (int q = 0; q < 9; q++) S.segs[pid].acc[q] += Q.acc[q];
                        S.segs[pid].nr_pts += Q.nr_pts;
                        S.label[k] = pid;
                        expanded = true;
                    } else S.assoc[r * np + k] = 0;
                }
                if (expanded) seg_fit(S.segs[pid]);
            }
        }
        __syncthreads();
        /* masks of the merged planes: eroded core, dilated minus eroded = boundary cells (CAPE.cpp:247-293) */
        for (int i = 0; i < np && status == 0; i++) {
            if (i != S.label[i]) continue;
            if (lane == 0) S.anyEroded = 0;
            for (int k = lane; k < ncell; k += 64) {
                const int gm = S.gridMap[k];
                S.mask[k] = (gm > i && S.label[gm - 1] == S.label[i]) ? 1 : 0;
            }
            __syncthreads();
            for (int k = lane; k < ncell; k += 64) {
                const int r = k / nh, c = k - r * nh;
                int e2 = S.mask[k];
                if (c > 0) e2 = min(e2, (int)S.mask[k - 1]);
                if (c < nh - 1) e2 = min(e2, (int)S.mask[k + 1]);
                if (r > 0) e2 = min(e2, (int)S.mask[k - nh]);
                if (r < nv - 1) e2 = min(e2, (int)S.mask[k + nh]);
                S.er[k] = (uint8_t)e2;
                if (e2) S.anyEroded = 1;
                int dl = 0;
                for (int dr = -1; dr <= 1; dr++)
                    for (int dc = -1; dc <= 1; dc++) {
                        const int rr = r + dr, kk = c + dc;
                        if (rr >= 0 && rr < nv && kk >= 0 && kk < nh) dl = max(dl, (int)S.mask[rr * nh + kk]);
                    }
                S.di[k] = (uint8_t)dl;
            }
            __syncthreads();
            if (!S.anyEroded) { __syncthreads(); continue; }
            if (nFinal >= CAPE_DEV_MAXP - 1) { status = CAPE_STATUS_CAPACITY; break; }
            if (lane == 0) {
                const DevSeg& sg = S.segs[i];
                drfe_cape_plane o;
                for (int k = 0; k < 3; k++) { o.normal[k] = sg.normal[k]; o.mean[k] = sg.mean[k]; }
                o.d = sg.d; o.mse = sg.MSE; o.score = sg.score; o.n_points = sg.nr_pts; o.pad = 0;
                planes[nFinal] = o;
                CapeRefinePlane p;
                p.nx = (float)sg.normal[0]; p.ny = (float)sg.normal[1]; p.nz = (float)sg.normal[2]; p.d = (float)sg.d; p.maxDist = 9 * sg.MSE;
                rp[nFinal] = p;
            }
            nFinal++;
            uint8_t* bnd = boundary + (size_t)(nFinal - 1) * ncell;
            for (int k = lane; k < ncell; k += 64) {
                if (S.er[k] > 0) gridEroded[k] = (uint8_t)nFinal;
                bnd[k] = ((int)S.di[k] - (int)S.er[k] > 0) ? 1 : 0;
            }
            __syncthreads();
        }
    }
    if (lane == 0) { CapeFrameOut o; o.nPlanes = status ? 0 : nFinal; o.status = status; o.pad0 = o.pad1 = 0; outBase[f] = o; }
}

hipError_t drfe_launch_cape_frames(const CapeCellRec* d_cells, int nh, int nv, float cosAngleMax, float maxMergeDist, int nframes,
                                   drfe_cape_plane* d_planes, uint8_t* d_tabs, size_t tabStride, CapeFrameOut* d_out, hipStream_t s)
{
    if (nframes <= 0) return hipSuccess;
    if (nh * nv > CAPE_DEV_MAXCELLS) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_cape_frame, dim3(nframes), dim3(64), 0, s, d_cells, nh, nv, cosAngleMax, maxMergeDist, d_planes, d_tabs, tabStride, d_out);
    return hipGetLastError();
}
