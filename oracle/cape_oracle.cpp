/* oracle/cape_oracle.cpp — TEST INFRASTRUCTURE (see oracle.h; parity unpinned vs Eigen/OpenCV).
 *
 * CPU restatement of the CAPE plane extractor as DR-SLAM wires it:
 *   PlaneDetection_CAPE::runPlaneDetection            reference src/PlaneExtractor.cpp:111-191
 *   CAPE::CAPE / process / getConnectedComponents / RegionGrowing   src/CAPE/CAPE.cpp
 *   PlaneSeg (cell fit, expandSegment, fitPlane)      src/CAPE/PlaneSeg.cpp
 *   Histogram                                         src/CAPE/Histogram.cpp
 * cylinder_detection is false in DR-SLAM (include/PlaneExtractor.h:113), so the cylinder branch is
 * not restated.  Reference bugs are preserved: min_MSE = Grid[i]->MSE with the loop
 * index, memset(distances, 100), mixed r/plane_id in the merge distance, mm-era constants.
 *
 * Canonicalised (the reference is host/UB dependent here):
 *  - Eigen's float32 `.sum()` reductions over a cell (PlaneSeg.cpp:78-86) depend on the SIMD width the
 *    reference was compiled for; the oracle accumulates sequentially in float32 in cell-local order
 *    and widens the result to double.
 *  - PlaneSeg::MSE/score/normal/mean/d of a non-planar cell are never written by the reference
 *    (uninitialised heap) yet `Grid[i]->MSE` is read for small i; the oracle treats them as 0.
 */
#include "cape_oracle.h"
#include "ahc_oracle.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>

namespace orc {

namespace {

struct Seg {   /* CAPE PlaneSeg */
    int nr_pts = 0, min_nr_pts = 0;
    double x_acc = 0, y_acc = 0, z_acc = 0, xx_acc = 0, yy_acc = 0, zz_acc = 0, xy_acc = 0, xz_acc = 0, yz_acc = 0;
    float score = 0, MSE = 0;
    bool planar = false;
    double mean[3] = {0, 0, 0}, normal[3] = {0, 0, 0}, d = 0;

    void expandSegment(const Seg* s)
    {
        x_acc += s->x_acc; y_acc += s->y_acc; z_acc += s->z_acc;
        xx_acc += s->xx_acc; yy_acc += s->yy_acc; zz_acc += s->zz_acc;
        xy_acc += s->xy_acc; xz_acc += s->xz_acc; yz_acc += s->yz_acc;
        nr_pts += s->nr_pts;
    }
    /* PlaneSeg::fitPlane, src/CAPE/PlaneSeg.cpp:110-142 */
    void fitPlane()
    {
        mean[0] = x_acc / nr_pts; mean[1] = y_acc / nr_pts; mean[2] = z_acc / nr_pts;
        double cov[3][3] = {{xx_acc - x_acc * x_acc / nr_pts, xy_acc - x_acc * y_acc / nr_pts, xz_acc - x_acc * z_acc / nr_pts},
                            {0, yy_acc - y_acc * y_acc / nr_pts, yz_acc - y_acc * z_acc / nr_pts},
                            {0, 0, zz_acc - z_acc * z_acc / nr_pts}};
        cov[1][0] = cov[0][1]; cov[2][0] = cov[0][2]; cov[2][1] = cov[1][2];
        double ev[3], V[3][3];
        eig33sym(cov, ev, V);
        const double v[3] = {V[0][0], V[1][0], V[2][0]};
        d = -(v[0] * mean[0] + v[1] * mean[1] + v[2] * mean[2]);
        if (d > 0) { normal[0] = v[0]; normal[1] = v[1]; normal[2] = v[2]; }
        else { normal[0] = -v[0]; normal[1] = -v[1]; normal[2] = -v[2]; d = -d; }
        MSE = (float)(ev[0] / nr_pts);
        score = (float)(ev[1] / ev[0]);
    }
};

/* PlaneSeg::PlaneSeg(cloud_array, cell_id, nr_pts_per_cell, cell_width), src/CAPE/PlaneSeg.cpp:8-94 */
static void initCell(Seg& s, const float* X, const float* Y, const float* Z, int n, int cell_width)
{
    s = Seg();
    s.min_nr_pts = n / 2;
    const int cell_height = n / cell_width;
    const double max_diff = 100;
    s.planar = true;
    int cnt = 0;
    for (int i = 0; i < n; i++) cnt += Z[i] > 0;
    s.nr_pts = cnt;
    if (s.nr_pts < s.min_nr_pts) { s.planar = false; return; }
    int jumps = 0;
    int i = cell_width * (cell_height / 2);
    int j = i + cell_width;
    float z = 0, z_last = std::max(Z[i], Z[i + 1]);
    i++;
    while (i < j) {
        z = Z[i];
        if (z > 0 && std::fabs(z - z_last) < max_diff) z_last = z;
        else if (z > 0) jumps++;
        i++;
    }
    if (jumps > 1) { s.planar = false; return; }
    i = cell_width / 2;
    j = n - i;
    z_last = std::max(Z[i], Z[i + cell_width]);
    i = i + cell_width;
    jumps = 0;
    while (i < j) {
        z = Z[i];
        if (z > 0 && std::fabs(z - z_last) < max_diff) z_last = z;
        else if (z > 0) jumps++;
        i += cell_width;
    }
    if (jumps > 1) { s.planar = false; return; }
    float sx = 0, sy = 0, sz = 0, sxx = 0, syy = 0, szz = 0, sxy = 0, sxz = 0, syz = 0;
    for (int k = 0; k < n; k++) {
        sx += X[k]; sy += Y[k]; sz += Z[k];
        sxx += X[k] * X[k]; syy += Y[k] * Y[k]; szz += Z[k] * Z[k];
        sxy += X[k] * Y[k]; sxz += X[k] * Z[k]; syz += Y[k] * Z[k];
    }
    s.x_acc = sx; s.y_acc = sy; s.z_acc = sz; s.xx_acc = sxx; s.yy_acc = syy; s.zz_acc = szz;
    s.xy_acc = sxy; s.xz_acc = sxz; s.yz_acc = syz;
    s.fitPlane();
    const double t = 0.000001425 * s.mean[2] * s.mean[2] + 10;   /* DEPTH_SIGMA_COEFF / MARGIN, Params.h:6-7 */
    if (s.MSE > t * t) s.planar = false;
}

struct Histogram {   /* src/CAPE/Histogram.cpp */
    std::vector<int> H, B;
    int nb, nbins, npts = 0;
    explicit Histogram(int n) : H(n * n, 0), nb(n), nbins(n * n) {}
    void init(const std::vector<double>& P0, const std::vector<double>& P1, const std::vector<bool>& flags)
    {
        npts = (int)P0.size();
        B.assign(npts, -1);
        const double min_X = 0, max_X = 3.14, min_Y = -3.14, max_Y = 3.14;
        for (int i = 0; i < npts; i++)
            if (flags[i]) {
                const int X_q = (int)((nb - 1) * (P0[i] - min_X) / (max_X - min_X));
                int Y_q = 0;
                if (X_q > 0) Y_q = (int)((nb - 1) * (P1[i] - min_Y) / (max_Y - min_Y));
                const int bin = Y_q * nb + X_q;
                B[i] = bin;
                H[bin]++;
            }
    }
    std::vector<int> mostFrequent() const
    {
        std::vector<int> ids;
        int best = -1, mx = 0;
        for (int i = 0; i < nbins; i++)
            if (H[i] > mx) { best = i; mx = H[i]; }
        if (mx > 0)
            for (int i = 0; i < npts; i++)
                if (B[i] == best) ids.push_back(i);
        return ids;
    }
    void remove(int id) { H[B[id]]--; B[id] = -1; }
};

struct Grower {
    int W, H;
    const std::vector<bool>* input;
    std::vector<bool>* output;
    const std::vector<Seg>* grid;
    const std::vector<float>* tols;
    float min_cos;
    /* CAPE::RegionGrowing, src/CAPE/CAPE.cpp:485-506 (recursive; visiting order matters) */
    void grow(int x, int y, const double* n1, double d)
    {
        const int index = x + W * y;
        if (!(*input)[index] || (*output)[index]) return;
        const Seg& c = (*grid)[index];
        const double* n2 = c.normal;
        const double* m = c.mean;
        const double v = n1[0] * m[0] + n1[1] * m[1] + n1[2] * m[2] + d;
        if (n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2] < min_cos || v * v > (*tols)[index]) return;
        (*output)[index] = true;
        if (x > 0) grow(x - 1, y, n2, c.d);
        if (x < W - 1) grow(x + 1, y, n2, c.d);
        if (y > 0) grow(x, y - 1, n2, c.d);
        if (y < H - 1) grow(x, y + 1, n2, c.d);
    }
};

} // namespace

CapeResult cape_run(const float* depth, int width, int height, const float K4[4], int patch, float cos_angle_max,
                    float max_merge_dist)
{
    CapeResult out;
    const int npx = width * height;
    const float fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
    const int nh = width / patch, nv = height / patch, ncell = nh * nv, ppc = patch * patch;
    /* cloud (float32 storage of float64 arithmetic), then cell-major re-layout (:111-152) */
    std::vector<float> X(npx, 0.f), Y(npx, 0.f), Z(npx, 0.f);
    for (int i = 0; i < height; i++)
        for (int j = 0; j < width; j++) {
            const double z = (double)depth[(size_t)i * width + j];
            const double x = ((double)j - cx) * z / fx;
            const double y = ((double)i - cy) * z / fy;
            /* pixels outside the cell grid (w or h not divisible by the patch) keep their row-major slot */
            const int cell_r = i / patch, local_r = i % patch, cell_c = j / patch, local_c = j % patch;
            const int id = (cell_r * nh + cell_c) * ppc + local_r * patch + local_c;
            if (id < npx) { X[id] = (float)x; Y[id] = (float)y; Z[id] = (float)z; }
        }
    out.seg.assign(npx, 0);
    std::vector<uint8_t> seg_stacked(npx, 0);
    float hugeF;
    { uint32_t bits = 0x64646464u; std::memcpy(&hugeF, &bits, 4); }   /* memset(distances_stacked, 100, ...) */
    std::vector<float> dist_stacked(npx, hugeF);

    /* planar cell fitting (:66-80) */
    std::vector<Seg> grid(ncell);
    std::vector<float> tols(ncell, 0.f);
    const float sin_cos = (float)std::sqrt(1 - std::pow((double)cos_angle_max, 2));
    for (int c = 0; c < ncell; c++) {
        const int off = c * ppc;
        initCell(grid[c], &X[off], &Y[off], &Z[off], ppc, patch);
        if (grid[c].planar) {
            const float dx = X[off + ppc - 1] - X[off], dy = Y[off + ppc - 1] - Y[off], dz = Z[off + ppc - 1] - Z[off];
            float sq = dx * dx;
            sq += dy * dy;
            sq += dz * dz;
            const float diameter = std::sqrt(sq);
            const float t = std::min(std::max(diameter * sin_cos, 20.0f), max_merge_dist);
            tols[c] = (float)((double)t * (double)t);
        }
    }
    out.cells.resize(ncell);
    for (int c = 0; c < ncell; c++) {
        CapeCell& o = out.cells[c];
        const Seg& s = grid[c];
        o.planar = s.planar; o.nr_pts = s.nr_pts;
        const double v[9] = {s.x_acc, s.y_acc, s.z_acc, s.xx_acc, s.yy_acc, s.zz_acc, s.xy_acc, s.xz_acc, s.yz_acc};
        std::copy(v, v + 9, o.sums);
        for (int k = 0; k < 3; k++) { o.mean[k] = s.mean[k]; o.normal[k] = s.normal[k]; }
        o.d = s.d; o.MSE = s.MSE; o.score = s.score; o.tol = tols[c];
    }

    /* histogram of normal directions (:81-104) */
    std::vector<double> C0(ncell, 0.0), C1(ncell, 0.0);
    std::vector<bool> planar_flags(ncell, false), unassigned(ncell, false), activation(ncell, false);
    int remaining = 0;
    for (int c = 0; c < ncell; c++)
        if (grid[c].planar) {
            const double nx = grid[c].normal[0], ny = grid[c].normal[1], nz = grid[c].normal[2];
            const double n_proj_norm = std::sqrt(nx * nx + ny * ny);
            C0[c] = std::acos(-nz);
            C1[c] = std::atan2(nx / n_proj_norm, ny / n_proj_norm);
            planar_flags[c] = true;
            remaining++;
        }
    Histogram Hs(20);
    Hs.init(C0, C1, planar_flags);
    std::vector<Seg> plane_segments;
    std::vector<int> grid_map(ncell, 0);
    for (int c = 0; c < ncell; c++) unassigned[c] = planar_flags[c];

    /* cell-wise region growing (:113-205) */
    while (remaining > 0) {
        const std::vector<int> cand = Hs.mostFrequent();
        if (cand.size() < 5) break;
        int seed_id = cand[0];
        float min_MSE = (float)INT_MAX;
        for (size_t i = 0; i < cand.size(); i++) {
            const int sc = cand[i];
            if (grid[sc].MSE < min_MSE) { seed_id = sc; min_MSE = grid[i].MSE; }   /* sic: Grid[i] */
        }
        Seg new_ps = grid[seed_id];
        const int y = seed_id / nh, x = seed_id % nh;
        std::fill(activation.begin(), activation.end(), false);
        Grower g{nh, nv, &unassigned, &activation, &grid, &tols, cos_angle_max};
        const double seed_n[3] = {new_ps.normal[0], new_ps.normal[1], new_ps.normal[2]};
        g.grow(x, y, seed_n, new_ps.d);
        int activated = 0;
        for (int i = 0; i < ncell; i++)
            if (activation[i]) {
                new_ps.expandSegment(&grid[i]);
                activated++;
                Hs.remove(i);
                unassigned[i] = false;
                remaining--;
            }
        if (activated < 4) continue;
        new_ps.fitPlane();
        if (new_ps.score > 100) {
            plane_segments.push_back(new_ps);
            const int nr = (int)plane_segments.size();
            for (int i = 0; i < ncell; i++)
                if (activation[i]) grid_map[i] = nr;
        }
    }

    /* plane merging (:208-245) */
    const int nr_planes = (int)plane_segments.size();
    std::vector<char> assoc((size_t)nr_planes * nr_planes, 0);
    for (int r = 0; r < nv - 1; r++)
        for (int c = 0; c < nh - 1; c++) {
            const int px = grid_map[r * nh + c];
            if (px > 0) {
                const int right = grid_map[r * nh + c + 1], below = grid_map[(r + 1) * nh + c];
                if (right > 0 && px != right) assoc[(size_t)(px - 1) * nr_planes + right - 1] = 1;
                if (below > 0 && px != below) assoc[(size_t)(px - 1) * nr_planes + below - 1] = 1;
            }
        }
    for (int r = 0; r < nr_planes; r++)
        for (int c = r + 1; c < nr_planes; c++)
            assoc[(size_t)r * nr_planes + c] = assoc[(size_t)r * nr_planes + c] || assoc[(size_t)c * nr_planes + r];
    std::vector<int> labels(nr_planes);
    for (int i = 0; i < nr_planes; i++) labels[i] = i;
    for (int r = 0; r < nr_planes; r++) {
        const int plane_id = labels[r];
        bool expanded = false;
        for (int c = r + 1; c < nr_planes; c++)
            if (assoc[(size_t)r * nr_planes + c]) {
                const Seg &P = plane_segments[plane_id], &Q = plane_segments[c];
                const double cos_angle = P.normal[0] * Q.normal[0] + P.normal[1] * Q.normal[1] + P.normal[2] * Q.normal[2];
                const double dv = plane_segments[r].normal[0] * Q.mean[0] + P.normal[1] * Q.mean[1] + P.normal[2] * Q.mean[2] + P.d;
                const double distance = dv * dv;
                if (cos_angle > cos_angle_max && distance < max_merge_dist) {
                    plane_segments[plane_id].expandSegment(&plane_segments[c]);
                    labels[c] = plane_id;
                    expanded = true;
                } else assoc[(size_t)r * nr_planes + c] = 0;
            }
        if (expanded) plane_segments[plane_id].fitPlane();
    }

    /* boundary refinement (:247-319) */
    std::vector<uint8_t> mask(ncell), eroded(ncell), dilated(ncell), diff(ncell), grid_eroded(ncell, 0);
    for (int i = 0; i < nr_planes; i++) {
        if (i != labels[i]) continue;
        std::fill(mask.begin(), mask.end(), 0);
        for (int j = i; j < nr_planes; j++)
            if (labels[j] == labels[i])
                for (int c = 0; c < ncell; c++)
                    if (grid_map[c] == j + 1) mask[c] = 1;
        /* cv::erode (3x3 cross) / cv::dilate (3x3 square), default border: outside pixels ignored */
        int mx = 0;
        for (int r = 0; r < nv; r++)
            for (int c = 0; c < nh; c++) {
                int e = mask[r * nh + c];
                if (c > 0) e = std::min<int>(e, mask[r * nh + c - 1]);
                if (c < nh - 1) e = std::min<int>(e, mask[r * nh + c + 1]);
                if (r > 0) e = std::min<int>(e, mask[(r - 1) * nh + c]);
                if (r < nv - 1) e = std::min<int>(e, mask[(r + 1) * nh + c]);
                eroded[r * nh + c] = (uint8_t)e;
                mx = std::max(mx, e);
                int dl = 0;
                for (int dr = -1; dr <= 1; dr++)
                    for (int dc = -1; dc <= 1; dc++) {
                        const int rr = r + dr, cc = c + dc;
                        if (rr >= 0 && rr < nv && cc >= 0 && cc < nh) dl = std::max<int>(dl, mask[rr * nh + cc]);
                    }
                dilated[r * nh + c] = (uint8_t)dl;
            }
        if (mx == 0) continue;
        out.planes.push_back(CapePlane());
        {
            CapePlane& p = out.planes.back();
            const Seg& s = plane_segments[i];
            for (int k = 0; k < 3; k++) { p.normal[k] = s.normal[k]; p.mean[k] = s.mean[k]; }
            p.d = s.d; p.MSE = s.MSE; p.score = s.score; p.nr_pts = s.nr_pts;
        }
        const uint8_t plane_nr = (uint8_t)out.planes.size();
        const float nx = (float)plane_segments[i].normal[0], ny = (float)plane_segments[i].normal[1];
        const float nz = (float)plane_segments[i].normal[2], d = (float)plane_segments[i].d;
        for (int c = 0; c < ncell; c++) {
            diff[c] = (uint8_t)std::max(0, (int)dilated[c] - (int)eroded[c]);
            if (eroded[c] > 0) grid_eroded[c] = plane_nr;
        }
        for (int c = 0; c < ncell; c++) {
            if (diff[c] == 0) continue;
            const int off = c * ppc;
            const float max_dist = 9 * plane_segments[i].MSE;
            for (int j = 0; j < ppc; j++) {
                const int pt = off + j;
                const float dv = X[pt] * nx + Y[pt] * ny + Z[pt] * nz + d;
                const float dist = (float)((double)dv * (double)dv);
                if (dist < max_dist && dist < dist_stacked[pt]) { dist_stacked[pt] = dist; seg_stacked[pt] = plane_nr; }
            }
        }
    }

    /* copying and rearranging segment data (:395-431) */
    for (int cr = 0; cr < nv; cr++)
        for (int cc = 0; cc < nh; cc++) {
            const int cell = cr * nh + cc;
            const int r0 = cr * patch, c0 = cc * patch;
            if (grid_eroded[cell] > 0) {
                for (int r = r0; r < r0 + patch; r++)
                    for (int c = c0; c < c0 + patch; c++) out.seg[(size_t)r * width + c] = grid_eroded[cell];
            } else {
                const uint8_t* sp = &seg_stacked[(size_t)ppc * cell];
                for (int r = r0; r < r0 + patch; r++)
                    for (int c = c0; c < c0 + patch; c++) {
                        if (*sp > 0) out.seg[(size_t)r * width + c] = *sp;
                        sp++;
                    }
            }
        }
    return out;
}

} // namespace orc
