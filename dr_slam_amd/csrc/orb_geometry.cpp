/* orb_geometry.cpp — host-side constants of the ORB path: everything the reference computes once in
 * ORBextractor::ORBextractor (src/ORBextractor.cc:410-470) and the per-level geometry it derives per
 * frame in ComputePyramid (:1111-1113) / ComputeKeyPointsOctTree (:773-806), flattened into device
 * tables so the kernels contain no float geometry of their own. */
#include "drfe_internal.h"
#include "../../include/drfe_math.h"

#include <algorithm>
#include <cmath>

static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }

int drfe_build_tables(drfe_ctx* c)
{
    const int nl = c->cfg.nlevels;
    /* the reference stores scaleFactor in a double member initialised from the float argument
     * (include/ORBextractor.h:98) and multiplies float tables by it (:419-431) */
    const double sfd = (double)c->cfg.scale_factor;
    c->scale.assign(nl, 1.f); c->sigma2.assign(nl, 1.f);
    c->invScale.assign(nl, 1.f); c->invSigma2.assign(nl, 1.f);
    for (int i = 1; i < nl; i++) {
        c->scale[i] = (float)(c->scale[i - 1] * sfd);
        c->sigma2[i] = c->scale[i] * c->scale[i];
    }
    for (int i = 0; i < nl; i++) {
        c->invScale[i] = 1.0f / c->scale[i];
        c->invSigma2[i] = 1.0f / c->sigma2[i];
    }
    /* per-level quotas, :434-446 */
    c->quota.assign(nl, 0);
    const float factor = (float)(1.0f / sfd);
    float want = c->cfg.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
    int sum = 0;
    for (int l = 0; l < nl - 1; l++) {
        c->quota[l] = drfe_round_half_even(want);
        sum += c->quota[l];
        want *= factor;
    }
    c->quota[nl - 1] = std::max(c->cfg.nfeatures - sum, 0);
    /* umax, :454-469 */
    const int hp = DRFE_HALF_PATCH;
    for (int v = 0; v <= hp; v++) c->umax[v] = 0;
    const int vmax = (int)std::floor(hp * std::sqrt(2.f) / 2 + 1);
    const int vmin = (int)std::ceil(hp * std::sqrt(2.f) / 2);
    const double hp2 = hp * hp;
    for (int v = 0; v <= vmax; ++v) c->umax[v] = drfe_round_half_even_d(std::sqrt(hp2 - v * v));
    for (int v = hp, v0 = 0; v >= vmin; --v) {
        while (c->umax[v0] == c->umax[v0 + 1]) ++v0;
        c->umax[v] = v0;
        ++v0;
    }
    return DRFE_OK;
}

/* cv::resize(INTER_LINEAR) coefficient generation for one axis: source index pair
 * and the two 11-bit weights per destination index. */
static void build_taps_axis(int src, int dst, std::vector<ResizeTap>* out)
{
    const double inv = (double)dst / src;
    const double scale = 1. / inv;
    for (int d = 0; d < dst; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        if (s < 0) { s = 0; f = 0; }
        if (s >= src - 1) { s = src - 1; f = 0; }
        ResizeTap t;
        t.s0 = (uint16_t)s;
        t.s1 = (uint16_t)std::min(s + 1, src - 1);
        int w0 = drfe_round_half_even((1.f - f) * 2048.f), w1 = drfe_round_half_even(f * 2048.f);
        t.w0 = (int16_t)std::min(32767, std::max(-32768, w0));
        t.w1 = (int16_t)std::min(32767, std::max(-32768, w1));
        out->push_back(t);
    }
}

/* The table the kernel reads is indexed by the BORDERED destination coordinate (copyMakeBorder
 * REFLECT_101 of the resized interior, :1122-1123, folded into the lookup), starts 16-byte aligned and
 * is padded to `padded` entries with zero-weight taps (-> output 0 in the pitch padding) that keep the
 * last source position, so the source window of a thread's four taps stays <= 9 pixels wide. */
static void build_taps(int src, int dst, int padded, std::vector<ResizeTap>* out)
{
    std::vector<ResizeTap> axis;
    build_taps_axis(src, dst, &axis);
    if (out->size() & 1) out->push_back(ResizeTap{0, 0, 0, 0});
    const int bordered = dst + 2 * DRFE_EDGE;
    for (int b = 0; b < padded; b++) {
        if (b >= bordered) { ResizeTap z = out->back(); z.w0 = z.w1 = 0; out->push_back(z); continue; }
        int p = b - DRFE_EDGE;
        if (p < 0) p = -p;
        if (p >= dst) p = 2 * (dst - 1) - p;
        out->push_back(axis[p]);
    }
}

int drfe_build_geometry(drfe_ctx* c, int w, int h, DevGeom* g, std::vector<FastCell>* cells,
                        std::vector<BlurTile>* tiles, std::vector<ResizeTap>* taps)
{
    const int nl = c->cfg.nlevels;
    g->nlevels = nl;
    g->iniTh = c->cfg.ini_th_fast;
    g->minTh = c->cfg.min_th_fast;
    g->imgW = w; g->imgH = h;
    g->fastMaxWh = 7;
    g->fastCols = 1;
    g->fastColsRows = 7;
    int pyrOff = 0, blurOff = 0, candOff = 0, kpOff = 0;
    if (cells) cells->clear();
    if (tiles) tiles->clear();
    if (taps) taps->clear();
    for (int l = 0; l < nl; l++) {
        DevLevel& L = g->lv[l];
        L.w = drfe_round_half_even((float)w * c->invScale[l]);
        L.h = drfe_round_half_even((float)h * c->invScale[l]);
        L.quota = c->quota[l];
        L.scale = c->scale[l];
        L.kpSize = (float)(int)(31 * c->scale[l]);
        L.minBX = DRFE_EDGE - 3; L.minBY = DRFE_EDGE - 3;
        L.maxBX = L.w - DRFE_EDGE + 3; L.maxBY = L.h - DRFE_EDGE + 3;
        const float width = (float)(L.maxBX - L.minBX), height = (float)(L.maxBY - L.minBY);
        L.nCols = (int)(width / 30.f);
        L.nRows = (int)(height / 30.f);
        if (L.nCols <= 0 || L.nRows <= 0) {
            c->err = "pyramid level " + std::to_string(l) + " is smaller than one 30-px FAST cell";
            return DRFE_ERR_INVALID;
        }
        L.wCell = (int)std::ceil(width / L.nCols);
        L.hCell = (int)std::ceil(height / L.nRows);
        if (L.wCell + 6 > DRFE_FAST_MAX_WIN || L.hCell + 6 > DRFE_FAST_MAX_WIN) {
            c->err = "FAST cell window exceeds the LDS tile";
            return DRFE_ERR_INVALID;
        }
        /* DistributeOctTree root nodes, :543-545 */
        L.nIni = (int)std::round((float)(L.maxBX - L.minBX) / (L.maxBY - L.minBY));
        if (L.nIni < 1) {
            c->err = "frame aspect ratio gives zero quadtree root nodes (reference divides by zero)";
            return DRFE_ERR_INVALID;
        }
        L.hX = (float)(L.maxBX - L.minBX) / L.nIni;
        L.pyrPitch = align_up(L.w + 2 * DRFE_EDGE, 64);
        L.pyrOff = pyrOff;
        pyrOff += align_up(L.pyrPitch * (L.h + 2 * DRFE_EDGE), 256);
        /* blurred level: 32 x 4-pixel tiles of 128 bytes (DRFE_BLUR_TILE_*), blurPitch = tiles per tile row.  Its only
         * reader gathers 37 x 37 patches: ~23 lines per patch instead of ~48 with row-major rows. */
        L.blurPitch = (L.w + DRFE_BTILE_W - 1) / DRFE_BTILE_W;
        L.blurOff = blurOff;
        blurOff += align_up(L.blurPitch * ((L.h + DRFE_BTILE_H - 1) / DRFE_BTILE_H) * 128, 256);
        /* FAST cells in the reference's loop order with its skip rules, :789-806 */
        L.cellBegin = cells ? (int)cells->size() : 0;
        int candCap = 0;
        for (int i = 0; i < L.nRows; i++) {
            const float iniY = (float)(L.minBY + i * L.hCell);
            float maxY = iniY + L.hCell + 6;
            if (iniY >= L.maxBY - 3) continue;
            if (maxY > L.maxBY) maxY = (float)L.maxBY;
            for (int j = 0; j < L.nCols; j++) {
                const float iniX = (float)(L.minBX + j * L.wCell);
                float maxX = iniX + L.wCell + 6;
                if (iniX >= L.maxBX - 6) continue;
                if (maxX > L.maxBX) maxX = (float)L.maxBX;
                FastCell fc;
                fc.x0 = (uint16_t)iniX; fc.y0 = (uint16_t)iniY;
                const int ww = (int)maxX - (int)iniX, wh = (int)maxY - (int)iniY;
                if (ww < 7 || wh < 7) continue; /* cv::FAST evaluates nothing */
                fc.ww = (uint8_t)ww; fc.wh = (uint8_t)wh;
                fc.level = (uint8_t)l;
                fc.off = (uint8_t)((fc.x0 + DRFE_EDGE) & 3);
                fc.srcOff = (uint32_t)(L.pyrOff + (fc.y0 + DRFE_EDGE) * L.pyrPitch + (fc.x0 + DRFE_EDGE - fc.off));
                fc.pitch = (uint32_t)L.pyrPitch;
                fc.candOff = 0; fc.candCap = 0;   /* filled once the level's capacity is known */
                fc.offX = (uint16_t)(j * L.wCell); fc.offY = (uint16_t)(i * L.hCell);
                fc.cellIdx = (uint32_t)(i * L.nCols + j);
                {
                    const int ew = ww - 6, eh = wh - 6;
                    const int ncp = (ew + 1) / 2, nrb0 = std::min(64 / ncp, eh);
                    const int rpl = (eh + nrb0 - 1) / nrb0;
                    const int nrb = (eh + rpl - 1) / rpl;      /* every block but the last one is full, the last one not empty */
                    fc.ncp = (uint8_t)ncp; fc.nrb = (uint8_t)nrb; fc.rpl = (uint8_t)rpl; fc.pad = 0;
                    fc.ncpMagic = (uint32_t)((65536 + ncp - 1) / ncp);
                    g->fastColsRows = std::max(g->fastColsRows, std::max(nrb * rpl + 6, wh));
                    if (ew > DRFE_FASTC_MAX_EW || rpl > DRFE_FASTC_MAX_RPL) g->fastCols = 0;
                }
                if (cells) cells->push_back(fc);
                if (wh > g->fastMaxWh) g->fastMaxWh = wh;
                /* strict 3x3 maxima: at most one per 2x2 block of the evaluated area */
                candCap += ((ww - 6 + 1) / 2) * ((wh - 6 + 1) / 2);
            }
        }
        L.cellEnd = cells ? (int)cells->size() : 0;
        L.candCap = align_up(std::max(candCap, 64), 64);
        L.candOff = candOff;
        if (cells)
            for (int k = L.cellBegin; k < L.cellEnd; k++) { (*cells)[k].candOff = (uint32_t)L.candOff; (*cells)[k].candCap = (uint32_t)L.candCap; }
        candOff += L.candCap;
        L.kpCap = std::max(L.quota, 4 * L.nIni) + 4;
        if (L.kpCap > DRFE_QT_MAX_NODES) {
            c->err = "nfeatures too large for the quadtree node pool";
            return DRFE_ERR_CAPACITY;
        }
        L.kpOff = kpOff;
        kpOff += L.kpCap;
        /* blur tiles */
        L.tileBegin = tiles ? (int)tiles->size() : 0;
        for (int ty = 0; ty < (L.h + DRFE_BLUR_TH - 1) / DRFE_BLUR_TH; ty++)
            for (int tx = 0; tx < (L.w + DRFE_BLUR_TW - 1) / DRFE_BLUR_TW; tx++) {
                BlurTile t; t.tx = (uint16_t)tx; t.ty = (uint16_t)ty; t.level = (uint16_t)l; t.pad = 0;
                if (tiles) tiles->push_back(t);
            }
        L.tileEnd = tiles ? (int)tiles->size() : 0;
        /* resize taps from level l-1 (cascade, :1120) */
        L.xtabOff = L.ytabOff = 0;
        L.xwinOff = L.ywinOff = 0;
        L.resizeLds = 0;
        if (l > 0 && taps) {
            if (taps->size() & 1) taps->push_back(ResizeTap{0, 0, 0, 0});
            L.xtabOff = (int)taps->size();
            build_taps(g->lv[l - 1].w, L.w, L.pyrPitch, taps);
            if (taps->size() & 1) taps->push_back(ResizeTap{0, 0, 0, 0});
            L.ytabOff = (int)taps->size();
            build_taps(g->lv[l - 1].h, L.h, align_up(L.h + 2 * DRFE_EDGE, 4), taps);   /* 4 rows per thread */
            /* source windows of the 256 x 16 output blocks of k_pyr_resize_lds */
            const int bh = L.h + 2 * DRFE_EDGE;
            const int nbx = (L.pyrPitch + 255) / 256, nby = (bh + 15) / 16;
            L.resizeLds = 1;
            L.xwinOff = (int)taps->size();
            for (int b = 0; b < nbx; b++) {
                int lo = 1 << 30, hi = -1;
                for (int x = b * 256; x < std::min(b * 256 + 256, L.pyrPitch); x++) {
                    const ResizeTap& t = (*taps)[L.xtabOff + x];
                    lo = std::min(lo, (int)t.s0); hi = std::max(hi, (int)t.s1);
                }
                const int ws = (lo + DRFE_EDGE) & ~3;
                if ((hi + DRFE_EDGE - ws) / 4 + 1 > DRFE_RESIZE_LDS_WD) L.resizeLds = 0;
                taps->push_back(ResizeTap{(uint16_t)lo, (uint16_t)hi, 0, 0});
            }
            L.ywinOff = (int)taps->size();
            for (int b = 0; b < nby; b++) {
                int lo = 1 << 30, hi = -1;
                for (int y = b * 16; y < std::min(b * 16 + 16, bh); y++) {
                    const ResizeTap& t = (*taps)[L.ytabOff + y];
                    lo = std::min(lo, (int)t.s0); hi = std::max(hi, (int)t.s1);
                }
                if (hi - lo + 1 > DRFE_RESIZE_LDS_ROWS) L.resizeLds = 0;
                taps->push_back(ResizeTap{(uint16_t)lo, (uint16_t)hi, 0, 0});
            }
            if (!L.resizeLds) {
                /* the register-only kernel reads a 12-byte window per four output columns: make sure it is enough */
                for (int x = 0; x + 3 < L.pyrPitch; x += 4) {
                    int lo = 1 << 30, hi = -1;
                    for (int k = 0; k < 4; k++) {
                        const ResizeTap& t = (*taps)[L.xtabOff + x + k];
                        lo = std::min(lo, (int)t.s0); hi = std::max(hi, (int)t.s1);
                    }
                    if (hi + DRFE_EDGE - ((lo + DRFE_EDGE) & ~3) > 11) {
                        c->err = "pyramid scale factor too large for the resize kernels (supported: up to 2.0)";
                        return DRFE_ERR_INVALID;
                    }
                }
            }
        }
    }
    /* k_fast_cells_cols keeps a lane's row scores in registers: the cells of at most 8 rows per lane (the four large levels
     * at 640x480) come first and run the 8-row instantiation, the rest the DRFE_FASTC_MAX_RPL one.  The table's order is
     * free: FastCell::cellIdx carries the emission order. */
    g->fastColsSmall = 0;
    if (cells) {
        std::stable_partition(cells->begin(), cells->end(), [](const FastCell& f) { return f.rpl <= 8; });
        for (const FastCell& f : *cells) g->fastColsSmall += f.rpl <= 8 ? 1 : 0;
    }
    g->pyrSlotBytes = pyrOff;
    g->blurSlotBytes = blurOff;
    g->candSlotElems = candOff;
    g->kpSlotElems = kpOff;
    g->totalCells = cells ? (int)cells->size() : 0;
    g->totalTiles = tiles ? (int)tiles->size() : 0;
    return DRFE_OK;
}
