/* oracle/capi.cpp — TEST INFRASTRUCTURE: flat C entry points (ctypes) over the CPU oracle. */
#include "oracle.h"
#include "match_oracle.h"
#include "ahc_oracle.h"
#include "cape_oracle.h"
#include "bow_oracle.h"
#include "lsd_oracle.h"
#include "oracle_math.h"

#include <chrono>
#include <cstring>
#include <exception>
#include <string>

using namespace orc;

static thread_local std::string g_err;

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

void* orc_orb_create(int nf, float sf, int nl, int ini, int mn) { return new OrbExtractor(nf, sf, nl, ini, mn); }
void orc_orb_destroy(void* h) { delete (OrbExtractor*)h; }

int orc_orb_extract(void* h, const uint8_t* gray, int w, int hh, long stride)
{
    try { return ((OrbExtractor*)h)->extract(gray, w, hh, (size_t)stride); }
    catch (const std::exception& e) { g_err = e.what(); return -1; }
}
void orc_orb_get_keypoints(void* h, KeyPoint* out)
{
    OrbExtractor* o = (OrbExtractor*)h;
    if (!o->keypoints.empty()) std::memcpy(out, o->keypoints.data(), o->keypoints.size() * sizeof(KeyPoint));
}
void orc_orb_get_descriptors(void* h, uint8_t* out)
{
    OrbExtractor* o = (OrbExtractor*)h;
    if (!o->descriptors.empty()) std::memcpy(out, o->descriptors.data(), o->descriptors.size());
}
void orc_orb_tables(void* h, float* scale, float* invScale, float* sigma2, float* invSigma2, int* quota, int* umax)
{
    OrbExtractor* o = (OrbExtractor*)h;
    for (int i = 0; i < o->nlevels; i++) {
        scale[i] = o->scale[i]; invScale[i] = o->invScale[i];
        sigma2[i] = o->sigma2[i]; invSigma2[i] = o->invSigma2[i];
        quota[i] = o->quota[i];
    }
    for (int i = 0; i <= kHalfPatch; i++) umax[i] = o->umax[i];
}
int orc_orb_geometry(void* h, int w, int hh, int* out /* nlevels x 11 */)
{
    OrbExtractor* o = (OrbExtractor*)h;
    try { o->computeGeometry(w, hh); }
    catch (const std::exception& e) { g_err = e.what(); return -1; }
    for (int l = 0; l < o->nlevels; l++) {
        const LevelGeom& g = o->geom[l];
        int* p = out + l * 11;
        p[0] = g.w; p[1] = g.h; p[2] = g.quota; p[3] = g.minBX; p[4] = g.minBY; p[5] = g.maxBX; p[6] = g.maxBY;
        p[7] = g.nCols; p[8] = g.nRows; p[9] = g.wCell; p[10] = g.hCell;
    }
    return 0;
}
void orc_orb_get_pyramid(void* h, int l, uint8_t* out)
{
    const Image& im = ((OrbExtractor*)h)->pyramid[l];
    std::memcpy(out, im.px.data(), im.px.size());
}
int orc_orb_get_blurred(void* h, int l, uint8_t* out)
{
    const Image& im = ((OrbExtractor*)h)->blurred[l];
    if (im.px.empty()) return 0;
    std::memcpy(out, im.px.data(), im.px.size());
    return 1;
}
int orc_orb_num_candidates(void* h, int l) { return (int)((OrbExtractor*)h)->candidates[l].size(); }
void orc_orb_get_candidates(void* h, int l, int32_t* out /* n x 3 */)
{
    const auto& c = ((OrbExtractor*)h)->candidates[l];
    for (size_t i = 0; i < c.size(); i++) { out[3 * i] = c[i].x; out[3 * i + 1] = c[i].y; out[3 * i + 2] = c[i].response; }
}

/* stage functions */
void orc_resize_linear_u8(const uint8_t* s, int sw, int sh, long ss, uint8_t* d, int dw, int dh, long ds)
{
    resize_linear_u8(s, sw, sh, (size_t)ss, d, dw, dh, (size_t)ds);
}
int orc_reflect101(int p, int n) { return reflect101(p, n); }
int orc_fast_detect(const uint8_t* img, int w, int h, long stride, int thr, int32_t* out, int cap)
{
    std::vector<Candidate> c;
    fast_detect(img, w, h, (size_t)stride, thr, c);
    for (size_t i = 0; i < c.size() && (int)i < cap; i++) { out[3 * i] = c[i].x; out[3 * i + 1] = c[i].y; out[3 * i + 2] = c[i].response; }
    return (int)c.size();
}
/* raw strength map: cornerScore<16>(p, 0) on rows/cols [3,n-3), -1 elsewhere */
void orc_fast_score_map(const uint8_t* img, int w, int h, long stride, int32_t* out)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            out[(size_t)y * w + x] = (y >= 3 && y < h - 3 && x >= 3 && x < w - 3)
                                         ? fast_score_9_16(img + (size_t)y * stride + x, (size_t)stride) : -1;
}
void orc_gaussian_blur(const uint8_t* s, int w, int h, long ss, uint8_t* d, long ds)
{
    gaussian_blur_7x7_s2_u8(s, w, h, (size_t)ss, d, (size_t)ds);
}
float orc_fast_atan2(float y, float x) { return fast_atan2_deg(y, x); }
void orc_sincos(float r, float* s, float* c) { sincos_f(r, s, c); }
float orc_ic_angle(void* h, const uint8_t* img, long stride, int x, int y)
{
    return ic_angle(img + (size_t)y * stride + x, (size_t)stride, ((OrbExtractor*)h)->umax);
}
void orc_orb_descriptor(const uint8_t* img, long stride, int x, int y, float angle, uint8_t* desc)
{
    orb_descriptor(img + (size_t)y * stride + x, (size_t)stride, angle, desc);
}
int orc_distribute_octtree(void* h, const int32_t* keys3, int n, int minX, int maxX, int minY, int maxY, int N,
                           int32_t* out)
{
    std::vector<Candidate> K(n);
    for (int i = 0; i < n; i++) K[i] = {keys3[3 * i], keys3[3 * i + 1], keys3[3 * i + 2]};
    try {
        std::vector<int> r = ((OrbExtractor*)h)->distributeOctTree(K, minX, maxX, minY, maxY, N);
        for (size_t i = 0; i < r.size(); i++) out[i] = r[i];
        return (int)r.size();
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int orc_hamming_swar(const uint8_t* a, const uint8_t* b) { return descriptor_distance_swar(a, b); }

/* imDepth.convertTo(depth, CV_32F, factor), src/Frame.cc:113-115: float(src)*float(scale) (+0) */
void orc_depth_to_float(const uint16_t* d, long n, float factor, float* out)
{
    for (long i = 0; i < n; i++) out[i] = (float)d[i] * factor;
}

/* Frame (no-distortion path: mvKeysUn = mvKeys, bounds = image, src/Frame.cc:836-838, 884-889) */
void* orc_frame_create(const KeyPoint* kps, const uint8_t* desc, int N, const float* depth, int dw, int dh,
                       const float K[4], float bf, int imw, int imh, const float* scaleFactors, int nlevels)
{
    Frame* f = new Frame();
    f->N = N;
    f->keys.assign(kps, kps + N);
    f->keysUn = f->keys;
    f->desc.assign(desc, desc + (size_t)N * 32);
    f->scaleFactors.assign(scaleFactors, scaleFactors + nlevels);
    f->fx = K[0]; f->fy = K[1]; f->cx = K[2]; f->cy = K[3];
    f->bf = bf; f->mb = bf / f->fx;
    f->minX = 0.f; f->maxX = (float)imw; f->minY = 0.f; f->maxY = (float)imh;
    f->gridInvW = (float)kGridCols / (float)(f->maxX - f->minX);
    f->gridInvH = (float)kGridRows / (float)(f->maxY - f->minY);
    f->computeStereoFromRGBD(depth, dw, dh);
    f->assignFeaturesToGrid();
    return f;
}
/* Frame with radial/tangential distortion: mvKeysUn through cv::undistortPoints, bounds through
 * ComputeImageBounds (src/Frame.cc:835-891); everything downstream reads mvKeysUn and the bounds */
void* orc_frame_create_dist(const KeyPoint* kps, const uint8_t* desc, int N, const float* depth, int dw, int dh,
                            const float K[4], float bf, int imw, int imh, const float* scaleFactors, int nlevels,
                            const float* dist, int nd)
{
    Frame* f = new Frame();
    f->N = N;
    f->keys.assign(kps, kps + N);
    f->keysUn = f->keys;
    if (nd > 0 && dist[0] != 0.0f && N > 0) {
        std::vector<float> xy(2 * (size_t)N), un(2 * (size_t)N);
        for (int i = 0; i < N; i++) { xy[2 * i] = kps[i].x; xy[2 * i + 1] = kps[i].y; }
        undistort_points(xy.data(), N, K, dist, nd, un.data());
        for (int i = 0; i < N; i++) { f->keysUn[i].x = un[2 * i]; f->keysUn[i].y = un[2 * i + 1]; }
    }
    f->desc.assign(desc, desc + (size_t)N * 32);
    f->scaleFactors.assign(scaleFactors, scaleFactors + nlevels);
    f->fx = K[0]; f->fy = K[1]; f->cx = K[2]; f->cy = K[3];
    f->bf = bf; f->mb = bf / f->fx;
    float b[4];
    image_bounds(imw, imh, K, dist, nd, b);
    f->minX = b[0]; f->maxX = b[1]; f->minY = b[2]; f->maxY = b[3];
    f->gridInvW = (float)kGridCols / (float)(f->maxX - f->minX);
    f->gridInvH = (float)kGridRows / (float)(f->maxY - f->minY);
    f->computeStereoFromRGBD(depth, dw, dh);
    f->assignFeaturesToGrid();
    return f;
}
void orc_frame_get_keys_un(void* h, KeyPoint* out)
{
    Frame* f = (Frame*)h;
    if (f->N) std::memcpy(out, f->keysUn.data(), sizeof(KeyPoint) * f->N);
}
void orc_frame_get_bounds(void* h, float* out) { Frame* f = (Frame*)h; out[0] = f->minX; out[1] = f->maxX; out[2] = f->minY; out[3] = f->maxY; }
void orc_undistort_points(const float* xy, int n, const float* K, const float* dist, int nd, float* out)
{
    undistort_points(xy, n, K, dist, nd, out);
}
void orc_image_bounds(int cols, int rows, const float* K, const float* dist, int nd, float* out) { image_bounds(cols, rows, K, dist, nd, out); }
void orc_frame_destroy(void* h) { delete (Frame*)h; }
void orc_frame_get_stereo(void* h, float* uRight, float* depth)
{
    Frame* f = (Frame*)h;
    std::memcpy(uRight, f->uRight.data(), f->N * sizeof(float));
    std::memcpy(depth, f->depth.data(), f->N * sizeof(float));
}
/* CSR in the reference's iteration order: cell id = ix*48 + iy (mGrid[ix][iy]) */
void orc_frame_grid_csr(void* h, int32_t* offsets /* 64*48+1 */, int32_t* indices /* N */)
{
    Frame* f = (Frame*)h;
    int o = 0;
    for (int ix = 0; ix < kGridCols; ix++)
        for (int iy = 0; iy < kGridRows; iy++) {
            offsets[ix * kGridRows + iy] = o;
            for (int idx : f->grid[ix][iy]) indices[o++] = idx;
        }
    offsets[kGridCols * kGridRows] = o;
}
int orc_frame_features_in_area(void* h, float x, float y, float r, int minL, int maxL, int32_t* out, int cap)
{
    std::vector<int> v;
    ((Frame*)h)->getFeaturesInArea(x, y, r, minL, maxL, v);
    for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = v[i];
    return (int)v.size();
}
/* Frame::UnprojectStereo, src/Frame.cc:913-923: world = Rwc*x3Dc + Ow (float gemm small path) */
void orc_frame_unproject(void* h, const float Twc[16], float* world /* N x 3 */, uint8_t* valid)
{
    Frame* f = (Frame*)h;
    const float invfx = 1.0f / f->fx, invfy = 1.0f / f->fy;
    for (int i = 0; i < f->N; i++) {
        const float z = f->depth[i];
        valid[i] = z > 0;
        if (!(z > 0)) { world[3 * i] = world[3 * i + 1] = world[3 * i + 2] = 0; continue; }
        const float u = f->keysUn[i].x, v = f->keysUn[i].y;
        const float x = (u - f->cx) * z * invfx;
        const float y = (v - f->cy) * z * invfy;
        const float p[3] = {x, y, z};
        for (int r = 0; r < 3; r++) {
            const float d = Twc[r * 4 + 0] * p[0] + Twc[r * 4 + 1] * p[1] + Twc[r * 4 + 2] * p[2];
            world[3 * i + r] = (float)((double)d * 1.0 + (double)Twc[r * 4 + 3] * 1.0);
        }
    }
}

int orc_search_by_projection_last(void* cur, void* last, const float* TcwCur, const float* TcwLast,
                                  const MapPointRec* lastMP, float th, int bMono, int checkOri,
                                  const uint8_t* curObs, int32_t* curMP)
{
    return search_by_projection_last(*(Frame*)cur, *(Frame*)last, TcwCur, TcwLast, lastMP, th, bMono != 0,
                                     checkOri != 0, curObs, curMP);
}
int orc_search_by_projection_map(void* f, const TrackedPointRec* mps, int M, float th, float nnratio,
                                 const uint8_t* claimObs, int32_t* frameMP)
{
    return search_by_projection_map(*(Frame*)f, mps, M, th, nnratio, claimObs, frameMP);
}
void orc_bf_knn_hamming(const uint8_t* Q, int nq, const uint8_t* T, int nt, int k, int32_t* idx, int32_t* dist)
{
    bf_knn_hamming(Q, nq, T, nt, k, idx, dist);
}
int orc_match_orb_points(const uint8_t* cd, int cn, const uint8_t* ld, int ln, const int32_t* lastMP,
                         const uint8_t* lastOutlier, int32_t* curMP)
{
    return match_orb_points(cd, cn, ld, ln, lastMP, lastOutlier, curMP);
}

/* AHC planes. Returns an opaque result; accessors copy out. */
struct AhcHandle { AhcResult r; std::vector<AhcBlock> blocks; };
void* orc_ahc_run(const uint16_t* depth, int w, int h, const float* K4, float depthfactor)
{
    AhcHandle* H = new AhcHandle();
    try { H->r = ahc_run(depth, w, h, K4, depthfactor, &H->blocks); }
    catch (const std::exception& e) { g_err = e.what(); delete H; return nullptr; }
    return H;
}
void orc_ahc_thresholds(int phase, double z, double* out3) { ahc_thresholds(phase, z, out3); }
void orc_ahc_disjoint_set(int n, const int32_t* pairs, int npairs, int32_t* unionRet, int32_t* findOut, int32_t* sizeOut)
{
    ahc_disjoint_set(n, pairs, npairs, unionRet, findOut, sizeOut);
}
void orc_ahc_free(void* h) { delete (AhcHandle*)h; }
int orc_ahc_num_planes(void* h) { return (int)((AhcHandle*)h)->r.planes.size(); }
int orc_ahc_num_blocks(void* h) { return (int)((AhcHandle*)h)->blocks.size(); }
void orc_ahc_get_planes(void* h, double* out /* n x 8: normal3 center3 mse curvature */, int32_t* nrid /* n x 2 */)
{
    const auto& P = ((AhcHandle*)h)->r.planes;
    for (size_t i = 0; i < P.size(); i++) {
        for (int k = 0; k < 3; k++) { out[8 * i + k] = P[i].normal[k]; out[8 * i + 3 + k] = P[i].center[k]; }
        out[8 * i + 6] = P[i].mse; out[8 * i + 7] = P[i].curvature;
        nrid[2 * i] = P[i].N; nrid[2 * i + 1] = P[i].rid;
    }
}
void orc_ahc_get_seg(void* h, uint8_t* out) { const auto& s = ((AhcHandle*)h)->r.seg; std::memcpy(out, s.data(), s.size()); }
int orc_ahc_member_count(void* h, int i) { return (int)((AhcHandle*)h)->r.membership[i].size(); }
void orc_ahc_get_members(void* h, int i, int32_t* out)
{
    const auto& m = ((AhcHandle*)h)->r.membership[i];
    for (size_t k = 0; k < m.size(); k++) out[k] = m[k];
}
void orc_ahc_get_blocks(void* h, double* out /* n x 17: sums9 center3 normal3 mse curvature */, int32_t* vn /* n x 2 */)
{
    const auto& B = ((AhcHandle*)h)->blocks;
    for (size_t i = 0; i < B.size(); i++) {
        for (int k = 0; k < 9; k++) out[17 * i + k] = B[i].sums[k];
        for (int k = 0; k < 3; k++) { out[17 * i + 9 + k] = B[i].center[k]; out[17 * i + 12 + k] = B[i].normal[k]; }
        out[17 * i + 15] = B[i].mse; out[17 * i + 16] = B[i].curvature;
        vn[2 * i] = B[i].valid; vn[2 * i + 1] = B[i].N;
    }
}
void orc_eig33sym(const double* K9, double* s3, double* V9)
{
    double K[3][3], V[3][3];
    for (int i = 0; i < 9; i++) K[i / 3][i % 3] = K9[i];
    eig33sym(K, s3, V);
    for (int i = 0; i < 9; i++) V9[i] = V[i / 3][i % 3];
}

/* CAPE planes */
void* orc_cape_run(const float* depth, int w, int h, const float* K4, int patch, float cos_angle_max, float max_merge_dist)
{
    CapeResult* R = new CapeResult();
    try { *R = cape_run(depth, w, h, K4, patch, cos_angle_max, max_merge_dist); }
    catch (const std::exception& e) { g_err = e.what(); delete R; return nullptr; }
    return R;
}
void orc_cape_free(void* h) { delete (CapeResult*)h; }
int orc_cape_num_planes(void* h) { return (int)((CapeResult*)h)->planes.size(); }
int orc_cape_num_cells(void* h) { return (int)((CapeResult*)h)->cells.size(); }
void orc_cape_get_planes(void* h, double* out /* n x 7: normal3 mean3 d */, float* ms /* n x 2 */, int32_t* npts)
{
    const auto& P = ((CapeResult*)h)->planes;
    for (size_t i = 0; i < P.size(); i++) {
        for (int k = 0; k < 3; k++) { out[7 * i + k] = P[i].normal[k]; out[7 * i + 3 + k] = P[i].mean[k]; }
        out[7 * i + 6] = P[i].d;
        ms[2 * i] = P[i].MSE; ms[2 * i + 1] = P[i].score;
        npts[i] = P[i].nr_pts;
    }
}
void orc_cape_get_seg(void* h, uint8_t* out) { const auto& s = ((CapeResult*)h)->seg; std::memcpy(out, s.data(), s.size()); }
void orc_cape_get_cells(void* h, double* out /* n x 16: sums9 mean3 normal3 d */, float* mst /* n x 3 */, int32_t* pn /* n x 2 */)
{
    const auto& C = ((CapeResult*)h)->cells;
    for (size_t i = 0; i < C.size(); i++) {
        for (int k = 0; k < 9; k++) out[16 * i + k] = C[i].sums[k];
        for (int k = 0; k < 3; k++) { out[16 * i + 9 + k] = C[i].mean[k]; out[16 * i + 12 + k] = C[i].normal[k]; }
        out[16 * i + 15] = C[i].d;
        mst[3 * i] = C[i].MSE; mst[3 * i + 1] = C[i].score; mst[3 * i + 2] = C[i].tol;
        pn[2 * i] = C[i].planar; pn[2 * i + 1] = C[i].nr_pts;
    }
}

/* DBoW2 vocabulary / transform / SearchByBoW */
void* orc_voc_create(const char* text)
{
    Vocabulary* v = new Vocabulary();
    try { v->loadFromText(text); }
    catch (const std::exception& e) { g_err = e.what(); delete v; return nullptr; }
    return v;
}
void orc_voc_free(void* h) { delete (Vocabulary*)h; }
void orc_voc_info(void* h, int32_t* out5)
{
    Vocabulary* v = (Vocabulary*)h;
    out5[0] = v->k; out5[1] = v->L; out5[2] = v->scoring; out5[3] = v->weighting; out5[4] = (int)v->nodes.size();
}
void orc_voc_get_nodes(void* h, int32_t* parent, int32_t* word_id, uint8_t* desc, double* weight)
{
    Vocabulary* v = (Vocabulary*)h;
    for (size_t i = 0; i < v->nodes.size(); i++) {
        parent[i] = v->nodes[i].parent; word_id[i] = v->nodes[i].word_id; weight[i] = v->nodes[i].weight;
        std::memcpy(desc + 32 * i, v->nodes[i].desc, 32);
    }
}
void orc_voc_transform_each(void* h, const uint8_t* desc, int n, int levelsup, int32_t* word, double* weight, int32_t* nid)
{
    Vocabulary* v = (Vocabulary*)h;
    for (int i = 0; i < n; i++) v->transformOne(desc + (size_t)i * 32, levelsup, word[i], weight[i], nid[i]);
}
/* BowVector as (ids, values) in key order; returns its size */
int orc_voc_transform_bow(void* h, const uint8_t* desc, int n, int levelsup, int32_t* ids, double* vals, int cap)
{
    std::map<int, double> bow;
    std::map<int, std::vector<unsigned>> fv;
    ((Vocabulary*)h)->transform(desc, n, levelsup, bow, fv);
    int i = 0;
    for (auto& kv : bow) { if (i < cap) { ids[i] = kv.first; vals[i] = kv.second; } i++; }
    return i;
}
/* nid arrays: node id at the FeatureVector level per feature, -1 when the word is stopped (weight <= 0) */
int orc_search_by_bow(const int32_t* nidKF, int nKF, const int32_t* nidF, int nF, const uint8_t* descKF,
                      const float* angleKF, const int32_t* kfMP, const uint8_t* descF, const float* angleF,
                      float nnratio, int checkOri, int32_t* out)
{
    std::map<int, std::vector<unsigned>> fvKF, fvF;
    for (int i = 0; i < nKF; i++) if (nidKF[i] >= 0) fvKF[nidKF[i]].push_back((unsigned)i);
    for (int i = 0; i < nF; i++) if (nidF[i] >= 0) fvF[nidF[i]].push_back((unsigned)i);
    return search_by_bow(fvKF, fvF, descKF, angleKF, kfMP, descF, angleF, nF, nnratio, checkOri != 0, out);
}

/* ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12): out2[idx2] = idx1 (vpMatches12[idx1] = map point of idx2) */
int orc_search_by_bow_kf(const int32_t* nid1, int n1, const int32_t* nid2, int n2, const uint8_t* desc1, const float* angle1,
                         const int32_t* mp1, const uint8_t* desc2, const float* angle2, const int32_t* mp2, float nnratio,
                         int checkOri, int32_t* out2)
{
    std::map<int, std::vector<unsigned>> fv1, fv2;
    for (int i = 0; i < n1; i++) if (nid1[i] >= 0) fv1[nid1[i]].push_back((unsigned)i);
    for (int i = 0; i < n2; i++) if (nid2[i] >= 0) fv2[nid2[i]].push_back((unsigned)i);
    return search_by_bow(fv1, fv2, desc1, angle1, mp1, desc2, angle2, n2, nnratio, checkOri != 0, out2, mp2, true);
}

/* per keyframe: kp = N x (x, y, angle, uRight) floats, io = N x (octave, mp, nid) ints (nid < 0: stopped word) */
int orc_search_for_triangulation(const float* kp1, const int32_t* io1, const uint8_t* desc1, int n1, const float* kp2,
                                 const int32_t* io2, const uint8_t* desc2, int n2, const float* F12, float ex, float ey,
                                 const float* scaleFactors, const float* levelSigma2, int onlyStereo, int checkOri,
                                 int32_t* out12)
{
    std::map<int, std::vector<unsigned>> fv1, fv2;
    std::vector<float> x1(n1), y1(n1), a1(n1), u1(n1), x2(n2), y2(n2), a2(n2), u2(n2);
    std::vector<int32_t> o1(n1), m1(n1), o2(n2), m2(n2);
    for (int i = 0; i < n1; i++) {
        x1[i] = kp1[4 * i]; y1[i] = kp1[4 * i + 1]; a1[i] = kp1[4 * i + 2]; u1[i] = kp1[4 * i + 3];
        o1[i] = io1[3 * i]; m1[i] = io1[3 * i + 1];
        if (io1[3 * i + 2] >= 0) fv1[io1[3 * i + 2]].push_back((unsigned)i);
    }
    for (int i = 0; i < n2; i++) {
        x2[i] = kp2[4 * i]; y2[i] = kp2[4 * i + 1]; a2[i] = kp2[4 * i + 2]; u2[i] = kp2[4 * i + 3];
        o2[i] = io2[3 * i]; m2[i] = io2[3 * i + 1];
        if (io2[3 * i + 2] >= 0) fv2[io2[3 * i + 2]].push_back((unsigned)i);
    }
    const TriKeyFrame k1 = {&fv1, x1.data(), y1.data(), a1.data(), u1.data(), o1.data(), m1.data(), desc1, n1};
    const TriKeyFrame k2 = {&fv2, x2.data(), y2.data(), a2.data(), u2.data(), o2.data(), m2.data(), desc2, n2};
    return search_for_triangulation(k1, k2, F12, ex, ey, scaleFactors, levelSigma2, onlyStereo != 0, checkOri != 0, out12);
}

/* LSD + LBD lines */
struct LineHandle { LineResult r; LsdStages st; };
void* orc_lines_run(const uint8_t* img, int w, int h, int maxLines)
{
    LineHandle* H = new LineHandle();
    try { H->r = extract_lines(img, w, h, maxLines, &H->st); }
    catch (const std::exception& e) { g_err = e.what(); delete H; return nullptr; }
    return H;
}
/* rect_nfa's reading chosen per call: 0 literal OpenCV 3.4 (default of orc_lines_run), 1 the real-valued one */
void* orc_lines_run_mode(const uint8_t* img, int w, int h, int maxLines, int rectMode)
{
    LineHandle* H = new LineHandle();
    try { H->r = extract_lines(img, w, h, maxLines, &H->st, rectMode); }
    catch (const std::exception& e) { g_err = e.what(); delete H; return nullptr; }
    return H;
}
/* the detector's raw output: number of rect_nfa calls and of segments; then their (total_pts, alg_pts) pairs / x1 y1 x2 y2 */
void orc_lines_trace_info(void* h, int32_t* out2)
{
    LineHandle* H = (LineHandle*)h;
    out2[0] = (int)(H->st.rectCounts.size() / 2); out2[1] = (int)(H->st.segments.size() / 4);
}
void orc_lines_trace_get_info(void* h, double* info3)
{
    LineHandle* H = (LineHandle*)h;
    if (info3 && !H->st.segInfo.empty()) std::memcpy(info3, H->st.segInfo.data(), H->st.segInfo.size() * sizeof(double));
}
void orc_lines_trace_get(void* h, int32_t* counts, float* segs)
{
    LineHandle* H = (LineHandle*)h;
    if (counts && !H->st.rectCounts.empty()) std::memcpy(counts, H->st.rectCounts.data(), H->st.rectCounts.size() * sizeof(int));
    if (segs && !H->st.segments.empty()) std::memcpy(segs, H->st.segments.data(), H->st.segments.size() * sizeof(float));
}
void orc_lines_free(void* h) { delete (LineHandle*)h; }
void orc_lines_info(void* h, int32_t* out4)
{
    LineHandle* H = (LineHandle*)h;
    out4[0] = (int)H->r.lines.size(); out4[1] = H->r.detected; out4[2] = H->st.sw; out4[3] = H->st.sh;
}
int orc_sizeof_keyline() { return (int)sizeof(KeyLine); }
void orc_lines_get(void* h, KeyLine* kl, uint8_t* desc, float* descf, double* lineF)
{
    LineHandle* H = (LineHandle*)h;
    const size_t n = H->r.lines.size();
    if (n) {
        std::memcpy(kl, H->r.lines.data(), n * sizeof(KeyLine));
        std::memcpy(desc, H->r.desc.data(), n * 32);
        std::memcpy(descf, H->r.descf.data(), n * 72 * sizeof(float));
        std::memcpy(lineF, H->r.lineF.data(), n * 3 * sizeof(double));
    }
}
void orc_lines_get_stages(void* h, uint8_t* scaled, double* modgrad, double* angles, int16_t* gx, int16_t* gy)
{
    LineHandle* H = (LineHandle*)h;
    if (scaled) std::memcpy(scaled, H->st.scaled.data(), H->st.scaled.size());
    if (modgrad) std::memcpy(modgrad, H->st.modgrad.data(), H->st.modgrad.size() * 8);
    if (angles) std::memcpy(angles, H->st.angles.data(), H->st.angles.size() * 8);
    if (gx && !H->st.gx.empty()) std::memcpy(gx, H->st.gx.data(), H->st.gx.size() * 2);
    if (gy && !H->st.gy.empty()) std::memcpy(gy, H->st.gy.data(), H->st.gy.size() * 2);
}

/* LSDmatcher */
void orc_line_descriptor_mad(const int32_t* dist, int nq, double* out2) { line_descriptor_mad(dist, nq, out2[0], out2[1]); }
int orc_lsd_search_by_descriptor(const uint8_t* dk, int nk, const uint8_t* has, const uint8_t* df, int nf, int32_t* out)
{
    return lsd_search_by_descriptor(dk, nk, has, df, nf, out);
}
int orc_lsd_search_for_triangulation(const uint8_t* d1, int n1, const uint8_t* d2, int n2, const uint8_t* h1, const uint8_t* h2,
                                     int32_t* out)
{
    return lsd_search_for_triangulation(d1, n1, d2, n2, h1, h2, out);
}
int orc_lsd_search_by_gap(const uint8_t* dq, int nq, const uint8_t* dt, int nt, const uint8_t* has, int32_t* out)
{
    return lsd_search_by_gap(dq, nq, dt, nt, has, out);
}

int orc_lsd_search_by_projection_last(const float* cam9, const float* TcwCur, const float* TcwLast, const float* scale,
                                      const MapLineRec* last, int nLast, const LineRec* cur, const uint8_t* curDesc,
                                      int nCur, float th, int mono, float nnratio, const uint8_t* curObs, int32_t* curML)
{
    const LineCamera cam = {cam9[0], cam9[1], cam9[2], cam9[3], cam9[4], cam9[5], cam9[6], cam9[7], cam9[8]};
    return lsd_search_by_projection_last(cam, TcwCur, TcwLast, scale, last, nLast, cur, curDesc, nCur, th, mono != 0,
                                         nnratio, curObs, curML);
}
int orc_lsd_search_by_projection_map(const float* scale, const TrackedLineRec* lines, int n, const LineRec* cur,
                                     const uint8_t* curDesc, int nCur, float th, float nnratio, const uint8_t* curObs,
                                     int32_t* curML)
{
    return lsd_search_by_projection_map(scale, lines, n, cur, curDesc, nCur, th, nnratio, curObs, curML);
}
void orc_is_in_frustum(const float* cam9, float bf, const float* Tcw, float logScale, int nLevels, const FrustumPointRec* pts,
                       int n, float limit, FrustumOut* out)
{
    const LineCamera cam = {cam9[0], cam9[1], cam9[2], cam9[3], cam9[4], cam9[5], cam9[6], cam9[7], cam9[8]};
    is_in_frustum(cam, bf, Tcw, logScale, nLevels, pts, n, limit, out);
}
void orc_is_in_frustum_lines(const float* cam9, const float* Tcw, float logScale, const FrustumLineRec* lines, int n,
                             float limit, FrustumLineOut* out)
{
    const LineCamera cam = {cam9[0], cam9[1], cam9[2], cam9[3], cam9[4], cam9[5], cam9[6], cam9[7], cam9[8]};
    is_in_frustum_lines(cam, Tcw, logScale, lines, n, limit, out);
}
float orc_logf(float x) { return log_f(x); }
void orc_fuse_search(void* frame, const float* Tcw, const float* invSigma2, float logScale, int nLevels,
                     const FrustumPointRec* pts, const uint8_t* descs, const uint8_t* skip, int n, float th, int32_t* bestIdx,
                     int32_t* bestDist)
{
    fuse_search(*(Frame*)frame, Tcw, invSigma2, logScale, nLevels, pts, descs, skip, n, th, bestIdx, bestDist);
}
void orc_lsd_fuse_search(const float* cam9, const float* Tcw, float logScale, const float* scaleFactors, int nLevels,
                         const FrustumLineRec* lines, const uint8_t* descs, const uint8_t* skip, int n, const LineRec* kf,
                         const uint8_t* kfDesc, int nKF, float th, int32_t* bestIdx, int32_t* bestDist)
{
    LineCamera cam;
    std::memcpy(&cam, cam9, sizeof(cam));
    lsd_fuse_search(cam, Tcw, logScale, scaleFactors, nLevels, lines, descs, skip, n, kf, kfDesc, nKF, th, bestIdx, bestDist);
}
void orc_lsd_fuse_search_sim3(const float* cam9, const float* Scw, float logScale, const float* scaleFactors, int nLevels,
                              const FrustumLineRec* lines, const uint8_t* descs, const uint8_t* skip, int n, const LineRec* kf,
                              const uint8_t* kfDesc, int nKF, float th, int32_t* bestIdx, int32_t* bestDist)
{
    LineCamera cam;
    std::memcpy(&cam, cam9, sizeof(cam));
    lsd_fuse_search_sim3(cam, Scw, logScale, scaleFactors, nLevels, lines, descs, skip, n, kf, kfDesc, nKF, th, bestIdx, bestDist);
}
int orc_lsd_search_by_projection_kf(const float* cam9, const float* Scw, float logScale, const float* scaleFactors, int nLevels,
                                    const FrustumLineRec* lines, const uint8_t* descs, const uint8_t* skip, int n, const LineRec* kf,
                                    const uint8_t* kfDesc, int nKF, const uint8_t* matched, int th, int32_t* newMatch)
{
    LineCamera cam;
    std::memcpy(&cam, cam9, sizeof(cam));
    return lsd_search_by_projection_kf(cam, Scw, logScale, scaleFactors, nLevels, lines, descs, skip, n, kf, kfDesc, nKF, matched, th, newMatch);
}
int orc_lsd_search_by_sim3(const float* cam9, const float* T1w, const float* T2w, float s12, const float* R12, const float* t12,
                           float logScale, const float* scaleFactors, int nLevels, const FrustumLineRec* lines1, const uint8_t* descs1,
                           const uint8_t* skip1, const LineRec* kf1, const uint8_t* kf1Desc, int n1, const FrustumLineRec* lines2,
                           const uint8_t* descs2, const uint8_t* skip2, const LineRec* kf2, const uint8_t* kf2Desc, int n2, float th,
                           int32_t* out12)
{
    LineCamera cam;
    std::memcpy(&cam, cam9, sizeof(cam));
    return lsd_search_by_sim3(cam, T1w, T2w, s12, R12, t12, logScale, scaleFactors, nLevels, lines1, descs1, skip1, kf1, kf1Desc, n1,
                              lines2, descs2, skip2, kf2, kf2Desc, n2, th, out12);
}
int orc_search_by_projection_kf(void* kf, const float* Scw, float logScale, int nLevels, const FrustumPointRec* pts,
                                const uint8_t* descs, const uint8_t* skip, int n, const uint8_t* matched, float th, int32_t* newMatch)
{
    return search_by_projection_kf(*(Frame*)kf, Scw, logScale, nLevels, pts, descs, skip, n, matched, th, newMatch);
}
int orc_search_for_initialization(void* f1, void* f2, float* prevMatched, int windowSize, float nnratio, int checkOri, int32_t* matches12)
{
    return search_for_initialization(*(Frame*)f1, *(Frame*)f2, prevMatched, windowSize, nnratio, checkOri != 0, matches12);
}

int orc_search_by_projection_reloc(void* cur, const float* Tcw, float logScale, int nLevels, const FrustumPointRec* pts,
                                   const uint8_t* descs, const float* kfAngles, const uint8_t* skip, int n, const uint8_t* matched,
                                   float th, int orbDist, int checkOri, int32_t* newMatch)
{
    return search_by_projection_reloc(*(Frame*)cur, Tcw, logScale, nLevels, pts, descs, kfAngles, skip, n, matched, th, orbDist,
                                      checkOri != 0, newMatch);
}
int orc_search_by_sim3(void* kf1, void* kf2, const float* T1w, const float* T2w, float s12, const float* R12, const float* t12,
                       float logScale, int nLevels, const FrustumPointRec* pts1, const uint8_t* descs1, const uint8_t* skip1,
                       const FrustumPointRec* pts2, const uint8_t* descs2, const uint8_t* skip2, float th, int32_t* out12)
{
    return search_by_sim3(*(Frame*)kf1, *(Frame*)kf2, T1w, T2w, s12, R12, t12, logScale, nLevels, pts1, descs1, skip1, pts2,
                          descs2, skip2, th, out12);
}
void orc_fuse_search_sim3(void* frame, const float* Scw, const float* invSigma2, float logScale, int nLevels,
                          const FrustumPointRec* pts, const uint8_t* descs, const uint8_t* skip, int n, float th,
                          int32_t* bestIdx, int32_t* bestDist)
{
    float Tcw[16];
    decompose_sim3(Scw, Tcw);
    fuse_search(*(Frame*)frame, Tcw, invSigma2, logScale, nLevels, pts, descs, skip, n, th, bestIdx, bestDist, true);
}
int orc_sizeof_maplinerec() { return (int)sizeof(MapLineRec); }
int orc_sizeof_trackedlinerec() { return (int)sizeof(TrackedLineRec); }

int orc_sizeof_keypoint() { return (int)sizeof(KeyPoint); }
int orc_sizeof_mappointrec() { return (int)sizeof(MapPointRec); }
int orc_sizeof_trackedpointrec() { return (int)sizeof(TrackedPointRec); }

} // extern "C"
