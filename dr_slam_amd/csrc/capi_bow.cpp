/* capi_bow.cpp — C-ABI of the bag-of-words step (include/drfe.h): vocabulary upload, Frame::ComputeBoW
 * device part, ORBmatcher::SearchByBoW. */
#include "drfe_internal.h"
#include "bow_internal.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <new>

#define HIPCHK(c, call)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (c)->err = std::string(#call) + ": " + hipGetErrorString(e__);                      \
            return DRFE_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

void drfe_bow_free(drfe_ctx* c)
{
    BowState* b = c->bow;
    if (!b) return;
    for (void* p : b->d_vocBlob)
        if (p) (void)hipFree(p);
    void* ptrs[] = {b->d_word, b->d_weight, b->d_nid, b->d_groups, b->d_kfIdx, b->d_fIdx, b->d_kfMP, b->d_fMP, b->d_match,
                    b->d_counters, b->d_hist, b->d_entries};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete[] b->slotDone;
    delete b;
    c->bow = nullptr;
}

void drfe_bow_slot_invalidate(drfe_ctx* c, int slot)
{
    BowState* b = c->bow;
    if (b && slot >= 0 && slot < b->nSlots) b->slotDone[slot] = 0;
}

extern "C" {

int drfe_voc_upload(drfe_ctx* c, int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent,
                    const uint8_t* desc, const double* weight, const uint8_t* is_leaf)
{
    if (!c || !parent || !desc || !weight || !is_leaf || n_nodes < 2) return DRFE_ERR_INVALID;
    /* the limits TemplatedVocabulary::loadFromTextFile enforces (:1359) */
    if (k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3) {
        c->err = "voc_upload: not a valid vocabulary header";
        return DRFE_ERR_INVALID;
    }
    HIPCHK(c, hipSetDevice(c->device));
    drfe_bow_free(c);
    BowState* b = new (std::nothrow) BowState();
    if (!b) return DRFE_ERR_INVALID;
    std::memset(b, 0, sizeof(*b));
    c->bow = b;
    b->nSlots = c->cfg.max_batch;
    b->slotDone = new (std::nothrow) uint8_t[(size_t)b->nSlots]();
    if (!b->slotDone) return DRFE_ERR_INVALID;
    /* children lists in node-id order, as the loader's m_nodes[pid].children.push_back(nid) builds them */
    std::vector<int> cnt(n_nodes + 1, 0), children(n_nodes - 1), wordId(n_nodes, -1);
    for (int i = 1; i < n_nodes; i++) {
        if (parent[i] < 0 || parent[i] >= n_nodes) { c->err = "voc_upload: bad parent id"; return DRFE_ERR_INVALID; }
        cnt[parent[i] + 1]++;
    }
    for (int i = 0; i < n_nodes; i++) {
        if (cnt[i + 1] > 32) { c->err = "voc_upload: more than 32 children per node"; return DRFE_ERR_INVALID; }
        cnt[i + 1] += cnt[i];
    }
    std::vector<int> fill(cnt.begin(), cnt.end() - 1);
    int nwords = 0;
    for (int i = 1; i < n_nodes; i++) {
        children[fill[parent[i]]++] = i;
        if (is_leaf[i]) wordId[i] = nwords++;
    }
    const size_t n = (size_t)n_nodes;
    uint8_t* d_desc; double* d_w; int *d_word, *d_cb, *d_ch;
    HIPCHK(c, hipMalloc((void**)&d_desc, n * 32)); b->d_vocBlob[0] = d_desc;
    HIPCHK(c, hipMalloc((void**)&d_w, n * 8)); b->d_vocBlob[1] = d_w;
    HIPCHK(c, hipMalloc((void**)&d_word, n * 4)); b->d_vocBlob[2] = d_word;
    HIPCHK(c, hipMalloc((void**)&d_cb, (n + 1) * 4)); b->d_vocBlob[3] = d_cb;
    HIPCHK(c, hipMalloc((void**)&d_ch, n * 4)); b->d_vocBlob[4] = d_ch;
    HIPCHK(c, hipMemcpy(d_desc, desc, n * 32, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(d_w, weight, n * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(d_word, wordId.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(d_cb, cnt.data(), (n + 1) * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(d_ch, children.data(), (n - 1) * 4, hipMemcpyHostToDevice));
    b->voc.k = k; b->voc.L = L; b->voc.nNodes = n_nodes;
    b->voc.desc = d_desc; b->voc.weight = d_w; b->voc.wordId = d_word; b->voc.childBegin = d_cb; b->voc.children = d_ch;
    b->scoring = scoring; b->weighting = weighting;
    const size_t B = (size_t)c->cfg.max_batch, M = (size_t)c->maxKp;
    HIPCHK(c, hipMalloc((void**)&b->d_word, B * M * 4));
    HIPCHK(c, hipMalloc((void**)&b->d_weight, B * M * 8));
    HIPCHK(c, hipMalloc((void**)&b->d_nid, B * M * 4));
    HIPCHK(c, hipMalloc((void**)&b->d_groups, M * sizeof(BowGroup)));
    HIPCHK(c, hipMalloc((void**)&b->d_kfIdx, M * 4));
    HIPCHK(c, hipMalloc((void**)&b->d_fIdx, M * 4));
    HIPCHK(c, hipMalloc((void**)&b->d_kfMP, M * 4));
    HIPCHK(c, hipMalloc((void**)&b->d_fMP, M * 4));
    HIPCHK(c, hipMalloc((void**)&b->d_match, M * 4));
    HIPCHK(c, hipMalloc((void**)&b->d_counters, 8));
    HIPCHK(c, hipMalloc((void**)&b->d_hist, 30 * 4));
    HIPCHK(c, hipMalloc((void**)&b->d_entries, M * 4));
    return DRFE_OK;
}

int drfe_bow_transform_batch(drfe_ctx* c, int levelsup, int nframes, void* stream)
{
    if (!c) return DRFE_ERR_INVALID;
    if (!c->bow) { c->err = "bow_transform: upload a vocabulary first"; return DRFE_ERR_STATE; }
    if (nframes < 1 || nframes > c->lastBatch) { c->err = "bow_transform: extract the batch first"; return DRFE_ERR_STATE; }
    HIPCHK(c, hipSetDevice(c->device));
    BowState* b = c->bow;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    HIPCHK(c, drfe_launch_bow_transform(c, b->voc, levelsup, nframes, b->d_word, b->d_weight, b->d_nid, s));
    if (b->levelsup != levelsup) std::memset(b->slotDone, 0, (size_t)b->nSlots);      /* node ids of another level */
    b->levelsup = levelsup;
    for (int f = 0; f < nframes && f < b->nSlots; f++) b->slotDone[f] = 1;
    return DRFE_OK;
}

/* The same for ONE slot (a frame put there by drfe_frame_submit / drfe_frame_load): Frame::ComputeBoW / KeyFrame::ComputeBoW of the
 * frame in `slot`.  The transform is a pure function of the descriptors and the vocabulary, so a keyframe loaded from the host
 * gets the mFeatVec / mBowVec entries it stored when it was created. */
int drfe_bow_transform_slot(drfe_ctx* c, int levelsup, int slot, void* stream)
{
    if (!c) return DRFE_ERR_INVALID;
    if (!c->bow) { c->err = "bow_transform: upload a vocabulary first"; return DRFE_ERR_STATE; }
    if (slot < 0 || slot >= c->lastBatch) { c->err = "bow_transform_slot: the slot holds no frame"; return DRFE_ERR_STATE; }
    HIPCHK(c, hipSetDevice(c->device));
    BowState* b = c->bow;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const size_t o = (size_t)slot * c->maxKp;
    {
        SlotShift shift(c, slot);
        HIPCHK(c, drfe_launch_bow_transform(c, b->voc, levelsup, 1, b->d_word + o, b->d_weight + o, b->d_nid + o, s));
    }
    if (b->levelsup != levelsup) std::memset(b->slotDone, 0, (size_t)b->nSlots);
    b->levelsup = levelsup;
    b->slotDone[slot] = 1;
    return DRFE_OK;
}

int drfe_bow_download(drfe_ctx* c, int slot, int32_t* word, double* weight, int32_t* nid, int cap)
{
    if (!c || !drfe_bow_slot_done(c->bow, slot)) return c ? DRFE_ERR_STATE : DRFE_ERR_INVALID;
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    int n = 0;
    HIPCHK(c, hipMemcpy(&n, c->d_kpCount + slot, sizeof(int), hipMemcpyDeviceToHost));
    if (n > cap) return DRFE_ERR_CAPACITY;
    const size_t o = (size_t)slot * c->maxKp;
    if (n && word) HIPCHK(c, hipMemcpy(word, c->bow->d_word + o, sizeof(int) * n, hipMemcpyDeviceToHost));
    if (n && weight) HIPCHK(c, hipMemcpy(weight, c->bow->d_weight + o, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (n && nid) HIPCHK(c, hipMemcpy(nid, c->bow->d_nid + o, sizeof(int) * n, hipMemcpyDeviceToHost));
    return DRFE_OK;
}

/* shared body of SearchByBoW(KF, Frame) (f_mp == NULL, `<= TH_LOW`) and SearchByBoW(KF1, KF2) (f_mp given, `< TH_LOW`) */
static int search_by_bow_impl(drfe_ctx* c, int kf_slot, int f_slot, const int32_t* kf_mp, int n_kf, const int32_t* f_mp,
                              int th_low, float nnratio, int check_ori, int32_t* f_match, int n_f, int* nmatches)
{
    if (!c || !kf_mp || !f_match || !nmatches) return DRFE_ERR_INVALID;
    BowState* b = c->bow;
    if (!drfe_bow_slot_done(b, kf_slot) || !drfe_bow_slot_done(b, f_slot)) {
        c->err = "search_by_bow: both slots need drfe_bow_transform_batch first";
        return DRFE_ERR_STATE;
    }
    HIPCHK(c, hipSetDevice(c->device));
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    int counts[2];
    HIPCHK(c, hipMemcpy(&counts[0], c->d_kpCount + kf_slot, sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&counts[1], c->d_kpCount + f_slot, sizeof(int), hipMemcpyDeviceToHost));
    if (n_kf != counts[0] || n_f != counts[1]) { c->err = "search_by_bow: N mismatch"; return DRFE_ERR_INVALID; }
    *nmatches = 0;
    for (int i = 0; i < n_f; i++) f_match[i] = -1;
    if (n_kf == 0 || n_f == 0) return DRFE_OK;
    /* FeatureVectors of both frames: node -> feature indices in feature order, stopped words skipped
     * (FeatureVector::addFeature is called only when w > 0, TemplatedVocabulary.h:1158-1162) */
    std::vector<int> nidKF(n_kf), nidF(n_f);
    std::vector<double> wKF(n_kf), wF(n_f);
    const size_t ok = (size_t)kf_slot * c->maxKp, of = (size_t)f_slot * c->maxKp;
    HIPCHK(c, hipMemcpy(nidKF.data(), b->d_nid + ok, sizeof(int) * n_kf, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(wKF.data(), b->d_weight + ok, sizeof(double) * n_kf, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(nidF.data(), b->d_nid + of, sizeof(int) * n_f, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(wF.data(), b->d_weight + of, sizeof(double) * n_f, hipMemcpyDeviceToHost));
    std::map<int, std::vector<int>> fvKF, fvF;
    for (int i = 0; i < n_kf; i++) if (wKF[i] > 0) fvKF[nidKF[i]].push_back(i);
    for (int i = 0; i < n_f; i++) if (wF[i] > 0) fvF[nidF[i]].push_back(i);
    std::vector<BowGroup> groups;
    std::vector<int> kfIdx, fIdx;
    for (auto& kv : fvKF) {                                   /* ascending node id == the merge walk of :181-265 */
        auto it = fvF.find(kv.first);
        if (it == fvF.end()) continue;
        BowGroup g;
        g.kfBegin = (int)kfIdx.size(); kfIdx.insert(kfIdx.end(), kv.second.begin(), kv.second.end()); g.kfEnd = (int)kfIdx.size();
        g.fBegin = (int)fIdx.size(); fIdx.insert(fIdx.end(), it->second.begin(), it->second.end()); g.fEnd = (int)fIdx.size();
        groups.push_back(g);
    }
    hipStream_t s = c->stream;
    if (!groups.empty()) {
        HIPCHK(c, hipMemcpy(b->d_groups, groups.data(), groups.size() * sizeof(BowGroup), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(b->d_kfIdx, kfIdx.data(), kfIdx.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(b->d_fIdx, fIdx.data(), fIdx.size() * 4, hipMemcpyHostToDevice));
    }
    HIPCHK(c, hipMemcpy(b->d_kfMP, kf_mp, sizeof(int) * n_kf, hipMemcpyHostToDevice));
    if (f_mp) HIPCHK(c, hipMemcpy(b->d_fMP, f_mp, sizeof(int) * n_f, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemsetAsync(b->d_match, 0xFF, sizeof(int) * n_f, s));
    HIPCHK(c, hipMemsetAsync(b->d_counters, 0, 8, s));
    HIPCHK(c, hipMemsetAsync(b->d_hist, 0, 30 * 4, s));
    HIPCHK(c, drfe_launch_bow_match(c, kf_slot, f_slot, b->d_groups, (int)groups.size(), b->d_kfIdx, b->d_fIdx, b->d_kfMP,
                                    f_mp ? b->d_fMP : nullptr, th_low, nnratio, check_ori, b->d_match, b->d_counters, b->d_hist,
                                    b->d_entries, s));
    HIPCHK(c, hipStreamSynchronize(s));
    HIPCHK(c, hipMemcpy(f_match, b->d_match, sizeof(int) * n_f, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(nmatches, b->d_counters, sizeof(int), hipMemcpyDeviceToHost));
    return DRFE_OK;
}

int drfe_search_by_bow(drfe_ctx* c, int kf_slot, int f_slot, const int32_t* kf_mp, int n_kf, float nnratio, int check_ori,
                       int32_t* f_match, int n_f, int* nmatches)
{
    return search_by_bow_impl(c, kf_slot, f_slot, kf_mp, n_kf, nullptr, 50, nnratio, check_ori, f_match, n_f, nmatches);
}

/* ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12), src/ORBmatcher.cc:526-660 (LoopClosing::ComputeSim3): both sides need
 * a good map point, `bestDist1 < TH_LOW`.  match2[keypoint of KF2] = keypoint of KF1 (vpMatches12[idx1] = vpMapPoints2[idx2]). */
int drfe_search_by_bow_kf(drfe_ctx* c, int slot1, int slot2, const int32_t* mp1, int n1, const int32_t* mp2, int n2,
                          float nnratio, int check_ori, int32_t* match2, int* nmatches)
{
    if (!mp2) return DRFE_ERR_INVALID;
    return search_by_bow_impl(c, slot1, slot2, mp1, n1, mp2, 49, nnratio, check_ori, match2, n2, nmatches);
}


/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo), src/ORBmatcher.cc:661-827 */
int drfe_search_for_triangulation(drfe_ctx* c, int slot1, int slot2, const int32_t* mp1, int n1, const int32_t* mp2, int n2,
                                  const float* F12, const float* Cw1, const float* T2w, const drfe_camera* cam2,
                                  int only_stereo, int check_ori, int32_t* matches12, int* nmatches)
{
    if (!c || !mp1 || !mp2 || !F12 || !Cw1 || !T2w || !cam2 || !matches12 || !nmatches) return DRFE_ERR_INVALID;
    BowState* b = c->bow;
    if (!drfe_bow_slot_done(b, slot1) || !drfe_bow_slot_done(b, slot2) || !c->glueValid) {
        c->err = "search_for_triangulation: both slots need the glue and drfe_bow_transform_batch first";
        return DRFE_ERR_STATE;
    }
    HIPCHK(c, hipSetDevice(c->device));
    int rc = drfe_stream_sync(c);
    if (rc != DRFE_OK) return rc;
    int counts[2];
    HIPCHK(c, hipMemcpy(&counts[0], c->d_kpCount + slot1, sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(&counts[1], c->d_kpCount + slot2, sizeof(int), hipMemcpyDeviceToHost));
    if (n1 != counts[0] || n2 != counts[1]) { c->err = "search_for_triangulation: N mismatch"; return DRFE_ERR_INVALID; }
    *nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (n1 == 0 || n2 == 0) return DRFE_OK;
    /* epipole of KF1's centre in KF2, :667-674: C2 = R2w*Cw + t2w through cv::Mat's float small-matrix product */
    TriParams P;
    std::memset(&P, 0, sizeof(P));
    float C2[3];
    for (int r = 0; r < 3; r++) {
        const float d = T2w[r * 4 + 0] * Cw1[0] + T2w[r * 4 + 1] * Cw1[1] + T2w[r * 4 + 2] * Cw1[2];
        C2[r] = d + T2w[r * 4 + 3];
    }
    const float invz = 1.0f / C2[2];
    P.ex = cam2->fx * C2[0] * invz + cam2->cx;
    P.ey = cam2->fy * C2[1] * invz + cam2->cy;
    std::memcpy(P.F, F12, 36);
    if (c->cfg.nlevels > 16) { c->err = "search_for_triangulation: more than 16 pyramid levels"; return DRFE_ERR_INVALID; }
    for (int l = 0; l < c->cfg.nlevels; l++) { P.scale[l] = c->scale[l]; P.sigma2[l] = c->sigma2[l]; }
    P.onlyStereo = only_stereo ? 1 : 0;
    P.checkOri = check_ori ? 1 : 0;
    std::vector<int> nid1(n1), nid2(n2);
    std::vector<double> w1(n1), w2(n2);
    const size_t o1 = (size_t)slot1 * c->maxKp, o2 = (size_t)slot2 * c->maxKp;
    HIPCHK(c, hipMemcpy(nid1.data(), b->d_nid + o1, sizeof(int) * n1, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(w1.data(), b->d_weight + o1, sizeof(double) * n1, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(nid2.data(), b->d_nid + o2, sizeof(int) * n2, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(w2.data(), b->d_weight + o2, sizeof(double) * n2, hipMemcpyDeviceToHost));
    std::map<int, std::vector<int>> fv1, fv2;
    for (int i = 0; i < n1; i++) if (w1[i] > 0) fv1[nid1[i]].push_back(i);
    for (int i = 0; i < n2; i++) if (w2[i] > 0) fv2[nid2[i]].push_back(i);
    std::vector<BowGroup> groups;
    std::vector<int> idx1, idx2;
    for (auto& kv : fv1) {
        auto it = fv2.find(kv.first);
        if (it == fv2.end()) continue;
        BowGroup g;
        g.kfBegin = (int)idx1.size(); idx1.insert(idx1.end(), kv.second.begin(), kv.second.end()); g.kfEnd = (int)idx1.size();
        g.fBegin = (int)idx2.size(); idx2.insert(idx2.end(), it->second.begin(), it->second.end()); g.fEnd = (int)idx2.size();
        groups.push_back(g);
    }
    hipStream_t s = c->stream;
    if (!groups.empty()) {
        HIPCHK(c, hipMemcpy(b->d_groups, groups.data(), groups.size() * sizeof(BowGroup), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(b->d_kfIdx, idx1.data(), idx1.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(b->d_fIdx, idx2.data(), idx2.size() * 4, hipMemcpyHostToDevice));
    }
    HIPCHK(c, hipMemcpy(b->d_kfMP, mp1, sizeof(int) * n1, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(b->d_fMP, mp2, sizeof(int) * n2, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemsetAsync(b->d_match, 0xFF, sizeof(int) * n1, s));
    HIPCHK(c, hipMemsetAsync(b->d_counters, 0, 8, s));
    HIPCHK(c, hipMemsetAsync(b->d_hist, 0, 30 * 4, s));
    HIPCHK(c, drfe_launch_bow_triangulation(c, slot1, slot2, b->d_groups, (int)groups.size(), b->d_kfIdx, b->d_fIdx, b->d_kfMP,
                                            b->d_fMP, P, b->d_match, b->d_counters, b->d_hist, b->d_entries, s));
    HIPCHK(c, hipStreamSynchronize(s));
    HIPCHK(c, hipMemcpy(matches12, b->d_match, sizeof(int) * n1, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(nmatches, b->d_counters, sizeof(int), hipMemcpyDeviceToHost));
    return DRFE_OK;
}

} /* extern "C" */
