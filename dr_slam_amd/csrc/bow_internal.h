/* bow_internal.h — device vocabulary and SearchByBoW records */
#ifndef DRFE_BOW_INTERNAL_H
#define DRFE_BOW_INTERNAL_H
#include "drfe_internal.h"

/* flattened DBoW2 vocabulary in HBM (k=10, L=6: 1 111 111 nodes ~ 50 MB) */
struct VocDev {
    int k, L, nNodes;
    const uint8_t* desc;      /* [nNodes][32] */
    const double* weight;     /* [nNodes] */
    const int* wordId;        /* [nNodes], -1 for inner nodes */
    const int* childBegin;    /* [nNodes+1] into children */
    const int* children;      /* child node ids in m_nodes[].children order */
};

struct BowGroup { int kfBegin, kfEnd, fBegin, fEnd; };   /* one vocabulary node common to both FeatureVectors */

struct BowState {
    VocDev voc;
    int scoring, weighting;
    void* d_vocBlob[5];       /* owning pointers of the five arrays */
    int* d_word; double* d_weight; int* d_nid;            /* [slot][maxKp] transform outputs */
    int levelsup;
    uint8_t* slotDone; int nSlots;                         /* [max_batch]: 1 = the slot's descriptors went through the transform (at `levelsup`) since they were last written */
    BowGroup* d_groups; int* d_kfIdx; int* d_fIdx; int* d_kfMP; int* d_fMP; int* d_match; int* d_counters; int* d_hist;
    uint16_t* d_entries;
};

hipError_t drfe_launch_bow_transform(drfe_ctx* c, const VocDev& voc, int levelsup, int nframes, int* d_word,
                                     double* d_weight, int* d_nid, hipStream_t s);
hipError_t drfe_launch_bow_match(drfe_ctx* c, int kfSlot, int fSlot, const BowGroup* d_groups, int ngroups,
                                 const int* d_kfIdx, const int* d_fIdx, const int* d_kfMP, const int* d_fMP, int thLow,
                                 float nnratio, int checkOri, int* d_match, int* d_counters, int* d_hist,
                                 uint16_t* d_entries, hipStream_t s);
struct TriParams { float F[9]; float ex, ey; float scale[16], sigma2[16]; int onlyStereo, checkOri; };
hipError_t drfe_launch_bow_triangulation(drfe_ctx* c, int slot1, int slot2, const BowGroup* d_groups, int ngroups,
                                         const int* d_idx1, const int* d_idx2, const int* d_mp1, const int* d_mp2,
                                         const TriParams& P, int* d_match12, int* d_counters, int* d_hist,
                                         uint16_t* d_entries, hipStream_t s);
void drfe_bow_free(drfe_ctx* c);
static inline bool drfe_bow_slot_done(const BowState* b, int slot) { return b && slot >= 0 && slot < b->nSlots && b->slotDone[slot]; }
#endif
