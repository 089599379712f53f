/* match_internal.h — device records shared by match_kernels.hip and capi_match.cpp */
#ifndef DRFE_MATCH_INTERNAL_H
#define DRFE_MATCH_INTERNAL_H
#include "drfe_internal.h"

/* one matcher invocation: queries (map points) against the keypoints of slot `curSlot` */
struct MatchPair {
    int curSlot, lastSlot;
    int mpSlot;      /* >= 0: one query per keypoint of lastSlot (count read on device); < 0: nQueries */
    int nQueries;
    int queryBase;   /* element offset into the query / candidate-count arrays */
    int mpBase;      /* element offset into the map-point array */
    int forward, backward;   /* bForward / bBackward, reference src/ORBmatcher.cc:1413-1414 */
    float Tcw[16];   /* CurrentFrame.mTcw */
};

/* one search window: what the candidate loop of the reference reads per map point */
struct MatchQuery {
    float u, v, radius;   /* GetFeaturesInArea(u, v, radius, minLevel, maxLevel) */
    float ur, thrR;       /* stereo gate: skip if |ur - mvuRight[i2]| > thrR when mvuRight[i2] > 0 */
    int minLevel, maxLevel;
    int valid;
    int obs;              /* Observations() > 0 of this map point (claims by it are permanent) */
    uint32_t desc[8];
};

struct MatchBuffers {
    MatchPair* d_pairs;
    MatchQuery* d_queries;
    drfe_map_point* d_mps;
    float* d_scale;           /* mvScaleFactors */
    uint32_t* d_candIdx;      /* [query][DRFE_MATCH_MAX_CAND]: keypoint index | octave << 24 */
    uint32_t* d_candKey;      /* distance << 22 | visit position */
    int* d_candCnt;
    uint2* d_candBest;        /* per query: (min key, its keypoint index | octave << 24) */
    uint16_t* d_hist;         /* rotation histogram entries per pair: (bin, idx) */
    uint8_t* d_initObs;       /* [maxKp] staging of caller-supplied claim flags */
    int32_t* d_bfIdx; int32_t* d_bfDist; uint8_t* d_bfQ; uint8_t* d_bfT; size_t bfCap; /* elements / bytes */
    size_t queryCap;          /* queries the buffers hold */
    /* pinned staging ring of drfe_match_consecutive_batch: pairs + poses of a batch travel from here, so the call needs
     * no stream synchronisation; an event per slot guards its reuse two calls later */
    void* h_stage[2]; hipEvent_t stageEv[2]; size_t stageBytes; int stageNext;
};

hipError_t drfe_launch_window_match(drfe_ctx* c, const MatchBuffers& mb, const drfe_camera& cam, int npairs,
                                    int maxQueries, int mode, float th, float nnratio, int checkOri,
                                    const uint8_t* d_initObs, hipStream_t s, int statusWord = 1);
hipError_t drfe_launch_fill_i32(int* d_p, int n, int v, hipStream_t s);
hipError_t drfe_launch_fill_i32_and_word(int* d_p, int n, int v, int* d_q, int w, hipStream_t s);
/* up to 10 dword ranges laid end to end (match_kernels.hip: k_pack_segments) */
struct DrfePackArgs { const uint32_t* src[10]; uint32_t dwords[10]; int n; };
hipError_t drfe_launch_pack_segments(const DrfePackArgs& A, uint32_t* d_dst, hipStream_t s);
/* bForward / bBackward of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame), src/ORBmatcher.cc:1406-1414 (capi_match.cpp) */
void drfe_motion_flags(const float* TcwCur, const float* TcwLast, float mb, int mono, int* fwd, int* bwd);
hipError_t drfe_launch_window_candidates(drfe_ctx* c, const MatchBuffers& mb, const drfe_camera& cam, int npairs, int maxQueries,
                                         hipStream_t s);
hipError_t drfe_launch_mappoints_last(drfe_ctx* c, const MatchBuffers& mb, const drfe_camera& cam, const float* d_Twc,
                                      int nframes, hipStream_t s);
hipError_t drfe_launch_bf_knn(const uint8_t* dQ, int nq, const uint8_t* dT, int nt, int k, int* dIdx, int* dDist,
                              hipStream_t s);

/* one GetLinesInArea + best/second scan of LSDmatcher::SearchByProjection (src/LSDmatcher.cpp:76-136) */
struct LineQuery {
    int valid, obs;               /* obs: Observations() > 0 of the map line (its claim blocks later lines) */
    int minLevel, maxLevel;
    float x1, y1, x2, y2, r;
    uint32_t desc[8];
};
struct LineCur { float ptX, ptY, angle; int octave; };   /* the KeyLine fields the search reads */

hipError_t drfe_launch_line_projection(const drfe_map_line* d_lines, int n, const float* d_TcwCur, const drfe_camera& cam,
                                       int forward, int backward, const float* d_scale, float th, LineQuery* d_q,
                                       hipStream_t s);
hipError_t drfe_launch_line_search(const LineQuery* d_q, int n, const LineCur* d_cur, const uint8_t* d_desc, int nCur,
                                   float nnratio, uint8_t* d_claim, int* d_curMl, int* d_nmatches, hipStream_t s);

struct FrustumPose { float T[16]; float Ow[3]; float bf; float logScale; int nLevels; float limit; };
hipError_t drfe_launch_frustum_points(const drfe_frustum_point* d_pts, int n, const FrustumPose& P, const drfe_camera& cam,
                                      drfe_tracked_point* d_out, hipStream_t s);
hipError_t drfe_launch_frustum_lines(const drfe_frustum_line* d_lines, int n, const FrustumPose& P, const drfe_camera& cam,
                                     drfe_tracked_line* d_out, hipStream_t s);

/* LSDmatcher::Fuse search: one wavefront per map line over the keyframe's key lines; bestIdx -2 = predicted level outside
 * the pyramid */
/* the similarity chained onto the source keyframe's pose in LSDmatcher::SearchBySim3: x2 = sR * x1 + t */
struct LineSim3 { float sR[9], t[3]; };
/* sim3 != NULL: the SearchBySim3 direction (P.T = pose of the source keyframe); d_distRow != NULL: [n][nKF] Hamming distance of
 * every key line that passed the gates, -1 otherwise (first-come replay on the host) */
hipError_t drfe_launch_line_fuse_search(const drfe_frustum_line* d_lines, const uint8_t* d_descs, const uint8_t* d_skip, int n,
                                        const FrustumPose& P, const drfe_camera& cam, const float* d_scale, float th,
                                        const LineCur* d_kf, const uint8_t* d_kfDesc, int nKF, int* d_bestIdx, int* d_bestDist,
                                        hipStream_t s, const LineSim3* sim3 = nullptr, int* d_distRow = nullptr);

struct FuseParams { float T[16]; float Ow[3]; float bf, logScale, th; int nLevels; float scale[16], invSigma2[16];
                    int sim3; /* 1: the Scw overload of Fuse: no chi-square gate, invz = (float)(1.0 / z);
                                 2: one direction of SearchBySim3: the point goes through T and then (sR2 | t2), the
                                    distance is the norm of the result, no viewing-cone test, no chi-square gate;
                                 3: SearchByProjection(KeyFrame*, Scw, ...): float 1/z, viewing cone kept, no chi-square
                                    gate; keypoints with taken[idx] != 0 are not candidates and the FUSE_LIST_K best
                                    candidates within listTh are listed in (distance, visit order);
                                 4: the relocalisation SearchByProjection(Frame&, KeyFrame*, ...): no depth test,
                                    double 1/z, inclusive image bounds, no viewing cone, octave window level-1..level+1,
                                    taken[] and candidate lists like mode 3 */
                    float sR2[9], t2[3];
                    int listTh; };
#define FUSE_LIST_K 8
/* d_taken / d_list / d_listCount are the mode-3 extras (nullptr otherwise): taken[keypoint], list[n][FUSE_LIST_K] =
 * (keypoint, distance), listCount[n] = how many candidates were within listTh (may exceed what the list holds) */
hipError_t drfe_launch_fuse_search(drfe_ctx* c, int slot, const drfe_frustum_point* d_pts, const uint8_t* d_descs,
                                   const uint8_t* d_skip, int n, const FuseParams& P, const drfe_camera& cam, int* d_bestIdx,
                                   int* d_bestDist, hipStream_t s, const uint8_t* d_taken = nullptr, int2* d_list = nullptr,
                                   int* d_listCount = nullptr);

MatchBuffers* drfe_match_buffers(drfe_ctx* c);   /* lazily allocated, owned by the context */
/* the two halves of drfe_match_consecutive_batch (capi_match.cpp): host staging + copies, then memsets + kernels (capturable) */
int drfe_match_consecutive_stage(drfe_ctx* c, const float* Tcw, const float* Twc, const drfe_camera* cam, int mono, int nframes, hipStream_t s);
hipError_t drfe_match_consecutive_enqueue(drfe_ctx* c, const drfe_camera* cam, float th, int check_ori, int nframes, hipStream_t s);
void drfe_match_buffers_free(drfe_ctx* c);
#endif
