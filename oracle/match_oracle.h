/* oracle/match_oracle.h — TEST INFRASTRUCTURE (see oracle.h). Frame glue + matcher restatement. */
#ifndef DRFE_MATCH_ORACLE_H
#define DRFE_MATCH_ORACLE_H
#include "oracle.h"

namespace orc {

constexpr int kGridCols = 64; /* FRAME_GRID_COLS, include/Frame.h:40 */
constexpr int kGridRows = 48; /* FRAME_GRID_ROWS, include/Frame.h:39 */

/* what the matchers read from a MapPoint* reached through LastFrame.mvpMapPoints[i] */
struct MapPointRec {
    uint8_t valid;       /* pMP != NULL && !mvbOutlier[i] */
    uint8_t obsPositive; /* pMP->Observations() > 0 */
    uint8_t pad[2];
    float world[3];      /* GetWorldPos() */
    uint8_t desc[32];    /* GetDescriptor() */
};

/* what SearchByProjection(Frame&, vector<MapPoint*>) reads (fields set by Frame::isInFrustum) */
struct TrackedPointRec {
    uint8_t trackInView, bad, obsPositive, pad;
    int32_t level;      /* mnTrackScaleLevel */
    float projX, projY, projXR, viewCos;
    uint8_t desc[32];
};

struct Frame {
    int N = 0;
    std::vector<KeyPoint> keys, keysUn;
    std::vector<uint8_t> desc;
    std::vector<float> uRight, depth;
    std::vector<float> scaleFactors;
    float minX = 0, maxX = 0, minY = 0, maxY = 0, gridInvW = 0, gridInvH = 0;
    float fx = 0, fy = 0, cx = 0, cy = 0, bf = 0, mb = 0;
    std::vector<int> grid[kGridCols][kGridRows];

    void computeStereoFromRGBD(const float* depth, int dw, int dh);
    void assignFeaturesToGrid();
    void getFeaturesInArea(float x, float y, float r, int minLevel, int maxLevel, std::vector<int>& out) const;
};

/* cv::undistortPoints(src, dst, K, distCoef, Mat(), K) as Frame::UndistortKeyPoints / ComputeImageBounds call
 * it (src/Frame.cc:835-888): OpenCV 3.4 cvUndistortPointsInternal, 5 fixed-point iterations in double.
 * dist = (k1, k2, p1, p2[, k3]). */
void undistort_points(const float* xy, int n, const float K[4], const float* dist, int nd, float* out);
/* Frame::ComputeImageBounds: out = (mnMinX, mnMaxX, mnMinY, mnMaxY) */
void image_bounds(int cols, int rows, const float K[4], const float* dist, int nd, float out[4]);

int search_by_projection_last(const Frame& Cur, const Frame& Last, const float TcwCur[16], const float TcwLast[16],
                              const MapPointRec* lastMP, float th, bool bMono, bool checkOri,
                              const uint8_t* curClaimObsPositive, int* curMP);
int search_by_projection_map(const Frame& F, const TrackedPointRec* mps, int M, float th, float nnratio,
                             const uint8_t* claimObsPositive, int* frameMP);
void bf_knn_hamming(const uint8_t* Q, int nq, const uint8_t* T, int nt, int k, int32_t* idx, int32_t* dist);
int match_orb_points(const uint8_t* curDesc, int curN, const uint8_t* lastDesc, int lastN, const int32_t* lastMP,
                     const uint8_t* lastOutlier, int32_t* curMP);
void line_descriptor_mad(const int32_t* dist, int nq, double& nn_mad, double& nn12_mad);
int lsd_search_by_descriptor(const uint8_t* descKF, int nKF, const uint8_t* kfHasLine, const uint8_t* descF, int nF,
                             int32_t* out);
int lsd_search_by_gap(const uint8_t* descQ, int nQ, const uint8_t* descT, int nT, const uint8_t* trainHasLine, int32_t* out);
/* LSDmatcher::SearchForTriangulation(pKF1, pKF2, vMatchedPairs), src/LSDmatcher.cpp:334-367: out12[q] = t or -1 */
int lsd_search_for_triangulation(const uint8_t* desc1, int n1, const uint8_t* desc2, int n2, const uint8_t* has1,
                                 const uint8_t* has2, int32_t* out12);


/* --- LSDmatcher::SearchByProjection, src/LSDmatcher.cpp:20-211 --------------- */
/* the KeyLine fields Frame::GetLinesInArea and the matcher read (src/Frame.cc:781-813) */
struct LineRec { float ptX, ptY, angle; int32_t octave; };
/* what the (Frame, Frame) variant reads through LastFrame.mvpMapLines[i] / mvKeylinesUn[i] */
struct MapLineRec {
    int32_t valid;        /* pML && !pML->isBad() && !mvbLineOutlier[i] */
    int32_t octave;       /* LastFrame.mvKeylinesUn[i].octave */
    int32_t obsPositive;  /* pML->Observations() > 0 (a claim by it blocks later lines, :101-103) */
    int32_t pad;
    double world[6];      /* GetWorldPos(): start xyz, end xyz */
    uint8_t desc[32];
};
/* what the (Frame, vector<MapLine*>) variant reads (fields set by Frame::isInFrustum for lines) */
struct TrackedLineRec {
    int32_t inView;       /* pML && !isBad() && mbTrackInView */
    int32_t level;        /* mnTrackScaleLevel */
    int32_t obsPositive;
    float x1, y1, x2, y2; /* mTrackProjX1/Y1/X2/Y2 */
    float viewCos;
    uint8_t desc[32];
};
struct LineCamera { float fx, fy, cx, cy, mb, minX, maxX, minY, maxY; };

void get_lines_in_area(const LineRec* lines, int n, float x1, float y1, float x2, float y2, float r, int minLevel,
                       int maxLevel, std::vector<int>& out);
/* curML in/out: >= 0 = holds a map line (curObs[i] says whether Observations() > 0; NULL = all positive);
 * new matches are written as the index i of the matched record.  Returns nmatches. */
int lsd_search_by_projection_last(const LineCamera& cam, const float TcwCur[16], const float TcwLast[16],
                                  const float* scaleFactors, const MapLineRec* last, int nLast, const LineRec* cur,
                                  const uint8_t* curDesc, int nCur, float th, bool bMono, float nnratio,
                                  const uint8_t* curObs, int32_t* curML);
int lsd_search_by_projection_map(const float* scaleFactors, const TrackedLineRec* lines, int n, const LineRec* cur,
                                 const uint8_t* curDesc, int nCur, float th, float nnratio, const uint8_t* curObs,
                                 int32_t* curML);


/* --- Frame::isInFrustum, src/Frame.cc:602-657 (MapPoint) and :659-727 (MapLine) -------------------- */
struct FrustumPointRec { float world[3], normal[3], minDistance, maxDistance; };      /* mfMinDistance / mfMaxDistance raw */
struct FrustumLineRec { double world[6], normal[3]; float minDistance, maxDistance; };
struct FrustumOut { int32_t inView, level; float projX, projY, projXR, viewCos; };    /* mbTrackInView ... mTrackViewCos */
struct FrustumLineOut { int32_t inView, level; float x1, y1, x2, y2, viewCos; };
/* logScaleFactor = Frame::mfLogScaleFactor; nLevels = mnScaleLevels.  log() is the canonical log_f. */
void is_in_frustum(const LineCamera& cam, float bf, const float Tcw[16], float logScaleFactor, int nLevels,
                   const FrustumPointRec* pts, int n, float viewingCosLimit, FrustumOut* out);
void is_in_frustum_lines(const LineCamera& cam, const float Tcw[16], float logScaleFactor, const FrustumLineRec* lines,
                         int n, float viewingCosLimit, FrustumLineOut* out);


/* --- ORBmatcher::Fuse(KeyFrame*, vector<MapPoint*>, th), src/ORBmatcher.cc:829-985: the search part ------------ */
/* For every map point (skip[i] = !pMP || isBad() || IsInKeyFrame(pKF)): projection, KeyFrame::IsInImage, distance band,
 * 60-degree viewing cone, PredictScale, KeyFrame::GetFeaturesInArea(u, v, th * scale[level]) (no level filter, src/
 * KeyFrame.cc:707-746), octave in [level-1, level], chi-square reprojection gate (7.8 stereo / 5.99 mono), first
 * minimum of the Hamming distance.  bestIdx[i] = keyframe keypoint or -1, bestDist[i] = its distance (256 if none);
 * what the caller does with it (Replace / AddObservation when bestDist <= TH_LOW) touches the map graph and stays on
 * the host. */
void fuse_search(const Frame& KF, const float Tcw[16], const float* invLevelSigma2, float logScaleFactor, int nLevels,
                 const FrustumPointRec* pts, const uint8_t* descs, const uint8_t* skip, int n, float th, int32_t* bestIdx,
                 int32_t* bestDist, bool sim3 = false);
/* Fuse(KeyFrame*, cv::Mat Scw, points, th, vpReplacePoint), :981-1107 (LoopClosing::SearchAndFuse): the pose is the
 * similarity Scw decomposed as in :989-994 (scale = |first row|, Rcw = sRcw / s, tcw = t / s), there is no chi-square
 * gate, and invz is computed as 1.0 / z in double.  Tcw receives the decomposed [Rcw | tcw]. */
void decompose_sim3(const float Scw[16], float Tcw[16]);
/* LSDmatcher::Fuse(KeyFrame* pKF, const vector<MapLine*>& vpMapLines, th), src/LSDmatcher.cpp:884-1010 (LocalMapping::
 * SearchInNeighbors): the search per map line (projection of both end points, distance band, 60-degree cone, unclamped
 * PredictScale, KeyFrame::GetLinesInArea, octave window, first minimum of the LBD distance).  bestIdx = -1: nothing,
 * -2: the predicted level is outside the pyramid (the reference indexes mvScaleFactors out of bounds there).
 * The caller applies TH_LOW and does the Replace / AddObservation surgery. */
void lsd_fuse_search(const LineCamera& cam, const float Tcw[16], float logScaleFactor, const float* scaleFactors, int nLevels,
                     const FrustumLineRec* lines, const uint8_t* descs, const uint8_t* skip, int n, const LineRec* kf,
                     const uint8_t* kfDesc, int nKF, float th, int32_t* bestIdx, int32_t* bestDist);
/* ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, th), src/ORBmatcher.cc:294-407
 * (LoopClosing::ComputeSim3 after the Sim3 optimisation, th = 10): the points are visited in order, a keypoint whose
 * vpMatched entry is set (on entry, or by an earlier point of this call) is not a candidate, a point is accepted when its
 * best remaining candidate is within TH_LOW.  matched[idx] != 0 on entry = vpMatched[idx] != NULL; newMatch[idx] = the
 * point that claimed keypoint idx in this call or -1.  Returns nmatches. */
int search_by_projection_kf(const Frame& KF, const float Scw[16], float logScaleFactor, int nLevels, const FrustumPointRec* pts,
                            const uint8_t* descs, const uint8_t* skip, int n, const uint8_t* matched, float th, int32_t* newMatch);
/* ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th, ORBdist), src/ORBmatcher.cc:1537-1664
 * (Tracking::Relocalization).  pts / descs / kfAngles / skip per keyframe keypoint; matched[k] = CurrentFrame.mvpMapPoints[k]
 * != NULL; newMatch[k] = keyframe index or -1.  Returns nmatches. */
int search_by_projection_reloc(const Frame& Cur, const float Tcw[16], float logScaleFactor, int nLevels, const FrustumPointRec* pts,
                               const uint8_t* descs, const float* kfAngles, const uint8_t* skip, int n, const uint8_t* matched,
                               float th, int orbDist, bool checkOri, int32_t* newMatch);
/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th), src/ORBmatcher.cc:1106-1330 (LoopClosing::ComputeSim3):
 * every unmatched map point of KF1 is moved into KF2 with the similarity (and vice versa), searched in a th * scale
 * window (octave level-1..level, distance <= TH_HIGH, first minimum), and a pair is accepted when both directions agree.
 * pts / descs / skip are per keypoint of the keyframe (skip = !pMP || isBad() || already matched).  out12[i1] = i2 or -1. */
int search_by_sim3(const Frame& KF1, const Frame& KF2, const float T1w[16], const float T2w[16], float s12,
                   const float R12[9], const float t12[3], float logScaleFactor, int nLevels, const FrustumPointRec* pts1,
                   const uint8_t* descs1, const uint8_t* skip1, const FrustumPointRec* pts2, const uint8_t* descs2,
                   const uint8_t* skip2, float th, int32_t* out12);

/* ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:409-524: matches12[F1.N] (index into F2 or -1), prevMatched updated */
int search_for_initialization(const Frame& F1, const Frame& F2, float* prevMatched, int windowSize, float nnratio, bool checkOri,
                              int32_t* matches12);

/* LSDmatcher::Fuse(KF, Scw, ...) (search), SearchByProjection(KF, Scw, ...) and SearchBySim3, src/LSDmatcher.cpp:377-882
 * (public API without a caller in the reference) */
void lsd_fuse_search_sim3(const LineCamera& cam, const float Scw[16], float logScaleFactor, const float* scaleFactors, int nLevels,
                          const FrustumLineRec* lines, const uint8_t* descs, const uint8_t* skip, int n, const LineRec* kf,
                          const uint8_t* kfDesc, int nKF, float th, int32_t* bestIdx, int32_t* bestDist);
int lsd_search_by_projection_kf(const LineCamera& cam, const float Scw[16], float logScaleFactor, const float* scaleFactors, int nLevels,
                                const FrustumLineRec* lines, const uint8_t* descs, const uint8_t* skip, int n, const LineRec* kf,
                                const uint8_t* kfDesc, int nKF, const uint8_t* matched, int th, int32_t* newMatch);
int lsd_search_by_sim3(const LineCamera& cam, const float T1w[16], const float T2w[16], float s12, const float R12[9],
                       const float t12[3], float logScaleFactor, const float* scaleFactors, int nLevels, const FrustumLineRec* lines1,
                       const uint8_t* descs1, const uint8_t* skip1, const LineRec* kf1, const uint8_t* kf1Desc, int n1,
                       const FrustumLineRec* lines2, const uint8_t* descs2, const uint8_t* skip2, const LineRec* kf2,
                       const uint8_t* kf2Desc, int n2, float th, int32_t* out12);

} // namespace orc

#endif
