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

} // namespace orc
#endif
