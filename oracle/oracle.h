/* oracle/oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of DR-SLAM's per-frame feature path in plain C++17, no OpenCV /
 * Eigen / PCL.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (dr_slam_amd/) never links, imports or falls back to it.
 *
 * PARITY UNPINNED at the OpenCV/Eigen boundary: the reference has no tests, golden vectors or
 * fixtures and cannot be built here (no OpenCV 3.4 / Eigen / PCL in the image), so
 * reference-owned logic follows the reference sources line by line (cited per function) and library
 * calls follow the OpenCV 3.4.4 semantics restated in SURVEY.md §10.  What IS pinned: the
 * known-answer tables of SURVEY.md §8 (level sizes, quotas, cell grids, umax) and independent
 * brute-force definitions (tests/test_oracle_*.py).
 */
#ifndef DRFE_ORACLE_H
#define DRFE_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include <vector>

namespace orc {

/* cv::KeyPoint layout (7 x 4 bytes). */
struct KeyPoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
};

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> px; /* w*h, stride == w */
    uint8_t at(int y, int x) const { return px[(size_t)y * w + x]; }
};

constexpr int kEdge = 19;      /* EDGE_THRESHOLD, src/ORBextractor.cc:72 */
constexpr int kHalfPatch = 15; /* HALF_PATCH_SIZE, :71 */
constexpr int kPatch = 31;     /* PATCH_SIZE, :70 */

struct LevelGeom {
    int w, h;           /* interior size */
    int quota;          /* mnFeaturesPerLevel */
    int minBX, minBY, maxBX, maxBY;
    int nCols, nRows, wCell, hCell;
};

struct Candidate { int x, y, response; }; /* coords relative to (minBorderX, minBorderY) */

class OrbExtractor {
public:
    OrbExtractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);

    /* ORBextractor::operator(), src/ORBextractor.cc:1043-1105. Returns N. */
    int extract(const uint8_t* gray, int w, int h, size_t stride);

    int nfeatures, nlevels, iniTh, minTh;
    double scaleFactorD; /* member is declared double, include/ORBextractor.h:98 */
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> quota;
    std::vector<int> umax;

    /* per-call state, kept for stage-by-stage parity checks */
    std::vector<LevelGeom> geom;
    std::vector<Image> pyramid;  /* bordered levels, (w+38)x(h+38) */
    std::vector<Image> blurred;  /* interior only, w x h (empty if level had no keypoints) */
    std::vector<std::vector<Candidate>> candidates; /* vToDistributeKeys per level, emission order */
    std::vector<KeyPoint> keypoints;
    std::vector<uint8_t> descriptors; /* N x 32 */

    void computeGeometry(int w, int h);
    void computePyramid(const uint8_t* gray, int w, int h, size_t stride);
    void computeCandidates(int level);
    std::vector<int> distributeOctTree(const std::vector<Candidate>& keys, int minX, int maxX, int minY,
                                       int maxY, int N) const;
    void blurLevel(int level);
};

/* stand-alone stage functions (unit-tested one by one) */
void resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, int dw, int dh,
                      size_t dstride);
int reflect101(int p, int n);
int fast_score_9_16(const uint8_t* p, size_t stride); /* cornerScore<16> with threshold 0 */
void fast_detect(const uint8_t* img, int w, int h, size_t stride, int threshold,
                 std::vector<Candidate>& out); /* cv::FAST(img, kps, threshold, true) */
void gaussian_blur_7x7_s2_u8(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride);
float ic_angle(const uint8_t* center, size_t stride, const std::vector<int>& umax);
void orb_descriptor(const uint8_t* center, size_t stride, float angle_deg, uint8_t* desc32);
int descriptor_distance_swar(const uint8_t* a, const uint8_t* b);

} // namespace orc

#endif
