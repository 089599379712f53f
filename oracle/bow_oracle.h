/* oracle/bow_oracle.h — TEST INFRASTRUCTURE (see oracle.h). DBoW2 vocabulary / transform / SearchByBoW. */
#ifndef DRFE_BOW_ORACLE_H
#define DRFE_BOW_ORACLE_H
#include <map>
#include <stdint.h>
#include <string>
#include <vector>

namespace orc {

struct VocNode {
    int parent = 0, word_id = -1;
    uint8_t desc[32] = {0};
    double weight = 0;
    std::vector<int> children;
};

struct Vocabulary {
    int k = 0, L = 0, scoring = 0, weighting = 0;
    std::vector<VocNode> nodes;
    void loadFromText(const std::string& text);
    void transformOne(const uint8_t* feature, int levelsup, int& word_id, double& weight, int& nid) const;
    void transform(const uint8_t* desc, int n, int levelsup, std::map<int, double>& bow,
                   std::map<int, std::vector<unsigned>>& fv) const;
};

/* fMP / strictLow select the keyframe-keyframe overload (src/ORBmatcher.cc:526-660, LoopClosing::ComputeSim3): only
 * keypoints of the second keyframe that HAVE a good map point take part (fMP[i] >= 0) and the distance test is
 * `bestDist1 < TH_LOW` instead of `<=`. */
int search_by_bow(const std::map<int, std::vector<unsigned>>& fvKF, const std::map<int, std::vector<unsigned>>& fvF,
                  const uint8_t* descKF, const float* angleKF, const int32_t* kfMP, const uint8_t* descF,
                  const float* angleF, int nF, float nnratio, bool checkOri, int32_t* out, const int32_t* fMP = nullptr,
                  bool strictLow = false);

/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo), src/ORBmatcher.cc:661-827, with
 * CheckDistEpipolarLine (:141-158).  Per-keypoint inputs of both keyframes: undistorted position (x, y), octave,
 * angle, uRight (>= 0 = stereo), map-point flag (>= 0 = has one).  epipole = projection of KF1's centre into KF2
 * (:667-674).  out12[i1] = matched keypoint of KF2 or -1.  Note: this reference never sets vbMatched2, so keypoints
 * of KF2 can be matched more than once and the loop has no order dependence between KF1 keypoints. */
struct TriKeyFrame {
    const std::map<int, std::vector<unsigned>>* fv;
    const float *x, *y, *angle, *uRight;
    const int32_t *octave, *mp;
    const uint8_t* desc;
    int N;
};
int search_for_triangulation(const TriKeyFrame& k1, const TriKeyFrame& k2, const float F12[9], float ex, float ey,
                             const float* scaleFactors, const float* levelSigma2, bool onlyStereo, bool checkOri,
                             int32_t* out12);

} // namespace orc
#endif
