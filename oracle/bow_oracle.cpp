/* oracle/bow_oracle.cpp — TEST INFRASTRUCTURE (see oracle.h).
 *
 * CPU restatement of the bag-of-words step of the path, following the
 * vendored DBoW2 and the matcher line by line:
 *   TemplatedVocabulary::loadFromTextFile             Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1424
 *   TemplatedVocabulary::transform (feature / set)    :1127-1190, :1216-1259
 *   FORB::distance                                    Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101 (== popcount)
 *   BowVector::addWeight/addIfNotExist/normalize      Thirdparty/DBoW2/DBoW2/BowVector.cpp
 *   FeatureVector::addFeature                         Thirdparty/DBoW2/DBoW2/FeatureVector.cpp
 *   Frame::ComputeBoW (levelsup = 4)                  src/Frame.cc:828-833
 *   ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...)   src/ORBmatcher.cc:160-292
 * The ORB vocabulary blob is missing from the reference (.MISSING_LARGE_BLOBS), so vocabularies are
 * synthetic trees in the same node format.
 */
#include "bow_oracle.h"
#include "oracle.h"

#include <cmath>
#include <sstream>
#include <stdexcept>

namespace orc {

/* loadFromTextFile: header "k L scoring weighting", then one line per node:
 * "parent isLeaf d0 .. d31 weight"; node ids are line numbers (root = 0). */
void Vocabulary::loadFromText(const std::string& text)
{
    std::istringstream f(text);
    std::string s;
    std::getline(f, s);
    std::stringstream ss(s);
    int n1, n2;
    ss >> k >> L >> n1 >> n2;
    if (k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3)
        throw std::runtime_error("Vocabulary loading failure: This is not a correct text file!");
    scoring = n1; weighting = n2;
    nodes.clear();
    nodes.resize(1);
    int nwords = 0;
    while (std::getline(f, s)) {
        if (s.empty()) continue;
        std::stringstream sn(s);
        VocNode nd;
        int isLeaf;
        sn >> nd.parent >> isLeaf;
        for (int i = 0; i < 32; i++) { int v; sn >> v; nd.desc[i] = (uint8_t)v; }
        sn >> nd.weight;
        const int nid = (int)nodes.size();
        nd.word_id = isLeaf > 0 ? nwords++ : -1;
        nodes.push_back(nd);
        nodes[nd.parent].children.push_back(nid);
    }
}

/* transform(feature, word_id, weight, nid, levelsup), TemplatedVocabulary.h:1216-1259 */
void Vocabulary::transformOne(const uint8_t* feature, int levelsup, int& word_id, double& weight, int& nid) const
{
    const int nid_level = L - levelsup;
    nid = 0;   /* `if(nid_level <= 0 && nid != NULL) *nid = 0;` — and the caller's variable otherwise */
    int final_id = 0, current_level = 0;
    do {
        ++current_level;
        const std::vector<int>& ch = nodes[final_id].children;
        final_id = ch[0];
        double best_d = (double)descriptor_distance_swar(feature, nodes[final_id].desc);
        for (size_t c = 1; c < ch.size(); c++) {
            const double d = (double)descriptor_distance_swar(feature, nodes[ch[c]].desc);
            if (d < best_d) { best_d = d; final_id = ch[c]; }
        }
        if (current_level == nid_level) nid = final_id;
    } while (!nodes[final_id].children.empty());
    word_id = nodes[final_id].word_id;
    weight = nodes[final_id].weight;
}

/* transform(features, BowVector, FeatureVector, levelsup), :1127-1190 */
void Vocabulary::transform(const uint8_t* desc, int n, int levelsup, std::map<int, double>& bow,
                           std::map<int, std::vector<unsigned>>& fv) const
{
    bow.clear();
    fv.clear();
    if (nodes.size() <= 1) return;
    /* scoring 0 = L1_NORM, 1 = L2_NORM, 2 = CHI_SQUARE (L1), 3 = KL, 4 = BHATTACHARYYA, 5 = DOT_PRODUCT */
    const bool must = scoring == 0 || scoring == 1 || scoring == 2 || scoring == 3 || scoring == 4;
    const bool l1 = scoring != 1;
    const bool tf = weighting == 0 || weighting == 1;   /* TF_IDF = 0, TF = 1, IDF = 2, BINARY = 3 */
    for (int i = 0; i < n; i++) {
        int id, nid;
        double w;
        transformOne(desc + (size_t)i * 32, levelsup, id, w, nid);
        if (w > 0) {
            auto it = bow.lower_bound(id);
            if (tf) {
                if (it != bow.end() && !(id < it->first)) it->second += w;
                else bow.insert(it, {id, w});
            } else if (it == bow.end() || id < it->first) bow.insert(it, {id, w});
            fv[nid].push_back((unsigned)i);
        }
    }
    if (tf && !bow.empty() && !must) {
        const double nd = (double)bow.size();
        for (auto& kv : bow) kv.second /= nd;
    }
    if (must) {
        double norm = 0.0;
        if (l1) { for (auto& kv : bow) norm += std::fabs(kv.second); }
        else { for (auto& kv : bow) norm += kv.second * kv.second; norm = std::sqrt(norm); }
        if (norm > 0.0) for (auto& kv : bow) kv.second /= norm;
    }
}

/* ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches), src/ORBmatcher.cc:160-292.
 * kfMP[i] >= 0: pKF has a good (non-bad) map point at keypoint i.  out[f] = KF keypoint index or -1. */
int search_by_bow(const std::map<int, std::vector<unsigned>>& fvKF, const std::map<int, std::vector<unsigned>>& fvF,
                  const uint8_t* descKF, const float* angleKF, const int32_t* kfMP, const uint8_t* descF,
                  const float* angleF, int nF, float nnratio, bool checkOri, int32_t* out, const int32_t* fMP, bool strictLow)
{
    const int TH_LOW = 50, HISTO_LENGTH = 30;
    for (int i = 0; i < nF; i++) out[i] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    auto KFit = fvKF.begin(), KFend = fvKF.end();
    auto Fit = fvF.begin(), Fend = fvF.end();
    while (KFit != KFend && Fit != Fend) {
        if (KFit->first == Fit->first) {
            const std::vector<unsigned>& iKFs = KFit->second;
            const std::vector<unsigned>& iFs = Fit->second;
            for (size_t a = 0; a < iKFs.size(); a++) {
                const unsigned realIdxKF = iKFs[a];
                if (kfMP[realIdxKF] < 0) continue;
                const uint8_t* dKF = descKF + (size_t)realIdxKF * 32;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (size_t b = 0; b < iFs.size(); b++) {
                    const unsigned realIdxF = iFs[b];
                    if (out[realIdxF] >= 0) continue;
                    if (fMP && fMP[realIdxF] < 0) continue;          /* KF-KF overload: !pMP2 || pMP2->isBad() */
                    const int dist = descriptor_distance_swar(dKF, descF + (size_t)realIdxF * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = (int)realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (strictLow ? bestDist1 < TH_LOW : bestDist1 <= TH_LOW) {
                    if ((float)bestDist1 < nnratio * (float)bestDist2) {
                        out[bestIdxF] = (int32_t)realIdxKF;
                        if (checkOri) {
                            float rot = angleKF[realIdxKF] - angleF[bestIdxF];
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)std::round(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back(bestIdxF);
                        }
                        nmatches++;
                    }
                }
            }
            ++KFit;
            ++Fit;
        } else if (KFit->first < Fit->first) KFit = fvKF.lower_bound(Fit->first);
        else Fit = fvF.lower_bound(KFit->first);
    }
    if (checkOri) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int s = (int)rotHist[i].size();
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : rotHist[i]) { out[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}


/* ORBmatcher::CheckDistEpipolarLine, src/ORBmatcher.cc:141-158 (float32 throughout; the threshold compares in double) */
static bool check_dist_epipolar_line(float x1, float y1, float x2, float y2, const float F[9], float sigma2)
{
    const float a = x1 * F[0] + y1 * F[3] + F[6];
    const float b = x1 * F[1] + y1 * F[4] + F[7];
    const float c = x1 * F[2] + y1 * F[5] + F[8];
    const float num = a * x2 + b * y2 + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return (double)dsqr < 3.84 * (double)sigma2;
}

int search_for_triangulation(const TriKeyFrame& k1, const TriKeyFrame& k2, const float F12[9], float ex, float ey,
                             const float* scaleFactors, const float* levelSigma2, bool onlyStereo, bool checkOri,
                             int32_t* out12)
{
    const int TH_LOW = 50, HISTO_LENGTH = 30;
    int nmatches = 0;
    for (int i = 0; i < k1.N; i++) out12[i] = -1;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    auto f1it = k1.fv->begin(), f1end = k1.fv->end();
    auto f2it = k2.fv->begin(), f2end = k2.fv->end();
    while (f1it != f1end && f2it != f2end) {
        if (f1it->first == f2it->first) {
            for (unsigned idx1 : f1it->second) {
                if (k1.mp[idx1] >= 0) continue;                    /* already a MapPoint */
                const bool bStereo1 = k1.uRight[idx1] >= 0;
                if (onlyStereo && !bStereo1) continue;
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (unsigned idx2 : f2it->second) {
                    if (k2.mp[idx2] >= 0) continue;                /* vbMatched2 is never set in this reference */
                    const bool bStereo2 = k2.uRight[idx2] >= 0;
                    if (onlyStereo && !bStereo2) continue;
                    const int dist = descriptor_distance_swar(k1.desc + (size_t)idx1 * 32, k2.desc + (size_t)idx2 * 32);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ex - k2.x[idx2], distey = ey - k2.y[idx2];
                        if (distex * distex + distey * distey < 100 * scaleFactors[k2.octave[idx2]]) continue;
                    }
                    if (check_dist_epipolar_line(k1.x[idx1], k1.y[idx1], k2.x[idx2], k2.y[idx2], F12, levelSigma2[k2.octave[idx2]])) {
                        bestIdx2 = (int)idx2;
                        bestDist = dist;
                    }
                }
                if (bestIdx2 >= 0) {
                    out12[idx1] = bestIdx2;
                    nmatches++;
                    if (checkOri) {
                        float rot = k1.angle[idx1] - k2.angle[bestIdx2];
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)std::round(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back((int)idx1);
                    }
                }
            }
            ++f1it;
            ++f2it;
        } else if (f1it->first < f2it->first) f1it = k1.fv->lower_bound(f2it->first);
        else f2it = k2.fv->lower_bound(f1it->first);
    }
    if (checkOri) {
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) {
            const int s = (int)rotHist[i].size();
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) ind3 = -1;
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx : rotHist[i]) { out12[idx] = -1; nmatches--; }
        }
    }
    return nmatches;
}

} // namespace orc
