/* oracle/ref_shim.cpp — TEST INFRASTRUCTURE.  C entry points over the pieces of the REFERENCE that compile here from their own
 * sources, where they lie under /root/reference (nothing is copied; oracle/Makefile's `ref` target is the recipe, the output
 * goes to oracle/_ref/):
 *   Thirdparty/DBoW2/DBoW2/BowVector.cpp, FeatureVector.cpp   BowVector::addWeight / addIfNotExist / normalize,
 *                                                             FeatureVector::addFeature (the containers Frame::ComputeBoW fills)
 *   include/peac/AHCParamSet.hpp                              ahc::ParamSet::T_mse / T_ang / T_dz (PlaneFitter thresholds)
 *   include/peac/DisjointSet.hpp                              the block-membership union-find of ahCluster
 * Everything else on the hot path needs OpenCV 3.4 / Eigen / PCL and cannot be built in this image.
 * tests/test_ref_pins.py compares the oracle's restatements with these, bit for bit. */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "BowVector.h"          /* -I/root/reference/Thirdparty/DBoW2/DBoW2 */
#include "FeatureVector.h"
#include "AHCParamSet.hpp"      /* -I/root/reference/include/peac */
#include "DisjointSet.hpp"

extern "C" {

/* The loop of TemplatedVocabulary::transform(features, v, fv, levelsup) (TemplatedVocabulary.h:1127-1190: needs OpenCV, so
 * the loop is restated here) over per-feature (word id, weight, node id) triples, on the reference's own containers.
 * weighting: TF_IDF 0, TF 1, IDF 2, BINARY 3; scoring: L1_NORM 0, L2_NORM 1, CHI_SQUARE 2, KL 3, BHATTACHARYYA 4, DOT_PRODUCT 5
 * (ScoringObject.cpp: mustNormalize is L1 for 0, 2, 3, 4, L2 for 1, false for 5).
 * Outputs: the BowVector in map order (ids, values, *n_words) and the FeatureVector flattened in map order
 * (fv_nodes[*n_nodes], fv_counts[*n_nodes], fv_features = the feature lists back to back). */
int ref_bow_containers(const int32_t* word, const double* weight, const int32_t* node, int n, int weighting, int scoring,
                       int32_t* ids, double* values, int* n_words, int32_t* fv_nodes, int32_t* fv_counts, int32_t* fv_features,
                       int* n_nodes)
{
    DBoW2::BowVector v;
    DBoW2::FeatureVector fv;
    const bool must = scoring != 5;
    const DBoW2::LNorm norm = scoring == 1 ? DBoW2::L2 : DBoW2::L1;
    if (weighting == DBoW2::TF || weighting == DBoW2::TF_IDF) {
        for (int i = 0; i < n; i++)
            if (weight[i] > 0) { v.addWeight((DBoW2::WordId)word[i], weight[i]); fv.addFeature((DBoW2::NodeId)node[i], (unsigned)i); }
        if (!v.empty() && !must) {
            const double nd = v.size();
            for (DBoW2::BowVector::iterator vit = v.begin(); vit != v.end(); vit++) vit->second /= nd;
        }
    } else {
        for (int i = 0; i < n; i++)
            if (weight[i] > 0) { v.addIfNotExist((DBoW2::WordId)word[i], weight[i]); fv.addFeature((DBoW2::NodeId)node[i], (unsigned)i); }
    }
    if (must) v.normalize(norm);
    int k = 0;
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++k) { ids[k] = (int32_t)it->first; values[k] = it->second; }
    *n_words = k;
    int m = 0, f = 0;
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it, ++m) {
        fv_nodes[m] = (int32_t)it->first;
        fv_counts[m] = (int32_t)it->second.size();
        for (size_t j = 0; j < it->second.size(); j++) fv_features[f++] = (int32_t)it->second[j];
    }
    *n_nodes = m;
    return 0;
}

/* ahc::ParamSet as PlaneFitter constructs it (DR-SLAM never changes a field: src/PlaneExtractor.cpp) */
void ref_ahc_thresholds(int phase, double z, double* out3)
{
    const ahc::ParamSet p;
    out3[0] = p.T_mse((ahc::ParamSet::Phase)phase, z);
    out3[1] = p.T_ang((ahc::ParamSet::Phase)phase, z);
    out3[2] = p.T_dz(z);
}

void ref_ahc_disjoint_set(int n, const int32_t* pairs, int npairs, int32_t* unionRet, int32_t* findOut, int32_t* sizeOut)
{
    DisjointSet ds(n);
    for (int i = 0; i < npairs; i++) unionRet[i] = ds.Union(pairs[2 * i], pairs[2 * i + 1]);
    for (int i = 0; i < n; i++) { findOut[i] = ds.Find(i); sizeOut[i] = ds.getSetSize(i); }
}

} /* extern "C" */
