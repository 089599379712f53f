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

int search_by_bow(const std::map<int, std::vector<unsigned>>& fvKF, const std::map<int, std::vector<unsigned>>& fvF,
                  const uint8_t* descKF, const float* angleKF, const int32_t* kfMP, const uint8_t* descF,
                  const float* angleF, int nF, float nnratio, bool checkOri, int32_t* out);

} // namespace orc
#endif
