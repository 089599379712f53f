/* oracle/ahc_oracle.h — TEST INFRASTRUCTURE (see oracle.h). AHC plane extraction restatement. */
#ifndef DRFE_AHC_ORACLE_H
#define DRFE_AHC_ORACLE_H
#include <stdint.h>
#include <vector>

namespace orc {

struct AhcBlock {   /* one 10x10 init block: PlaneSeg ctor result + whether it entered the graph */
    int valid = 0, N = 0;
    double sums[9] = {0};   /* sx sy sz sxx syy szz sxy syz sxz */
    double center[3] = {0}, normal[3] = {0}, mse = 0, curvature = 0;
};
struct AhcPlane {
    double normal[3], center[3], mse, curvature;
    int N, rid;
};
struct AhcResult {
    std::vector<AhcPlane> planes;                 /* plane_filter.extractedPlanes (sorted by N desc) */
    std::vector<std::vector<int>> membership;     /* plane_vertices_ */
    std::vector<uint8_t> seg;                     /* seg_output: plid+1, 0 = none */
};

void eig33sym(const double K[3][3], double s[3], double V[3][3]);
AhcResult ahc_run(const uint16_t* depth, int w, int h, const float K4[4], float depthfactor,
                  std::vector<AhcBlock>* blocksOut = nullptr);

void ahc_thresholds(int phase, double z, double out3[3]);      /* T_mse, T_ang, T_dz of the default ParamSet */
void ahc_disjoint_set(int n, const int32_t* pairs, int npairs, int32_t* unionRet, int32_t* findOut, int32_t* sizeOut);

} // namespace orc
#endif
