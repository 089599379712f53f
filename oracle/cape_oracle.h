/* oracle/cape_oracle.h — TEST INFRASTRUCTURE (see oracle.h). CAPE plane extraction restatement. */
#ifndef DRFE_CAPE_ORACLE_H
#define DRFE_CAPE_ORACLE_H
#include <stdint.h>
#include <vector>

namespace orc {

struct CapeCell {   /* one PATCH x PATCH cell after the PlaneSeg constructor */
    int planar = 0, nr_pts = 0;
    double sums[9] = {0};   /* x y z xx yy zz xy xz yz */
    double mean[3] = {0}, normal[3] = {0}, d = 0;
    float MSE = 0, score = 0, tol = 0;
};
struct CapePlane {
    double normal[3], mean[3], d;
    float MSE, score;
    int nr_pts;
};
struct CapeResult {
    std::vector<CapePlane> planes;   /* plane_params */
    std::vector<uint8_t> seg;        /* seg_output */
    std::vector<CapeCell> cells;
};

CapeResult cape_run(const float* depth_m, int width, int height, const float K4[4], int patch, float cos_angle_max,
                    float max_merge_dist);

} // namespace orc
#endif
