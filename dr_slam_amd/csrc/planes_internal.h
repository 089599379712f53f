/* planes_internal.h — records shared by planes_kernels.hip and planes_ahc.cpp */
#ifndef DRFE_PLANES_INTERNAL_H
#define DRFE_PLANES_INTERNAL_H
#include "drfe_internal.h"

#define AHC_WIN 10            /* windowWidth == windowHeight, AHCPlaneFitter.hpp:155 */
#define AHC_MIN_SUPPORT 3000  /* minSupport, :154 */

struct AhcBlockRec {          /* result of one PlaneSeg init block */
    double sums[9];           /* sx sy sz sxx syy szz sxy syz sxz */
    double center[3], normal[3], mse, curvature;
    int valid;                /* window valid AND mse < T_mse(P_INIT): enters the graph */
    int N;                    /* 100 for a valid window, else 0 */
};

struct CapeCellRec {          /* CAPE PlaneSeg of one PATCH x PATCH cell (src/CAPE/PlaneSeg.cpp:8-94) */
    double acc[9];            /* x y z xx yy zz xy xz yz (float32 sums widened) */
    double mean[3], normal[3], d;
    float MSE, score, tol;    /* tol = cell_distance_tols[cell] (src/CAPE/CAPE.cpp:73) */
    int planar, nr_pts;
};

struct PlanesScratch {
    AhcBlockRec* d_blocks; size_t blocksCap;   /* device, [slot][Nw*Nh] */
    uint16_t* d_depth; size_t depthCap;        /* staging for the host-buffer API */
};

hipError_t drfe_launch_ahc_blocks(const uint16_t* d_depth, size_t frameStride, size_t rowStride, int w, int h,
                                  const float K4[4], float depthFactor, int nframes, AhcBlockRec* d_out, hipStream_t s);
void drfe_planes_free(drfe_ctx* c);
#endif
