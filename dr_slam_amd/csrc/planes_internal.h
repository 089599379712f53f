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

/* device scratch of drfe_planes_cape, kept by the context between frames (grow-only) */
struct CapeScratch {
    float* d_depth; size_t depthCap;           /* w*h metres */
    CapeCellRec* d_cells; size_t cellCap;
    uint8_t* d_seg; size_t segCap;             /* w*h labels */
    uint8_t* d_tab; size_t tabCap;             /* CapeRefinePlane[n] | gridEroded[ncell] | boundary[n][ncell] */
    float* h_depth; CapeCellRec* h_cells; uint8_t* h_seg;   /* pinned mirrors (sized with the device buffers): the copy calls neither stage nor pin */
};

/* ---- CAPE's cell stage on the device (cape_frame_kernels.hip): histogram seeding, cell growing, merging, masks ---- */
#define CAPE_DEV_MAXP 64             /* plane segments / final planes per frame the device path holds */
#define CAPE_DEV_MAXCELLS 3072       /* 64 x 48 cells */
#define CAPE_STATUS_UNCERTAIN 1      /* a histogram bin the device's acos / atan2 could not certify */
#define CAPE_STATUS_CAPACITY 2       /* more segments than CAPE_DEV_MAXP, or an iteration bound hit */
struct CapeFrameOut { int nPlanes, status, pad0, pad1; };
/* bytes of one frame's table block: CapeRefinePlane[CAPE_DEV_MAXP] | gridEroded[ncell] | boundary[CAPE_DEV_MAXP][ncell] */
static inline size_t drfe_cape_tab_bytes(int ncell) { return ((size_t)CAPE_DEV_MAXP * 20 + (size_t)ncell * (1 + CAPE_DEV_MAXP) + 255) & ~(size_t)255; }
struct drfe_cape_plane;
hipError_t drfe_launch_cape_cells_batch(const float* d_depth, size_t frameStride, size_t rowStride, int w, int h, const float K4[4], int patch,
                                        float sinCos, float maxMergeDist, int nframes, CapeCellRec* d_out, hipStream_t s);
/* CAPE::process between the cell fits and the per-pixel refinement (src/CAPE/CAPE.cpp:81-293) for nframes frames, one
 * wavefront each: planes of frame f at d_planes[f * CAPE_DEV_MAXP ..], its refinement tables at d_tabs + f * tabStride,
 * d_out[f] = {planes, status} */
hipError_t drfe_launch_cape_frames(const CapeCellRec* d_cells, int nh, int nv, float cosAngleMax, float maxMergeDist, int nframes,
                                   drfe_cape_plane* d_planes, uint8_t* d_tabs, size_t tabStride, CapeFrameOut* d_out, hipStream_t s);
hipError_t drfe_launch_cape_refine_batch(const float* d_depth, size_t frameStride, size_t rowStride, int w, int h, const float K4[4], int patch,
                                         const uint8_t* d_tabs, size_t tabStride, const CapeFrameOut* d_frameOut, int nframes, uint8_t* d_seg,
                                         hipStream_t s);

/* one final plane of the boundary refinement (src/CAPE/CAPE.cpp:294-319): float copies of normal and d, 9 * MSE */
struct CapeRefinePlane { float nx, ny, nz, d, maxDist; };

/* seg_output of CAPE: cells of a plane's eroded mask take its number; every pixel of a boundary cell (dilated minus eroded
 * mask of plane p) goes to the plane with the least squared distance below 9 * MSE, planes in extraction order, strict <
 * (first plane wins ties).  One thread per pixel. */
hipError_t drfe_launch_cape_refine(const float* d_depth, size_t rowStride, int w, int h, const float K4[4], int patch,
                                   const CapeRefinePlane* d_planes, int nplanes, const uint8_t* d_gridEroded,
                                   const uint8_t* d_boundary, uint8_t* d_seg, hipStream_t s);

/* ---- the whole extractor on the device (ahc_frame_kernels.hip) ---- */
struct AhcDevParams {
    int w, h, Nw, Nh, NB;
    int maxNodes, poolCap, rfCap, planeCap;
    double fx, fy, cx, cy, factor;       /* K floats promoted, depth factor: PlaneDetection::readDepthImage */
    double cos60, cos30;                 /* similarityTh_merge / _refine, from the host's libm as the host path's constants */
    float maxPointDist;                  /* Frame::ComputePlanes' gather: points beyond it stay out of the plane's cloud */
};
#define AHC_HANDOFF_INTS 272
/* one frame of a drfe_launch_ahc_frames launch (k_ahc_cluster, then k_ahc_refine): inputs, per-frame scratch, outputs */
struct AhcDevFrame {
    const uint16_t* depth; size_t rowStride;                 /* device CV_16U image */
    const AhcBlockRec* blocks;                               /* k_ahc_blocks' records of this frame */
    double* nodeS; double* nodeFit; int* nodeN; int* nodeRid; uint8_t* nodeNouse; int* nbOff; int* nbLen; int* nbPool;
    int* dsParent; int* dsSize; int* G; int* blkMap; int* ridToPlid;
    int16_t* membership; float* distMap; uint32_t* rf;
    int* handoff;                                            /* AHC_HANDOFF_INTS words from k_ahc_cluster to k_ahc_refine */
    drfe_plane* planes; uint8_t* seg; int* memberOff; int* memberIdx;      /* final planes, label image, member lists */
    float* pts; int2* jobs; int ptsBase;                     /* plane clouds for k_voxel_grid: pts = arena cloud + 3 * ptsBase, jobs[planeCap] (or null) */
    int* out;                                                /* out[0] = planes, out[1] = status (0 = done; else: redo on the host) */
};
/* ev3 (optional): events recorded after k_ahc_cluster, after k_ahc_refine and after the three k_ahc_labels_* kernels */
hipError_t drfe_launch_ahc_frames(const AhcDevFrame* d_frames, int nframes, const AhcDevParams& P, hipStream_t s, hipEvent_t* ev3 = nullptr);
/* frames the device extractor takes: up to 12 800 init blocks and 2^21 pixels (1280 x 960 = BASELINE config 5 is 12 288 / 1 228 800) */
int drfe_ahc_device_fits(int w, int h);

hipError_t drfe_launch_ahc_blocks(const uint16_t* d_depth, size_t frameStride, size_t rowStride, int w, int h,
                                  const float K4[4], float depthFactor, int nframes, AhcBlockRec* d_out, hipStream_t s);
void drfe_planes_free(drfe_ctx* c);
void drfe_ahc_arena_free(drfe_ctx* c);        /* planes_ahc.cpp: frame slots of the device extractor */
#endif
