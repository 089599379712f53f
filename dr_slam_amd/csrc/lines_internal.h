/* lines_internal.h — scratch of the line-feature path */
#ifndef DRFE_LINES_INTERNAL_H
#define DRFE_LINES_INTERNAL_H
#include "drfe_internal.h"

struct LineTaps { int n; int t[9]; };   /* 8.8 fixed-point Gaussian taps */

struct LinesScratch {
    int w, h, sw, sh;               /* input and 0.8-scaled sizes */
    uint8_t* d_img; uint8_t* d_blur; uint8_t* d_scaled;
    uint16_t* d_tmp16;
    double* d_modgrad; double* d_angles;
    float2* d_cs;                   /* (cos, sin) of float(angle) per scaled pixel, 0 where the angle is undefined */
    unsigned long long* d_maxGrad;
    int16_t* d_gx; int16_t* d_gy;
    struct RectCand* d_cands; int2* d_counts; size_t candCap;   /* grow-only scratch of the NFA rounds */
};

/* one rectangle whose aligned pixels are to be counted: the fields cv::LineSegmentDetectorImpl::rect_nfa reads */
struct RectCand { double x1, y1, x2, y2, width, dx, dy, theta, prec; };

/* (pixels inside the rectangle, pixels among them aligned with theta up to prec) for n rectangles: the pixel loop of
 * rect_nfa, one wavefront per rectangle.  d_angles = the level-line angle field k_ll_angle left on the device. */
hipError_t drfe_launch_rect_counts(const RectCand* d_cands, int n, const double* d_angles, int W, int H, int2* d_counts,
                                   hipStream_t s);

hipError_t drfe_launch_lines_passes(const uint8_t* d_img, int w, int h, const LineTaps& lsdTaps, const LineTaps& lbdTaps,
                                    LinesScratch* sc, double threshold, hipStream_t s);
void drfe_lines_free(drfe_ctx* c);
#endif
