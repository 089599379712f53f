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
    struct LbdLine* d_lbdLines; uint8_t* d_lbdOut; size_t lbdCap;
};

/* one rectangle whose aligned pixels are to be counted: the fields cv::LineSegmentDetectorImpl::rect_nfa reads */
struct RectCand { double x1, y1, x2, y2, width, dx, dy, theta, prec; };

/* (pixels inside the rectangle, pixels among them aligned with theta up to prec) for n rectangles: the pixel loop of
 * rect_nfa, one wavefront per rectangle.  d_angles = the level-line angle field k_ll_angle left on the device. */
hipError_t drfe_launch_rect_counts(const RectCand* d_cands, int n, const double* d_angles, int W, int H, int2* d_counts,
                                   hipStream_t s);

/* one key line as BinaryDescriptor::computeLBD reads it (octave 0); dL = (cos, sin) of the line angle from the host's libm */
struct LbdLine { float midX, midY, dL0, dL1; int len, pad; };
struct LbdTables { float coefG[63], coefL[21]; };      /* the Gaussian band weights, float(double exp(..)) */
/* LBD descriptors (32 bytes each) of n lines from the Sobel images of the LBD input: one wavefront per line, lane = one of
 * the 63 band rows for the row sums, lane = (moment, band) for the band accumulation; every float sum in the reference's order */
hipError_t drfe_launch_lbd(const LbdLine* d_lines, int n, const int16_t* d_gx, const int16_t* d_gy, int w, int h,
                           const LbdTables& tab, uint8_t* d_out, hipStream_t s);

hipError_t drfe_launch_lines_passes(const uint8_t* d_img, int w, int h, const LineTaps& lsdTaps, const LineTaps& lbdTaps,
                                    LinesScratch* sc, double threshold, hipStream_t s);
void drfe_lines_free(drfe_ctx* c);
#endif
