/* lines_internal.h — scratch of the line-feature path */
#ifndef DRFE_LINES_INTERNAL_H
#define DRFE_LINES_INTERNAL_H
#include "drfe_internal.h"

struct LineTaps { int n; int t[9]; };   /* 8.8 fixed-point Gaussian taps */

/* device arrays of `frames` consecutive frame slots (dense: slot f of an array starts at f x the per-frame element count) */
struct LinesScratch {
    int w, h, sw, sh;               /* input and 0.8-scaled sizes */
    int frames;                     /* frame slots (1 for the single-frame entry) */
    uint8_t* d_img; uint8_t* d_blur; uint8_t* d_scaled;
    uint16_t* d_tmp16;
    double* d_modgrad; double* d_angles;
    float2* d_cs;                   /* (cos, sin) of float(angle) per scaled pixel, 0 where the angle is undefined */
    float2* d_cs0;                  /* batch arena: float(cos(angle)), float(sin(angle)) per scaled pixel - a seed's direction (k_lsd_keys) */
    unsigned long long* d_meta;     /* per slot two words: [0] bits of the largest gradient magnitude, [1] smallest gradient bin of a pixel with an angle */
    int16_t* d_gx; int16_t* d_gy;
    struct RectCand* d_cands; int2* d_counts; size_t candCap;   /* grow-only scratch of the NFA rounds (one lane's) */
    struct RectCand* h_cands; int2* h_counts;                   /* their pinned mirrors: no staging / pinning inside the copy calls */
    struct LbdLine* d_lbdLines; uint8_t* d_lbdOut; size_t lbdCap;
    /* device region growing (batch entry): per slot the ordering keys / sorted ordering, member list, shrink scratch,
     * accepted rectangles, (count, status); the launch's frame table; pinned host mirrors */
    uint32_t* d_order; uint32_t* d_reg; uint32_t* d_tmp; struct LsdRect* d_rects; int* d_out; struct LsdGrowFrame* d_frames;
    uint32_t* d_notdef;             /* one bit per scaled pixel: no level-line angle (k_lsd_notdef): the start of the growth kernels' `used` map */
    uint32_t* d_regMw; uint32_t* d_tmpMw; uint32_t* d_gbmMw; int regCapMw;      /* k_lsd_grow_mw: member list / shrink scratch of each of a frame's four wavefronts (regCapMw entries each) */
    int* d_ordStatus; int* h_ordStatus;   /* k_lsd_order's status word per slot */
    uint32_t* h_order; unsigned long long* h_meta; struct LsdRect* h_rects; int* h_out; struct LsdGrowFrame* h_frames;
    int rectCap;
    /* rect_improve + NFA on the device: validated segments per rectangle slot, pinned mirror, the host-filled log-gamma table */
    struct LsdSegOut* d_segs; struct LsdSegOut* h_segs; double* d_lgamma; int lgammaN;
    /* key lines on the device (k_lsd_keylines + k_lbd batch form): klCap slots per frame; pinned mirrors of what the caller receives */
    int klCap;
    drfe_keyline* d_kl; double* d_klLineF; struct LbdLine* d_klLbd; uint8_t* d_klDesc; int* d_klOut;
    drfe_keyline* h_kl; double* h_klLineF; uint8_t* h_klDesc; int* h_klOut;
};

/* one rectangle whose aligned pixels are to be counted: the fields cv::LineSegmentDetectorImpl::rect_nfa reads */
struct RectCand { double x1, y1, x2, y2, width, dx, dy, theta, prec; };

/* (pixels inside the rectangle, pixels among them aligned with theta up to prec) for n rectangles: the pixel loop of
 * rect_nfa, one wavefront per rectangle.  d_angles = the level-line angle field k_ll_angle left on the device. */
hipError_t drfe_launch_rect_counts(const RectCand* d_cands, int n, const double* d_angles, int W, int H, int rectMode,
                                   int2* d_counts, hipStream_t s);

/* one key line as BinaryDescriptor::computeLBD reads it (octave 0); dL = (cos, sin) of the line angle from the host's libm */
struct LbdLine { float midX, midY, dL0, dL1; int len, pad; };
struct LbdTables { float coefG[63], coefL[21]; };      /* the Gaussian band weights, float(double exp(..)) */
/* LBD descriptors (32 bytes each) of n lines from the Sobel images of the LBD input: one wavefront per line, lane = one of
 * the 63 band rows for the row sums, lane = (moment, band) for the band accumulation; every float sum in the reference's order */
hipError_t drfe_launch_lbd(const LbdLine* d_lines, int n, const int16_t* d_gx, const int16_t* d_gy, int w, int h,
                           const LbdTables& tab, uint8_t* d_out, hipStream_t s);

/* ---- rect_improve + NFA on the device (lsd_nfa_kernels.hip) ---- */
/* the constants of cv::LineSegmentDetectorImpl::nfa as the HOST's libm computes them: the device never evaluates a log-gamma
 * or a log of p itself.  p[j] = p0 / 2^j (rect_improve halves the precision at most ten times); lgamma[i] = log_gamma(double(i))
 * for every i a W x H field can ask for (i <= W x H + 1) */
struct LsdNfaTables {
    double logNT;
    double p[11], logP[11], log1mP[11], log10P[11];
    const double* lgamma; int lgammaN;
    int lgammaFirst;         /* nfa()'s first term: 0 = `double(n) + 1` as OpenCV 3.4's lsd.cpp spells it, 1 = log_gamma(n + 1) (the LSD paper) */
};
/* drfe_lsd_configure_rect's modes.  0 (default) = the OpenCV 3.4 source text throughout: rect_nfa's integer corners / step quotients /
 * (y - tailp->p.x) denominators AND nfa()'s `log1term = (double(n) + 1) - log_gamma(k + 1) - log_gamma(n - k + 1) + ...` (the first
 * log_gamma of the LSD paper is missing in the library: nearly every rectangle with k > n p then passes the NFA test at once);
 * 1 = the LSD paper's reading of both (real-valued corners, log_gamma(n + 1)): rounds 2-3; 2 = integer corners with log_gamma(n + 1):
 * round 4's default. */
static inline int lsd_walk_mode(int rectMode) { return rectMode == 1 ? 1 : 0; }
static inline int lsd_lgamma_first(int rectMode) { return rectMode != 0 ? 1 : 0; }
/* one validated rectangle: the segment LineSegmentDetectorImpl::detect would emit (input-image scale), flag 1 = kept */
struct LsdSegOut { float x1, y1, x2, y2; int flag; };
#define DRFE_LSD_NFA_UNCERTAIN 1        /* LsdGrowFrame::out[2]: a decision of rect_improve the device could not certify */
struct LsdGrowFrame;
/* rect_improve for every rectangle of nframes frames (d_frames[f].rects / out[0]): d_segs[f * rectCap + i] per rectangle in
 * seed order; d_frames[f].out[2] |= DRFE_LSD_NFA_UNCERTAIN when a decision was too close to certify */
hipError_t drfe_launch_rect_improve(const LsdGrowFrame* d_frames, int nframes, int W, int H, int rectMode, const LsdNfaTables& tab,
                                    int rectCap, LsdSegOut* d_segs, hipStream_t s);

/* key lines, response cut, LBD parameters and line equations of nframes frames from k_rect_improve's segments (one wavefront per
 * frame): d_kl / d_lineF / d_lbd hold klCap slots per frame; d_frameOut[4 f] = {lines kept, lines detected, status (0 = done,
 * otherwise the host finishes the frame), pad} */
hipError_t drfe_launch_lsd_keylines(const LsdGrowFrame* d_frames, const LsdSegOut* d_segs, int rectCap, int nframes, int w, int h, int maxLines,
                                    int klCap, drfe_keyline* d_kl, double* d_lineF, LbdLine* d_lbd, int* d_frameOut, hipStream_t s);
hipError_t drfe_launch_lbd_batch(const LbdLine* d_lines, const int* d_frameCounts, int klCap, int nframes, const int16_t* d_gx, const int16_t* d_gy,
                                 int w, int h, const LbdTables& tab, uint8_t* d_out, hipStream_t s);

/* ---- device region growing (lsd_grow_kernels.hip) ---- */
/* a rectangle as cv::LineSegmentDetectorImpl::rect carries it (what region2rect fills; prec, p of the detection) */
struct LsdRect { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; };
#define DRFE_LSD_STATUS_UNCERTAIN 1     /* a cos / sin whose correct rounding could not be certified (cr_sincos.h) */
#define DRFE_LSD_STATUS_OVERFLOW 2      /* more accepted regions than rectCap */
#define DRFE_LSD_OUT_NEXT_RECT 36       /* out[36]: the next rectangle k_rect_improve's wavefronts take (zeroed by k_lsd_grow) */
#define DRFE_LSD_OUT_INTS 40             /* ints per frame in LsdGrowFrame::out: count, status, pad, pad, then 16 x u64 phase counters of LSD_PROFILE builds */
/* one frame of a k_lsd_grow launch: its level-line fields, the sorted pseudo-ordering, scratch and outputs */
struct LsdGrowFrame {
    const double* ang; const float2* cs; const double* mod;   /* W x H fields of k_ll_angle */
    const float2* cs0;                                        /* W x H: float(cos), float(sin) of the pixel's angle (k_lsd_keys): a region's first direction */
    const uint32_t* order;                                    /* keys bin << 22 | y << 11 | x in std::sort's order */
    const uint32_t* notdef;                                   /* (W x H + 31) / 32 words, bit q set = pixel q has no angle (k_lsd_notdef); null: the growth kernel reads the angles itself */
    uint32_t* reg; uint32_t* tmp;                             /* W x H entries each: member list (y << 16 | x), shrink scratch */
    uint32_t* regMw; uint32_t* tmpMw;                         /* k_lsd_grow_mw: 4 x regCap entries each, one share per wavefront of the frame's workgroup */
    uint32_t* gbm;                                            /* k_lsd_grow_mw: a W x H bitmap in HBM, the overlay of a region too large for an LDS table */
    LsdRect* rects; int* out;                                 /* accepted rectangles in seed order; out[0] = count, out[1] = status (DRFE_LSD_OUT_INTS ints per frame) */
    int nOrder; uint32_t minSeedBin;
    const unsigned long long* meta;                           /* non-null: minSeedBin = 1024 - low word of meta[1] (k_lsd_keys), read on the device */
};
size_t drfe_lsd_grow_lds_bytes(int W, int H);
/* keys of nframes consecutive slots: d_mod / d_ang / d_meta / d_keys point at the first of them */
hipError_t drfe_launch_lsd_keys(const double* d_mod, const double* d_ang, int W, int H, unsigned long long* d_meta, uint32_t* d_keys,
                                float2* d_cs0, uint32_t* d_notdef, int nframes, hipStream_t s);
/* std::sort's permutation of nframes key arrays (n keys each, keyStride apart) in place: introsort's moves on the device
 * (lsd_order_kernels.hip).  d_posL / d_posR: scratch of >= n entries per frame, posStride apart.  d_status[f * statusStride]:
 * 0, or 1 = a range ran out of introsort's depth limit (heap sort in libstdc++), 2 = internal queue overflow: order on the host. */
hipError_t drfe_launch_lsd_order(uint32_t* d_keys, size_t keyStride, int n, uint32_t* d_posL, uint32_t* d_posR, size_t posStride,
                                 int* d_status, int statusStride, int nframes, hipStream_t s, int depthOverride = -1);
/* regCapMw > 0: the multi-wave kernel (four wavefronts per frame, speculation with in-order commit: k_lsd_grow_mw), each wavefront
 * owning regCapMw entries of regMw / tmpMw; 0: one wavefront per frame (k_lsd_grow).  Identical results. */
hipError_t drfe_launch_lsd_grow(const LsdGrowFrame* d_frames, int nframes, int W, int H, double prec, double p, int minReg,
                                double densityTh, int rectCap, hipStream_t s, int regCapMw = 0);
size_t drfe_lsd_grow_mw_lds_bytes(int W, int H);

/* the image passes for slots frame0 .. frame0 + nframes - 1 of sc (d_img = slot frame0's input image, frames w x h apart) */
hipError_t drfe_launch_lines_passes(const uint8_t* d_img, int w, int h, const LineTaps& lsdTaps, const LineTaps& lbdTaps,
                                    LinesScratch* sc, int frame0, int nframes, double threshold, hipStream_t s);
void drfe_lines_free(drfe_ctx* c);
#endif
