/* oracle/lsd_oracle.h — TEST INFRASTRUCTURE (see oracle.h). LSD + LBD line features restatement. */
#ifndef DRFE_LSD_ORACLE_H
#define DRFE_LSD_ORACLE_H
#include <stdint.h>
#include <vector>

namespace orc {

/* cv::line_descriptor::KeyLine fields the path produces / consumes */
struct KeyLine {
    float angle = 0;
    int class_id = 0, octave = 0;
    float ptX = 0, ptY = 0, response = 0, size = 0;
    float startPointX = 0, startPointY = 0, endPointX = 0, endPointY = 0;
    float sPointInOctaveX = 0, sPointInOctaveY = 0, ePointInOctaveX = 0, ePointInOctaveY = 0;
    float lineLength = 0;
    int numOfPixels = 0;
};

struct LsdStages {   /* intermediates for stage-by-stage parity of the device image passes */
    int sw = 0, sh = 0;
    std::vector<uint8_t> scaled;          /* Gaussian + 0.8 downscale */
    std::vector<double> modgrad, angles;  /* ll_angle */
    std::vector<int16_t> gx, gy;          /* Sobel of the 5x5-blurred image (LBD input) */
    std::vector<int> rectCounts;          /* (total_pts, alg_pts) of every rect_nfa call in call order */
    std::vector<float> segments;          /* the detector's x1 y1 x2 y2 per accepted segment, before the key-line stage */
    std::vector<double> segInfo;          /* per accepted segment: width, p (precision / pi), -log10(NFA) - detect()'s optional outputs */
};

struct LineResult {
    int detected = 0;                 /* lines before the 40-highest-response cut */
    std::vector<KeyLine> lines;
    std::vector<uint8_t> desc;        /* NL x 32 */
    std::vector<float> descf;         /* NL x 72 (float LBD before binarisation) */
    std::vector<double> lineF;        /* NL x 3 */
};

void gaussian_blur_q8(const uint8_t* src, int w, int h, const std::vector<int>& taps, uint8_t* dst);
void resize_linear_exact_08(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh);
void sobel3_s16(const uint8_t* src, int w, int h, int16_t* gx, int16_t* gy);
void lbd_descriptor(const int16_t* dxImg, const int16_t* dyImg, int realWidth, int realHeight, const KeyLine& kl,
                    float* desVec72, uint8_t* desc32);
/* rectMode: the reading of OpenCV 3.4's lsd.cpp - 0 the source text (rect_nfa's integer corners and step quotients, nfa()'s
 * `double(n) + 1` first term; default), 1 the LSD paper's reading of both (rounds 2-3), 2 integer corners with log_gamma(n + 1)
 * (round 4); see lsd_oracle.cpp */
LineResult extract_lines(const uint8_t* img, int w, int h, int maxLines = 40, LsdStages* stages = nullptr, int rectMode = 0);

} // namespace orc
#endif
