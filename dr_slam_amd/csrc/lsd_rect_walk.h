/* lsd_rect_walk.h - the scan-line bookkeeping of cv::LineSegmentDetectorImpl::rect_nfa (OpenCV 3.4 imgproc/src/lsd.cpp, the
 * detector behind reference src/LSDextractor.cpp:14-17), shared by k_rect_counts (lines_kernels.hip) and k_rect_improve
 * (lsd_nfa_kernels.hip).  Device code. */
#ifndef DRFE_LSD_RECT_WALK_H
#define DRFE_LSD_RECT_WALK_H
#include "lines_internal.h"

struct RectWalk { int loX, loY, hiY, leftY, rightY; double fl, sl, fr, sr; };

__device__ __forceinline__ RectWalk rect_walk_setup(const RectCand& rc, int rectMode)
{
    const double hw = rc.width / 2.0, dyhw = rc.dy * hw, dxhw = rc.dx * hw;
    int cx[4] = {(int)(rc.x1 - dyhw), (int)(rc.x2 - dyhw), (int)(rc.x2 + dyhw), (int)(rc.x1 + dyhw)};
    int cy[4] = {(int)(rc.y1 + dxhw), (int)(rc.y2 + dxhw), (int)(rc.y2 - dxhw), (int)(rc.y1 - dxhw)};
    /* ascending (x, y): five compare-exchanges */
#define CSWAP(a, b)                                                                     \
    if (cx[a] > cx[b] || (cx[a] == cx[b] && cy[a] > cy[b])) {                           \
        const int tx = cx[a], ty = cy[a]; cx[a] = cx[b]; cy[a] = cy[b]; cx[b] = tx; cy[b] = ty; \
    }
    CSWAP(0, 1) CSWAP(2, 3) CSWAP(0, 2) CSWAP(1, 3) CSWAP(1, 2)
#undef CSWAP
    int lo = 0, hi = 0;
    for (int i = 1; i < 4; i++) { if (cy[lo] > cy[i]) lo = i; if (cy[hi] < cy[i]) hi = i; }
    bool taken[4] = {false, false, false, false};
    taken[lo] = true;
    int left = -1, right = -1, tail = -1;
    for (int i = 0; i < 4; i++) if (!taken[i] && (left < 0 || cx[left] > cx[i])) left = i;
    taken[left] = true;
    for (int i = 0; i < 4; i++) if (!taken[i] && (right < 0 || cx[right] < cx[i])) right = i;
    taken[right] = true;
    for (int i = 0; i < 4; i++) if (!taken[i] && (tail < 0 || cx[tail] > cx[i])) tail = i;
    const int loX = cx[lo], loY = cy[lo], leftX = cx[left], leftY = cy[left], rightX = cx[right], rightY = cy[right],
              tailX = cx[tail], tailY = cy[tail];
    RectWalk w;
    w.loX = loX; w.loY = loY; w.hiY = cy[hi]; w.leftY = leftY; w.rightY = rightY;
    if (rectMode == 0) {
        w.fl = (loY != leftY) ? (double)((loX - leftX) / (loY - leftY)) : 0.0;
        w.sl = (leftY != tailX) ? (double)((leftX - tailX) / (leftY - tailX)) : 0.0;
        w.fr = (loY != rightY) ? (double)((loX - rightX) / (loY - rightY)) : 0.0;
        w.sr = (rightY != tailX) ? (double)((rightX - tailX) / (rightY - tailX)) : 0.0;
    } else {
        w.fl = (loY != leftY) ? (double)(loX - leftX) / (double)(loY - leftY) : 0.0;
        w.sl = (leftY != tailX) ? (double)(leftX - tailX) / (double)(leftY - tailY) : 0.0;
        w.fr = (loY != rightY) ? (double)(loX - rightX) / (double)(loY - rightY) : 0.0;
        w.sr = (rightY != tailX) ? (double)(rightX - tailX) / (double)(rightY - tailY) : 0.0;
        if (!isfinite(w.sl)) w.sl = 0;
        if (!isfinite(w.sr)) w.sr = 0;
    }
    return w;
}


#endif
