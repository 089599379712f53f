/* undistort_math.h — cv::undistortPoints(src, dst, K, distCoef, Mat(), K) for one point, as
 * Frame::UndistortKeyPoints / ComputeImageBounds use it (reference src/Frame.cc:835-888; OpenCV 3.4
 * cvUndistortPointsInternal: R = identity, P = K, no tilt, TermCriteria(MAX_ITER, 5, 0.01) = five fixed-point
 * iterations in double).  Shared by the device kernel and the host (image bounds); compiled with
 * -ffp-contract=off so both evaluate the library's rounding sequence.  The zero-coefficient terms of the
 * 14-coefficient model are kept on purpose: they add exact zeros in the library too. */
#ifndef DRFE_UNDISTORT_MATH_H
#define DRFE_UNDISTORT_MATH_H

#if defined(__HIPCC__)
#define DRFE_UHD __host__ __device__
#else
#define DRFE_UHD
#endif

struct DrfeDistortion {
    double fx, fy, cx, cy;   /* mK (CV_32F) converted to double */
    double k[5];             /* k1, k2, p1, p2, k3 */
    int enabled;             /* mDistCoef.at<float>(0) != 0 */
};

DRFE_UHD static inline void drfe_undistort_point(const DrfeDistortion& D, float px, float py, float* ox, float* oy)
{
    const double ifx = 1. / D.fx, ify = 1. / D.fy;
    const double k0 = D.k[0], k1 = D.k[1], k2 = D.k[2], k3 = D.k[3], k4 = D.k[4], zero = 0.0;
    double x = ((double)px - D.cx) * ifx;
    double y = ((double)py - D.cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((zero * r2 + zero) * r2 + zero) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + zero * r2 + zero * r2 * r2;
        const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y + zero * r2 + zero * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = D.fx * x + zero * y + D.cx;
    const double yy = zero * x + D.fy * y + D.cy;
    const double ww = 1. / (zero * x + zero * y + 1.0);
    *ox = (float)(xx * ww);
    *oy = (float)(yy * ww);
}

#endif
