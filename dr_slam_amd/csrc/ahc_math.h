/* ahc_math.h — float64 plane-fit arithmetic used by the device block kernel and the host clustering of
 * the AHC path (product code; the oracle has its own restatement).
 *
 * plane_from_sums == ahc::PlaneSeg::Stats::compute (reference include/peac/AHCPlaneSeg.hpp:125-156):
 * covariance from the nine sums, Eigen 3.3.7 SelfAdjointEigenSolver<Matrix3d> (scaling, closed-form 3x3
 * Householder tridiagonalisation, implicit symmetric QR with Wilkinson shift, ascending sort), normal =
 * eigenvector of the least eigenvalue oriented towards the camera.  Only + - * / sqrt fabs in IEEE
 * double, compiled without FMA contraction on both sides.
 */
#ifndef DRFE_AHC_MATH_H
#define DRFE_AHC_MATH_H

#include <math.h>

#if defined(__HIPCC__)
#define AHC_HD __host__ __device__ static inline
#else
#define AHC_HD static inline
#endif

struct AhcFit {
    double center[3], normal[3], mse, curvature;
};

/* Eigen's JacobiRotation::makeGivens (real case: Jacobi.h) without lane-divergent branches.  Its two general cases are one
 * computation with the roles of p and q exchanged - t = small / big, u = +-sqrt(1 + t t) with the sign of `big`, r = 1 / u, m = t r:
 *   |p| > |q|:  c = 1 / u = r,           s = -t c     = -m
 *   else:       s = -1 / u = -r,         c = -t s     = (-t)(-r) = m
 * (sign changes are exact in IEEE arithmetic, so -1.0 / u == -(1.0 / u) and (-t) c == -(t c): the same bits as the branches) - and
 * the two degenerate cases override the result.  On the device the trial merges run one plane fit per lane: with the branches, a
 * wavefront whose lanes disagree on the case executes the division / square root / division sequence once per case (round 6). */
AHC_HD void ahc_givens(double p, double q, double* c, double* s)
{
    const bool pBig = fabs(p) > fabs(q);
    const double big = pBig ? p : q, small = pBig ? q : p;
    const double t = small / big;
    double u = sqrt(1.0 + t * t);
    if (big < 0.0) u = -u;
    const double r = 1.0 / u;
    const double m = t * r;
    double cc = pBig ? r : m, ss = pBig ? -m : -r;
    if (p == 0.0) { cc = 0.0; ss = q < 0.0 ? 1.0 : -1.0; }
    if (q == 0.0) { cc = p < 0.0 ? -1.0 : 1.0; ss = 0.0; }
    *c = cc;
    *s = ss;
}

/* eigen-decomposition of the symmetric matrix given by its lower triangle; ev ascending,
 * Q column-major (column k = eigenvector k) */
AHC_HD void ahc_eig3(double m00, double m10, double m20, double m11, double m21, double m22, double ev[3], double Q[9])
{
    double scale = fabs(m00);
    scale = fmax(scale, fabs(m10)); scale = fmax(scale, fabs(m20)); scale = fmax(scale, fabs(m11));
    scale = fmax(scale, fabs(m21)); scale = fmax(scale, fabs(m22));
    if (scale == 0.0) scale = 1.0;
    m00 /= scale; m10 /= scale; m20 /= scale; m11 /= scale; m21 /= scale; m22 /= scale;
    double d[3], e[2];
    const double tiny = 2.2250738585072014e-308;   /* DBL_MIN */
    d[0] = m00;
    const double v1norm2 = m20 * m20;
    for (int i = 0; i < 9; i++) Q[i] = 0.0;
    Q[0] = 1.0;
    if (v1norm2 <= tiny) {
        d[1] = m11; d[2] = m22; e[0] = m10; e[1] = m21;
        Q[4] = 1.0; Q[8] = 1.0;
    } else {
        const double beta = sqrt(m10 * m10 + v1norm2);
        const double invBeta = 1.0 / beta;
        const double m01 = m10 * invBeta, m02 = m20 * invBeta;
        const double q = 2.0 * m01 * m21 + m02 * (m22 - m11);
        d[1] = m11 + m02 * q;
        d[2] = m22 - m02 * q;
        e[0] = beta;
        e[1] = m21 - m01 * q;
        Q[4] = m01; Q[5] = m02; Q[7] = m02; Q[8] = -m01;
    }
    int end = 2, start = 0, iter = 0;
    const double prec = 2.0 * 2.220446049250313e-16;
    while (end > 0) {
        for (int i = start; i < end; ++i)
            if (fabs(e[i]) <= (fabs(d[i]) + fabs(d[i + 1])) * prec || fabs(e[i]) <= tiny) e[i] = 0.0;
        while (end > 0 && e[end - 1] == 0.0) end--;
        if (end <= 0) break;
        iter++;
        if (iter > 90) break;
        start = end - 1;
        while (start > 0 && e[start - 1] != 0.0) start--;
        /* one implicit QR step on [start, end] */
        const double td = (d[end - 1] - d[end]) * 0.5;
        const double ee = e[end - 1];
        double mu = d[end];
        if (td == 0.0) mu -= fabs(ee);
        else {
            const double e2 = ee * ee;
            const double ax = fabs(td), ay = fabs(ee);
            const double p = ax > ay ? ax : ay, lo = ax > ay ? ay : ax;       /* numext::hypot's scaling: one division whichever is larger */
            const double qp = lo / p;
            const double h = (p == 0.0) ? 0.0 : p * sqrt(1.0 + qp * qp);
            if (e2 == 0.0) mu -= (ee / (td + (td > 0.0 ? 1.0 : -1.0))) * (ee / h);
            else mu -= e2 / (td + (td > 0.0 ? h : -h));
        }
        double x = d[start] - mu, z = e[start];
        for (int k = start; k < end; ++k) {
            double c, s;
            ahc_givens(x, z, &c, &s);
            const double sdk = s * d[k] + c * e[k];
            const double dkp1 = s * e[k] + c * d[k + 1];
            d[k] = c * (c * d[k] - s * e[k]) - s * (c * e[k] - s * d[k + 1]);
            d[k + 1] = s * sdk + c * dkp1;
            e[k] = c * sdk - s * dkp1;
            if (k > start) e[k - 1] = c * e[k - 1] - s * z;
            x = e[k];
            if (k < end - 1) { z = -s * e[k + 1]; e[k + 1] = c * e[k + 1]; }
            for (int i = 0; i < 3; i++) {
                const double xi = Q[k * 3 + i], yi = Q[(k + 1) * 3 + i];
                Q[k * 3 + i] = c * xi - s * yi;
                Q[(k + 1) * 3 + i] = s * xi + c * yi;
            }
        }
    }
    if (iter <= 90) {
        for (int i = 0; i < 2; ++i) {
            int k = 0;
            for (int j = 1; j < 3 - i; j++)
                if (d[i + j] < d[i + k]) k = j;
            if (k > 0) {
                const double t = d[i]; d[i] = d[k + i]; d[k + i] = t;
                for (int r = 0; r < 3; r++) { const double u = Q[i * 3 + r]; Q[i * 3 + r] = Q[(k + i) * 3 + r]; Q[(k + i) * 3 + r] = u; }
            }
        }
    }
    for (int i = 0; i < 3; i++) ev[i] = d[i] * scale;
}

/* sums: sx sy sz sxx syy szz sxy syz sxz */
AHC_HD void ahc_plane_from_sums(const double* S, int N, AhcFit* f)
{
    const double sc = 1.0 / N;
    f->center[0] = S[0] * sc; f->center[1] = S[1] * sc; f->center[2] = S[2] * sc;
    const double k00 = S[3] - S[0] * S[0] * sc, k01 = S[6] - S[0] * S[1] * sc, k02 = S[8] - S[0] * S[2] * sc;
    const double k11 = S[4] - S[1] * S[1] * sc, k12 = S[7] - S[1] * S[2] * sc, k22 = S[5] - S[2] * S[2] * sc;
    double ev[3], Q[9];
    ahc_eig3(k00, k01, k02, k11, k12, k22, ev, Q);
    const double v0 = Q[0], v1 = Q[1], v2 = Q[2];   /* eigenvector of the least eigenvalue */
    if (v0 * f->center[0] + v1 * f->center[1] + v2 * f->center[2] <= 0) { f->normal[0] = v0; f->normal[1] = v1; f->normal[2] = v2; }
    else { f->normal[0] = -v0; f->normal[1] = -v1; f->normal[2] = -v2; }
    f->mse = ev[0] * sc;
    f->curvature = ev[0] / (ev[0] + ev[1] + ev[2]);
}

#endif
