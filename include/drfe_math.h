/* drfe_math.h — scalar arithmetic primitives whose results must be bit-identical on host and gfx950.
 *
 * Every function here is plain IEEE-754 add/mul/div/compare/convert, written so that a compiler that
 * does NOT contract a*b+c into an FMA (-ffp-contract=off on g++ AND hipcc) yields the same bits on
 * x86-64 and on CDNA4.  They restate library calls the reference makes into un-vendored OpenCV/glibc
 * ("parity unpinned" against a real OpenCV 3.4.4 build):
 *
 *   drfe_round_half_even  <- cvRound()            (reference src/ORBextractor.cc:81,119-120,1111)
 *   drfe_fast_atan2       <- cv::fastAtan2()      (reference src/ORBextractor.cc:103)
 *   drfe_sincos           <- cos(float)/sin(float) (reference src/ORBextractor.cc:113-114); the
 *                            reference calls glibc cosf/sinf whose last bit is host dependent, so the
 *                            build canonicalises on this routine.
 *   drfe_logf             <- log(float) of MapPoint::PredictScale / MapLine::PredictScale (reference
 *                            src/MapPoint.cc:456, src/MapLine.cpp:389) — same libm caveat, same remedy.
 */
#ifndef DRFE_MATH_H
#define DRFE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define DRFE_HD __host__ __device__ static inline
#else
#define DRFE_HD static inline
#endif

/* cvRound(float): SSE cvtss2si == round-half-to-even. rintf() honours the (default) RNE mode on both
 * sides (v_rndne_f32 on gfx950). */
DRFE_HD int drfe_round_half_even(float v) { return (int)rintf(v); }
DRFE_HD int drfe_round_half_even_d(double v) { return (int)rint(v); }

/* cv::fastAtan2(y, x) of OpenCV 3.x, degrees in [0,360). float32 throughout, Horner form, no FMA. */
DRFE_HD float drfe_fast_atan2(float y, float x)
{
    const float scale = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16; /* (float)DBL_EPSILON */
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* cos/sin of a float32 angle in radians, evaluated in float64 (Cody-Waite pi/2 reduction + Taylor
 * polynomials, |r| <= pi/4, truncation error < 1e-19) and rounded once to float32.  Valid for
 * |rad| < 1e4 (the path only produces [0, 2*pi]). */
DRFE_HD void drfe_sincos(float rad, float* s_out, float* c_out)
{
    const double x = (double)rad;
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632673412561417e+00; /* 33 significant bits */
    const double pio2_lo = 6.07710050650619224932e-11;
    const double kd = rint(x * two_over_pi);
    const int k = (int)kd;
    double r = x - kd * pio2_hi;
    r = r - kd * pio2_lo;
    const double z = r * r;
    /* sin(r) = r * (1 + z*(s1 + z*(s2 + ...)))   cos(r) = 1 + z*(c1 + z*(c2 + ...)) */
    double ps = -8.22063524662432971696e-18;              /* -1/19! */
    ps = ps * z + 2.81145725434552076320e-15;             /*  1/17! */
    ps = ps * z + -7.64716373181981647590e-13;            /* -1/15! */
    ps = ps * z + 1.60590438368216145994e-10;             /*  1/13! */
    ps = ps * z + -2.50521083854417187751e-08;            /* -1/11! */
    ps = ps * z + 2.75573192239858906526e-06;             /*  1/9!  */
    ps = ps * z + -1.98412698412698412698e-04;            /* -1/7!  */
    ps = ps * z + 8.33333333333333333333e-03;             /*  1/5!  */
    ps = ps * z + -1.66666666666666666667e-01;            /* -1/3!  */
    const double sn = r + r * (z * ps);
    double pc = 4.11031762331216485848e-19;               /*  1/20! */
    pc = pc * z + -1.56192069685862264622e-16;            /* -1/18! */
    pc = pc * z + 4.77947733238738529744e-14;             /*  1/16! */
    pc = pc * z + -1.14707455977297247139e-11;            /* -1/14! */
    pc = pc * z + 2.08767569878680989792e-09;             /*  1/12! */
    pc = pc * z + -2.75573192239858906526e-07;            /* -1/10! */
    pc = pc * z + 2.48015873015873015873e-05;             /*  1/8!  */
    pc = pc * z + -1.38888888888888888889e-03;            /* -1/6!  */
    pc = pc * z + 4.16666666666666666667e-02;             /*  1/4!  */
    pc = pc * z + -5.00000000000000000000e-01;            /* -1/2!  */
    const double cs = 1.0 + z * pc;
    double s, c;
    switch (k & 3) {
    case 0: s = sn; c = cs; break;
    case 1: s = cs; c = -sn; break;
    case 2: s = -sn; c = -cs; break;
    default: s = -cs; c = sn; break;
    }
    *s_out = (float)s;
    *c_out = (float)c;
}

/* 256-bit Hamming distance: the reference's 8x32-bit SWAR popcount (src/ORBmatcher.cc:1712-1728,
 * src/LSDmatcher.cpp:316-332) equals popcount(xor) over 4x u64. */
/* log of a positive finite float32, evaluated in float64 and rounded once to float32:
 * x = m * 2^e with m in [sqrt(1/2), sqrt(2)), log m = 2 atanh((m-1)/(m+1)) as an odd series to t^25
 * (|t| <= 0.1716: truncation < 1e-20), plus e * ln 2.  Zero, negative, inf and NaN follow logf(). */
DRFE_HD float drfe_logf(float xf)
{
    if (!(xf > 0.0f)) return xf == 0.0f ? -INFINITY : NAN;
    if (xf == INFINITY) return xf;
    double x = (double)xf;
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7FF) - 1023;          /* floats promoted to double are never subnormal */
    bits = (bits & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m;
    memcpy(&m, &bits, 8);                                 /* [1, 2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double t = (m - 1.0) / (m + 1.0), t2 = t * t;
    double p = 1.0 / 25.0;
    p = p * t2 + 1.0 / 23.0; p = p * t2 + 1.0 / 21.0; p = p * t2 + 1.0 / 19.0; p = p * t2 + 1.0 / 17.0;
    p = p * t2 + 1.0 / 15.0; p = p * t2 + 1.0 / 13.0; p = p * t2 + 1.0 / 11.0; p = p * t2 + 1.0 / 9.0;
    p = p * t2 + 1.0 / 7.0;  p = p * t2 + 1.0 / 5.0;  p = p * t2 + 1.0 / 3.0;  p = p * t2 + 1.0;
    return (float)(2.0 * t * p + (double)e * 0.6931471805599453);
}

DRFE_HD int drfe_hamming256(const uint64_t* a, const uint64_t* b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(a[0] ^ b[0]) + __popcll(a[1] ^ b[1]) + __popcll(a[2] ^ b[2]) + __popcll(a[3] ^ b[3]);
#else
    return __builtin_popcountll(a[0] ^ b[0]) + __builtin_popcountll(a[1] ^ b[1]) +
           __builtin_popcountll(a[2] ^ b[2]) + __builtin_popcountll(a[3] ^ b[3]);
#endif
}

#endif /* DRFE_MATH_H */
