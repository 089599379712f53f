/* cr_sincos.h — cos / sin of a double, CORRECTLY ROUNDED, from + - * fma only: identical bits on x86-64 and gfx950.
 *
 * Why: cv::LineSegmentDetectorImpl calls the host's libm in two places that decide output bits - `dx = cos(theta),
 * dy = sin(theta)` of region2rect and `sumdx = float(cos(reg_angle))` at the start of region_grow (OpenCV 3.4 lsd.cpp, behind
 * reference src/LSDextractor.cpp:12-43).  Once region growing and the rectangle fit run on the device (lsd_grow_kernels.hip)
 * there is no libm to call.  glibc's double sin / cos stay within 0.55 ulp, so they return the correctly rounded value except
 * when the exact result lies within ~0.05 ulp of a rounding boundary; this routine returns the correctly rounded value always
 * (or says that it cannot certify it, ~2^-44 of the arguments, and the caller falls back to the host path).
 * tests/test_host_cpu.py compares it with the host's libm and with a 60-digit decimal evaluation.
 *
 * Method: x in [0, 64): k = round(x * 2/pi), r = x - k * pi/2 as a double-double (pi/2 in three doubles, products by fma,
 * |error| < 2^-150); sin r and cos r by Taylor series, the small terms in double, the leading ones in double-double; quadrant
 * swap; the final double is the high word after normalisation, certified by Ziv's test (the value moved by its error bound
 * either way rounds to the same double): first with three double-double terms (error 2^-64), then, for the one call in ~250
 * that lands within that of a rounding boundary, with eight (2^-100). */
#ifndef DRFE_CR_SINCOS_H
#define DRFE_CR_SINCOS_H

#include <math.h>

#if defined(__HIPCC__)
#define DRFE_CR_HD __host__ __device__ static inline
#else
#define DRFE_CR_HD static inline
#endif

struct drfe_dd { double h, l; };

DRFE_CR_HD drfe_dd drfe_dd_two_sum(double a, double b)
{
    const double s = a + b, bb = s - a;
    const double e = (a - (s - bb)) + (b - bb);
    drfe_dd r = {s, e};
    return r;
}
DRFE_CR_HD drfe_dd drfe_dd_fast_two_sum(double a, double b)       /* |a| >= |b| or a == 0 */
{
    const double s = a + b;
    drfe_dd r = {s, b - (s - a)};
    return r;
}
DRFE_CR_HD drfe_dd drfe_dd_two_prod(double a, double b)
{
    const double p = a * b;
    drfe_dd r = {p, fma(a, b, -p)};
    return r;
}
DRFE_CR_HD drfe_dd drfe_dd_add(drfe_dd a, drfe_dd b)
{
    drfe_dd s = drfe_dd_two_sum(a.h, b.h);
    const drfe_dd t = drfe_dd_two_sum(a.l, b.l);
    s.l += t.h;
    s = drfe_dd_fast_two_sum(s.h, s.l);
    s.l += t.l;
    return drfe_dd_fast_two_sum(s.h, s.l);
}
DRFE_CR_HD drfe_dd drfe_dd_mul(drfe_dd a, drfe_dd b)
{
    drfe_dd p = drfe_dd_two_prod(a.h, b.h);
    p.l += a.h * b.l;
    p.l += a.l * b.h;
    return drfe_dd_fast_two_sum(p.h, p.l);
}
DRFE_CR_HD drfe_dd drfe_dd_neg(drfe_dd a) { drfe_dd r = {-a.h, -a.l}; return r; }

/* (-1)^n / (2n+1)! and (-1)^n / (2n)! as double-doubles, n = 0..14 */
#define DRFE_CR_SIN_COEFS                                                                                          \
    {0x1.0000000000000p+0, 0x0.0p+0}, {-0x1.5555555555555p-3, -0x1.5555555555555p-57},                             \
    {0x1.1111111111111p-7, 0x1.1111111111111p-63}, {-0x1.a01a01a01a01ap-13, -0x1.a01a01a01a01ap-73},               \
    {0x1.71de3a556c734p-19, -0x1.c154f8ddc6c00p-73}, {-0x1.ae64567f544e4p-26, 0x1.c062e06d1f209p-80},              \
    {0x1.6124613a86d09p-33, 0x1.f28e0cc748ebep-87}, {-0x1.ae7f3e733b81fp-41, -0x1.1d8656b0ee8cbp-97},              \
    {0x1.952c77030ad4ap-49, 0x1.ac981465ddc6cp-103}, {-0x1.2f49b46814157p-57, -0x1.2650f61dbdcb4p-112},            \
    {0x1.71b8ef6dcf572p-66, -0x1.d043ae40c4647p-120}, {-0x1.761b41316381ap-75, 0x1.3423c7d91404fp-130},            \
    {0x1.3f3ccdd165fa9p-84, -0x1.58ddadf344487p-139}, {-0x1.d1ab1c2dccea3p-94, -0x1.054d0c78aea14p-149},           \
    {0x1.259f98b4358adp-103, 0x1.eaf8c39dd9bc5p-157}
#define DRFE_CR_COS_COEFS                                                                                          \
    {0x1.0000000000000p+0, 0x0.0p+0}, {-0x1.0000000000000p-1, 0x0.0p+0},                                           \
    {0x1.5555555555555p-5, 0x1.5555555555555p-59}, {-0x1.6c16c16c16c17p-10, 0x1.f49f49f49f49fp-65},                \
    {0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-76}, {-0x1.27e4fb7789f5cp-22, -0x1.cbbc05b4fa99ap-76},              \
    {0x1.1eed8eff8d898p-29, -0x1.2aec959e14c06p-83}, {-0x1.93974a8c07c9dp-37, -0x1.05d6f8a2efd1fp-92},             \
    {0x1.ae7f3e733b81fp-45, 0x1.1d8656b0ee8cbp-101}, {-0x1.6827863b97d97p-53, -0x1.eec01221a8b0bp-107},            \
    {0x1.e542ba4020225p-62, 0x1.ea72b4afe3c2fp-120}, {-0x1.0ce396db7f853p-70, 0x1.aebcdbd20331cp-124},             \
    {0x1.f2cf01972f578p-80, -0x1.9ada5fcc1ab14p-135}, {-0x1.88e85fc6a4e5ap-89, 0x1.71c37ebd16540p-143},            \
    {0x1.0a18a2635085dp-98, 0x1.b9e2e28e1aa54p-153}

/* sum_{n} c[n] z^n with z = r^2 as a double-double: the terms n >= ddTerms in double, the leading ddTerms in double-double.
 * ddTerms = 8: every term below 2^-53 of the leading one is in the double part for |r| <= 0.8 (relative error < 2^-100);
 * ddTerms = 3: the double part is below 2^-14 of the leading term (relative error < 2^-64), at a third of the cost */
DRFE_CR_HD drfe_dd drfe_cr_series(const drfe_dd* c, drfe_dd z, int ddTerms)
{
    double t = c[14].h;
    for (int n = 13; n >= ddTerms; n--) t = t * z.h + c[n].h;
    drfe_dd acc = {t, 0.0};
    for (int n = ddTerms - 1; n >= 0; n--) acc = drfe_dd_add(drfe_dd_mul(acc, z), c[n]);
    return acc;
}

/* sin and cos of x as double-doubles.  Returns 0 if x is outside [0, 64) or not finite. */
DRFE_CR_HD int drfe_sincos_dd(double x, drfe_dd* s_out, drfe_dd* c_out, int ddTerms)
{
    if (!(x >= 0.0 && x < 64.0)) return 0;
    const double kd = rint(x * 0.6366197723675814);         /* 2/pi */
    const int k = (int)kd;
    /* r = x - k * (P1 + P2 + P3): k * P1 by fma (exact pair); x - high word exact (Sterbenz, k >= 1) */
    const drfe_dd p1 = drfe_dd_two_prod(kd, 0x1.921fb54442d18p+0);
    drfe_dd r = {x - p1.h, 0.0};
    r = drfe_dd_add(r, drfe_dd_neg(drfe_dd_two_prod(kd, 0x1.1a62633145c07p-54)));
    drfe_dd low = {-p1.l, kd * 0x1.f1976b7ed8fbcp-110};     /* - k P1 low part, - k P3 (P3 is negative) */
    r = drfe_dd_add(r, low);
    const drfe_dd z = drfe_dd_mul(r, r);
    const drfe_dd sc[15] = {DRFE_CR_SIN_COEFS};
    const drfe_dd cc[15] = {DRFE_CR_COS_COEFS};
    const drfe_dd sr = drfe_dd_mul(drfe_cr_series(sc, z, ddTerms), r);
    const drfe_dd cr = drfe_cr_series(cc, z, ddTerms);
    switch (k & 3) {
    case 0: *s_out = sr; *c_out = cr; break;
    case 1: *s_out = cr; *c_out = drfe_dd_neg(sr); break;
    case 2: *s_out = drfe_dd_neg(sr); *c_out = drfe_dd_neg(cr); break;
    default: *s_out = drfe_dd_neg(cr); *c_out = sr; break;
    }
    return 1;
}

/* 1 if the high word is certainly the correctly rounded double of the value h + l approximates to within relErr * |h| */
DRFE_CR_HD int drfe_cr_certain(drfe_dd v, double relErr)
{
    const double e = fabs(v.h) * relErr;
    return (v.h + (v.l + e) == v.h) && (v.h + (v.l - e) == v.h);
}

/* correctly rounded sin(x), cos(x); returns 0 when the rounding cannot be certified (the caller must not use the values).
 * Ziv's strategy: the cheap evaluation decides unless the value lies within 2^-60 of a rounding boundary (one call in ~250),
 * then the full one (2^-96; undecided for ~2^-43 of the arguments). */
DRFE_CR_HD int drfe_cr_sincos(double x, double* s_out, double* c_out)
{
    drfe_dd s, c;
    if (!drfe_sincos_dd(x, &s, &c, 3)) return 0;
    if (!(drfe_cr_certain(s, 0x1p-60) && drfe_cr_certain(c, 0x1p-60))) {
        (void)drfe_sincos_dd(x, &s, &c, 8);
        *s_out = s.h; *c_out = c.h;
        return drfe_cr_certain(s, 0x1p-96) && drfe_cr_certain(c, 0x1p-96);
    }
    *s_out = s.h; *c_out = c.h;
    return 1;
}

/* float(sin(x)), float(cos(x)) as C computes them from a correctly rounded double libm: round to double, then to float.
 * First in plain double (fma reduction by pi/2 in two words, Taylor to r^19: relative error < 2^-50, also next to a zero
 * of the function, where r itself is the small result): accepted when the value scaled by 1 +- 2^-47 rounds to the same
 * float, which places the exact value and its double rounding strictly inside that float's interval.  Otherwise (2^-22 of
 * the arguments) through the correctly rounded doubles. */
DRFE_CR_HD int drfe_cr_sincos_f(double x, float* s_out, float* c_out)
{
    if (x >= 0.0 && x < 64.0) {
        const double kd = rint(x * 0.6366197723675814);
        const int k = (int)kd;
        double r = fma(-kd, 0x1.921fb54442d18p+0, x);
        r = fma(-kd, 0x1.1a62633145c07p-54, r);
        const double z = r * r;
        double ps = -0x1.2f49b46814157p-57;            /* -1/19! */
        ps = ps * z + 0x1.952c77030ad4ap-49;
        ps = ps * z + -0x1.ae7f3e733b81fp-41;
        ps = ps * z + 0x1.6124613a86d09p-33;
        ps = ps * z + -0x1.ae64567f544e4p-26;
        ps = ps * z + 0x1.71de3a556c734p-19;
        ps = ps * z + -0x1.a01a01a01a01ap-13;
        ps = ps * z + 0x1.1111111111111p-7;
        ps = ps * z + -0x1.5555555555555p-3;
        const double sr = r + r * (z * ps);
        double pc = 0x1.e542ba4020225p-62;             /* 1/20! */
        pc = pc * z + -0x1.6827863b97d97p-53;
        pc = pc * z + 0x1.ae7f3e733b81fp-45;
        pc = pc * z + -0x1.93974a8c07c9dp-37;
        pc = pc * z + 0x1.1eed8eff8d898p-29;
        pc = pc * z + -0x1.27e4fb7789f5cp-22;
        pc = pc * z + 0x1.a01a01a01a01ap-16;
        pc = pc * z + -0x1.6c16c16c16c17p-10;
        pc = pc * z + 0x1.5555555555555p-5;
        const double cr = 1.0 + z * (-0.5 + z * pc);
        double sv, cv;
        switch (k & 3) {
        case 0: sv = sr; cv = cr; break;
        case 1: sv = cr; cv = -sr; break;
        case 2: sv = -sr; cv = -cr; break;
        default: sv = -cr; cv = sr; break;
        }
        const float sf = (float)sv, cf = (float)cv;
        const double up = 1.0 + 0x1p-47, dn = 1.0 - 0x1p-47;
        if ((float)(sv * up) == sf && (float)(sv * dn) == sf && (float)(cv * up) == cf && (float)(cv * dn) == cf) {
            *s_out = sf; *c_out = cf;
            return 1;
        }
    }
    double s, c;
    const int ok = drfe_cr_sincos(x, &s, &c);
    *s_out = (float)s; *c_out = (float)c;
    return ok;
}

#endif
