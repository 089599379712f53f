/* ahc_math_simd.h — ahc_plane_from_sums (ahc_math.h) for W trial merges at once on the HOST's vector unit.
 *
 * ahCluster tries ~21 neighbour merges per step, 31 600 per 640x480 frame, each a 3x3 symmetric eigen-solve whose cost is its
 * dependent chain of divisions and square roots (planes_ahc.cpp).  The trials of a step are independent, so W of them run in
 * the lanes of one vector: EVERY LANE EXECUTES THE SCALAR ROUTINE'S OPERATION SEQUENCE - the same IEEE + - * / sqrt in the
 * same order, no FMA contraction (the translation unit is built -ffp-contract=off and the FMA feature is not enabled) - and
 * the scalar routine's branches become lane masks, so the results are bit-identical to ahc_plane_from_sums
 * (tests/test_host_cpu.py compares them on random and degenerate inputs).  Lanes that have converged idle through the remaining
 * QL sweeps of the slowest lane; operations of a branch a lane does not take may divide by zero in that lane - the result is
 * discarded by the blend and floating-point exceptions are masked.
 *
 * Written with the compilers' generic vector types (no intrinsics): instantiated for W = 8 under target("avx2") (two registers per value) and under
 * target("avx512f"), chosen at run time; any other host keeps the scalar path. */
#ifndef DRFE_AHC_MATH_SIMD_H
#define DRFE_AHC_MATH_SIMD_H

#include "ahc_math.h"
#include <stdint.h>

namespace ahc_simd {

template <int W> struct Vec;
template <> struct Vec<4> {
    typedef double d __attribute__((vector_size(32)));
    typedef long long i __attribute__((vector_size(32)));
};
template <> struct Vec<8> {
    typedef double d __attribute__((vector_size(64)));
    typedef long long i __attribute__((vector_size(64)));
};

#define AHC_SIMD_INLINE static inline __attribute__((always_inline))

template <int W> struct Ops {
    typedef typename Vec<W>::d vd;
    typedef typename Vec<W>::i vi;
    AHC_SIMD_INLINE vd splat(double x) { vd r; for (int k = 0; k < W; k++) r[k] = x; return r; }
    AHC_SIMD_INLINE vi splati(long long x) { vi r; for (int k = 0; k < W; k++) r[k] = x; return r; }
    /* m ? a : b per lane (m is all-ones / all-zeros per lane) */
    AHC_SIMD_INLINE vd sel(vi m, vd a, vd b) { return (vd)(((vi)a & m) | ((vi)b & ~m)); }
    AHC_SIMD_INLINE vi seli(vi m, vi a, vi b) { return (a & m) | (b & ~m); }
    AHC_SIMD_INLINE vd vabs(vd a) { return (vd)((vi)a & splati(0x7FFFFFFFFFFFFFFFll)); }
    AHC_SIMD_INLINE vd vsqrt(vd a) { vd r; for (int k = 0; k < W; k++) r[k] = __builtin_sqrt(a[k]); return r; }   /* one vsqrtpd */
    AHC_SIMD_INLINE vd vmax(vd a, vd b) { return sel(a < b, b, a); }                                      /* fmax on non-NaN operands */
    AHC_SIMD_INLINE bool any(vi m) { long long r = 0; for (int k = 0; k < W; k++) r |= m[k]; return r != 0; }
};

/* ahc_givens for W lanes: (c, s) of the rotation that zeroes q against p; the four scalar branches as blends */
template <int W>
AHC_SIMD_INLINE void givens(typename Vec<W>::d p, typename Vec<W>::d q, typename Vec<W>::d* c, typename Vec<W>::d* s)
{
    typedef Ops<W> O; typedef typename Vec<W>::d vd; typedef typename Vec<W>::i vi;
    const vd zero = O::splat(0.0), one = O::splat(1.0), mone = O::splat(-1.0);
    const vi qz = q == zero, pz = p == zero, pbig = O::vabs(p) > O::vabs(q);
    /* |p| > |q|: t = q / p, u = +-sqrt(1 + t^2) (sign of p), c = 1 / u, s = -t c;  else: t = p / q, u signed by q, s = -1 / u,
     * c = -t s.  One division, one square root and one reciprocal serve both branches: the operands are blended first, the
     * operations and their order per lane are the scalar routine's */
    const vd num = O::sel(pbig, q, p), den = O::sel(pbig, p, q);
    const vd t = num / den;
    vd u = O::vsqrt(one + t * t);
    u = O::sel(den < zero, -u, u);
    const vd r = O::sel(pbig, one, mone) / u;      /* c of the first branch, s of the second */
    const vd w = -t * r;                           /* s of the first branch, c of the second */
    vd cc = O::sel(pbig, r, w), ss = O::sel(pbig, w, r);
    cc = O::sel(pz, zero, cc); ss = O::sel(pz, O::sel(q < zero, one, mone), ss);
    cc = O::sel(qz, O::sel(p < zero, mone, one), cc); ss = O::sel(qz, zero, ss);
    *c = cc; *s = ss;
}

/* ahc_eig3 for W lanes: ev[3] ascending, Q[9] column-major */
template <int W>
AHC_SIMD_INLINE void eig3(typename Vec<W>::d m00, typename Vec<W>::d m10, typename Vec<W>::d m20, typename Vec<W>::d m11,
                          typename Vec<W>::d m21, typename Vec<W>::d m22, typename Vec<W>::d ev[3], typename Vec<W>::d Q[9])
{
    typedef Ops<W> O; typedef typename Vec<W>::d vd; typedef typename Vec<W>::i vi;
    const vd zero = O::splat(0.0), one = O::splat(1.0), half = O::splat(0.5), two = O::splat(2.0);
    const vd tiny = O::splat(2.2250738585072014e-308), prec = O::splat(2.0 * 2.220446049250313e-16);
    vd scale = O::vabs(m00);
    scale = O::vmax(scale, O::vabs(m10)); scale = O::vmax(scale, O::vabs(m20)); scale = O::vmax(scale, O::vabs(m11));
    scale = O::vmax(scale, O::vabs(m21)); scale = O::vmax(scale, O::vabs(m22));
    scale = O::sel(scale == zero, one, scale);
    m00 /= scale; m10 /= scale; m20 /= scale; m11 /= scale; m21 /= scale; m22 /= scale;
    vd d[3], e[2];
    d[0] = m00;
    const vd v1norm2 = m20 * m20;
    const vi noHouse = v1norm2 <= tiny;
    {
        const vd beta = O::vsqrt(m10 * m10 + v1norm2);
        const vd invBeta = one / beta;
        const vd m01 = m10 * invBeta, m02 = m20 * invBeta;
        const vd q = two * m01 * m21 + m02 * (m22 - m11);
        d[1] = O::sel(noHouse, m11, m11 + m02 * q);
        d[2] = O::sel(noHouse, m22, m22 - m02 * q);
        e[0] = O::sel(noHouse, m10, beta);
        e[1] = O::sel(noHouse, m21, m21 - m01 * q);
        for (int i = 0; i < 9; i++) Q[i] = zero;
        Q[0] = one;
        Q[4] = O::sel(noHouse, one, m01); Q[5] = O::sel(noHouse, zero, m02);
        Q[7] = O::sel(noHouse, zero, m02); Q[8] = O::sel(noHouse, one, -m01);
    }
    /* per-lane loop state of `while (end > 0)`: end in {2, 1, 0}, iter, live */
    vi end = O::splati(2), iter = O::splati(0), live = O::splati(-1);
    const vi i0 = O::splati(0), i1 = O::splati(1), i2 = O::splati(2), i90 = O::splati(90);
    vi start = i0;
    for (int sweep = 0; sweep < 92 && O::any(live); sweep++) {
        /* deflation of e[i], start <= i < end */
        for (int i = 0; i < 2; i++) {
            const vi in = live & (start <= O::splati(i)) & (end > O::splati(i));
            const vd ae = O::vabs(e[i]);
            const vi small = (ae <= (O::vabs(d[i]) + O::vabs(d[i + 1])) * prec) | (ae <= tiny);
            e[i] = O::sel(in & small, zero, e[i]);
        }
        /* while (end > 0 && e[end - 1] == 0) end-- */
        end = O::seli(live & (end == i2) & (e[1] == zero), i1, end);
        end = O::seli(live & (end == i1) & (e[0] == zero), i0, end);
        live = live & (end > i0);
        iter = O::seli(live, iter + i1, iter);
        live = live & ~(iter > i90);                                   /* `if (iter > 90) break`: iter stays 91 */
        if (!O::any(live)) break;
        /* start = end - 1; while (start > 0 && e[start - 1] != 0) start-- */
        start = O::seli(live, O::seli((end == i2) & (e[0] == zero), i1, i0), start);
        const vi end2 = end == i2;
        /* Wilkinson shift from the trailing 2x2 block of [start, end] */
        const vd dEm1 = O::sel(end2, d[1], d[0]), dE = O::sel(end2, d[2], d[1]), ee = O::sel(end2, e[1], e[0]);
        const vd td = (dEm1 - dE) * half;
        vd mu;
        {
            const vd e2 = ee * ee, ax = O::vabs(td), ay = O::vabs(ee);
            const vi axBig = ax > ay;
            const vd p = O::sel(axBig, ax, ay), qp = O::sel(axBig, ay, ax) / p;
            const vd h = O::sel(p == zero, zero, p * O::vsqrt(one + qp * qp));
            const vi tdPos = td > zero, e2z = e2 == zero;
            const vd muA = dE - O::vabs(ee);                                                        /* td == 0 */
            /* e2 == 0: mu -= (ee / (td +- 1)) * (ee / h);  else: mu -= e2 / (td +- h): one blended division */
            const vd hs = O::sel(e2z, one, h);
            const vd quo = O::sel(e2z, ee, e2) / (td + O::sel(tdPos, hs, -hs));
            const vd muB = dE - quo * (ee / h);
            const vd muC = dE - quo;
            mu = O::sel(td == zero, muA, O::sel(e2 == zero, muB, muC));
        }
        const vi start0 = start == i0;
        vd x = O::sel(start0, d[0], d[1]) - mu, z = O::sel(start0, e[0], e[1]);
        /* k = 0 (lanes with start == 0), then k = 1 (lanes with end == 2) */
        for (int k = 0; k < 2; k++) {
            const vi act = live & (k == 0 ? start0 : end2);
            if (!O::any(act)) continue;                                /* no lane rotates at this k: every update below is blended by act */
            vd c, s;
            givens<W>(x, z, &c, &s);
            const vd sdk = s * d[k] + c * e[k];
            const vd dkp1 = s * e[k] + c * d[k + 1];
            const vd ndk = c * (c * d[k] - s * e[k]) - s * (c * e[k] - s * d[k + 1]);
            const vd ndk1 = s * sdk + c * dkp1;
            const vd nek = c * sdk - s * dkp1;
            d[k] = O::sel(act, ndk, d[k]); d[k + 1] = O::sel(act, ndk1, d[k + 1]); e[k] = O::sel(act, nek, e[k]);
            if (k == 1) {                                              /* if (k > start) e[k - 1] = c * e[k - 1] - s * z */
                const vi m = act & start0;
                e[0] = O::sel(m, c * e[0] - s * z, e[0]);
            }
            x = O::sel(act, e[k], x);
            if (k == 0) {                                              /* if (k < end - 1) { z = -s * e[k + 1]; e[k + 1] = c * e[k + 1]; } */
                const vi m = act & end2;
                z = O::sel(m, -s * e[1], z);
                e[1] = O::sel(m, c * e[1], e[1]);
            }
            for (int i = 0; i < 3; i++) {
                const vd xi = Q[k * 3 + i], yi = Q[(k + 1) * 3 + i];
                Q[k * 3 + i] = O::sel(act, c * xi - s * yi, xi);
                Q[(k + 1) * 3 + i] = O::sel(act, s * xi + c * yi, yi);
            }
        }
    }
    /* ascending sort of the converged lanes (iter <= 90) */
    const vi sortable = ~(iter > i90);
    for (int i = 0; i < 2; ++i) {
        /* k = index (relative to i) of the least of d[i..2], first minimum wins */
        vd best = d[i];
        vi kk = i0;
        for (int j = 1; j < 3 - i; j++) {
            const vi lt = d[i + j] < best;
            best = O::sel(lt, d[i + j], best);
            kk = O::seli(lt, O::splati(j), kk);
        }
        for (int j = 1; j < 3 - i; j++) {
            const vi sw = sortable & (kk == O::splati(j));
            const vd t = d[i];
            d[i] = O::sel(sw, d[i + j], d[i]); d[i + j] = O::sel(sw, t, d[i + j]);
            for (int r = 0; r < 3; r++) {
                const vd u = Q[i * 3 + r];
                Q[i * 3 + r] = O::sel(sw, Q[(i + j) * 3 + r], Q[i * 3 + r]);
                Q[(i + j) * 3 + r] = O::sel(sw, u, Q[(i + j) * 3 + r]);
            }
        }
    }
    for (int i = 0; i < 3; i++) ev[i] = d[i] * scale;
}

/* ahc_plane_from_sums for W trials: S[k][lane] (k = 0..8: sx sy sz sxx syy szz sxy syz sxz), N[lane] -> fits */
template <int W>
AHC_SIMD_INLINE void plane_from_sums(const double S[9][W], const int N[W], AhcFit out[W])
{
    typedef Ops<W> O; typedef typename Vec<W>::d vd; typedef typename Vec<W>::i vi;
    vd s[9], n;
    for (int k = 0; k < 9; k++)
        for (int l = 0; l < W; l++) s[k][l] = S[k][l];
    for (int l = 0; l < W; l++) n[l] = (double)N[l];
    const vd sc = O::splat(1.0) / n;
    const vd cx = s[0] * sc, cy = s[1] * sc, cz = s[2] * sc;
    const vd k00 = s[3] - s[0] * s[0] * sc, k01 = s[6] - s[0] * s[1] * sc, k02 = s[8] - s[0] * s[2] * sc;
    const vd k11 = s[4] - s[1] * s[1] * sc, k12 = s[7] - s[1] * s[2] * sc, k22 = s[5] - s[2] * s[2] * sc;
    vd ev[3], Q[9];
    eig3<W>(k00, k01, k02, k11, k12, k22, ev, Q);
    const vd v0 = Q[0], v1 = Q[1], v2 = Q[2];
    const vi keep = (v0 * cx + v1 * cy + v2 * cz) <= O::splat(0.0);
    const vd nx = O::sel(keep, v0, -v0), ny = O::sel(keep, v1, -v1), nz = O::sel(keep, v2, -v2);
    const vd mse = ev[0] * sc, curv = ev[0] / (ev[0] + ev[1] + ev[2]);
    for (int l = 0; l < W; l++) {
        out[l].center[0] = cx[l]; out[l].center[1] = cy[l]; out[l].center[2] = cz[l];
        out[l].normal[0] = nx[l]; out[l].normal[1] = ny[l]; out[l].normal[2] = nz[l];
        out[l].mse = mse[l]; out[l].curvature = curv[l];
    }
}

}  // namespace ahc_simd
#endif
