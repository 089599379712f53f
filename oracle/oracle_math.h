/* oracle/oracle_math.h — TEST INFRASTRUCTURE.  The oracle's OWN scalar primitives: nothing here is shared with the product
 * (include/drfe_math.h is compiled into libdrfe.so only), so a wrong coefficient or rounding rule on either side shows up
 * as a HIP-vs-oracle mismatch instead of cancelling out.
 *
 *   round_he            cvRound(): round half to even, spelled out with floor + the tie rule (the product uses rintf)
 *   fast_atan2_deg       cv::fastAtan2 of OpenCV 3.4 (core/src/mathfuncs_core.simd.hpp, atan_f32): the same published
 *                        polynomial - it is the definition - evaluated from a coefficient table
 *   sincos_f, log_f the float results the reference takes from glibc's cosf/sinf/logf are host dependent in the last
 *                        bit; canonical value on both sides = the exactly rounded one.  Here: glibc's DOUBLE cos/sin/log
 *                        rounded once to float (the product evaluates its own double polynomials); the two agree unless a
 *                        double result falls within ~1e-16 of a float rounding boundary.
 */
#ifndef ORC_ORACLE_MATH_H
#define ORC_ORACLE_MATH_H
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <emmintrin.h>

namespace orc {

/* cvRound as OpenCV 3.4 defines it on x86-64 (core/include/opencv2/core/fast_math.hpp): the SSE conversion instruction,
 * round-half-to-even under the default MXCSR mode */
inline int round_he_d(double v) { return _mm_cvtsd_si32(_mm_set_sd(v)); }
inline int round_he(float v) { return _mm_cvtss_si32(_mm_set_ss(v)); }

inline float fast_atan2_deg(float y, float x)
{
    static const float k = (float)(180.0 / 3.1415926535897932384626433832795);
    static const float poly[4] = {-0.04432655554792128f * k, 0.1555786518463281f * k, -0.3258083974640975f * k,
                                  0.9997878412794807f * k};                  /* p7, p5, p3, p1 in degrees */
    const float ax = std::fabs(x), ay = std::fabs(y);
    const bool steep = !(ax >= ay);
    const float num = steep ? ax : ay, den = (steep ? ay : ax) + (float)DBL_EPSILON;
    const float c = num / den, c2 = c * c;
    float acc = poly[0];
    for (int i = 1; i < 4; i++) acc = acc * c2 + poly[i];
    float a = acc * c;
    if (steep) a = 90.f - a;
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

inline void sincos_f(float rad, float* s, float* c)
{
    *s = (float)std::sin((double)rad);
    *c = (float)std::cos((double)rad);
}

inline float log_f(float x) { return (float)std::log((double)x); }

}  // namespace orc
#endif
