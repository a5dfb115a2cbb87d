// flvis_amd: deterministic fp64 elementary functions shared by the HIP kernels and the CPU oracle.
//
// The closed-loop front-end is a chaotic recurrence in the discrete sense: a 1-ulp difference in a pose flips, sooner or
// later, a float cast of an LK start point or a threshold test, and from then on two correct runs track different landmark
// sets.  Device libm (ocml) and glibc agree to ~1 ulp, not bit for bit, so "feature indices bit-exact" for the PATH (not just
// per kernel) needs one definition of sin / cos / atan / atan2 / log that both sides execute with the same IEEE operations
// in the same order.  These are the classic fdlibm algorithms (Sun Microsystems' freely distributable libm: argument
// reduction by Cody-Waite with a three-part pi/2, minimax polynomials), written with + - * / and bit moves only, and compiled
// on both sides with -ffp-contract=off.  Accuracy: < 1 ulp on the ranges the path uses (tests/test_det_math.py checks them
// against libm); domain of sin/cos: |x| < 2^19 * pi/2 (rotation angles and Euler angles: |x| <= 2 pi here).
//
// pow() in the reference appears twice on the path with small integer exponents (g2o's Levenberg step control
// `1 - pow(2 rho - 1, 3)`, optimization_algorithm_levenberg.cpp:124; cv::RANSACUpdateNumIters `pow(1 - ep, modelPoints)`):
// det_powi multiplies, left to right.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define DETM_FN __host__ __device__ inline
#else
#define DETM_FN inline
#endif

namespace detm {

DETM_FN uint64_t dbits(double x) {
  union {
    double d;
    uint64_t u;
  } v;
  v.d = x;
  return v.u;
}
DETM_FN double from_dbits(uint64_t u) {
  union {
    double d;
    uint64_t u;
  } v;
  v.u = u;
  return v.d;
}
DETM_FN int32_t dhi(double x) { return (int32_t)(dbits(x) >> 32); }
DETM_FN uint32_t dlo(double x) { return (uint32_t)dbits(x); }
DETM_FN double with_hi(double x, int32_t hi) { return from_dbits(((uint64_t)(uint32_t)hi << 32) | (dbits(x) & 0xffffffffull)); }
DETM_FN double dabs(double x) { return from_dbits(dbits(x) & 0x7fffffffffffffffull); }

// sin on [-pi/4, pi/4]; y is the tail of x (k_sin.c)
DETM_FN double k_sin(double x, double y, int iy) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const int32_t ix = dhi(x) & 0x7fffffff;
  if (ix < 0x3e400000) {  // |x| < 2^-27
    if ((int)x == 0) return x;
  }
  const double z = x * x;
  const double v = z * x;
  const double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  if (iy == 0) return x + v * (S1 + z * r);
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

// cos on [-pi/4, pi/4] (k_cos.c)
DETM_FN double k_cos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const int32_t ix = dhi(x) & 0x7fffffff;
  if (ix < 0x3e400000) {
    if ((int)x == 0) return 1.0;
  }
  const double z = x * x;
  const double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  if (ix < 0x3FD33333) return 1.0 - (0.5 * z - (z * r - x * y));
  double qx;
  if (ix > 0x3fe90000)
    qx = 0.28125;
  else
    qx = from_dbits((uint64_t)(uint32_t)(ix - 0x00200000) << 32);  // x / 4
  const double hz = 0.5 * z - qx;
  const double a = 1.0 - qx;
  return a - (hz - (z * r - x * y));
}

// x = n * pi/2 + (y0 + y1), |y0 + y1| <= pi/4; returns n (e_rem_pio2.c, the Cody-Waite path: |x| < 2^19 * pi/2)
DETM_FN int rem_pio2(double x, double& y0, double& y1) {
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11,
               pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21, pio2_3 = 2.02226624871116645580e-21,
               pio2_3t = 8.47842766036889956997e-32;
  const int32_t hx = dhi(x);
  const int32_t ix = hx & 0x7fffffff;
  const double t = dabs(x);
  const int n = (int)(t * invpio2 + 0.5);
  const double fn = (double)n;
  double r = t - fn * pio2_1;
  double w = fn * pio2_1t;  // 1st round, good to 85 bits
  const int32_t j = ix >> 20;
  y0 = r - w;
  int32_t i = j - ((dhi(y0) >> 20) & 0x7ff);
  if (i > 16) {  // 2nd iteration needed, good to 118 bits
    double t2 = r;
    w = fn * pio2_2;
    r = t2 - w;
    w = fn * pio2_2t - ((t2 - r) - w);
    y0 = r - w;
    i = j - ((dhi(y0) >> 20) & 0x7ff);
    if (i > 49) {  // 3rd iteration, 151 bits
      t2 = r;
      w = fn * pio2_3;
      r = t2 - w;
      w = fn * pio2_3t - ((t2 - r) - w);
      y0 = r - w;
    }
  }
  y1 = (r - y0) - w;
  if (hx < 0) {
    y0 = -y0;
    y1 = -y1;
    return -n;
  }
  return n;
}

DETM_FN double det_sin(double x) {
  const int32_t ix = dhi(x) & 0x7fffffff;
  if (ix <= 0x3fe921fb) return k_sin(x, 0.0, 0);
  if (ix >= 0x7ff00000) return x - x;
  double y0, y1;
  const int n = rem_pio2(x, y0, y1);
  switch (n & 3) {
    case 0: return k_sin(y0, y1, 1);
    case 1: return k_cos(y0, y1);
    case 2: return -k_sin(y0, y1, 1);
    default: return -k_cos(y0, y1);
  }
}

DETM_FN double det_cos(double x) {
  const int32_t ix = dhi(x) & 0x7fffffff;
  if (ix <= 0x3fe921fb) return k_cos(x, 0.0);
  if (ix >= 0x7ff00000) return x - x;
  double y0, y1;
  const int n = rem_pio2(x, y0, y1);
  switch (n & 3) {
    case 0: return k_cos(y0, y1);
    case 1: return -k_sin(y0, y1, 1);
    case 2: return -k_cos(y0, y1);
    default: return k_sin(y0, y1, 1);
  }
}

// s_atan.c
DETM_FN double det_atan(double x) {
  const double atanhi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01, 9.82793723247329054082e-01,
                            1.57079632679489655800e+00};
  const double atanlo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17, 1.39033110312309984516e-17,
                            6.12323399573676603587e-17};
  const double aT[11] = {3.33333333333329318027e-01,  -1.99999999998764832476e-01, 1.42857142725034663711e-01,
                         -1.11111104054623557880e-01, 9.09088713343650656196e-02,  -7.69187620504482999495e-02,
                         6.66107313738753120669e-02,  -5.83357013379057348645e-02, 4.97687799461593236017e-02,
                         -3.65315727442169155270e-02, 1.62858201153657823623e-02};
  const int32_t hx = dhi(x);
  const int32_t ix = hx & 0x7fffffff;
  int id;
  if (ix >= 0x44100000) {  // |x| >= 2^66
    if (ix > 0x7ff00000 || (ix == 0x7ff00000 && dlo(x) != 0)) return x + x;  // NaN
    return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3fdc0000) {  // |x| < 0.4375
    if (ix < 0x3e200000) return x;  // |x| < 2^-29
    id = -1;
  } else {
    x = dabs(x);
    if (ix < 0x3ff30000) {    // |x| < 1.1875
      if (ix < 0x3fe60000) {  // 7/16 <= |x| < 11/16
        id = 0;
        x = (2.0 * x - 1.0) / (2.0 + x);
      } else {  // 11/16 <= |x| < 19/16
        id = 1;
        x = (x - 1.0) / (x + 1.0);
      }
    } else {
      if (ix < 0x40038000) {  // |x| < 2.4375
        id = 2;
        x = (x - 1.5) / (1.0 + 1.5 * x);
      } else {  // 2.4375 <= |x| < 2^66
        id = 3;
        x = -1.0 / x;
      }
    }
  }
  double z = x * x;
  const double w = z * z;
  const double s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  const double s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
  return hx < 0 ? -z : z;
}

// e_atan2.c
DETM_FN double det_atan2(double y, double x) {
  const double tiny = 1.0e-300, pi_o_4 = 7.8539816339744827900E-01, pi_o_2 = 1.5707963267948965580E+00,
               pi = 3.1415926535897931160E+00, pi_lo = 1.2246467991473531772E-16;
  const int32_t hx = dhi(x), hy = dhi(y);
  const uint32_t lx = dlo(x), ly = dlo(y);
  const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if (((uint32_t)ix | ((lx | (0u - lx)) >> 31)) > 0x7ff00000u || ((uint32_t)iy | ((ly | (0u - ly)) >> 31)) > 0x7ff00000u)
    return x + y;                                                        // x or y is NaN
  if ((((uint32_t)hx - 0x3ff00000u) | lx) == 0) return det_atan(y);  // x = 1.0
  const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);                 // 2 * sign(x) + sign(y)
  if (((uint32_t)iy | ly) == 0) {                                    // y = 0
    switch (m) {
      case 0:
      case 1: return y;
      case 2: return pi + tiny;
      default: return -pi - tiny;
    }
  }
  if (((uint32_t)ix | lx) == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;  // x = 0
  if (ix == 0x7ff00000) {                                                          // x is INF
    if (iy == 0x7ff00000) {
      switch (m) {
        case 0: return pi_o_4 + tiny;
        case 1: return -pi_o_4 - tiny;
        case 2: return 3.0 * pi_o_4 + tiny;
        default: return -3.0 * pi_o_4 - tiny;
      }
    } else {
      switch (m) {
        case 0: return 0.0;
        case 1: return -0.0;
        case 2: return pi + tiny;
        default: return -pi - tiny;
      }
    }
  }
  if (iy == 0x7ff00000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;  // y is INF
  const int32_t k = (iy - ix) >> 20;
  double z;
  if (k > 60)
    z = pi_o_2 + 0.5 * pi_lo;  // |y/x| > 2^60
  else if (hx < 0 && k < -60)
    z = 0.0;  // |y|/x < -2^60
  else
    z = det_atan(dabs(y / x));
  switch (m) {
    case 0: return z;                  // atan(+,+)
    case 1: return -z;                 // atan(-,+)
    case 2: return pi - (z - pi_lo);   // atan(+,-)
    default: return (z - pi_lo) - pi;  // atan(-,-)
  }
}

// e_log.c
DETM_FN double det_log(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10, two54 = 1.80143985094819840000e+16,
               Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  int32_t hx = dhi(x);
  const uint32_t lx = dlo(x);
  int32_t k = 0;
  if (hx < 0x00100000) {  // x < 2^-1022
    if ((((uint32_t)hx & 0x7fffffffu) | lx) == 0) return -two54 / 0.0;  // log(+-0) = -inf
    if (hx < 0) return (x - x) / 0.0;                                     // log(-#) = NaN
    k -= 54;
    x *= two54;  // subnormal: scale up
    hx = dhi(x);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  int32_t i = (hx + 0x95f64) & 0x100000;
  x = with_hi(x, hx | (i ^ 0x3ff00000));  // normalise x or x/2
  k += (i >> 20);
  const double f = x - 1.0;
  if ((0x000fffff & (2 + hx)) < 3) {  // |f| < 2^-20
    if (f == 0.0) {
      if (k == 0) return 0.0;
      const double dk = (double)k;
      return dk * ln2_hi + dk * ln2_lo;
    }
    const double R = f * f * (0.5 - 0.33333333333333333 * f);
    if (k == 0) return f - R;
    const double dk = (double)k;
    return dk * ln2_hi - ((R - dk * ln2_lo) - f);
  }
  const double s = f / (2.0 + f);
  const double dk = (double)k;
  const double z = s * s;
  i = hx - 0x6147a;
  const double w = z * z;
  const int32_t j = 0x6b851 - hx;
  const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  i |= j;
  const double R = t2 + t1;
  if (i > 0) {
    const double hfsq = 0.5 * f * f;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  }
  if (k == 0) return f - s * (f - R);
  return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

// x^n for a small non-negative integer n: n - 1 multiplications, left to right
DETM_FN double det_powi(double x, int n) {
  if (n <= 0) return 1.0;
  double r = x;
  for (int i = 1; i < n; i++) r = r * x;
  return r;
}

// e_acos.c (cv::solveCubic's three-real-root branch and cv::p3p's solve_deg3 take the arc cosine of R / sqrt(Q^3))
DETM_FN double det_acos(double x) {
  const double one = 1.0, pi = 3.14159265358979311600e+00, pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
               pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
               pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
               qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
               qS4 = 7.70381505559019352791e-02;
  const int32_t hx = dhi(x);
  const int32_t ix = hx & 0x7fffffff;
  if (ix >= 0x3ff00000) {  // |x| >= 1
    if (((uint32_t)(ix - 0x3ff00000) | dlo(x)) == 0) return hx > 0 ? 0.0 : pi + 2.0 * pio2_lo;
    return (x - x) / (x - x);  // NaN
  }
  if (ix < 0x3fe00000) {  // |x| < 0.5
    if (ix <= 0x3c600000) return pio2_hi + pio2_lo;
    const double z = x * x;
    const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const double q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const double r = p / q;
    return pio2_hi - (x - (pio2_lo - r * x));
  }
  if (hx < 0) {  // x < -0.5
    const double z = (one + x) * 0.5;
    const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const double q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const double s = sqrt(z);
    const double r = p / q;
    const double w = r * s - pio2_lo;
    return pi - 2.0 * (s + w);
  }
  const double z = (one - x) * 0.5;  // x > 0.5
  const double s = sqrt(z);
  const double df = from_dbits(dbits(s) & 0xffffffff00000000ull);
  const double c = (z - df * df) / (s + df);
  const double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  const double q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  const double r = p / q;
  const double w = r * s + c;
  return 2.0 * (df + w);
}

// s_cbrt.c (normal numbers; 0, inf and NaN returned as they are)
DETM_FN double det_cbrt(double x) {
  const uint32_t B1 = 715094163u;
  const double C = 5.42857142857142815906e-01, D = -7.05306122448979611050e-01, E = 1.41428571428571436819e+00,
               F = 1.60714285714285720630e+00, G = 3.57142857142857150787e-01;
  int32_t hx = dhi(x);
  const uint32_t sign = (uint32_t)hx & 0x80000000u;
  hx ^= (int32_t)sign;
  if (hx >= 0x7ff00000) return x + x;
  if (((uint32_t)hx | dlo(x)) == 0) return x;
  x = with_hi(x, hx);  // |x|
  double t;
  if (hx < 0x00100000) {  // subnormal
    t = with_hi(0.0, 0x43500000);
    t *= x;
    t = with_hi(t, (int32_t)((uint32_t)dhi(t) / 3u + 696219795u));
  } else {
    t = with_hi(0.0, (int32_t)((uint32_t)hx / 3u + B1));
  }
  double r = t * t / x;
  double s = C + r * t;
  t *= G + F / (s + E + D / s);
  t = from_dbits(((uint64_t)(uint32_t)(dhi(t) + 1)) << 32);  // chopped to 20 bits, made larger than cbrt(x)
  s = t * t;  // exact
  r = x / s;
  const double w = t + t;
  r = (r - t) / (w + r);
  t = t + t * r;
  return from_dbits(dbits(t) | ((uint64_t)sign << 32));
}

}  // namespace detm
