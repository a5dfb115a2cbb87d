// flvis_amd: the minimal solvers inside the two RANSACs of LKORBTracking::tracking (lkorb_tracking.cpp:134-135, 170-177), restated from
// the algorithms OpenCV 3.x runs there -- one header for the HIP kernels and for the CPU checker of the tests, plain C++ with + - * / sqrt
// and the det_math.hpp functions, compiled with -ffp-contract=off on both sides, so that both execute the same IEEE operations in the
// same order.  No OpenCV exists in this image: what is restated is the published structure of its code (calib3d/src/fundam.cpp
// run7Point, core/src/lapack.cpp JacobiSVDImpl_, core/src/mathfuncs.cpp solveCubic, calib3d/src/p3p.cpp + polynom_solver.cpp) -- the
// order of the operations, the stopping rules, the order in which solutions are reported; it cannot be pinned on the library itself.
//
//   run7point      A (7 x 9, rows (x2 x1, x2 y1, x2, y2 x1, y2 y1, y2, x1, y1, 1) from the Point2f coordinates, unnormalised) -> its two
//                  null vectors as cv::SVDecomp(A, FULL_UV) produces them: one-sided Jacobi on the rows of A (cyclic pairs (i, j), rotation
//                  from gamma = hypot(2p, a - b), stop after a sweep without a rotation above eps = 10 DBL_EPSILON, at most 30 sweeps),
//                  rows sorted by decreasing norm, and rows 7 and 8 of Vt completed the way JacobiSVDImpl_ completes a basis: +-1/9
//                  entries drawn from cv::RNG(0x12345678), two passes of projection against all earlier rows with an L1 renormalisation,
//                  then unit L2 norm.  f1 = Vt[7], f2 = Vt[8]; det(lambda f1 + (1 - lambda) f2) = 0 -> cubic -> solveCubic; for every
//                  root F = lambda f1 + mu f2 scaled to F[8] = 1.
//   solve_cubic    cv::solveCubic (OpenCV 3.2 form: `d >= 0` -> three roots via acos / cos, else one via pow(., 0.333333333333))
//   p3p_solve      cv::p3p::solve (Gao, Hou, Tang, Chang 2003): quartic in x = |PA| / |PC| by Ferrari's method (polynom_solver.cpp:
//                  solve_deg4 / solve_deg3 / solve_deg2), y from the b1 / b0 closed form, lengths, absolute orientation by Horn's
//                  quaternion method with the 4 x 4 cyclic Jacobi eigen-solver of p3p.cpp; the four-point form picks the pose that
//                  reprojects the fourth point best.
//
// Departures that remain (stated in the checker's README): std::hypot is cvs::hypot2 below (scaled sqrt, not glibc's), acos / cos / cbrt are
// det_math's fdlibm forms, pow(x, 0.333333333333) is cbrt(x) (1 + (0.333333333333 - 1/3) log x) (first order; the neglected term is 1e-25),
// pow(x, 1/3.) in solve_deg3 is cbrt(x).
#pragma once
#include <float.h>
#include <math.h>
#include <stdint.h>

#include "det_math.hpp"

#if defined(__HIPCC__)
#define CVS_FN __host__ __device__ inline
#else
#define CVS_FN inline
#endif

#if defined(__clang__)
#define CVS_UNROLL _Pragma("unroll")
#else
#define CVS_UNROLL _Pragma("GCC unroll 9")
#endif

namespace flvis {
namespace cvs {

constexpr double kPi = 3.1415926535897932384626433832795;  // CV_PI

// cv::RNG::next (core.hpp: the multiply-with-carry generator)
CVS_FN uint32_t rng_next(uint64_t& state) {
  state = (uint64_t)(uint32_t)state * 4164903690u + (uint32_t)(state >> 32);
  return (uint32_t)state;
}

// stand-in for std::hypot (see the header)
CVS_FN double hypot2(double a, double b) {
  a = fabs(a);
  b = fabs(b);
  if (a < b) {
    const double t = a;
    a = b;
    b = t;
  }
  if (a == 0) return 0;
  const double r = b / a;
  return a * sqrt(1 + r * r);
}

// ------------------------------------------------------------------------------------------------ cv::solveCubic
// coefficients c[0] x^3 + c[1] x^2 + c[2] x + c[3]; returns the number of roots written to r[0..2] (-1: every x is a root)
CVS_FN int solve_cubic(const double* c, double* r) {
  double a0 = c[0], a1 = c[1], a2 = c[2], a3 = c[3];
  double x0 = 0., x1 = 0., x2 = 0.;
  int n = 0;
  if (a0 == 0) {
    if (a1 == 0) {
      if (a2 == 0) {
        n = a3 == 0 ? -1 : 0;
      } else {
        x0 = -a3 / a2;  // linear equation
        n = 1;
      }
    } else {
      double d = a2 * a2 - 4 * a1 * a3;  // quadratic equation
      if (d >= 0) {
        d = sqrt(d);
        const double q1 = (-a2 + d) * 0.5;
        const double q2 = (a2 + d) * -0.5;
        if (fabs(q1) > fabs(q2)) {
          x0 = q1 / a1;
          x1 = a3 / q1;
        } else {
          x0 = q2 / a1;
          x1 = a3 / q2;
        }
        n = d > 0 ? 2 : 1;
      }
    }
  } else {
    a0 = 1. / a0;
    a1 *= a0;
    a2 *= a0;
    a3 *= a0;
    const double Q = (a1 * a1 - 3 * a2) * (1. / 9);
    const double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
    const double Qcubed = Q * Q * Q;
    double d = Qcubed - R * R;
    if (d >= 0) {
      const double theta = detm::det_acos(R / sqrt(Qcubed));
      const double sqrtQ = sqrt(Q);
      const double t0 = -2 * sqrtQ;
      const double t1 = theta * (1. / 3);
      const double t2 = a1 * (1. / 3);
      x0 = t0 * detm::det_cos(t1) - t2;
      x1 = t0 * detm::det_cos(t1 + (2. * kPi / 3)) - t2;
      x2 = t0 * detm::det_cos(t1 + (4. * kPi / 3)) - t2;
      n = 3;
    } else {
      d = sqrt(-d);
      const double v = d + fabs(R);
      double e = detm::det_cbrt(v) * (1.0 + (0.333333333333 - 1.0 / 3.0) * detm::det_log(v));  // pow(v, 0.333333333333)
      if (R > 0) e = -e;
      x0 = (e + Q / e) - a1 * (1. / 3);
      n = 1;
    }
  }
  r[0] = x0;
  r[1] = x1;
  r[2] = x2;
  return n;
}

// ------------------------------------------------------------------------------------------------ run7Point
// Workspace of one solve: 9 rows x 9 doubles (rows 0..6: A, then the right singular vectors; rows 7, 8: the completed null basis) + 7
// squared row norms; element e of the workspace lives at wk[e * WS] (WS = 1 on the host; the kernel interleaves 64 lanes' workspaces).
constexpr int SP_WORK = 9 * 9 + 7;
constexpr int SVD_MAX_SWEEPS = 30;  // max(m, 30), m = 9

#define CVS_A(i, k) wk[(9 * (i) + (k)) * WS]
#define CVS_W(i) wk[(81 + (i)) * WS]
// form a linear system: i-th row of A represents the equation (m2[i], 1)' F (m1[i], 1) = 0; then the squared row norms
template <int WS>
CVS_FN void sp_fill(const double (*x1)[2], const double (*x2)[2], double* wk) {
  for (int i = 0; i < 7; i++) {
    const double px0 = x1[i][0], py0 = x1[i][1];
    const double px1 = x2[i][0], py1 = x2[i][1];
    CVS_A(i, 0) = px1 * px0;
    CVS_A(i, 1) = px1 * py0;
    CVS_A(i, 2) = px1;
    CVS_A(i, 3) = py1 * px0;
    CVS_A(i, 4) = py1 * py0;
    CVS_A(i, 5) = py1;
    CVS_A(i, 6) = px0;
    CVS_A(i, 7) = py0;
    CVS_A(i, 8) = 1;
  }
  for (int k = 0; k < 9; k++) CVS_A(7, k) = 0, CVS_A(8, k) = 0;
  for (int i = 0; i < 7; i++) {
    double sd = 0;
    for (int k = 0; k < 9; k++) {
      const double t = CVS_A(i, k);
      sd += t * t;
    }
    CVS_W(i) = sd;
  }
}

// One pair (i, j) of a Jacobi sweep of JacobiSVDImpl_<double> (eps = 10 DBL_EPSILON): rows i and j of A are rotated so that they become
// orthogonal.  true: rotated.  Pairs that share no row commute exactly -- the kernel runs the pairs of a sweep (and the head of the next)
// on their anti-diagonals i + j, three at a time, and obtains the bits of the cyclic order (i, j) = (0, 1), (0, 2) .. (5, 6) executed here.
template <int WS>
CVS_FN bool sp_pair(double* wk, int i, int j) {
  const double eps = DBL_EPSILON * 10;
  double ai[9], aj[9];
  CVS_UNROLL
  for (int k = 0; k < 9; k++) ai[k] = CVS_A(i, k), aj[k] = CVS_A(j, k);
  double a = CVS_W(i), p = 0, b = CVS_W(j);
  CVS_UNROLL
  for (int k = 0; k < 9; k++) p += ai[k] * aj[k];
  if (fabs(p) <= eps * sqrt(a * b)) return false;
  p *= 2;
  const double beta = a - b, gamma = hypot2(p, beta);
  double c, s;
  if (beta < 0) {
    const double delta = (gamma - beta) * 0.5;
    s = sqrt(delta / gamma);
    c = p / (gamma * s * 2);
  } else {
    c = sqrt((gamma + beta) / (gamma * 2));
    s = p / (gamma * c * 2);
  }
  a = b = 0;
  CVS_UNROLL
  for (int k = 0; k < 9; k++) {
    const double t0 = c * ai[k] + s * aj[k];
    const double t1 = -s * ai[k] + c * aj[k];
    CVS_A(i, k) = t0;
    CVS_A(j, k) = t1;
    a += t0 * t0;
    b += t1 * t1;
  }
  CVS_W(i) = a;
  CVS_W(j) = b;
  return true;
}

// behind the sweeps: singular values, their order, the completed basis, the cubic, the matrices.  Returns the number of matrices.
template <int WS, typename Stamp>
CVS_FN int sp_finish(double* wk, double (*F)[9], Stamp&& stamp) {
  const double eps = DBL_EPSILON * 10;
  stamp(1);
  for (int i = 0; i < 7; i++) {
    double sd = 0;
    for (int k = 0; k < 9; k++) {
      const double t = CVS_A(i, k);
      sd += t * t;
    }
    CVS_W(i) = sqrt(sd);
  }
  for (int i = 0; i < 6; i++) {  // selection sort, decreasing
    int j = i;
    for (int k = i + 1; k < 7; k++)
      if (CVS_W(j) < CVS_W(k)) j = k;
    if (i != j) {
      const double t = CVS_W(i);
      CVS_W(i) = CVS_W(j);
      CVS_W(j) = t;
      for (int k = 0; k < 9; k++) {
        const double u = CVS_A(i, k);
        CVS_A(i, k) = CVS_A(j, k);
        CVS_A(j, k) = u;
      }
    }
  }
  // unit rows; a row without a direction of its own (the two beyond the rank always; a vanishing singular value too) is drawn and
  // orthogonalised against all rows before it
  uint64_t rng = 0x12345678ull;
  for (int i = 0; i < 9; i++) {
    double sd = i < 7 ? CVS_W(i) : 0;
    double row[9];  // (row i stays in registers while it is built: the same operations, without a round trip through the workspace per step)
    CVS_UNROLL
    for (int k = 0; k < 9; k++) row[k] = CVS_A(i, k);
    for (int ii = 0; ii < 100 && sd <= DBL_MIN; ii++) {
      const double val0 = 1. / 9;
      CVS_UNROLL
      for (int k = 0; k < 9; k++) row[k] = (rng_next(rng) & 256) != 0 ? val0 : -val0;
      for (int it = 0; it < 2; it++)
        for (int j = 0; j < i; j++) {
          double rj[9];
          CVS_UNROLL
          for (int k = 0; k < 9; k++) rj[k] = CVS_A(j, k);
          sd = 0;
          CVS_UNROLL
          for (int k = 0; k < 9; k++) sd += row[k] * rj[k];
          double asum = 0;
          CVS_UNROLL
          for (int k = 0; k < 9; k++) {
            const double t = row[k] - sd * rj[k];
            row[k] = t;
            asum += fabs(t);
          }
          asum = asum > eps * 100 ? 1 / asum : 0;
          CVS_UNROLL
          for (int k = 0; k < 9; k++) row[k] *= asum;
        }
      sd = 0;
      CVS_UNROLL
      for (int k = 0; k < 9; k++) {
        const double t = row[k];
        sd += t * t;
      }
      sd = sqrt(sd);
    }
    const double s = sd > DBL_MIN ? 1 / sd : 0.;
    CVS_UNROLL
    for (int k = 0; k < 9; k++) CVS_A(i, k) = row[k] * s;
  }
  stamp(2);
  // f1, f2: a basis of the null space; f ~ lambda f1 + (1 - lambda) f2; det(f) = 0 is a cubic in lambda
  double f1[9], f2[9];
  for (int k = 0; k < 9; k++) f1[k] = CVS_A(7, k), f2[k] = CVS_A(8, k);
  for (int k = 0; k < 9; k++) f1[k] -= f2[k];
  double c[4], r[3];
  double t0 = f2[4] * f2[8] - f2[5] * f2[7];
  double t1 = f2[3] * f2[8] - f2[5] * f2[6];
  double t2 = f2[3] * f2[7] - f2[4] * f2[6];
  c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
  c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 - f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) + f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) -
         f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) + f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
         f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
  t0 = f1[4] * f1[8] - f1[5] * f1[7];
  t1 = f1[3] * f1[8] - f1[5] * f1[6];
  t2 = f1[3] * f1[7] - f1[4] * f1[6];
  c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
  c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 - f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) + f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) -
         f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) + f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
         f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
  const int n = solve_cubic(c, r);
  stamp(3);
  if (n < 1 || n > 3) return 0;
  for (int k = 0; k < 3; k++) {
    if (k >= n) break;
    // for each root form the fundamental matrix, normalised so that F(3,3) == 1
    double lambda = r[k], mu = 1.;
    const double s = f1[8] * r[k] + f2[8];
    double f8;
    if (fabs(s) > DBL_EPSILON) {
      mu = 1. / s;
      lambda *= mu;
      f8 = 1.;
    } else {
      f8 = 0.;
    }
    for (int i = 0; i < 8; i++) {
      const double v = f1[i] * lambda + f2[i] * mu;
      if (k == 0) F[0][i] = v;
      else if (k == 1) F[1][i] = v;
      else F[2][i] = v;
    }
    if (k == 0) F[0][8] = f8;
    else if (k == 1) F[1][8] = f8;
    else F[2][8] = f8;
  }
  stamp(4);
  return n;
}
#undef CVS_A
#undef CVS_W

// The schedule of the kernel.  Pairs on one anti-diagonal i + j share no row, and of two pairs that do share a row the one with the
// smaller i + j is the earlier one of the cyclic order: running the anti-diagonals 1 .. 11 of a sweep in order, all pairs of one at once,
// reproduces the cyclic order's bits.  The same holds across sweeps when sweep s + 1 starts seven steps behind sweep s (its first pairs
// touch rows the last pairs of sweep s have left).  Step T = 7 s_hi + sigma (sigma = 1 .. 7) therefore holds at most three pairs: those of
// sweep s_hi on anti-diagonal sigma and those of sweep s_hi - 1 on anti-diagonal sigma + 7.  A sweep that follows an unchanged one
// rotates nothing (it tests the same rows), so starting it before its predecessor's verdict is known changes nothing either.
struct SpSlot {
  signed char ds, i, j;  // sweep = s_hi + ds (ds = 0 or -1); i < 0: no pair
};
CVS_FN SpSlot sp_slot(int sigma, int q) {
  // sigma = 1 .. 7, q = 0 .. 2
  switch (sigma * 4 + q) {
    case 1 * 4 + 0: return SpSlot{0, 0, 1};
    case 1 * 4 + 1: return SpSlot{-1, 2, 6};
    case 1 * 4 + 2: return SpSlot{-1, 3, 5};
    case 2 * 4 + 0: return SpSlot{0, 0, 2};
    case 2 * 4 + 1: return SpSlot{-1, 3, 6};
    case 2 * 4 + 2: return SpSlot{-1, 4, 5};
    case 3 * 4 + 0: return SpSlot{0, 0, 3};
    case 3 * 4 + 1: return SpSlot{0, 1, 2};
    case 3 * 4 + 2: return SpSlot{-1, 4, 6};
    case 4 * 4 + 0: return SpSlot{0, 0, 4};
    case 4 * 4 + 1: return SpSlot{0, 1, 3};
    case 4 * 4 + 2: return SpSlot{-1, 5, 6};
    case 5 * 4 + 0: return SpSlot{0, 0, 5};
    case 5 * 4 + 1: return SpSlot{0, 1, 4};
    case 5 * 4 + 2: return SpSlot{0, 2, 3};
    case 6 * 4 + 0: return SpSlot{0, 0, 6};
    case 6 * 4 + 1: return SpSlot{0, 1, 5};
    case 6 * 4 + 2: return SpSlot{0, 2, 4};
    case 7 * 4 + 0: return SpSlot{0, 1, 6};
    case 7 * 4 + 1: return SpSlot{0, 2, 5};
    case 7 * 4 + 2: return SpSlot{0, 3, 4};
    default: return SpSlot{0, -1, -1};
  }
}
// the sweeps in the kernel's schedule, one step after the other (what three lanes do at once): the CPU statement of that schedule, which
// the tests hold against run7point bit for bit
template <int WS>
CVS_FN void sp_sweeps_scheduled(double* wk) {
  bool chg_prev = false, chg_cur = false;
  for (int T = 1;; T++) {
    const int s_hi = (T - 1) / 7, sigma = T - 7 * s_hi;
    for (int q = 0; q < 3; q++) {
      const SpSlot e = sp_slot(sigma, q);
      const int sw = s_hi + e.ds;
      if (e.i < 0 || sw < 0 || sw >= SVD_MAX_SWEEPS) continue;
      const bool rot = sp_pair<WS>(wk, e.i, e.j);
      if (e.ds == 0) chg_cur = chg_cur || rot;
      else chg_prev = chg_prev || rot;
    }
    if (sigma == 4 && s_hi >= 1 && (!chg_prev || s_hi == SVD_MAX_SWEEPS)) return;  // sweep s_hi - 1 is complete: unchanged, or the last one
    if (sigma == 7) {
      chg_prev = chg_cur;
      chg_cur = false;
    }
  }
}
template <int WS, typename Stamp>
CVS_FN int run7point_scheduled(const double (*x1)[2], const double (*x2)[2], double* wk, double (*F)[9], Stamp&& stamp) {
  sp_fill<WS>(x1, x2, wk);
  sp_sweeps_scheduled<WS>(wk);
  return sp_finish<WS>(wk, F, stamp);
}

// x1, x2: the seven correspondences (Point2f values as doubles).  F: up to three matrices, row-major.  Returns their number (0: none).
template <int WS, typename Stamp>
CVS_FN int run7point(const double (*x1)[2], const double (*x2)[2], double* wk, double (*F)[9], Stamp&& stamp) {
  sp_fill<WS>(x1, x2, wk);
  // ---- JacobiSVDImpl_<double>(At = A (n = 7 rows of m = 9), n1 = 9, minval = DBL_MIN, eps = 10 DBL_EPSILON); the left factor is not formed
  for (int iter = 0; iter < SVD_MAX_SWEEPS; iter++) {
    bool changed = false;
    for (int i = 0; i < 6; i++)
      for (int j = i + 1; j < 7; j++) changed = sp_pair<WS>(wk, i, j) || changed;
    if (!changed) break;
  }
  return sp_finish<WS>(wk, F, stamp);
}

// ------------------------------------------------------------------------------------------------ polynom_solver.cpp
CVS_FN int solve_deg2(double a, double b, double c, double& x1, double& x2) {
  const double delta = b * b - 4 * a * c;
  if (delta < 0) return 0;
  const double inv_2a = 0.5 / a;
  if (delta == 0) {
    x1 = -b * inv_2a;
    x2 = x1;
    return 1;
  }
  const double sqrt_delta = sqrt(delta);
  x1 = (-b + sqrt_delta) * inv_2a;
  x2 = (-b - sqrt_delta) * inv_2a;
  return 2;
}

CVS_FN int solve_deg3(double a, double b, double c, double d, double& x0, double& x1, double& x2) {
  if (a == 0) {
    if (b == 0) {  // first order system
      if (c == 0) return 0;
      x0 = -d / c;
      return 1;
    }
    x2 = 0;
    return solve_deg2(b, c, d, x0, x1);
  }
  // the normalized form x^3 + a2 x^2 + a1 x + a0 = 0
  const double inv_a = 1. / a;
  const double b_a = inv_a * b, b_a2 = b_a * b_a;
  const double c_a = inv_a * c;
  const double d_a = inv_a * d;
  const double Q = (3 * c_a - b_a2) / 9;
  const double R = (9 * b_a * c_a - 27 * d_a - 2 * b_a * b_a2) / 54;
  const double Q3 = Q * Q * Q;
  const double D = Q3 + R * R;
  const double b_a_3 = (1. / 3.) * b_a;
  if (Q == 0) {
    if (R == 0) {
      x0 = x1 = x2 = -b_a_3;
      return 3;
    }
    x0 = detm::det_cbrt(2 * R) - b_a_3;  // pow(2 * R, 1 / 3.0)
    return 1;
  }
  if (D <= 0) {  // three real roots
    const double theta = detm::det_acos(R / sqrt(-Q3));
    const double sqrt_Q = sqrt(-Q);
    x0 = 2 * sqrt_Q * detm::det_cos(theta / 3.0) - b_a_3;
    x1 = 2 * sqrt_Q * detm::det_cos((theta + 2 * kPi) / 3.0) - b_a_3;
    x2 = 2 * sqrt_Q * detm::det_cos((theta + 4 * kPi) / 3.0) - b_a_3;
    return 3;
  }
  // D > 0, only one real root
  const double AD = detm::det_cbrt(fabs(R) + sqrt(D)) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
  const double BD = (AD == 0) ? 0 : -Q / AD;
  x0 = AD + BD - b_a_3;
  return 1;
}

CVS_FN int solve_deg4(double a, double b, double c, double d, double e, double& x0, double& x1, double& x2, double& x3) {
  if (a == 0) {
    x3 = 0;
    return solve_deg3(b, c, d, e, x0, x1, x2);
  }
  const double inv_a = 1. / a;
  b *= inv_a;
  c *= inv_a;
  d *= inv_a;
  e *= inv_a;
  const double b2 = b * b, bc = b * c, b3 = b2 * b;
  // resultant cubic
  double r0 = 0, r1 = 0, r2 = 0;
  const int n = solve_deg3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, r0, r1, r2);
  if (n == 0) return 0;
  const double R2 = 0.25 * b2 - c + r0;
  if (R2 < 0) return 0;
  const double R = sqrt(R2);
  const double inv_R = 1. / R;
  int nb_real_roots = 0;
  double D2, E2;
  if (R < 10E-12) {
    const double temp = r0 * r0 - 4 * e;
    if (temp < 0) {
      D2 = E2 = -1;
    } else {
      const double sqrt_temp = sqrt(temp);
      D2 = 0.75 * b2 - 2 * c + 2 * sqrt_temp;
      E2 = D2 - 4 * sqrt_temp;
    }
  } else {
    const double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * inv_R * (4 * bc - 8 * d - b3);
    D2 = u + v;
    E2 = u - v;
  }
  const double b_4 = 0.25 * b, R_2 = 0.5 * R;
  if (D2 >= 0) {
    const double D = sqrt(D2);
    nb_real_roots = 2;
    const double D_2 = 0.5 * D;
    x0 = R_2 + D_2 - b_4;
    x1 = x0 - D;
  }
  if (E2 >= 0) {
    const double E = sqrt(E2);
    const double E_2 = 0.5 * E;
    if (nb_real_roots == 0) {
      x0 = -R_2 + E_2 - b_4;
      x1 = x0 - E;
      nb_real_roots = 2;
    } else {
      x2 = -R_2 + E_2 - b_4;
      x3 = x2 - E;
      nb_real_roots = 4;
    }
  }
  return nb_real_roots;
}

// ------------------------------------------------------------------------------------------------ p3p.cpp
// the cyclic Jacobi eigen-solver of p3p.cpp (Numerical Recipes' `jacobi` for n = 4): A symmetric, row-major, destroyed; D eigenvalues;
// U eigenvectors in columns
CVS_FN bool jacobi_4x4(double* A, double* D, double* U) {
  double B[4], Z[4];
  for (int i = 0; i < 16; i++) U[i] = 0;
  U[0] = U[5] = U[10] = U[15] = 1.;
  B[0] = A[0], B[1] = A[5], B[2] = A[10], B[3] = A[15];
  for (int i = 0; i < 4; i++) D[i] = B[i], Z[i] = 0;
  for (int iter = 0; iter < 50; iter++) {
    const double sum = fabs(A[1]) + fabs(A[2]) + fabs(A[3]) + fabs(A[6]) + fabs(A[7]) + fabs(A[11]);
    if (sum == 0.0) return true;
    const double tresh = (iter < 3) ? 0.2 * sum / 16. : 0.0;
    CVS_UNROLL
    for (int i = 0; i < 3; i++) {
      int pij = 5 * i + 1;
      CVS_UNROLL
      for (int j = i + 1; j < 4; j++) {
        const double Aij = A[pij];
        const double eps_machine = 100.0 * fabs(Aij);
        if (iter > 3 && fabs(D[i]) + eps_machine == fabs(D[i]) && fabs(D[j]) + eps_machine == fabs(D[j])) {
          A[pij] = 0.0;
        } else if (fabs(Aij) > tresh) {
          double hh = D[j] - D[i], t;
          if (fabs(hh) + eps_machine == fabs(hh)) {
            t = Aij / hh;
          } else {
            const double theta = 0.5 * hh / Aij;
            t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
            if (theta < 0.0) t = -t;
          }
          hh = t * Aij;
          Z[i] -= hh;
          Z[j] += hh;
          D[i] -= hh;
          D[j] += hh;
          A[pij] = 0.0;
          const double c = 1.0 / sqrt(1 + t * t);
          const double s = t * c;
          const double tau = s / (1.0 + c);
          CVS_UNROLL
          for (int k = 0; k <= i - 1; k++) {
            const double g = A[k * 4 + i], h = A[k * 4 + j];
            A[k * 4 + i] = g - s * (h + g * tau);
            A[k * 4 + j] = h + s * (g - h * tau);
          }
          CVS_UNROLL
          for (int k = i + 1; k <= j - 1; k++) {
            const double g = A[i * 4 + k], h = A[k * 4 + j];
            A[i * 4 + k] = g - s * (h + g * tau);
            A[k * 4 + j] = h + s * (g - h * tau);
          }
          CVS_UNROLL
          for (int k = j + 1; k < 4; k++) {
            const double g = A[i * 4 + k], h = A[j * 4 + k];
            A[i * 4 + k] = g - s * (h + g * tau);
            A[j * 4 + k] = h + s * (g - h * tau);
          }
          CVS_UNROLL
          for (int k = 0; k < 4; k++) {
            const double g = U[k * 4 + i], h = U[k * 4 + j];
            U[k * 4 + i] = g - s * (h + g * tau);
            U[k * 4 + j] = h + s * (g - h * tau);
          }
        }
        pij++;
      }
    }
    for (int i = 0; i < 4; i++) B[i] += Z[i];
    for (int i = 0; i < 4; i++) D[i] = B[i], Z[i] = 0;
  }
  return false;
}

// p3p::align: the rotation and translation that take the world points X[k] to the camera-frame points M[k] (Horn's quaternion method)
CVS_FN bool p3p_align(const double M_end[3][3], const double X[3][3], double R[9], double T[3]) {
  double C_start[3], C_end[3];
  for (int i = 0; i < 3; i++) C_end[i] = (M_end[0][i] + M_end[1][i] + M_end[2][i]) / 3;
  for (int i = 0; i < 3; i++) C_start[i] = (X[0][i] + X[1][i] + X[2][i]) / 3;
  double s[9];  // covariance
  for (int j = 0; j < 3; j++) {
    s[0 * 3 + j] = (X[0][0] * M_end[0][j] + X[1][0] * M_end[1][j] + X[2][0] * M_end[2][j]) / 3 - C_end[j] * C_start[0];
    s[1 * 3 + j] = (X[0][1] * M_end[0][j] + X[1][1] * M_end[1][j] + X[2][1] * M_end[2][j]) / 3 - C_end[j] * C_start[1];
    s[2 * 3 + j] = (X[0][2] * M_end[0][j] + X[1][2] * M_end[1][j] + X[2][2] * M_end[2][j]) / 3 - C_end[j] * C_start[2];
  }
  double Qs[16], evs[4], U[16];
  Qs[0 * 4 + 0] = s[0 * 3 + 0] + s[1 * 3 + 1] + s[2 * 3 + 2];
  Qs[1 * 4 + 1] = s[0 * 3 + 0] - s[1 * 3 + 1] - s[2 * 3 + 2];
  Qs[2 * 4 + 2] = s[1 * 3 + 1] - s[2 * 3 + 2] - s[0 * 3 + 0];
  Qs[3 * 4 + 3] = s[2 * 3 + 2] - s[0 * 3 + 0] - s[1 * 3 + 1];
  Qs[1 * 4 + 0] = Qs[0 * 4 + 1] = s[1 * 3 + 2] - s[2 * 3 + 1];
  Qs[2 * 4 + 0] = Qs[0 * 4 + 2] = s[2 * 3 + 0] - s[0 * 3 + 2];
  Qs[3 * 4 + 0] = Qs[0 * 4 + 3] = s[0 * 3 + 1] - s[1 * 3 + 0];
  Qs[2 * 4 + 1] = Qs[1 * 4 + 2] = s[1 * 3 + 0] + s[0 * 3 + 1];
  Qs[3 * 4 + 1] = Qs[1 * 4 + 3] = s[2 * 3 + 0] + s[0 * 3 + 2];
  Qs[3 * 4 + 2] = Qs[2 * 4 + 3] = s[2 * 3 + 1] + s[1 * 3 + 2];
  jacobi_4x4(Qs, evs, U);
  // the largest eigenvalue's vector is the quaternion
  int i_ev = 0;
  double ev_max = evs[i_ev];
  for (int i = 1; i < 4; i++)
    if (evs[i] > ev_max) ev_max = evs[i_ev = i];
  double q[4];
  for (int i = 0; i < 4; i++) q[i] = U[i * 4 + i_ev];
  const double q02 = q[0] * q[0], q12 = q[1] * q[1], q22 = q[2] * q[2], q32 = q[3] * q[3];
  const double q0_1 = q[0] * q[1], q0_2 = q[0] * q[2], q0_3 = q[0] * q[3];
  const double q1_2 = q[1] * q[2], q1_3 = q[1] * q[3];
  const double q2_3 = q[2] * q[3];
  R[0] = q02 + q12 - q22 - q32;
  R[1] = 2. * (q1_2 - q0_3);
  R[2] = 2. * (q1_3 + q0_2);
  R[3] = 2. * (q1_2 + q0_3);
  R[4] = q02 + q22 - q12 - q32;
  R[5] = 2. * (q2_3 - q0_1);
  R[6] = 2. * (q1_3 - q0_2);
  R[7] = 2. * (q2_3 + q0_1);
  R[8] = q02 + q32 - q12 - q22;
  for (int i = 0; i < 3; i++) T[i] = C_end[i] - (R[3 * i] * C_start[0] + R[3 * i + 1] * C_start[1] + R[3 * i + 2] * C_start[2]);
  return true;
}

// p3p::solve_for_lengths: |PA|, |PB|, |PC| (up to four solutions) from the distances |BC|, |AC|, |AB| and the cosines of the angles BPC,
// APC, APB.  Only the main branch of Gao's classification, as in OpenCV ("NOT ALL THE DEGENERATE CASES ARE IMPLEMENTED").
CVS_FN int p3p_lengths(double lengths[4][3], const double distances[3], const double cosines[3]) {
  const double p = cosines[0] * 2;
  const double q = cosines[1] * 2;
  const double r = cosines[2] * 2;
  const double inv_d22 = 1. / (distances[2] * distances[2]);
  const double a = inv_d22 * (distances[0] * distances[0]);
  const double b = inv_d22 * (distances[1] * distances[1]);
  const double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
  const double pr = p * r, pqr = q * pr;
  // reality condition (the four points should not be coplanar)
  if (p2 + q2 + r2 - pqr - 1 == 0) return 0;
  const double ab = a * b, a_2 = 2 * a;
  const double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
  if (A == 0) return 0;
  const double a_4 = 4 * a;
  const double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
  const double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
  const double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
  const double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
  const double temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr);
  const double b0 = b * temp * temp;
  if (b0 == 0) return 0;
  double real_roots[4] = {0, 0, 0, 0};
  const int n = solve_deg4(A, B, C, D, E, real_roots[0], real_roots[1], real_roots[2], real_roots[3]);
  if (n == 0) return 0;
  int nb_solutions = 0;
  const double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q;
  const double inv_b0 = 1. / b0;
  for (int i = 0; i < n; i++) {
    const double x = real_roots[i];
    if (x <= 0) continue;
    const double x2 = x * x;
    const double b1 =
        ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) *
        (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
          (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) + pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) * x2 +
         (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) + r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
          pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) * x +
         2 * r3q * (a_2 - b - a2 + ab - 1) + pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
         p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));
    if (b1 <= 0) continue;
    const double y = inv_b0 * b1;
    const double v = x2 + y * y - x * y * r;
    if (v <= 0) continue;
    const double Z = distances[2] / sqrt(v);
    const double X = x * Z;
    const double Y = y * Z;
    lengths[nb_solutions][0] = X;
    lengths[nb_solutions][1] = Y;
    lengths[nb_solutions][2] = Z;
    nb_solutions++;
  }
  return nb_solutions;
}

struct P3PCamera {
  double fx, fy, cx, cy, inv_fx, inv_fy, cx_fx, cy_fy;
};
CVS_FN P3PCamera p3p_camera(double fx, double fy, double cx, double cy) {
  P3PCamera c;
  c.fx = fx, c.fy = fy, c.cx = cx, c.cy = cy;
  c.inv_fx = 1. / fx;
  c.inv_fy = 1. / fy;
  c.cx_fx = cx / fx;
  c.cy_fy = cy / fy;
  return c;
}

// p3p::solve (three points): image points (mu, mv) in pixels, world points X.  Up to four poses (R row-major, t).
CVS_FN int p3p_solve3(const P3PCamera& cam, const double uv[3][2], const double X[3][3], double R[4][9], double t[4][3]) {
  double m[3][3];  // unit bearings
  for (int k = 0; k < 3; k++) {
    double mu = cam.inv_fx * uv[k][0] - cam.cx_fx;
    double mv = cam.inv_fy * uv[k][1] - cam.cy_fy;
    const double norm = sqrt(mu * mu + mv * mv + 1);
    const double mk = 1. / norm;
    mu *= mk;
    mv *= mk;
    m[k][0] = mu, m[k][1] = mv, m[k][2] = mk;
  }
  double distances[3];
  distances[0] = sqrt((X[1][0] - X[2][0]) * (X[1][0] - X[2][0]) + (X[1][1] - X[2][1]) * (X[1][1] - X[2][1]) + (X[1][2] - X[2][2]) * (X[1][2] - X[2][2]));
  distances[1] = sqrt((X[0][0] - X[2][0]) * (X[0][0] - X[2][0]) + (X[0][1] - X[2][1]) * (X[0][1] - X[2][1]) + (X[0][2] - X[2][2]) * (X[0][2] - X[2][2]));
  distances[2] = sqrt((X[0][0] - X[1][0]) * (X[0][0] - X[1][0]) + (X[0][1] - X[1][1]) * (X[0][1] - X[1][1]) + (X[0][2] - X[1][2]) * (X[0][2] - X[1][2]));
  double cosines[3];
  cosines[0] = m[1][0] * m[2][0] + m[1][1] * m[2][1] + m[1][2] * m[2][2];
  cosines[1] = m[0][0] * m[2][0] + m[0][1] * m[2][1] + m[0][2] * m[2][2];
  cosines[2] = m[0][0] * m[1][0] + m[0][1] * m[1][1] + m[0][2] * m[1][2];
  double lengths[4][3];
  const int n = p3p_lengths(lengths, distances, cosines);
  int nb_solutions = 0;
  for (int i = 0; i < n; i++) {
    double M_orig[3][3];
    for (int k = 0; k < 3; k++)
      for (int j = 0; j < 3; j++) M_orig[k][j] = lengths[i][k] * m[k][j];
    if (!p3p_align(M_orig, X, R[nb_solutions], t[nb_solutions])) continue;
    nb_solutions++;
  }
  return nb_solutions;
}

// p3p::solve (four points): the pose among those of the first three points that reprojects the fourth best (first minimum)
CVS_FN bool p3p_solve4(const P3PCamera& cam, const double uv[4][2], const double X[4][3], double R[9], double t[3]) {
  double Rs[4][9], ts[4][3];
  const int n = p3p_solve3(cam, uv, X, Rs, ts);
  if (n == 0) return false;
  int ns = 0;
  double min_reproj = 0;
  for (int i = 0; i < n; i++) {
    const double X3p = Rs[i][0] * X[3][0] + Rs[i][1] * X[3][1] + Rs[i][2] * X[3][2] + ts[i][0];
    const double Y3p = Rs[i][3] * X[3][0] + Rs[i][4] * X[3][1] + Rs[i][5] * X[3][2] + ts[i][1];
    const double Z3p = Rs[i][6] * X[3][0] + Rs[i][7] * X[3][1] + Rs[i][8] * X[3][2] + ts[i][2];
    const double mu3p = cam.cx + cam.fx * X3p / Z3p;
    const double mv3p = cam.cy + cam.fy * Y3p / Z3p;
    const double reproj = (mu3p - uv[3][0]) * (mu3p - uv[3][0]) + (mv3p - uv[3][1]) * (mv3p - uv[3][1]);
    if (i == 0 || min_reproj > reproj) {
      ns = i;
      min_reproj = reproj;
    }
  }
  for (int i = 0; i < 9; i++) R[i] = Rs[ns][i];
  for (int i = 0; i < 3; i++) t[i] = ts[ns][i];
  return true;
}

// ================================================================================================ SOLVEPNP_ITERATIVE on the inliers
// cv::solvePnP(..., useExtrinsicGuess = false, SOLVEPNP_ITERATIVE) = cvFindExtrinsicCameraParams2 (calib3d/src/calibration.cpp), the final
// solve of cv::solvePnPRansac(ITERATIVE) on its inliers (lkorb_tracking.cpp:172): a DLT start (non-planar point sets) and CvLevMarq (at most
// 20 iterations, stop when the relative change of (rvec, tvec) falls below FLT_EPSILON) on the reprojection error, rotation as a Rodrigues
// vector.  By default the kernels and the checker refine the RANSAC's winning model by Gauss-Newton to the same minimum; this tail is what
// `make -C oracle TAIL=cv` builds into the checker and what FLVIS_PNP_TAIL=cv makes the device run behind k_ransac_pnp (k_pnp_tail_cv: this
// very function, one wave per stream -- LANES below --, in lockstep with that checker; +2.2 ms per frame at 64 streams: the Jacobi SVDs walk
// their matrices in private memory); what it buys is measured by the tests (the two tails agree to ~1e-7
// relative, CvLevMarq's own stopping tolerance).  Zero distortion (rectified images).

// cv::JacobiSVDImpl_<double>: At holds n rows of m doubles (the columns of the matrix to decompose); on return row i = sigma_i u_i scaled to
// unit length (for i < n1; rows beyond the rank are completed), W the singular values in decreasing order, Vt (n x n, may be null) the
// right singular vectors in rows.  At must have room for n1 rows.
CVS_FN void jacobi_svd(double* At, int astep, double* W, double* Vt, int vstep, int m, int n, int n1) {
  const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
  const int max_iter = m > 30 ? m : 30;
  for (int i = 0; i < n; i++) {
    double sd = 0;
    for (int k = 0; k < m; k++) {
      const double t = At[i * astep + k];
      sd += t * t;
    }
    W[i] = sd;
    if (Vt) {
      for (int k = 0; k < n; k++) Vt[i * vstep + k] = 0;
      Vt[i * vstep + i] = 1;
    }
  }
  for (int iter = 0; iter < max_iter; iter++) {
    bool changed = false;
    for (int i = 0; i < n - 1; i++)
      for (int j = i + 1; j < n; j++) {
        double *Ai = At + i * astep, *Aj = At + j * astep;
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < m; k++) p += Ai[k] * Aj[k];
        if (fabs(p) <= eps * sqrt(a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = hypot2(p, beta);
        double c, s;
        if (beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          s = sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = b = 0;
        for (int k = 0; k < m; k++) {
          const double t0 = c * Ai[k] + s * Aj[k];
          const double t1 = -s * Ai[k] + c * Aj[k];
          Ai[k] = t0;
          Aj[k] = t1;
          a += t0 * t0;
          b += t1 * t1;
        }
        W[i] = a;
        W[j] = b;
        changed = true;
        if (Vt) {
          double *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;
          for (int k = 0; k < n; k++) {
            const double t0 = c * Vi[k] + s * Vj[k];
            const double t1 = -s * Vi[k] + c * Vj[k];
            Vi[k] = t0;
            Vj[k] = t1;
          }
        }
      }
    if (!changed) break;
  }
  for (int i = 0; i < n; i++) {
    double sd = 0;
    for (int k = 0; k < m; k++) {
      const double t = At[i * astep + k];
      sd += t * t;
    }
    W[i] = sqrt(sd);
  }
  for (int i = 0; i < n - 1; i++) {
    int j = i;
    for (int k = i + 1; k < n; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      const double t = W[i];
      W[i] = W[j];
      W[j] = t;
      if (Vt) {
        for (int k = 0; k < m; k++) {
          const double u = At[i * astep + k];
          At[i * astep + k] = At[j * astep + k];
          At[j * astep + k] = u;
        }
        for (int k = 0; k < n; k++) {
          const double u = Vt[i * vstep + k];
          Vt[i * vstep + k] = Vt[j * vstep + k];
          Vt[j * vstep + k] = u;
        }
      }
    }
  }
  if (!Vt) return;
  uint64_t rng = 0x12345678ull;
  for (int i = 0; i < n1; i++) {
    double sd = i < n ? W[i] : 0;
    for (int ii = 0; ii < 100 && sd <= minval; ii++) {
      const double val0 = 1. / m;
      for (int k = 0; k < m; k++) At[i * astep + k] = (rng_next(rng) & 256) != 0 ? val0 : -val0;
      for (int it = 0; it < 2; it++)
        for (int j = 0; j < i; j++) {
          sd = 0;
          for (int k = 0; k < m; k++) sd += At[i * astep + k] * At[j * astep + k];
          double asum = 0;
          for (int k = 0; k < m; k++) {
            const double t = At[i * astep + k] - sd * At[j * astep + k];
            At[i * astep + k] = t;
            asum += fabs(t);
          }
          asum = asum > eps * 100 ? 1 / asum : 0;
          for (int k = 0; k < m; k++) At[i * astep + k] *= asum;
        }
      sd = 0;
      for (int k = 0; k < m; k++) {
        const double t = At[i * astep + k];
        sd += t * t;
      }
      sd = sqrt(sd);
    }
    const double s = sd > minval ? 1 / sd : 0.;
    for (int k = 0; k < m; k++) At[i * astep + k] *= s;
  }
}

// cv::SVD::compute of a square n x n matrix A (row-major, n <= 12): w, u (columns = left vectors), vt (rows = right vectors)
CVS_FN void svd_square(const double* A, int n, double* w, double* u, double* vt) {
  double at[144];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) at[i * n + j] = A[j * n + i];  // transpose(src, temp_a)
  jacobi_svd(at, n, w, vt, n, n, n, n);
  if (u)
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) u[j * n + i] = at[i * n + j];  // transpose(temp_u, _u)
}

// cv::solve(A, b, x, DECOMP_SVD), A n x n (n <= 6): JacobiSVD on A^T + SVBkSb
CVS_FN void solve_svd(const double* A, const double* b, int n, double* x) {
  double at[36], w[6], vt[36];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) at[i * n + j] = A[j * n + i];
  jacobi_svd(at, n, w, vt, n, n, n, n);
  double threshold = 0;
  for (int i = 0; i < n; i++) threshold += w[i];
  threshold *= DBL_EPSILON * 2;
  for (int k = 0; k < n; k++) x[k] = 0;
  for (int i = 0; i < n; i++) {
    double wi = w[i];
    if (fabs(wi) <= threshold) continue;
    wi = 1 / wi;
    double sv = 0;
    for (int k = 0; k < n; k++) sv += at[i * n + k] * b[k];  // u_i . b
    sv *= wi;
    for (int k = 0; k < n; k++) x[k] += sv * vt[i * n + k];
  }
}

// cvRodrigues2, vector -> matrix, with dR/dr (3 x 9: row i = d vec(R) / d r_i) when J != null
CVS_FN void rodrigues(const double* rv, double* R, double* J) {
  const double theta = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
  if (theta < DBL_EPSILON) {
    for (int i = 0; i < 9; i++) R[i] = 0;
    R[0] = R[4] = R[8] = 1;
    if (J) {
      for (int i = 0; i < 27; i++) J[i] = 0;
      J[5] = J[15] = J[19] = -1;
      J[7] = J[11] = J[21] = 1;
    }
    return;
  }
  const double c = detm::det_cos(theta), s = detm::det_sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
  const double rx = rv[0] * itheta, ry = rv[1] * itheta, rz = rv[2] * itheta;
  const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
  const double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
  const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
  if (J) {
    const double drrt[27] = {rx + rx, ry, rz, ry, 0, 0, rz, 0, 0, 0, rx, 0, rx, ry + ry, rz, 0, rz, 0, 0, 0, rx, 0, 0, ry, rx, ry, rz + rz};
    const double d_r_x_[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; i++) {
      const double ri = i == 0 ? rx : i == 1 ? ry : rz;
      const double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
      const double a3 = (c - s * itheta) * ri, a4 = s * itheta;
      for (int k = 0; k < 9; k++) J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x_[i * 9 + k];
    }
  }
}

// cvRodrigues2, matrix -> vector (the matrix is first projected onto SO(3) by its SVD)
CVS_FN void rodrigues_inv(const double* Rin, double* rv) {
  double w[3], u[9], vt[9], R[9];
  svd_square(Rin, 3, w, u, vt);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[3 * i + j] = u[3 * i] * vt[j] + u[3 * i + 1] * vt[3 + j] + u[3 * i + 2] * vt[6 + j];
  double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
  const double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
  double c = (R[0] + R[4] + R[8] - 1) * 0.5;
  c = c > 1. ? 1. : c < -1. ? -1. : c;
  const double theta = detm::det_acos(c);
  if (s < 1e-5) {
    if (c > 0) {
      rx = ry = rz = 0;
    } else {
      double t;
      t = (R[0] + 1) * 0.5;
      rx = sqrt(t > 0. ? t : 0.);
      t = (R[4] + 1) * 0.5;
      ry = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
      t = (R[8] + 1) * 0.5;
      rz = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
      if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
      const double nn = sqrt(rx * rx + ry * ry + rz * rz);
      const double k = theta / nn;
      rx *= k, ry *= k, rz *= k;
    }
  } else {
    double vth = 1 / (2 * s);
    vth *= theta;
    rx *= vth, ry *= vth, rz *= vth;
  }
  rv[0] = rx, rv[1] = ry, rv[2] = rz;
}

// How the long loops of find_extrinsic_iterative are shared out.  The checker runs them as written (SerialLanes: one "lane").  On the device
// (k_pnp_tail_cv) every lane of a wave executes the function -- the small dense algebra redundantly, lane for lane the same values -- and
// the loops over the correspondences are dealt out: a lane takes whole points (rows of L, residuals, Jacobian rows) or whole SUMS (an entry
// of L^T L or of J^T J: still ONE lane adding its 2 n terms in row order), the results go through `work`, `sync` makes them visible.  The
// arithmetic, and its order, is the same either way.
struct SerialLanes {
  CVS_FN int lane() const { return 0; }
  CVS_FN int lanes() const { return 1; }
  CVS_FN void sync() const {}
};

// cvProjectPoints2 without distortion: residuals err[2 n] = projection - measurement and (J != null) the 2 n x 6 Jacobian (dp/dr, dp/dt)
template <class LANES>
CVS_FN void project_residuals(int n, const double* M, const double* m, const double* param, double fx, double fy, double cx, double cy, double* err,
                              double* J, LANES ln) {
  double R[9], dRdr[27];
  rodrigues(param, R, J ? dRdr : nullptr);
  const double* t = param + 3;
  for (int i = ln.lane(); i < n; i += ln.lanes()) {
    const double X = M[3 * i], Y = M[3 * i + 1], Z = M[3 * i + 2];
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z;
    y *= z;
    err[2 * i] = (x * fx + cx) - m[2 * i];
    err[2 * i + 1] = (y * fy + cy) - m[2 * i + 1];
    if (J) {
      double* Jx = J + (size_t)(2 * i) * 6;
      double* Jy = Jx + 6;
      const double dx0dr[3] = {X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2], X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11], X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20]};
      const double dy0dr[3] = {X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5], X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14], X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23]};
      const double dz0dr[3] = {X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8], X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17], X * dRdr[24] + Y * dRdr[25] + Z * dRdr[26]};
      for (int j = 0; j < 3; j++) {
        const double dxdr = z * (dx0dr[j] - x * dz0dr[j]);
        const double dydr = z * (dy0dr[j] - y * dz0dr[j]);
        Jx[j] = fx * dxdr;
        Jy[j] = fy * dydr;
      }
      const double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
      for (int j = 0; j < 3; j++) {
        Jx[3 + j] = fx * dxdt[j];
        Jy[3 + j] = fy * dydt[j];
      }
    }
  }
  ln.sync();
}

// CvLevMarq::step for six free parameters: (JtJ with its diagonal scaled by 1 + lambda) x = JtErr by SVD, param = prev - x
CVS_FN void levmarq_step(const double* JtJ, const double* JtErr, const double* prev, int lambdaLg10, double* param) {
  // exp(lambdaLg10 * log(10.)): the power of ten itself (the library's exp is within an ulp of it)
  const double p10[34] = {1e-16, 1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1e0,
                          1e1,   1e2,   1e3,   1e4,   1e5,   1e6,   1e7,   1e8,  1e9,  1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17};
  const double lambda = p10[lambdaLg10 + 16];
  double A[36], x[6];
  for (int i = 0; i < 36; i++) A[i] = JtJ[i];
  for (int i = 0; i < 6; i++) A[7 * i] *= 1. + lambda;
  solve_svd(A, JtErr, 6, x);
  for (int i = 0; i < 6; i++) param[i] = prev[i] - x[i];
}

CVS_FN double norm2(const double* v, int n) {
  double s = 0;
  for (int i = 0; i < n; i++) s += v[i] * v[i];
  return sqrt(s);
}

// cvFindExtrinsicCameraParams2(useExtrinsicGuess = 0) for n >= 6 non-planar points: M world points (3 n), m pixels (2 n); work: >= 24 n
// + 192 doubles (the last 192: the sums the lanes hand each other).  rvec / tvec out.  false: the point set is planar (OpenCV starts from a
// homography there: not restated) or n < 6.
template <class LANES>
CVS_FN bool find_extrinsic_iterative(int n, const double* M, const double* m, double fx, double fy, double cx, double cy, double* work, double* rvec,
                                     double* tvec, int* iterations_out, LANES ln) {
  if (n < 6) return false;
  double* const shared_sums = work + 24 * (size_t)n;  // [192]
  // planarity test: SVD of the covariance of the points
  double Mc[3] = {0, 0, 0};
  for (int i = 0; i < n; i++) Mc[0] += M[3 * i], Mc[1] += M[3 * i + 1], Mc[2] += M[3 * i + 2];
  for (int k = 0; k < 3; k++) Mc[k] /= n;
  double MM[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int a = 0; a < 3; a++)
    for (int b = a; b < 3; b++) {
      double sacc = 0;
      for (int i = 0; i < n; i++) sacc += (M[3 * i + a] - Mc[a]) * (M[3 * i + b] - Mc[b]);
      MM[3 * a + b] = MM[3 * b + a] = sacc;
    }
  double W3[3], V3[9];
  svd_square(MM, 3, W3, nullptr, V3);
  if (W3[2] / W3[1] < 1e-3) return false;
  // ---- DLT: L (2 n x 12), LL = L^T L, the right singular vector of the smallest singular value -> [R | t] up to scale
  double LL[144];
  for (int i = 0; i < 144; i++) LL[i] = 0;
  {
    // MulTransposed: entry (a, b) = sum over the rows of L, in row order
    double* L = work;  // 2 n x 12
    for (int i = ln.lane(); i < n; i += ln.lanes()) {
      const double xn = (m[2 * i] - cx) * (1. / fx), yn = (m[2 * i + 1] - cy) * (1. / fy);  // cvUndistortPoints, zero distortion
      const double x = -xn, y = -yn;
      double* r0 = L + (size_t)(2 * i) * 12;
      double* r1 = r0 + 12;
      r0[0] = r1[4] = M[3 * i];
      r0[1] = r1[5] = M[3 * i + 1];
      r0[2] = r1[6] = M[3 * i + 2];
      r0[3] = r1[7] = 1.;
      r0[4] = r0[5] = r0[6] = r0[7] = 0.;
      r1[0] = r1[1] = r1[2] = r1[3] = 0.;
      r0[8] = x * M[3 * i];
      r0[9] = x * M[3 * i + 1];
      r0[10] = x * M[3 * i + 2];
      r0[11] = x;
      r1[8] = y * M[3 * i];
      r1[9] = y * M[3 * i + 1];
      r1[10] = y * M[3 * i + 2];
      r1[11] = y;
    }
    ln.sync();
    for (int e = ln.lane(); e < 144; e += ln.lanes()) {  // (entry (a, b), a <= b: one lane, its 2 n terms in row order)
      const int a = e / 12, b = e - 12 * a;
      if (b < a) continue;
      double sacc = 0;
      for (int k = 0; k < 2 * n; k++) sacc += L[(size_t)k * 12 + a] * L[(size_t)k * 12 + b];
      shared_sums[e] = sacc;
    }
    ln.sync();
    for (int a = 0; a < 12; a++)
      for (int b = a; b < 12; b++) LL[12 * a + b] = LL[12 * b + a] = shared_sums[12 * a + b];
    ln.sync();  // (the buffer is written again below)
  }
  double LW[12], LV[144];
  svd_square(LL, 12, LW, nullptr, LV);
  double RRt[12];
  for (int k = 0; k < 12; k++) RRt[k] = LV[11 * 12 + k];
  double RR[9] = {RRt[0], RRt[1], RRt[2], RRt[4], RRt[5], RRt[6], RRt[8], RRt[9], RRt[10]};
  double tt[3] = {RRt[3], RRt[7], RRt[11]};
  const double det = RR[0] * (RR[4] * RR[8] - RR[5] * RR[7]) - RR[1] * (RR[3] * RR[8] - RR[5] * RR[6]) + RR[2] * (RR[3] * RR[7] - RR[4] * RR[6]);
  if (det < 0) {
    for (int k = 0; k < 9; k++) RR[k] = -RR[k];
    for (int k = 0; k < 3; k++) tt[k] = -tt[k];
  }
  const double sc = norm2(RR, 9);
  double W[3], U[9], Vt[9], R0[9];
  svd_square(RR, 3, W, U, Vt);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R0[3 * i + j] = U[3 * i] * Vt[j] + U[3 * i + 1] * Vt[3 + j] + U[3 * i + 2] * Vt[6 + j];
  const double tscale = norm2(R0, 9) / sc;
  double param[6], prev[6];
  rodrigues_inv(R0, param);
  for (int k = 0; k < 3; k++) param[3 + k] = tt[k] * tscale;
  // ---- CvLevMarq(6, 2 n, (20, FLT_EPSILON), completeSymmFlag = true), lambdaLg10 = -3
  double* err = work;               // 2 n
  double* J = work + 2 * (size_t)n; // 2 n x 6
  int lambdaLg10 = -3, iters = 0;
  double prevErrNorm = DBL_MAX, errNorm = 0;
  project_residuals(n, M, m, param, fx, fy, cx, cy, err, J, ln);
  for (;;) {
    double JtJ[36], JtErr[6];
    for (int e = ln.lane(); e < 42; e += ln.lanes()) {  // (36 entries of J^T J, of which a <= b are summed, and the 6 of J^T err)
      double sacc = 0;
      if (e < 36) {
        const int a = e / 6, b = e - 6 * a;
        if (b < a) continue;
        for (int k = 0; k < 2 * n; k++) sacc += J[(size_t)k * 6 + a] * J[(size_t)k * 6 + b];
      } else {
        const int a = e - 36;
        for (int k = 0; k < 2 * n; k++) sacc += J[(size_t)k * 6 + a] * err[k];
      }
      shared_sums[e] = sacc;
    }
    ln.sync();
    for (int a = 0; a < 6; a++) {
      for (int b = a; b < 6; b++) JtJ[6 * a + b] = JtJ[6 * b + a] = shared_sums[6 * a + b];
      JtErr[a] = shared_sums[36 + a];
    }
    for (int k = 0; k < 6; k++) prev[k] = param[k];
    levmarq_step(JtJ, JtErr, prev, lambdaLg10, param);
    if (iters == 0) prevErrNorm = norm2(err, 2 * n);
    for (;;) {
      ln.sync();  // (every lane has read the residuals it needed -- the sums above, the norms -- before they are written again)
      project_residuals(n, M, m, param, fx, fy, cx, cy, err, nullptr, ln);
      errNorm = norm2(err, 2 * n);
      if (errNorm > prevErrNorm) {
        if (++lambdaLg10 <= 16) {
          levmarq_step(JtJ, JtErr, prev, lambdaLg10, param);
          continue;
        }
      }
      break;
    }
    lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
    double dn = 0, pn = 0;
    for (int k = 0; k < 6; k++) dn += (param[k] - prev[k]) * (param[k] - prev[k]), pn += prev[k] * prev[k];
    if (++iters >= 20 || sqrt(dn) / sqrt(pn) < FLT_EPSILON) break;
    prevErrNorm = errNorm;
    ln.sync();  // (everybody has read err and the sums of this iteration)
    project_residuals(n, M, m, param, fx, fy, cx, cy, err, J, ln);
  }
  for (int k = 0; k < 3; k++) rvec[k] = param[k], tvec[k] = param[3 + k];
  if (iterations_out) *iterations_out = iters;
  return true;
}

CVS_FN bool find_extrinsic_iterative(int n, const double* M, const double* m, double fx, double fy, double cx, double cy, double* work, double* rvec,
                                     double* tvec, int* iterations_out) {
  return find_extrinsic_iterative(n, M, m, fx, fy, cx, cy, work, rvec, tvec, iterations_out, SerialLanes());
}

}  // namespace cvs
}  // namespace flvis
