// ORACLE (test infrastructure, NOT product code) -- point geometry of the FLVIS front-end.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
//   project_points       cv::projectPoints call sites   src/processing/lkorb_tracking.cpp:55-61, camera_frame.cpp:113-122
//   undistort_points     cv::undistortPoints            src/frontend/f2f_tracking.cpp:301,425; lkorb_tracking.cpp:87; camera_frame.cpp:130
//   triangulate_dlt      Triangulation::triangulationPt src/processing/triangulation.cpp:9-39 (Eigen::JacobiSVD -> one-sided Jacobi)
//   find_fundamental_ransac  cv::findFundamentalMat(FM_RANSAC, 5.0, 0.99)   lkorb_tracking.cpp:133-135
//   solve_pnp_ransac     cv::solvePnPRansac(...,100,3.0,0.99,ITERATIVE|P3P) lkorb_tracking.cpp:170-177
//   optimize_in_frame    OptimizeInFrame::optimize      src/processing/optimize_in_frame.cpp:10-91 + g2o LM
//                        (3rdPartLib/g2o/g2o/core/optimization_algorithm_levenberg.cpp:58-175, base_binary_edge.hpp:61-134,
//                         robust_kernel_impl.cpp:65-78, types/sba/types_six_dof_expmap.cpp:389-433)
//
// **parity unpinned** for the OpenCV-resident steps (no OpenCV here, no golden vectors in the reference).
// Where OpenCV's internals cannot be matched (RNG stream, EPnP/DLT initialisers) this file DEFINES the algorithm:
//   * both RANSACs keep OpenCV's control flow (RANSACPointSetRegistrator: subset draw with duplicate rejection, best model =
//     strictly more inliers, adaptive iteration count RANSACUpdateNumIters) but draw from a counter-based RNG (ref_math.hpp);
//   * 7-point F: Hartley-normalised points, null space by Gauss-Jordan with full pivoting, cubic by bracketing + bisection;
//   * PnP hypotheses: Grunert P3P on the first 3 sample points, remaining sample points disambiguate (OpenCV: EPnP on 5
//     points for ITERATIVE, P3P on 4 for P3P); refinement on the RANSAC inliers: 10 Gauss-Newton steps on reprojection error.
// Tests pin these against synthetic ground truth and scipy.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "ref_api.h"
#include "../flvis_amd/csrc/epnp_core.hpp"  // EPnP's arithmetic: the very functions the kernel runs, here with one lane
#include "ref_math.hpp"
#include "../flvis_amd/csrc/cv_solvers.hpp"  // the OpenCV-shaped minimal solvers: the very functions the kernels run

namespace ref {

// ------------------------------------------------------------------------------------------ polynomial real roots
static double poly_eval(const double* a, int deg, double x) {
  double r = a[deg];
  for (int i = deg - 1; i >= 0; i--) r = r * x + a[i];
  return r;
}

// Real roots of a[0] + a[1]x + ... + a[deg]x^deg (deg <= 4), ascending.  Deterministic: only + - * / sqrt and compares.
// Roots of even multiplicity (no sign change) are not reported.
int poly_real_roots(const double* a_in, int deg, double* roots) {
  double a[5];
  double amax = 0;
  for (int i = 0; i <= deg; i++) {
    a[i] = a_in[i];
    amax = std::max(amax, std::fabs(a[i]));
  }
  if (amax == 0) return 0;
  while (deg > 0 && std::fabs(a[deg]) <= 1e-14 * amax) deg--;
  if (deg == 0) return 0;
  if (deg == 1) {
    roots[0] = -a[0] / a[1];
    return 1;
  }
  if (deg == 2) {
    double disc = a[1] * a[1] - 4 * a[2] * a[0];
    if (disc < 0) return 0;
    double sq = std::sqrt(disc);
    double q = -0.5 * (a[1] + (a[1] >= 0 ? sq : -sq));
    double r0 = q / a[2];
    double r1 = (q != 0) ? a[0] / q : r0;
    if (disc == 0) {
      roots[0] = r0;
      return 1;
    }
    roots[0] = std::min(r0, r1);
    roots[1] = std::max(r0, r1);
    return 2;
  }
  double d[5] = {0, 0, 0, 0, 0};
  for (int i = 1; i <= deg; i++) d[i - 1] = a[i] * i;
  double crit[4];
  int nc = poly_real_roots(d, deg - 1, crit);
  double B = 0;
  for (int i = 0; i < deg; i++) B = std::max(B, std::fabs(a[i] / a[deg]));
  B += 1.0;
  double knots[6];
  int nk = 0;
  knots[nk++] = -B;
  for (int i = 0; i < nc; i++)
    if (crit[i] > -B && crit[i] < B) knots[nk++] = crit[i];
  knots[nk++] = B;
  int nr = 0;
  for (int i = 0; i + 1 < nk; i++) {
    double lo = knots[i], hi = knots[i + 1];
    double flo = poly_eval(a, deg, lo), fhi = poly_eval(a, deg, hi);
    if (flo == 0) {
      if (nr == 0 || roots[nr - 1] != lo) roots[nr++] = lo;
      continue;
    }
    if (fhi == 0) {
      if (i + 2 == nk) roots[nr++] = hi;  // interior knots are picked up as `lo` of the next interval
      continue;
    }
    if ((flo < 0) == (fhi < 0)) continue;
    for (int it = 0; it < 200; it++) {
      double mid = 0.5 * (lo + hi);
      if (mid == lo || mid == hi) break;
      double fm = poly_eval(a, deg, mid);
      if (fm == 0) {
        lo = hi = mid;
        break;
      }
      if ((fm < 0) == (flo < 0)) {
        lo = mid;
        flo = fm;
      } else {
        hi = mid;
      }
    }
    roots[nr++] = 0.5 * (lo + hi);
  }
  return nr;
}

// ------------------------------------------------------------------------------------------ camera maps
// cv::projectPoints(Point3f, rvec, tvec, K, D[k1 k2 p1 p2]) -> Point2f.  R is taken from the pose quaternion directly
// (the reference goes SE3 -> Rodrigues -> rvec -> Rodrigues, an identity up to rounding).
void project_points(const float* p3d, int n, const SE3& T, const double K[4], const double D[4], float* out) {
  Mat3 R = quat_to_mat(T.q);
  for (int i = 0; i < n; i++) {
    Vec3 P{(double)p3d[3 * i], (double)p3d[3 * i + 1], (double)p3d[3 * i + 2]};
    Vec3 X = R * P + T.t;
    double z = X.z ? 1. / X.z : 1;
    double x = X.x * z, y = X.y * z;
    double r2 = x * x + y * y, r4 = r2 * r2;
    double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    double cdist = 1 + D[0] * r2 + D[1] * r4;
    double xd = x * cdist + D[2] * a1 + D[3] * a2;
    double yd = y * cdist + D[2] * a3 + D[3] * a1;
    out[2 * i] = (float)(xd * K[0] + K[2]);
    out[2 * i + 1] = (float)(yd * K[1] + K[3]);
  }
}

// cv::undistortPoints(src, dst, K, D, R, P): 5 fixed-point iterations of the distortion model, then R and P.
void undistort_points(const float* src, int n, const double K[4], const double D[4], const Mat3& R, const double P[12],
                      float* dst) {
  for (int i = 0; i < n; i++) {
    double x = (src[2 * i] - K[2]) / K[0], y = (src[2 * i + 1] - K[3]) / K[1];
    double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
      double r2 = x * x + y * y;
      double icdist = 1. / (1 + (D[1] * r2 + D[0]) * r2);
      double deltaX = 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x);
      double deltaY = D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    double xx = R.m[0][0] * x + R.m[0][1] * y + R.m[0][2];
    double yy = R.m[1][0] * x + R.m[1][1] * y + R.m[1][2];
    double ww = 1. / (R.m[2][0] * x + R.m[2][1] * y + R.m[2][2]);
    x = xx * ww;
    y = yy * ww;
    dst[2 * i] = (float)(x * P[0] + P[2]);       // fx' = P(0,0), cx' = P(0,2)
    dst[2 * i + 1] = (float)(y * P[5] + P[6]);   // fy' = P(1,1), cy' = P(1,2)
  }
}

// ------------------------------------------------------------------------------------------ DLT triangulation
// Smallest right singular vector of the 4x4 A by one-sided (Hestenes) Jacobi; X = V[:,min] / V[3,min].
Vec3 triangulate_dlt(Vec2 pt1, Vec2 pt2, const double P1[12], const double P2[12]) {
  double A[4][4], V[4][4];
  for (int j = 0; j < 4; j++) {
    A[0][j] = pt1.y * P1[8 + j] - P1[4 + j];
    A[1][j] = P1[j] - pt1.x * P1[8 + j];
    A[2][j] = pt2.y * P2[8 + j] - P2[4 + j];
    A[3][j] = P2[j] - pt2.x * P2[8 + j];
  }
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) V[i][j] = (i == j);
  for (int sweep = 0; sweep < 30; sweep++) {
    double off = 0;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 4; q++) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < 4; i++) {
          alpha += A[i][p] * A[i][p];
          beta += A[i][q] * A[i][q];
          gamma += A[i][p] * A[i][q];
        }
        // orthogonal to 10 eps: the convergence test of OpenCV's Jacobi SVD, |p| <= 10 DBL_EPSILON sqrt(a b).  (A 1e-16 test is below
        // the rounding of the dot product itself: one problem in twelve then chatters through all 30 sweeps.)
        if (gamma * gamma <= 4.930380657631324e-30 * (alpha * beta)) continue;
        off = 1;
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 4; i++) {
          double ap = A[i][p], aq = A[i][q];
          A[i][p] = c * ap - s * aq;
          A[i][q] = s * ap + c * aq;
          double vp = V[i][p], vq = V[i][q];
          V[i][p] = c * vp - s * vq;
          V[i][q] = s * vp + c * vq;
        }
      }
    if (off == 0) break;
  }
  int best = 0;
  double bn = DBL_MAX;
  for (int j = 0; j < 4; j++) {
    double nn = 0;
    for (int i = 0; i < 4; i++) nn += A[i][j] * A[i][j];
    if (nn < bn) {
      bn = nn;
      best = j;
    }
  }
  return {V[0][best] / V[3][best], V[1][best] / V[3][best], V[2][best] / V[3][best]};
}

// Triangulation::triangulationPt(pt1, pt2, T_c_w1, T_c_w2, fx,fy,cx,cy)  triangulation.cpp:80-97
Vec3 triangulate_two_view(Vec2 pt1, Vec2 pt2, const SE3& T1, const SE3& T2, double fx, double fy, double cx, double cy) {
  double P1[12], P2[12];
  const SE3* Ts[2] = {&T1, &T2};
  double* Ps[2] = {P1, P2};
  for (int k = 0; k < 2; k++) {
    Mat3 R = quat_to_mat(Ts[k]->q);
    Vec3 t = Ts[k]->t;
    double T34[3][4] = {{R.m[0][0], R.m[0][1], R.m[0][2], t.x}, {R.m[1][0], R.m[1][1], R.m[1][2], t.y},
                        {R.m[2][0], R.m[2][1], R.m[2][2], t.z}};
    for (int j = 0; j < 4; j++) {
      Ps[k][j] = fx * T34[0][j] + 0 * T34[1][j] + cx * T34[2][j];
      Ps[k][4 + j] = 0 * T34[0][j] + fy * T34[1][j] + cy * T34[2][j];
      Ps[k][8 + j] = 0 * T34[0][j] + 0 * T34[1][j] + 1 * T34[2][j];
    }
  }
  return triangulate_dlt(pt1, pt2, P1, P2);
}

// ------------------------------------------------------------------------------------------ RANSAC plumbing
// cv::RANSACUpdateNumIters
int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
  p = std::max(p, 0.);
  p = std::min(p, 1.);
  ep = std::max(ep, 0.);
  ep = std::min(ep, 1.);
  double num = std::max(1. - p, DBL_MIN);
  double denom = 1. - detm::det_powi(1. - ep, modelPoints);
  if (denom < DBL_MIN) return 0;
  num = detm::det_log(num);
  denom = detm::det_log(denom);
  return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)lrint(num / denom);
}

// RANSACPointSetRegistrator::getSubset (calib3d/src/ptsetreg.cpp, OpenCV 3.x; checkPartialSubsets is false): per slot an index is
// drawn with rng.uniform(0, count) and redrawn while it repeats an earlier slot; the complete subset is then offered to the callback's
// checkSubset and redrawn as a whole (one more attempt) when that refuses it.  `check` may be null (PnPRansacCallback does not override
// checkSubset: always true).  false after maxAttempts refused subsets.
bool cv_get_subset(CvRNG& rng, int count, int modelPoints, int maxAttempts, bool (*check)(const int* idx, int m, const void* ctx),
                   const void* ctx, int* idx) {
  int iters = 0, i = 0;
  for (; iters < maxAttempts; iters++) {
    for (i = 0; i < modelPoints && iters < maxAttempts;) {
      int idx_i;
      for (;;) {
        idx_i = idx[i] = rng.uniform(0, count);
        int j = 0;
        for (; j < i; j++)
          if (idx_i == idx[j]) break;
        if (j == i) break;
      }
      i++;
    }
    if (i == modelPoints && check && !check(idx, i, ctx)) continue;
    break;
  }
  return i == modelPoints && iters < maxAttempts;
}

// haveCollinearPoints (fundam.cpp): the LAST point of the subset against every pair of earlier ones, on Point2f coordinates
static bool have_collinear_points(const float* m, const int* idx, int count) {
  const int i = count - 1;
  for (int j = 0; j < i; j++) {
    const double dx1 = m[2 * idx[j]] - m[2 * idx[i]], dy1 = m[2 * idx[j] + 1] - m[2 * idx[i] + 1];
    for (int k = 0; k < j; k++) {
      const double dx2 = m[2 * idx[k]] - m[2 * idx[i]], dy2 = m[2 * idx[k] + 1] - m[2 * idx[i] + 1];
      if (std::fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2))) return true;
    }
  }
  return false;
}
struct FmCheckCtx {
  const float *m1, *m2;
};
// FMEstimatorCallback::checkSubset
static bool fm_check_subset(const int* idx, int m, const void* ctx) {
  const FmCheckCtx* c = (const FmCheckCtx*)ctx;
  return !have_collinear_points(c->m1, idx, m) && !have_collinear_points(c->m2, idx, m);
}

// ------------------------------------------------------------------------------------------ 7-point fundamental matrix
static double det3(const double* r0, const double* r1, const double* r2) {
  return r0[0] * (r1[1] * r2[2] - r1[2] * r2[1]) - r0[1] * (r1[0] * r2[2] - r1[2] * r2[0]) +
         r0[2] * (r1[0] * r2[1] - r1[1] * r2[0]);
}

// Returns up to 3 fundamental matrices (row-major 3x3, x2^T F x1 = 0).
int seven_point(const double x1[][2], const double x2[][2], double F[3][9]) {
  // Hartley normalisation of the 7 points
  double c1[2] = {0, 0}, c2[2] = {0, 0};
  for (int i = 0; i < 7; i++) {
    c1[0] += x1[i][0];
    c1[1] += x1[i][1];
    c2[0] += x2[i][0];
    c2[1] += x2[i][1];
  }
  for (int k = 0; k < 2; k++) {
    c1[k] /= 7;
    c2[k] /= 7;
  }
  double d1 = 0, d2 = 0;
  for (int i = 0; i < 7; i++) {
    d1 += std::sqrt((x1[i][0] - c1[0]) * (x1[i][0] - c1[0]) + (x1[i][1] - c1[1]) * (x1[i][1] - c1[1]));
    d2 += std::sqrt((x2[i][0] - c2[0]) * (x2[i][0] - c2[0]) + (x2[i][1] - c2[1]) * (x2[i][1] - c2[1]));
  }
  if (d1 < 1e-12 || d2 < 1e-12) return 0;
  const double s1 = std::sqrt(2.0) * 7 / d1, s2 = std::sqrt(2.0) * 7 / d2;
  double A[7][9];
  for (int i = 0; i < 7; i++) {
    double u1 = (x1[i][0] - c1[0]) * s1, v1 = (x1[i][1] - c1[1]) * s1;
    double u2 = (x2[i][0] - c2[0]) * s2, v2 = (x2[i][1] - c2[1]) * s2;
    double row[9] = {u2 * u1, u2 * v1, u2, v2 * u1, v2 * v1, v2, u1, v1, 1};
    memcpy(A[i], row, sizeof(row));
  }
  // Gauss-Jordan with full pivoting
  int pivcol[7];
  bool used[9] = {false};
  for (int r = 0; r < 7; r++) {
    int br = -1, bc = -1;
    double bv = 0;
    for (int i = r; i < 7; i++)
      for (int j = 0; j < 9; j++)
        if (!used[j] && std::fabs(A[i][j]) > bv) {
          bv = std::fabs(A[i][j]);
          br = i;
          bc = j;
        }
    if (bv < 1e-12) return 0;  // rank deficient sample
    if (br != r)
      for (int j = 0; j < 9; j++) std::swap(A[r][j], A[br][j]);
    used[bc] = true;
    pivcol[r] = bc;
    double inv = 1.0 / A[r][bc];
    for (int j = 0; j < 9; j++) A[r][j] *= inv;
    for (int i = 0; i < 7; i++)
      if (i != r) {
        double f = A[i][bc];
        if (f != 0)
          for (int j = 0; j < 9; j++) A[i][j] -= f * A[r][j];
      }
  }
  int freec[2], nf = 0;
  for (int j = 0; j < 9; j++)
    if (!used[j]) freec[nf++] = j;
  double f1[9], f2[9];
  double* fs[2] = {f1, f2};
  for (int k = 0; k < 2; k++) {
    for (int j = 0; j < 9; j++) fs[k][j] = 0;
    fs[k][freec[k]] = 1;
    for (int r = 0; r < 7; r++) fs[k][pivcol[r]] = -A[r][freec[k]];
  }
  // det(f2 + lambda (f1 - f2)) = 0
  double Bm[9];
  for (int j = 0; j < 9; j++) Bm[j] = f1[j] - f2[j];
  const double *a0 = f2, *a1 = f2 + 3, *a2 = f2 + 6, *b0 = Bm, *b1 = Bm + 3, *b2 = Bm + 6;
  double c[4];
  c[0] = det3(a0, a1, a2);
  c[1] = det3(b0, a1, a2) + det3(a0, b1, a2) + det3(a0, a1, b2);
  c[2] = det3(b0, b1, a2) + det3(b0, a1, b2) + det3(a0, b1, b2);
  c[3] = det3(b0, b1, b2);
  double roots[4];
  int nr = poly_real_roots(c, 3, roots);
  int nm = 0;
  for (int k = 0; k < nr && nm < 3; k++) {
    double Fh[9];
    for (int j = 0; j < 9; j++) Fh[j] = f2[j] + roots[k] * Bm[j];
    // F = T2^T Fh T1 with T = [s 0 -s cx; 0 s -s cy; 0 0 1]
    double T1[9] = {s1, 0, -s1 * c1[0], 0, s1, -s1 * c1[1], 0, 0, 1};
    double T2[9] = {s2, 0, -s2 * c2[0], 0, s2, -s2 * c2[1], 0, 0, 1};
    double tmp[9], Fo[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int m = 0; m < 3; m++) s += Fh[3 * i + m] * T1[3 * m + j];
        tmp[3 * i + j] = s;
      }
    double nn = 0;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int m = 0; m < 3; m++) s += T2[3 * m + i] * tmp[3 * m + j];
        Fo[3 * i + j] = s;
        nn += s * s;
      }
    if (!(nn > 0) || !std::isfinite(nn)) continue;
    double inv = 1.0 / std::sqrt(nn);
    for (int j = 0; j < 9; j++) F[nm][j] = Fo[j] * inv;
    nm++;
  }
  return nm;
}

// The 7-point solver the RANSAC / LMedS loops run.  Default (round 6): OpenCV's run7Point as restated in flvis_amd/csrc/cv_solvers.hpp (SVD
// null space of the unnormalised system, closed-form cubic, solutions in solveCubic's order, F(3,3) = 1).  -DFLVIS_SOLVERS_PRODUCT
// (`make -C oracle SOLVERS=product` -> libflvis_ref_prod.so) keeps the product-defined solver of rounds 1-5 above buildable; the distance
// between the two is measured on the lockstep sequences (tests/test_oracle_tracking.py, table in oracle/README.md).
static int seven_point_sel(const double x1[][2], const double x2[][2], double F[3][9]) {
#ifdef FLVIS_SOLVERS_PRODUCT
  return seven_point(x1, x2, F);
#else
  double wk[flvis::cvs::SP_WORK];
  return flvis::cvs::run7point<1>(x1, x2, wk, F, [](int) {});
#endif
}

// OpenCV FMEstimatorCallback::computeError: max of the two squared point-line distances, as float
static inline float f_error(const double* F, double x1, double y1, double x2, double y2) {
  double a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
  double s2 = 1. / (a * a + b * b);
  double dd2 = x2 * a + y2 * b + c;
  a = F[0] * x2 + F[3] * y2 + F[6];
  b = F[1] * x2 + F[4] * y2 + F[7];
  c = F[2] * x2 + F[5] * y2 + F[8];
  double s1 = 1. / (a * a + b * b);
  double dd1 = x1 * a + y1 * b + c;
  return (float)std::max(dd1 * dd1 * s1, dd2 * dd2 * s2);
}

// cv::findFundamentalMat(m1, m2, FM_RANSAC, thr, conf, mask) -> mask only (the reference discards F).  Returns #inliers.
// fundam.cpp: fewer than 7 points: nothing; exactly 7: the 7-point solver alone (all points kept); 8 .. 14 points: the LMedS registrator
// (`(method & ~3) == FM_RANSAC && npoints >= 15` selects RANSAC); from 15 points on RANSAC with 1000 iterations at most.  `seed` is
// unused: both registrators draw from their own RNG((uint64)-1).
static void fm_sample(const float* m1, const float* m2, const int* idx, double x1[7][2], double x2[7][2]) {
  for (int k = 0; k < 7; k++) {
    x1[k][0] = m1[2 * idx[k]];
    x1[k][1] = m1[2 * idx[k] + 1];
    x2[k][0] = m2[2 * idx[k]];
    x2[k][1] = m2[2 * idx[k] + 1];
  }
}

// LMeDSPointSetRegistrator::run (ptsetreg.cpp): a fixed number of subsets, the model with the smallest median error wins, the inliers
// are the points within sigma = 2.5 * 1.4826 * (1 + 5 / (count - modelPoints)) * sqrt(median) of it
static int find_fundamental_lmeds(const float* m1, const float* m2, int n, double conf, uint8_t* mask) {
  const int modelPoints = 7;
  CvRNG rng;
  const FmCheckCtx cc{m1, m2};
  int niters = ransac_update_num_iters(conf, 0.45, modelPoints, 1000);
  niters = std::max(niters, 3);
  double minMedian = DBL_MAX;
  double best[9] = {0};
  std::vector<float> err(n);
  for (int iter = 0; iter < niters; iter++) {
    int idx[7];
    if (!cv_get_subset(rng, n, modelPoints, 1000, fm_check_subset, &cc, idx)) {
      if (iter == 0) return 0;
      break;
    }
    double x1[7][2], x2[7][2], F[3][9];
    fm_sample(m1, m2, idx, x1, x2);
    const int nm = seven_point_sel(x1, x2, F);
    for (int m = 0; m < nm; m++) {
      for (int i = 0; i < n; i++) err[i] = f_error(F[m], m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1]);
      std::sort((int32_t*)err.data(), (int32_t*)err.data() + n);  // `std::sort(errf.ptr<int>(), ...)`: ordered through the bit patterns
      const double median = n % 2 != 0 ? (double)err[n / 2] : ((double)(float)(err[n / 2 - 1] + err[n / 2])) * 0.5;
      if (median < minMedian) {
        minMedian = median;
        memcpy(best, F[m], sizeof(best));
      }
    }
  }
  if (!(minMedian < DBL_MAX)) return 0;
  double sigma = 2.5 * 1.4826 * (1 + 5. / (n - modelPoints)) * std::sqrt(minMedian);
  sigma = std::max(sigma, 0.001);
  const float t = (float)(sigma * sigma);
  int good = 0;
  for (int i = 0; i < n; i++) {
    mask[i] = f_error(best, m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1]) <= t;
    good += mask[i];
  }
  return good;
}

int find_fundamental_ransac(const float* m1, const float* m2, int n, double thr, double conf, uint64_t /*seed*/, uint8_t* mask) {
  for (int i = 0; i < n; i++) mask[i] = 0;
  if (n < 7) return 0;
  const int modelPoints = 7;
  if (n == 7) {  // OpenCV: exactly 7 points -> no RANSAC, all points kept
    for (int i = 0; i < n; i++) mask[i] = 1;
    return n;
  }
  if (n < 15) return find_fundamental_lmeds(m1, m2, n, conf, mask);
  int niters = 1000;
  const float t = (float)(thr * thr);
  int maxGood = 0;
  std::vector<uint8_t> cur(n);
  CvRNG rng;
  const FmCheckCtx cc{m1, m2};
  for (int iter = 0; iter < niters; iter++) {
    int idx[7];
    if (!cv_get_subset(rng, n, modelPoints, 10000, fm_check_subset, &cc, idx)) break;
    double x1[7][2], x2[7][2];
    fm_sample(m1, m2, idx, x1, x2);
    double F[3][9];
    int nm = seven_point_sel(x1, x2, F);
    for (int m = 0; m < nm; m++) {
      int good = 0;
      for (int i = 0; i < n; i++) {
        float e = f_error(F[m], m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1]);
        cur[i] = (e <= t);
        good += cur[i];
      }
      if (good > std::max(maxGood, modelPoints - 1)) {
        memcpy(mask, cur.data(), n);
        maxGood = good;
        niters = ransac_update_num_iters(conf, (double)(n - good) / n, modelPoints, niters);
      }
    }
  }
  return maxGood;
}

// ------------------------------------------------------------------------------------------ P3P (Grunert)
// World points P[3], unit bearings f[3] -> up to 4 poses (R, t) with X_cam = R P + t.
int p3p_grunert(const Vec3 P[3], const Vec3 f[3], Mat3 Rs[4], Vec3 ts[4]) {
  double a2 = dot(P[1] - P[2], P[1] - P[2]), b2 = dot(P[0] - P[2], P[0] - P[2]), c2 = dot(P[0] - P[1], P[0] - P[1]);
  if (b2 < 1e-20 || a2 < 1e-20 || c2 < 1e-20) return 0;
  double ca = dot(f[1], f[2]), cb = dot(f[0], f[2]), cg = dot(f[0], f[1]);
  double A = (a2 - c2) / b2, C = c2 / b2;
  double q[5];
  q[4] = A * A - 2 * A - 4 * C * ca * ca + 1;
  q[3] = -4 * A * A * cb + 4 * A * ca * cg + 4 * A * cb + 8 * C * ca * ca * cb + 8 * C * ca * cg - 4 * ca * cg;
  q[2] = 4 * A * A * cb * cb + 2 * A * A - 8 * A * ca * cb * cg - 4 * A * cg * cg - 4 * C * ca * ca - 16 * C * ca * cb * cg -
         4 * C * cg * cg + 4 * ca * ca + 4 * cg * cg - 2;
  q[1] = -4 * A * A * cb + 4 * A * ca * cg + 8 * A * cb * cg * cg - 4 * A * cb + 8 * C * ca * cg + 8 * C * cb * cg * cg -
         4 * ca * cg;
  q[0] = A * A - 4 * A * cg * cg + 2 * A - 4 * C * cg * cg + 1;
  double roots[4];
  int nr = poly_real_roots(q, 4, roots);
  int ns = 0;
  for (int k = 0; k < nr && ns < 4; k++) {
    double v = roots[k];
    if (!(v > 0)) continue;
    double den = 2 * (cg - v * ca);
    if (std::fabs(den) < 1e-12) continue;
    double u = ((A - 1) * v * v - 2 * A * cb * v + 1 + A) / den;
    if (!(u > 0)) continue;
    double dd = 1 + v * v - 2 * v * cb;
    if (!(dd > 0)) continue;
    double s1 = std::sqrt(b2 / dd), s2 = u * s1, s3 = v * s1;
    Vec3 X[3] = {s1 * f[0], s2 * f[1], s3 * f[2]};
    // rigid alignment by orthonormal triads
    Vec3 e1w = P[1] - P[0], e1c = X[1] - X[0];
    double n1w = norm(e1w), n1c = norm(e1c);
    if (n1w < 1e-12 || n1c < 1e-12) continue;
    e1w = (1 / n1w) * e1w;
    e1c = (1 / n1c) * e1c;
    Vec3 e3w = cross(e1w, P[2] - P[0]), e3c = cross(e1c, X[2] - X[0]);
    double n3w = norm(e3w), n3c = norm(e3c);
    if (n3w < 1e-12 || n3c < 1e-12) continue;
    e3w = (1 / n3w) * e3w;
    e3c = (1 / n3c) * e3c;
    Vec3 e2w = cross(e3w, e1w), e2c = cross(e3c, e1c);
    Mat3 R;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) R.m[i][j] = e1c[i] * e1w[j] + e2c[i] * e2w[j] + e3c[i] * e3w[j];
    Rs[ns] = R;
    ts[ns] = X[0] - R * P[0];
    ns++;
  }
  return ns;
}

static inline bool reproj(const Mat3& R, Vec3 t, Vec3 P, double fx, double fy, double cx, double cy, double& u, double& v) {
  Vec3 X = R * P + t;
  double z = X.z ? 1. / X.z : 1;  // cv::projectPoints convention
  u = fx * X.x * z + cx;
  v = fy * X.y * z + cy;
  return true;
}

// 6x6 SPD solve (Cholesky); false if not positive definite
bool solve_spd6(const double H[36], const double b[6], double x[6]) {
  double L[36] = {0};
  for (int j = 0; j < 6; j++) {
    double s = H[6 * j + j];
    for (int k = 0; k < j; k++) s -= L[6 * j + k] * L[6 * j + k];
    if (!(s > 0)) return false;
    L[6 * j + j] = std::sqrt(s);
    for (int i = j + 1; i < 6; i++) {
      double v = H[6 * i + j];
      for (int k = 0; k < j; k++) v -= L[6 * i + k] * L[6 * j + k];
      L[6 * i + j] = v / L[6 * j + j];
    }
  }
  double y[6];
  for (int i = 0; i < 6; i++) {
    double v = b[i];
    for (int k = 0; k < i; k++) v -= L[6 * i + k] * y[k];
    y[i] = v / L[6 * i + i];
  }
  for (int i = 5; i >= 0; i--) {
    double v = y[i];
    for (int k = i + 1; k < 6; k++) v -= L[6 * k + i] * x[k];
    x[i] = v / L[6 * i + i];
  }
  return true;
}

// g2o EdgeSE3ProjectXYZ: error and pose Jacobian (types_six_dof_expmap.cpp:389-433), tangent = (omega, upsilon)
static inline void proj_edge(const SE3& T, Vec3 pw, Vec2 z, double fx, double fy, double cx, double cy, double e[2],
                             double J[2][6]) {
  Vec3 X = g2o_map(T, pw);
  double x = X.x, y = X.y, zz = X.z, z2 = zz * zz;
  e[0] = z.x - (x / zz * fx + cx);
  e[1] = z.y - (y / zz * fy + cy);
  if (J) {
    J[0][0] = x * y / z2 * fx;
    J[0][1] = -(1 + (x * x / z2)) * fx;
    J[0][2] = y / zz * fx;
    J[0][3] = -1. / zz * fx;
    J[0][4] = 0;
    J[0][5] = x / z2 * fx;
    J[1][0] = (1 + y * y / z2) * fy;
    J[1][1] = -x * y / z2 * fy;
    J[1][2] = -x / zz * fy;
    J[1][3] = 0;
    J[1][4] = -1. / zz * fy;
    J[1][5] = y / z2 * fy;
  }
}

// Gauss-Newton refinement of T on the given correspondences (stand-in for OpenCV's final solvePnP on the inliers)
static void pnp_refine(SE3& T, const std::vector<Vec3>& pw, const std::vector<Vec2>& z, double fx, double fy, double cx,
                       double cy) {
  for (int it = 0; it < 10; it++) {
    double H[36] = {0}, b[6] = {0};
    for (size_t i = 0; i < pw.size(); i++) {
      double e[2], J[2][6];
      proj_edge(T, pw[i], z[i], fx, fy, cx, cy, e, J);
      for (int r = 0; r < 6; r++) {
        b[r] -= J[0][r] * e[0] + J[1][r] * e[1];
        for (int c = 0; c < 6; c++) H[6 * r + c] += J[0][r] * J[0][c] + J[1][r] * J[1][c];
      }
    }
    double dx[6];
    if (!solve_spd6(H, b, dx)) break;
    // CvLevMarq's termination test of cvFindExtrinsicCameraParams2 (criteria: 20 iterations / FLT_EPSILON): the relative change of
    // the six parameters (rotation vector, translation) -- evaluated on the increment against the norm of the parameters before
    // the step, the rotation vector's length taken as 2 |q_v|
    const double pn = 4.0 * (T.q.x * T.q.x + T.q.y * T.q.y + T.q.z * T.q.z) + (T.t.x * T.t.x + T.t.y * T.t.y + T.t.z * T.t.z);
    T = g2o_mul(g2o_exp(dx), T);
    double nn = 0;
    for (int k = 0; k < 6; k++) nn += dx[k] * dx[k];
    if (nn < 1e-20 || nn < 1.4210854715202004e-14 * pn) break;  // FLT_EPSILON^2
  }
}

// cv::solvePnP(..., SOLVEPNP_EPNP) on the correspondences idx[0..n) (idx == nullptr: all of 0..n): `undistortPoints` to normalised
// coordinates stored as float (zero distortion here), which epnp::init_points maps back with `x * fu + uc`; epnp::compute_pose.
static thread_local flvis::epnp::Work g_epnp_last;  // (tests look at the intermediate values of the last solve)
bool solve_epnp(const float* p3d, const float* p2d, const int* idx, int n, double fx, double fy, double cx, double cy, Mat3& R, Vec3& t) {
  flvis::epnp::Work& w = g_epnp_last;
  auto pw = [&](int i, double* p) {
    const int k = idx ? idx[i] : i;
    p[0] = (double)p3d[3 * k], p[1] = (double)p3d[3 * k + 1], p[2] = (double)p3d[3 * k + 2];
  };
  auto uv = [&](int i, double* z) {
    const int k = idx ? idx[i] : i;
    z[0] = (double)(float)(((double)p2d[2 * k] - cx) / fx) * fx + cx;
    z[1] = (double)(float)(((double)p2d[2 * k + 1] - cy) / fy) * fy + cy;
  };
  static thread_local double part[flvis::epnp::PART_DOUBLES];
  if (n > 1024) return false;  // (the chunk-sum scratch is sized for 1024 correspondences, the product's capacity)
  const flvis::epnp::Pose P = flvis::epnp::solve<1>(w, n, pw, uv, flvis::epnp::Camera{fx, fy, cx, cy}, part, 0, [] {});
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R.m[i][j] = P.R[3 * i + j];
  t = {P.t[0], P.t[1], P.t[2]};
  return P.ok;
}

// cv::solvePnPRansac(p3d, p2d, K_rect, D=0, r, t, false, iters, reprojErr, conf, inliers, ITERATIVE|P3P).
// The RANSAC kernel: SOLVEPNP_ITERATIVE -> EPnP on 5-point subsets; SOLVEPNP_P3P -> P3P on 4-point subsets (three points give up to four
// poses, the fourth picks the one that reprojects it best, as cv::p3p::solve does).  The final solve on the inliers:
// SOLVEPNP_ITERATIVE -> the iterative minimisation of the reprojection error (here: 10 Gauss-Newton steps from the winning model -- the
// minimum OpenCV's Levenberg-Marquardt converges to from its own DLT start); SOLVEPNP_P3P -> EPnP on all inliers, unrefined.
// p3d/p2d are the float-cast values (camera_frame.cpp:415-427).  Returns #inliers (0 = no model: T untouched).
int solve_pnp_ransac(const float* p3d, const float* p2d, int n, double fx, double fy, double cx, double cy,
                     bool iterative_flag, int iterations, double reproj_err, double conf, uint64_t /*seed*/, SE3& T,
                     uint8_t* mask) {
  for (int i = 0; i < n; i++) mask[i] = 0;
  const int modelPoints = iterative_flag ? 5 : 4;
  if (n < modelPoints) return 0;
  int niters = iterations;
  const float t2 = (float)(reproj_err * reproj_err);
  int maxGood = 0;
  Mat3 bestR = mat3_identity();
  Vec3 bestt{0, 0, 0};
  std::vector<uint8_t> cur(n);
  CvRNG rng;  // RANSACPointSetRegistrator::run: RNG rng((uint64)-1)
  if (n == modelPoints) niters = 1;  // (solvePnPRansac hands exactly model_points points to solvePnP directly: one hypothesis, no draw)
  for (int iter = 0; iter < niters; iter++) {
    int idx[5];
    if (n == modelPoints) {
      for (int k = 0; k < modelPoints; k++) idx[k] = k;
    } else if (!cv_get_subset(rng, n, modelPoints, 10000, nullptr, nullptr, idx)) {
      break;
    }
    Mat3 Rh = mat3_identity();
    Vec3 th{0, 0, 0};
    if (iterative_flag) {
      if (!solve_epnp(p3d, p2d, idx, 5, fx, fy, cx, cy, Rh, th)) continue;
    } else {
#ifdef FLVIS_SOLVERS_PRODUCT
      Vec3 P[3], f[3];
      for (int k = 0; k < 3; k++) {
        P[k] = {(double)p3d[3 * idx[k]], (double)p3d[3 * idx[k] + 1], (double)p3d[3 * idx[k] + 2]};
        Vec3 d{((double)p2d[2 * idx[k]] - cx) / fx, ((double)p2d[2 * idx[k] + 1] - cy) / fy, 1.0};
        f[k] = (1.0 / norm(d)) * d;
      }
      Mat3 Rs[4];
      Vec3 ts[4];
      int ns = p3p_grunert(P, f, Rs, ts);
      if (ns == 0) continue;
      int bk = -1;
      double be = DBL_MAX;
      for (int k = 0; k < ns; k++) {
        double e = 0;
        for (int m = 3; m < modelPoints; m++) {
          Vec3 Pm{(double)p3d[3 * idx[m]], (double)p3d[3 * idx[m] + 1], (double)p3d[3 * idx[m] + 2]};
          double u, v;
          reproj(Rs[k], ts[k], Pm, fx, fy, cx, cy, u, v);
          double du = u - (double)p2d[2 * idx[m]], dv = v - (double)p2d[2 * idx[m] + 1];
          e += du * du + dv * dv;
        }
        if (e < be) {
          be = e;
          bk = k;
        }
      }
      if (bk < 0) continue;
      Rh = Rs[bk];
      th = ts[bk];
#else
      // cv::solvePnP(SOLVEPNP_P3P) on the four sample points: undistortPoints to normalised float coordinates, p3p::extract_points maps
      // them back with x * fx + cx (p3p.cpp); Gao's solver on the first three, the fourth picks the pose (cv_solvers.hpp)
      const flvis::cvs::P3PCamera cam = flvis::cvs::p3p_camera(fx, fy, cx, cy);
      double uv[4][2], X[4][3], Rr[9], tr[3];
      for (int k = 0; k < 4; k++) {
        uv[k][0] = (double)(float)(((double)p2d[2 * idx[k]] - cx) / fx) * fx + cx;
        uv[k][1] = (double)(float)(((double)p2d[2 * idx[k] + 1] - cy) / fy) * fy + cy;
        for (int j = 0; j < 3; j++) X[k][j] = (double)p3d[3 * idx[k] + j];
      }
      if (!flvis::cvs::p3p_solve4(cam, uv, X, Rr, tr)) continue;
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rh.m[i][j] = Rr[3 * i + j];
      th = {tr[0], tr[1], tr[2]};
#endif
    }
    int good = 0;
    for (int i = 0; i < n; i++) {
      Vec3 Pi{(double)p3d[3 * i], (double)p3d[3 * i + 1], (double)p3d[3 * i + 2]};
      double u, v;
      reproj(Rh, th, Pi, fx, fy, cx, cy, u, v);
      float du = (float)u - p2d[2 * i], dv = (float)v - p2d[2 * i + 1];  // projectPoints -> Point2f, then L2SQR in float
      float e = du * du + dv * dv;
      cur[i] = (e <= t2);
      good += cur[i];
    }
    if (good > std::max(maxGood, modelPoints - 1)) {
      memcpy(mask, cur.data(), n);
      maxGood = good;
      bestR = Rh;
      bestt = th;
      niters = ransac_update_num_iters(conf, (double)(n - good) / n, modelPoints, niters);
    }
  }
  if (maxGood == 0) return 0;
  if (!iterative_flag) {  // SOLVEPNP_P3P: the final solvePnP on the inliers runs SOLVEPNP_EPNP
    std::vector<int> inl;
    for (int i = 0; i < n; i++)
      if (mask[i]) inl.push_back(i);
    Mat3 Rf;
    Vec3 tf;
    if (solve_epnp(p3d, p2d, inl.data(), (int)inl.size(), fx, fy, cx, cy, Rf, tf)) {
      const SE3 Te = g2o_from_mat(Rf, tf);
      T = se3_from_mat(quat_to_mat(Te.q), Te.t);  // SE3_from_rvec_tvec (common.h:151-158)
    } else {
      const SE3 Te = g2o_from_mat(bestR, bestt);
      T = se3_from_mat(quat_to_mat(Te.q), Te.t);
    }
    return maxGood;
  }
  SE3 Tb = g2o_from_mat(bestR, bestt);
  std::vector<Vec3> pw;
  std::vector<Vec2> z;
  for (int i = 0; i < n; i++)
    if (mask[i]) {
      pw.push_back({(double)p3d[3 * i], (double)p3d[3 * i + 1], (double)p3d[3 * i + 2]});
      z.push_back({(double)p2d[2 * i], (double)p2d[2 * i + 1]});
    }
#ifdef FLVIS_TAIL_CV
  // `make -C oracle TAIL=cv` (libflvis_ref_cvtail.so): the final solve as cv::solvePnP(ITERATIVE, useExtrinsicGuess = false) runs it -- a DLT
  // start and CvLevMarq on the inliers (cv_solvers.hpp: find_extrinsic_iterative), independent of the RANSAC's winning model.  Kept beside
  // the default (Gauss-Newton from the winning model, what the kernels run) to measure the distance (tests/test_oracle_tracking.py).
  {
    const int ni = (int)pw.size();
    std::vector<double> Mw(3 * (size_t)ni), mz(2 * (size_t)ni), work(24 * (size_t)ni + 192);
    for (int i = 0; i < ni; i++) {
      Mw[3 * i] = pw[i].x, Mw[3 * i + 1] = pw[i].y, Mw[3 * i + 2] = pw[i].z;
      mz[2 * i] = z[i].x, mz[2 * i + 1] = z[i].y;
    }
    double rv[3], tv[3];
    if (flvis::cvs::find_extrinsic_iterative(ni, Mw.data(), mz.data(), fx, fy, cx, cy, work.data(), rv, tv, nullptr)) {
      double Rm[9];
      flvis::cvs::rodrigues(rv, Rm, nullptr);
      Mat3 R;
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R.m[i][j] = Rm[3 * i + j];
      T = se3_from_mat(R, {tv[0], tv[1], tv[2]});
      return maxGood;
    }
  }
#endif
  pnp_refine(Tb, pw, z, fx, fy, cx, cy);
  // SE3_from_rvec_tvec: Rodrigues(rvec) -> R -> SE3(R,t)   (common.h:151-158)
  T = se3_from_mat(quat_to_mat(Tb.q), Tb.t);
  return maxGood;
}

// ------------------------------------------------------------------------------------------ pose-only LM (g2o)
// One g2o optimize(n_iter) call on a single free VertexSE3Expmap with fixed landmarks; edges given in active order.
struct PoseEdge {
  Vec3 pw;
  Vec2 z;
  int64_t id;
  bool alive;
};

// Sums over the active edges (chi2, H, b): g2o adds edge after edge; here -- an fp64 rounding-order choice, stated as a deviation in
// DESIGN.md -- 32 consecutive edges of the active-edge list (positions 32 c .. 32 c + 31, dead edges counted) are summed in order, then
// the chunk sums in order: the same minimum to rounding, and a dependent chain an order of magnitude shorter on the device, which
// follows this very definition (k_pose_lm, PL_CH).
// `make -C oracle REF_ORDER=g2o` (-DFLVIS_REF_ORDER_G2O -> libflvis_ref_g2o.so) keeps the reference's own order buildable: edge after edge
// over the id-sorted active edges, as g2o's BaseBinaryEdge::constructQuadraticForm adds them (base_binary_edge.hpp:61-134, sparse_optimizer.cpp:
// 493-498) -- one "chunk" holds every edge.  tests/test_oracle_tracking.py runs the lockstep sequences on both builds: identical discrete
// outputs, poses within 1e-12.
#ifdef FLVIS_REF_ORDER_G2O
constexpr int POSE_CHUNK = 1 << 30;
#else
constexpr int POSE_CHUNK = 32;
#endif

static double robust_chi2(const SE3& T, const std::vector<PoseEdge>& E, double fx, double fy, double cx, double cy) {
  double chi = 0, part = 0;
  for (size_t k = 0; k < E.size(); k++) {
    if (k > 0 && k % POSE_CHUNK == 0) {
      chi += part;
      part = 0;
    }
    const PoseEdge& e = E[k];
    if (!e.alive) continue;
    double er[2];
    proj_edge(T, e.pw, e.z, fx, fy, cx, cy, er, nullptr);
    double c2 = er[0] * er[0] + er[1] * er[1];
    if (c2 <= 1.0)
      part += c2;
    else
      part += 2 * std::sqrt(c2) * 1.0 - 1.0;
  }
  if (!E.empty()) chi += part;
  return chi;
}

static void g2o_pose_optimize(SE3& T, const std::vector<PoseEdge>& E, int iterations, double fx, double fy, double cx,
                              double cy) {
  double lambda = -1, ni = 2;
  for (int iteration = 0; iteration < iterations; iteration++) {
    double currentChi = robust_chi2(T, E, fx, fy, cx, cy);
    double H[36] = {0}, b[6] = {0}, Hp[36] = {0}, bp[6] = {0};  // totals and the running chunk (see POSE_CHUNK)
    for (size_t k = 0; k < E.size(); k++) {
      if (k > 0 && k % POSE_CHUNK == 0) {
        for (int j = 0; j < 36; j++) H[j] += Hp[j], Hp[j] = 0;
        for (int j = 0; j < 6; j++) b[j] += bp[j], bp[j] = 0;
      }
      const PoseEdge& e = E[k];
      if (!e.alive) continue;
      double er[2], J[2][6];
      proj_edge(T, e.pw, e.z, fx, fy, cx, cy, er, J);
      double c2 = er[0] * er[0] + er[1] * er[1];
      double w = (c2 <= 1.0) ? 1.0 : 1.0 / std::sqrt(c2);  // rho'
      double o0 = -er[0] * w, o1 = -er[1] * w;               // omega_r * rho[1]
      for (int r = 0; r < 6; r++) {
        bp[r] += J[0][r] * o0 + J[1][r] * o1;
        for (int c = 0; c < 6; c++) Hp[6 * r + c] += (J[0][r] * w) * J[0][c] + (J[1][r] * w) * J[1][c];
      }
    }
    if (!E.empty()) {
      for (int j = 0; j < 36; j++) H[j] += Hp[j];
      for (int j = 0; j < 6; j++) b[j] += bp[j];
    }
    // the linear solver reads ONE triangle: g2o's LinearSolverEigen factorises SimplicialLDLT<SparseMatrix, Eigen::Upper>
    // (solvers/eigen/linear_solver_eigen.h:44).  With a robust weight w != 1 the two triangles of A^T (w Omega) A differ in
    // the last bit ((J_r w) J_c vs (J_c w) J_r), so the upper triangle is mirrored before the solve.
    for (int r = 0; r < 6; r++)
      for (int c = r + 1; c < 6; c++) H[6 * c + r] = H[6 * r + c];
    if (iteration == 0) {
      double maxDiag = 0;
      for (int j = 0; j < 6; j++) maxDiag = std::max(std::fabs(H[7 * j]), maxDiag);
      lambda = 1e-5 * maxDiag;
      ni = 2;
    }
    double rho = 0;
    int qmax = 0;
    bool lambda_bad = false;
    do {
      SE3 backup = T;
      double Hl[36];
      memcpy(Hl, H, sizeof(Hl));
      for (int j = 0; j < 6; j++) Hl[7 * j] += lambda;
      double x[6] = {0};
      bool ok2 = solve_spd6(Hl, b, x);
      if (ok2) T = g2o_mul(g2o_exp(x), T);
      double tempChi = robust_chi2(T, E, fx, fy, cx, cy);
      if (!ok2) tempChi = DBL_MAX;
      rho = currentChi - tempChi;
      double scale = 0;
      for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - detm::det_powi((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        double scaleFactor = std::max(1. / 3., alpha);
        lambda *= scaleFactor;
        ni = 2;
        currentChi = tempChi;
      } else {
        lambda *= ni;
        ni *= 2;
        T = backup;
        if (!std::isfinite(lambda)) {
          lambda_bad = true;
          break;
        }
      }
      qmax++;
    } while (rho < 0 && qmax < 10);
    if (qmax == 10 || rho == 0 || lambda_bad) break;  // Terminate
  }
}

// OptimizeInFrame::optimize.  lm arrays are the valid inlier pairs in frame order.  Returns false if <10 edges.
bool optimize_in_frame(SE3& T_c_w, const Vec3* lm_3d_w, const Vec2* lm_2d, const int64_t* lm_id, int n, double fx,
                       double fy, double cx, double cy) {
  if (n < 10) return false;
  std::vector<PoseEdge> E(n);
  for (int i = 0; i < n; i++) E[i] = {lm_3d_w[i], lm_2d[i], lm_id[i], true};
  // active edges are processed in edge-id order (sparse_optimizer.cpp:493-498); edge id = lm_id
  std::stable_sort(E.begin(), E.end(), [](const PoseEdge& a, const PoseEdge& b) { return a.id < b.id; });
  SE3 T = g2o_from_mat(quat_to_mat(T_c_w.q), T_c_w.t);
  g2o_pose_optimize(T, E, 2, fx, fy, cx, cy);
  int alive = 0;
  for (auto& e : E) {
    double er[2];
    proj_edge(T, e.pw, e.z, fx, fy, cx, cy, er, nullptr);
    if (er[0] * er[0] + er[1] * er[1] > 3.0) e.alive = false;
    alive += e.alive;
  }
  if (alive < 10) return false;
  g2o_pose_optimize(T, E, 2, fx, fy, cx, cy);
  T_c_w = se3_from_mat(quat_to_mat(T.q), T.t);
  return true;
}

}  // namespace ref

// ------------------------------------------------------------------------------------------ C entry points (ctypes)
extern "C" {
// which summation order this build of the checker uses: 0 = the product's chunk sums (the default, what the lockstep tests compare
// with), 1 = the reference's (REF_ORDER=g2o)
int ref_sum_order(void) {
#ifdef FLVIS_REF_ORDER_G2O
  return 1;
#else
  return 0;
#endif
}
int ref_poly_real_roots(const double* a, int deg, double* roots) { return ref::poly_real_roots(a, deg, roots); }
// the first `nsub` subsets a RANSAC run over `count` points draws (no checkSubset), and the raw generator outputs
void ref_cv_subsets(int count, int modelPoints, int nsub, int* idx_out) {
  ref::CvRNG rng;
  for (int k = 0; k < nsub; k++) ref::cv_get_subset(rng, count, modelPoints, 10000, nullptr, nullptr, idx_out + (size_t)k * modelPoints);
}
void ref_cv_rng_outputs(int n, unsigned* out) {
  ref::CvRNG rng;
  for (int k = 0; k < n; k++) out[k] = rng.next();
}

void ref_project_points(const float* p3d, int n, const double* pose7 /*tx ty tz qx qy qz qw*/, const double* K,
                        const double* D, float* out) {
  ref::SE3 T{{pose7[6], pose7[3], pose7[4], pose7[5]}, {pose7[0], pose7[1], pose7[2]}};
  ref::project_points(p3d, n, T, K, D, out);
}
void ref_undistort_points(const float* src, int n, const double* K, const double* D, const double* R9, const double* P12,
                          float* dst) {
  ref::Mat3 R;
  for (int i = 0; i < 9; i++) R.m[i / 3][i % 3] = R9[i];
  ref::undistort_points(src, n, K, D, R, P12, dst);
}
void ref_triangulate_dlt(const double* pt1, const double* pt2, const double* P1, const double* P2, double* out) {
  ref::Vec3 X = ref::triangulate_dlt({pt1[0], pt1[1]}, {pt2[0], pt2[1]}, P1, P2);
  out[0] = X.x;
  out[1] = X.y;
  out[2] = X.z;
}
int ref_seven_point(const double* x1, const double* x2, double* F27) {
  double a[7][2], b[7][2], F[3][9];
  for (int i = 0; i < 7; i++) {
    a[i][0] = x1[2 * i];
    a[i][1] = x1[2 * i + 1];
    b[i][0] = x2[2 * i];
    b[i][1] = x2[2 * i + 1];
  }
  int n = ref::seven_point(a, b, F);
  memcpy(F27, F, sizeof(F));
  return n;
}
int ref_find_fundamental_ransac(const float* m1, const float* m2, int n, double thr, double conf, uint64_t seed,
                                uint8_t* mask) {
  return ref::find_fundamental_ransac(m1, m2, n, thr, conf, seed, mask);
}
int ref_p3p(const double* P9, const double* f9, double* R36, double* t12) {
  ref::Vec3 P[3], f[3];
  for (int i = 0; i < 3; i++) {
    P[i] = {P9[3 * i], P9[3 * i + 1], P9[3 * i + 2]};
    f[i] = {f9[3 * i], f9[3 * i + 1], f9[3 * i + 2]};
  }
  ref::Mat3 Rs[4];
  ref::Vec3 ts[4];
  int n = ref::p3p_grunert(P, f, Rs, ts);
  for (int k = 0; k < n; k++) {
    for (int i = 0; i < 9; i++) R36[9 * k + i] = Rs[k].m[i / 3][i % 3];
    t12[3 * k] = ts[k].x;
    t12[3 * k + 1] = ts[k].y;
    t12[3 * k + 2] = ts[k].z;
  }
  return n;
}
// the OpenCV-shaped solvers of flvis_amd/csrc/cv_solvers.hpp, one call each (tests/test_oracle_cv_solvers.py)
int ref_cv_seven_point(const double* x1, const double* x2, double* F27) {
  double a[7][2], b[7][2], F[3][9], wk[flvis::cvs::SP_WORK];
  for (int i = 0; i < 7; i++) {
    a[i][0] = x1[2 * i];
    a[i][1] = x1[2 * i + 1];
    b[i][0] = x2[2 * i];
    b[i][1] = x2[2 * i + 1];
  }
  memset(F, 0, sizeof(F));
  const int n = flvis::cvs::run7point<1>(a, b, wk, F, [](int) {});
  memcpy(F27, F, sizeof(F));
  return n;
}
// the same solve in the kernel's schedule (anti-diagonals of two overlapping sweeps): must equal ref_cv_seven_point bit for bit
int ref_cv_seven_point_scheduled(const double* x1, const double* x2, double* F27) {
  double a[7][2], b[7][2], F[3][9], wk[flvis::cvs::SP_WORK];
  for (int i = 0; i < 7; i++) {
    a[i][0] = x1[2 * i];
    a[i][1] = x1[2 * i + 1];
    b[i][0] = x2[2 * i];
    b[i][1] = x2[2 * i + 1];
  }
  memset(F, 0, sizeof(F));
  const int n = flvis::cvs::run7point_scheduled<1>(a, b, wk, F, [](int) {});
  memcpy(F27, F, sizeof(F));
  return n;
}
int ref_cv_solve_cubic(const double* c4, double* r3) { return flvis::cvs::solve_cubic(c4, r3); }
int ref_cv_solve_deg4(const double* c5, double* r4) {
  r4[0] = r4[1] = r4[2] = r4[3] = 0;
  return flvis::cvs::solve_deg4(c5[0], c5[1], c5[2], c5[3], c5[4], r4[0], r4[1], r4[2], r4[3]);
}
int ref_cv_p3p(const double* K4, const double* uv6, const double* X9, double* R36, double* t12) {
  const flvis::cvs::P3PCamera cam = flvis::cvs::p3p_camera(K4[0], K4[1], K4[2], K4[3]);
  double uv[3][2], X[3][3], R[4][9], t[4][3];
  for (int k = 0; k < 3; k++) {
    uv[k][0] = uv6[2 * k], uv[k][1] = uv6[2 * k + 1];
    for (int j = 0; j < 3; j++) X[k][j] = X9[3 * k + j];
  }
  const int n = flvis::cvs::p3p_solve3(cam, uv, X, R, t);
  for (int k = 0; k < n; k++) {
    memcpy(R36 + 9 * k, R[k], sizeof(double) * 9);
    memcpy(t12 + 3 * k, t[k], sizeof(double) * 3);
  }
  return n;
}
// cvFindExtrinsicCameraParams2 without a guess (cv_solvers.hpp): pose from n >= 6 non-planar correspondences; returns the LM iterations (0: refused)
// ... the same with its loops dealt to `lanes` host threads the way k_pnp_tail_cv deals them to the lanes of a wave (cv_solvers.hpp: a
// lane takes whole points or whole sums; `sync` is a barrier of the threads): the test expects the serial call's bits.
namespace {
struct ThreadLanes {
  int l, nl;
  struct Bar {
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0, gen = 0;
  }* bar;
  int lane() const { return l; }
  int lanes() const { return nl; }
  void sync() const {
    std::unique_lock<std::mutex> lk(bar->mu);
    const int g = bar->gen;
    if (++bar->waiting == nl) {
      bar->waiting = 0;
      bar->gen++;
      bar->cv.notify_all();
    } else {
      bar->cv.wait(lk, [&] { return bar->gen != g; });
    }
  }
};
}  // namespace
int ref_cv_find_extrinsic_lanes(int n, const double* M, const double* m, const double* K4, int lanes, double* rvec3, double* tvec3) {
  std::vector<double> work(24 * (size_t)n + 192);
  ThreadLanes::Bar bar;
  std::vector<int> its(lanes, 0), oks(lanes, 0);
  std::vector<double> rv(3 * (size_t)lanes), tv(3 * (size_t)lanes);
  std::vector<std::thread> th;
  for (int l = 0; l < lanes; l++)
    th.emplace_back([&, l] {
      oks[l] = flvis::cvs::find_extrinsic_iterative(n, M, m, K4[0], K4[1], K4[2], K4[3], work.data(), &rv[3 * l], &tv[3 * l], &its[l], ThreadLanes{l, lanes, &bar});
    });
  for (auto& t : th) t.join();
  for (int l = 1; l < lanes; l++)  // every lane leaves with the same answer
    if (oks[l] != oks[0] || its[l] != its[0] || memcmp(&rv[3 * l], &rv[0], 24) || memcmp(&tv[3 * l], &tv[0], 24)) return -1;
  memcpy(rvec3, &rv[0], 24);
  memcpy(tvec3, &tv[0], 24);
  return oks[0] ? its[0] : 0;
}
int ref_cv_find_extrinsic(int n, const double* M, const double* m, const double* K4, double* rvec3, double* tvec3) {
  std::vector<double> work(24 * (size_t)n + 192);
  int it = 0;
  if (!flvis::cvs::find_extrinsic_iterative(n, M, m, K4[0], K4[1], K4[2], K4[3], work.data(), rvec3, tvec3, &it)) return 0;
  return it;
}
void ref_cv_rodrigues(const double* r3, double* R9, double* J27) { flvis::cvs::rodrigues(r3, R9, J27); }
void ref_cv_rodrigues_inv(const double* R9, double* r3) { flvis::cvs::rodrigues_inv(R9, r3); }
void ref_cv_svd_square(const double* A, int n, double* w, double* u, double* vt) { flvis::cvs::svd_square(A, n, w, u, vt); }
int ref_cv_jacobi4(const double* A16, double* D4, double* U16) {
  double A[16];
  memcpy(A, A16, sizeof(A));
  return flvis::cvs::jacobi_4x4(A, D4, U16) ? 1 : 0;
}
int ref_solve_epnp(const float* p3d, const float* p2d, int n, const double* K4, double* R9, double* t3) {
  ref::Mat3 R;
  ref::Vec3 t;
  const bool ok = ref::solve_epnp(p3d, p2d, nullptr, n, K4[0], K4[1], K4[2], K4[3], R, t);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R9[3 * i + j] = R.m[i][j];
  t3[0] = t.x, t3[1] = t.y, t3[2] = t.z;
  return ok ? 1 : 0;
}
// the 12 x 12 symmetric eigen-decomposition of epnp_core.hpp on its own: eigenvalues (unsorted) and eigenvectors (columns of V)
void ref_epnp_last(double* betas12, double* err3, double* v48, double* L60, double* rho6) {
  const flvis::epnp::Work& w = ref::g_epnp_last;
  for (int q = 0; q < 3; q++) {
    err3[q] = w.err[q];
    for (int i = 0; i < 4; i++) betas12[4 * q + i] = w.betas[q][i];
  }
  for (int i = 0; i < 48; i++) v48[i] = w.v[i / 12][i % 12];
  for (int i = 0; i < 60; i++) L60[i] = w.L[i];
  for (int i = 0; i < 6; i++) rho6[i] = w.rho[i];
}
int ref_epnp_jacobi12(const double* A144, double* evals12, double* V144) {
  static flvis::epnp::Work w;
  for (int e = 0; e < 144; e++) w.AV[e] = A144[e], w.AV[144 + e] = (e / 12 == e % 12) ? 1.0 : 0.0;
  flvis::epnp::jacobi12_setup(w, 0);
  int steps = 0;
  flvis::epnp::jacobi12(w, 0, 1, [&] { steps++; });
  for (int i = 0; i < 12; i++) evals12[i] = w.AV[13 * i];
  for (int e = 0; e < 144; e++) V144[e] = w.AV[144 + e];
  return steps / 24;  // sweeps: 24 phase boundaries each
}
int ref_solve_pnp_ransac(const float* p3d, const float* p2d, int n, const double* K4, int iterative_flag, int iterations,
                         double reproj_err, double conf, uint64_t seed, double* pose7_inout, uint8_t* mask) {
  ref::SE3 T{{pose7_inout[6], pose7_inout[3], pose7_inout[4], pose7_inout[5]},
             {pose7_inout[0], pose7_inout[1], pose7_inout[2]}};
  int r = ref::solve_pnp_ransac(p3d, p2d, n, K4[0], K4[1], K4[2], K4[3], iterative_flag != 0, iterations, reproj_err,
                                conf, seed, T, mask);
  pose7_inout[0] = T.t.x;
  pose7_inout[1] = T.t.y;
  pose7_inout[2] = T.t.z;
  pose7_inout[3] = T.q.x;
  pose7_inout[4] = T.q.y;
  pose7_inout[5] = T.q.z;
  pose7_inout[6] = T.q.w;
  return r;
}
int ref_optimize_in_frame(double* pose7_inout, const double* lm3d, const double* lm2d, const int64_t* ids, int n,
                          const double* K4) {
  ref::SE3 T{{pose7_inout[6], pose7_inout[3], pose7_inout[4], pose7_inout[5]},
             {pose7_inout[0], pose7_inout[1], pose7_inout[2]}};
  std::vector<ref::Vec3> p3(n);
  std::vector<ref::Vec2> p2(n);
  for (int i = 0; i < n; i++) {
    p3[i] = {lm3d[3 * i], lm3d[3 * i + 1], lm3d[3 * i + 2]};
    p2[i] = {lm2d[2 * i], lm2d[2 * i + 1]};
  }
  bool ok = ref::optimize_in_frame(T, p3.data(), p2.data(), ids, n, K4[0], K4[1], K4[2], K4[3]);
  pose7_inout[0] = T.t.x;
  pose7_inout[1] = T.t.y;
  pose7_inout[2] = T.t.z;
  pose7_inout[3] = T.q.x;
  pose7_inout[4] = T.q.y;
  pose7_inout[5] = T.q.z;
  pose7_inout[6] = T.q.w;
  return ok ? 1 : 0;
}
}
