// flvis_amd: device-side point geometry (fp64): camera maps, DLT triangulation, 7-point F, Grunert P3P, projection
// edge + small SPD solves.  Reference call sites: src/processing/lkorb_tracking.cpp:55-61,87,133-135,170-177;
// src/processing/camera_frame.cpp:113-131; src/processing/triangulation.cpp:9-97;
// 3rdPartLib/g2o/g2o/types/sba/types_six_dof_expmap.cpp:389-433.
#pragma once
#include "dev_math.hpp"

namespace flvis {

// cv::projectPoints for one Point3f (k1 k2 p1 p2 model) -> Point2f
FD void project_point(const float* p3, const M3& R, V3 t, const double* K, const double* D, float* out) {
  V3 P{(double)p3[0], (double)p3[1], (double)p3[2]};
  V3 X = R * P + t;
  double z = X.z ? 1. / X.z : 1;
  double x = X.x * z, y = X.y * z;
  double r2 = x * x + y * y, r4 = r2 * r2;
  double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
  double cdist = 1 + D[0] * r2 + D[1] * r4;
  double xd = x * cdist + D[2] * a1 + D[3] * a2;
  double yd = y * cdist + D[2] * a3 + D[3] * a1;
  out[0] = (float)(xd * K[0] + K[2]);
  out[1] = (float)(yd * K[1] + K[3]);
}

// cv::undistortPoints for one Point2f with (K, D, R, P)
FD void undistort_point(const float* src, const double* K, const double* D, const double* R9, const double* P12,
                        float* dst) {
  double x = (src[0] - K[2]) / K[0], y = (src[1] - K[3]) / K[1];
  double x0 = x, y0 = y;
  for (int j = 0; j < 5; j++) {
    double r2 = x * x + y * y;
    double icdist = 1. / (1 + (D[1] * r2 + D[0]) * r2);
    double deltaX = 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x);
    double deltaY = D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  double xx = R9[0] * x + R9[1] * y + R9[2];
  double yy = R9[3] * x + R9[4] * y + R9[5];
  double ww = 1. / (R9[6] * x + R9[7] * y + R9[8]);
  x = xx * ww;
  y = yy * ww;
  dst[0] = (float)(x * P12[0] + P12[2]);
  dst[1] = (float)(y * P12[5] + P12[6]);
}

// Triangulation::triangulationPt: smallest right singular vector of the 4x4 DLT matrix (one-sided Jacobi)
__device__ inline V3 triangulate_dlt(double u1, double v1, double u2, double v2, const double* P1, const double* P2) {
  double A[4][4], V[4][4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    A[0][j] = v1 * P1[8 + j] - P1[4 + j];
    A[1][j] = P1[j] - u1 * P1[8 + j];
    A[2][j] = v2 * P2[8 + j] - P2[4 + j];
    A[3][j] = P2[j] - u2 * P2[8 + j];
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; sweep++) {
    int off = 0;
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int q = p + 1; q < 4; q++) {
        double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          alpha += A[i][p] * A[i][p];
          beta += A[i][q] * A[i][q];
          gamma += A[i][p] * A[i][q];
        }
        // orthogonal to 10 eps (OpenCV's Jacobi SVD test).  Round 2's 1e-16 was below the rounding of the dot product: one problem in
        // twelve chattered through all 30 sweeps, and a wave runs as long as its slowest lane -- nearly every wave ran 30 sweeps.
        if (gamma * gamma <= 4.930380657631324e-30 * (alpha * beta)) continue;
        off = 1;
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          double ap = A[i][p], aq = A[i][q];
          A[i][p] = c * ap - s * aq;
          A[i][q] = s * ap + c * aq;
          double vp = V[i][p], vq = V[i][q];
          V[i][p] = c * vp - s * vq;
          V[i][q] = s * vp + c * vq;
        }
      }
    if (!off) break;
  }
  int best = 0;
  double bn = 1.7976931348623157e308;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    double nn = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) nn += A[i][j] * A[i][j];
    if (nn < bn) {
      bn = nn;
      best = j;
    }
  }
  double vx = 0, vy = 0, vz = 0, vw = 1;
#pragma unroll
  for (int j = 0; j < 4; j++)
    if (j == best) {
      vx = V[0][j];
      vy = V[1][j];
      vz = V[2][j];
      vw = V[3][j];
    }
  return V3{vx / vw, vy / vw, vz / vw};
}

__device__ inline V3 triangulate_two_view(double u1, double v1, double u2, double v2, const SE3d& T1, const SE3d& T2,
                                          double fx, double fy, double cx, double cy) {
  double P1[12], P2[12];
  for (int k = 0; k < 2; k++) {
    const SE3d& T = k == 0 ? T1 : T2;
    double* P = k == 0 ? P1 : P2;
    M3 R = q_to_mat(T.q);
    double T34[3][4] = {{R.m[0][0], R.m[0][1], R.m[0][2], T.t.x}, {R.m[1][0], R.m[1][1], R.m[1][2], T.t.y},
                        {R.m[2][0], R.m[2][1], R.m[2][2], T.t.z}};
    for (int j = 0; j < 4; j++) {
      P[j] = fx * T34[0][j] + 0 * T34[1][j] + cx * T34[2][j];
      P[4 + j] = 0 * T34[0][j] + fy * T34[1][j] + cy * T34[2][j];
      P[8 + j] = 0 * T34[0][j] + 0 * T34[1][j] + 1 * T34[2][j];
    }
  }
  return triangulate_dlt(u1, v1, u2, v2, P1, P2);
}

// 7-point fundamental matrix: Hartley-normalised, Gauss-Jordan null space, cubic by bracketing.  x1/x2: [7][2]
// 7-point fundamental matrix (Hartley normalisation, Gauss-Jordan null space with full pivoting, cubic in the pencil
// parameter).  The 7x9 system lives in a per-lane LDS workspace `wk` (element (i,j) at wk[(9*i+j)*WS]): pivoting indexes
// rows and columns dynamically, and a dynamically indexed local array would be placed in scratch memory (the elimination
// then runs at memory latency: it was 85% of k_ransac_f).  Column bookkeeping uses bit masks / selects.
#ifndef SP_STAMP
#define SP_STAMP(i) do { } while (0)
#endif
// state of a 7-point solve between the cubic's coefficients and its roots
struct SevenPointMid {
  double f2[9], Bm[9];
  double s1, s2, c1[2], c2[2];
};
// first half: normalisation, null space, coefficients c[0..3] of det(f2 + x (f1 - f2)).  false: degenerate sample, no model
template <int WS>
__device__ inline bool seven_point_a(const double (*x1)[2], const double (*x2)[2], double* wk, SevenPointMid& mid, double* c) {
#define SP_A(i, j) wk[(9 * (i) + (j)) * WS]
  double c1[2] = {0, 0}, c2[2] = {0, 0};
#pragma unroll
  for (int i = 0; i < 7; i++) {
    c1[0] += x1[i][0];
    c1[1] += x1[i][1];
    c2[0] += x2[i][0];
    c2[1] += x2[i][1];
  }
#pragma unroll
  for (int k = 0; k < 2; k++) {
    c1[k] /= 7;
    c2[k] /= 7;
  }
  double d1 = 0, d2 = 0;
#pragma unroll
  for (int i = 0; i < 7; i++) {
    d1 += sqrt((x1[i][0] - c1[0]) * (x1[i][0] - c1[0]) + (x1[i][1] - c1[1]) * (x1[i][1] - c1[1]));
    d2 += sqrt((x2[i][0] - c2[0]) * (x2[i][0] - c2[0]) + (x2[i][1] - c2[1]) * (x2[i][1] - c2[1]));
  }
  if (d1 < 1e-12 || d2 < 1e-12) return false;
  const double s1 = sqrt(2.0) * 7 / d1, s2 = sqrt(2.0) * 7 / d2;
#pragma unroll
  for (int i = 0; i < 7; i++) {
    double u1 = (x1[i][0] - c1[0]) * s1, v1 = (x1[i][1] - c1[1]) * s1;
    double u2 = (x2[i][0] - c2[0]) * s2, v2 = (x2[i][1] - c2[1]) * s2;
    SP_A(i, 0) = u2 * u1;
    SP_A(i, 1) = u2 * v1;
    SP_A(i, 2) = u2;
    SP_A(i, 3) = v2 * u1;
    SP_A(i, 4) = v2 * v1;
    SP_A(i, 5) = v2;
    SP_A(i, 6) = u1;
    SP_A(i, 7) = v1;
    SP_A(i, 8) = 1;
  }
  SP_STAMP(0);
  int pivcol[7];
  unsigned used = 0;
#pragma unroll
  for (int r = 0; r < 7; r++) {
    int br = -1, bc = -1;
    double bv = 0;
#pragma unroll
    for (int i = r; i < 7; i++) {
      double av[9];  // (all loads of a row are issued before the compare chain)
#pragma unroll
      for (int j = 0; j < 9; j++) av[j] = fabs(SP_A(i, j));
#pragma unroll
      for (int j = 0; j < 9; j++)
        if (!((used >> j) & 1u) && av[j] > bv) {
          bv = av[j];
          br = i;
          bc = j;
        }
    }
    if (bv < 1e-12) return false;
    if (br != r) {
      double ra[9], rb[9];
#pragma unroll
      for (int j = 0; j < 9; j++) {
        ra[j] = SP_A(r, j);
        rb[j] = SP_A(br, j);
      }
#pragma unroll
      for (int j = 0; j < 9; j++) {
        SP_A(r, j) = rb[j];
        SP_A(br, j) = ra[j];
      }
    }
    used |= 1u << bc;
    pivcol[r] = bc;
    double inv = 1.0 / SP_A(r, bc);
    double prow[9];
#pragma unroll
    for (int j = 0; j < 9; j++) {
      prow[j] = SP_A(r, j) * inv;
      SP_A(r, j) = prow[j];
    }
    double fcol[7];
#pragma unroll
    for (int i = 0; i < 7; i++) fcol[i] = SP_A(i, bc);
#pragma unroll
    for (int i = 0; i < 7; i++)
      if (i != r) {
        const double f = fcol[i];
        if (f != 0) {
          double row[9];
#pragma unroll
          for (int j = 0; j < 9; j++) row[j] = SP_A(i, j);
#pragma unroll
          for (int j = 0; j < 9; j++) SP_A(i, j) = row[j] - f * prow[j];
        }
      }
  }
  SP_STAMP(1);
  int freec[2] = {0, 0}, nf = 0;
#pragma unroll
  for (int j = 0; j < 9; j++)
    if (!((used >> j) & 1u) && nf < 2) {
      if (nf == 0) freec[0] = j;
      else freec[1] = j;
      nf++;
    }
  double f1[9], f2[9];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    double fs[9];
#pragma unroll
    for (int j = 0; j < 9; j++) fs[j] = (j == freec[k]) ? 1.0 : 0.0;
#pragma unroll
    for (int r = 0; r < 7; r++) {
      const double v = -SP_A(r, freec[k]);
#pragma unroll
      for (int j = 0; j < 9; j++) fs[j] = (pivcol[r] == j) ? v : fs[j];
    }
#pragma unroll
    for (int j = 0; j < 9; j++) {
      if (k == 0) f1[j] = fs[j];
      else f2[j] = fs[j];
    }
  }
#undef SP_A
  double Bm[9];
#pragma unroll
  for (int j = 0; j < 9; j++) Bm[j] = f1[j] - f2[j];
  const double *a0 = f2, *a1 = f2 + 3, *a2 = f2 + 6, *b0 = Bm, *b1 = Bm + 3, *b2 = Bm + 6;
  c[0] = det3(a0, a1, a2);
  c[1] = det3(b0, a1, a2) + det3(a0, b1, a2) + det3(a0, a1, b2);
  c[2] = det3(b0, b1, a2) + det3(b0, a1, b2) + det3(a0, b1, b2);
  c[3] = det3(b0, b1, b2);
#pragma unroll
  for (int j = 0; j < 9; j++) {
    mid.f2[j] = f2[j];
    mid.Bm[j] = Bm[j];
  }
  mid.s1 = s1, mid.s2 = s2;
  mid.c1[0] = c1[0], mid.c1[1] = c1[1], mid.c2[0] = c2[0], mid.c2[1] = c2[1];
  SP_STAMP(2);
  return true;
}
// second half: one fundamental matrix per real root of the cubic (de-normalised, unit Frobenius norm)
__device__ inline int seven_point_b(const SevenPointMid& mid, const double* roots, int nr, double (*F)[9]) {
  const double* f2 = mid.f2;
  const double* Bm = mid.Bm;
  const double s1 = mid.s1, s2 = mid.s2;
  const double c1[2] = {mid.c1[0], mid.c1[1]}, c2[2] = {mid.c2[0], mid.c2[1]};
  int nm = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (k >= nr) break;
    double Fh[9];
#pragma unroll
    for (int j = 0; j < 9; j++) Fh[j] = f2[j] + roots[k] * Bm[j];
    double T1[9] = {s1, 0, -s1 * c1[0], 0, s1, -s1 * c1[1], 0, 0, 1};
    double T2[9] = {s2, 0, -s2 * c2[0], 0, s2, -s2 * c2[1], 0, 0, 1};
    double tmp[9], Fo[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        double s = 0;
#pragma unroll
        for (int m = 0; m < 3; m++) s += Fh[3 * i + m] * T1[3 * m + j];
        tmp[3 * i + j] = s;
      }
    double nn = 0;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        double s = 0;
#pragma unroll
        for (int m = 0; m < 3; m++) s += T2[3 * m + i] * tmp[3 * m + j];
        Fo[3 * i + j] = s;
        nn += s * s;
      }
    if (!(nn > 0) || !isfinite(nn)) continue;
    double inv = 1.0 / sqrt(nn);
    // (static destination index: nm is 0, 1 or 2)
#pragma unroll
    for (int j = 0; j < 9; j++) {
      const double v = Fo[j] * inv;
      if (nm == 0) F[0][j] = v;
      else if (nm == 1) F[1][j] = v;
      else F[2][j] = v;
    }
    nm++;
  }
  return nm;
}

template <int WS>
__device__ inline int seven_point(const double (*x1)[2], const double (*x2)[2], double (*F)[9], double* wk) {
  SevenPointMid mid;
  double c[4];
  if (!seven_point_a<WS>(x1, x2, wk, mid, c)) return 0;
  double roots[4];
  const int nr = poly_real_roots(c, 3, roots);
  SP_STAMP(3);
  return seven_point_b(mid, roots, nr, F);
}

// OpenCV FMEstimatorCallback::computeError (max of the two squared point-line distances, float)
FD float f_error(const double* F, double x1, double y1, double x2, double y2) {
  double a = F[0] * x1 + F[1] * y1 + F[2], b = F[3] * x1 + F[4] * y1 + F[5], c = F[6] * x1 + F[7] * y1 + F[8];
  double s2 = 1. / (a * a + b * b);
  double dd2 = x2 * a + y2 * b + c;
  a = F[0] * x2 + F[3] * y2 + F[6];
  b = F[1] * x2 + F[4] * y2 + F[7];
  c = F[2] * x2 + F[5] * y2 + F[8];
  double s1 = 1. / (a * a + b * b);
  double dd1 = x1 * a + y1 * b + c;
  return (float)fmax(dd1 * dd1 * s1, dd2 * dd2 * s2);
}

// Grunert P3P: up to 4 (R, t) with X_cam = R P + t
// Calls fn(R, t) for every solution (X_cam = R P + t), in root order; returns their number.  Everything is statically
// indexed (no solution arrays): indexed local arrays would live in scratch memory.
// state of a P3P solve between the quartic's coefficients and its roots
struct P3PMid {
  double b2, ca, cb, cg, A;
};
// first half: the quartic q[0..4] in v = s3 / s1.  false: degenerate triangle
__device__ inline bool p3p_grunert_a(const V3* P, const V3* f, P3PMid& mid, double* q) {
  double a2 = dot(P[1] - P[2], P[1] - P[2]), b2 = dot(P[0] - P[2], P[0] - P[2]), c2 = dot(P[0] - P[1], P[0] - P[1]);
  if (b2 < 1e-20 || a2 < 1e-20 || c2 < 1e-20) return false;
  double ca = dot(f[1], f[2]), cb = dot(f[0], f[2]), cg = dot(f[0], f[1]);
  double A = (a2 - c2) / b2, C = c2 / b2;
  q[4] = A * A - 2 * A - 4 * C * ca * ca + 1;
  q[3] = -4 * A * A * cb + 4 * A * ca * cg + 4 * A * cb + 8 * C * ca * ca * cb + 8 * C * ca * cg - 4 * ca * cg;
  q[2] = 4 * A * A * cb * cb + 2 * A * A - 8 * A * ca * cb * cg - 4 * A * cg * cg - 4 * C * ca * ca - 16 * C * ca * cb * cg -
         4 * C * cg * cg + 4 * ca * ca + 4 * cg * cg - 2;
  q[1] = -4 * A * A * cb + 4 * A * ca * cg + 8 * A * cb * cg * cg - 4 * A * cb + 8 * C * ca * cg + 8 * C * cb * cg * cg -
         4 * ca * cg;
  q[0] = A * A - 4 * A * cg * cg + 2 * A - 4 * C * cg * cg + 1;
  mid.b2 = b2, mid.ca = ca, mid.cb = cb, mid.cg = cg, mid.A = A;
  return true;
}
// second half: a pose per admissible root, in root order
template <class Fn>
__device__ inline int p3p_grunert_b(const V3* P, const V3* f, const P3PMid& mid, const double* roots, int nr, Fn&& fn) {
  const double b2 = mid.b2, ca = mid.ca, cb = mid.cb, cg = mid.cg, A = mid.A;
  int ns = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (k >= nr) break;
    double v = roots[k];
    if (!(v > 0)) continue;
    double den = 2 * (cg - v * ca);
    if (fabs(den) < 1e-12) continue;
    double u = ((A - 1) * v * v - 2 * A * cb * v + 1 + A) / den;
    if (!(u > 0)) continue;
    double dd = 1 + v * v - 2 * v * cb;
    if (!(dd > 0)) continue;
    double s1 = sqrt(b2 / dd), s2 = u * s1, s3 = v * s1;
    V3 X0 = s1 * f[0], X1 = s2 * f[1], X2 = s3 * f[2];
    V3 e1w = P[1] - P[0], e1c = X1 - X0;
    double n1w = norm(e1w), n1c = norm(e1c);
    if (n1w < 1e-12 || n1c < 1e-12) continue;
    e1w = (1 / n1w) * e1w;
    e1c = (1 / n1c) * e1c;
    V3 e3w = cross(e1w, P[2] - P[0]), e3c = cross(e1c, X2 - X0);
    double n3w = norm(e3w), n3c = norm(e3c);
    if (n3w < 1e-12 || n3c < 1e-12) continue;
    e3w = (1 / n3w) * e3w;
    e3c = (1 / n3c) * e3c;
    V3 e2w = cross(e3w, e1w), e2c = cross(e3c, e1c);
    M3 R;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++)
        R.m[i][j] = vget(e1c, i) * vget(e1w, j) + vget(e2c, i) * vget(e2w, j) + vget(e3c, i) * vget(e3w, j);
    fn(R, X0 - R * P[0]);
    ns++;
  }
  return ns;
}

template <class Fn>
__device__ inline int p3p_grunert_each(const V3* P, const V3* f, Fn&& fn) {
  P3PMid mid;
  double q[5];
  if (!p3p_grunert_a(P, f, mid, q)) return 0;
  double roots[4];
  const int nr = poly_real_roots(q, 4, roots);
  return p3p_grunert_b(P, f, mid, roots, nr, fn);
}

// g2o EdgeSE3ProjectXYZ error + pose Jacobian (tangent = omega, upsilon)
FD void proj_edge(const SE3d& T, V3 pw, double zu, double zv, double fx, double fy, double cx, double cy, double* e,
                  double (*J)[6]) {
  V3 X = g2o_map(T, pw);
  double x = X.x, y = X.y, zz = X.z, z2 = zz * zz;
  e[0] = zu - (x / zz * fx + cx);
  e[1] = zv - (y / zz * fy + cy);
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

// 6x6 SPD solve by Cholesky; false if not positive definite
__device__ inline bool solve_spd6(const double* H, const double* b, double* x) {
  double L[36];
  for (int i = 0; i < 36; i++) L[i] = 0;
  for (int j = 0; j < 6; j++) {
    double s = H[6 * j + j];
    for (int k = 0; k < j; k++) s -= L[6 * j + k] * L[6 * j + k];
    if (!(s > 0)) return false;
    L[6 * j + j] = sqrt(s);
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

FD double huber_rho(double e) { return e <= 1.0 ? e : 2 * sqrt(e) - 1.0; }
FD double huber_w(double e) { return e <= 1.0 ? 1.0 : 1.0 / sqrt(e); }

}  // namespace flvis
