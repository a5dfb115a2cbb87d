// flvis_amd: device-side small-matrix / Lie-group / polynomial kit (fp64, gfx950).
// Semantics follow the reference's Sophus / Eigen / g2o usage (3rdPartLib/Sophus/sophus/{se3,so3}.cpp,
// 3rdPartLib/g2o/g2o/types/slam3d/se3quat.h, src/utils/include/kinetic_math.h).  Everything is plain IEEE + - * / sqrt in a
// fixed order (the library is built with -ffp-contract=off), so scalar paths reproduce bit-for-bit across runs.
#pragma once
#include "det_math.hpp"  // sin / cos / atan / atan2 / log / small integer powers: one definition for the kernels and the oracle
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace flvis {

#define FD __device__ __forceinline__

struct V3 {
  double x, y, z;
};
struct Q4 {
  double w, x, y, z;
};
struct M3 {
  double m[3][3];
};
struct SE3d {
  Q4 q;
  V3 t;
};

FD V3 v3(double x, double y, double z) { return V3{x, y, z}; }
FD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
FD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
FD V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
FD V3 operator*(V3 a, double s) { return V3{s * a.x, s * a.y, s * a.z}; }
FD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
FD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
FD double norm(V3 a) { return sqrt(dot(a, a)); }
FD double vget(const V3& a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

FD M3 m3_identity() {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r.m[i][j] = (i == j) ? 1.0 : 0.0;
  return r;
}
FD M3 operator*(const M3& a, const M3& b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
FD V3 operator*(const M3& a, V3 v) {
  return V3{a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
            a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
FD M3 transpose(const M3& a) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
  return r;
}
FD M3 skew(V3 v) {
  M3 r;
  r.m[0][0] = 0;
  r.m[0][1] = -v.z;
  r.m[0][2] = v.y;
  r.m[1][0] = v.z;
  r.m[1][1] = 0;
  r.m[1][2] = -v.x;
  r.m[2][0] = -v.y;
  r.m[2][1] = v.x;
  r.m[2][2] = 0;
  return r;
}
FD M3 m3_add(const M3& a, const M3& b, double sb) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] + sb * b.m[i][j];
  return r;
}
FD void m3_inverse(const M3& a, M3& inv) {  // cofactors / determinant (Eigen fixed 3x3 inverse)
  double c00 = a.m[1][1] * a.m[2][2] - a.m[1][2] * a.m[2][1];
  double c01 = a.m[1][2] * a.m[2][0] - a.m[1][0] * a.m[2][2];
  double c02 = a.m[1][0] * a.m[2][1] - a.m[1][1] * a.m[2][0];
  double det = a.m[0][0] * c00 + a.m[0][1] * c01 + a.m[0][2] * c02;
  double id = 1.0 / det;
  inv.m[0][0] = c00 * id;
  inv.m[1][0] = c01 * id;
  inv.m[2][0] = c02 * id;
  inv.m[0][1] = (a.m[0][2] * a.m[2][1] - a.m[0][1] * a.m[2][2]) * id;
  inv.m[1][1] = (a.m[0][0] * a.m[2][2] - a.m[0][2] * a.m[2][0]) * id;
  inv.m[2][1] = (a.m[0][1] * a.m[2][0] - a.m[0][0] * a.m[2][1]) * id;
  inv.m[0][2] = (a.m[0][1] * a.m[1][2] - a.m[0][2] * a.m[1][1]) * id;
  inv.m[1][2] = (a.m[0][2] * a.m[1][0] - a.m[0][0] * a.m[1][2]) * id;
  inv.m[2][2] = (a.m[0][0] * a.m[1][1] - a.m[0][1] * a.m[1][0]) * id;
}

FD Q4 q_identity() { return Q4{1, 0, 0, 0}; }
FD Q4 q_mul(Q4 a, Q4 b) {
  return Q4{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
FD Q4 q_conj(Q4 q) { return Q4{q.w, -q.x, -q.y, -q.z}; }
FD double q_sqnorm(Q4 q) { return q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z; }
FD Q4 q_normalized(Q4 q) {
  double n = sqrt(q_sqnorm(q));
  return Q4{q.w / n, q.x / n, q.y / n, q.z / n};
}
FD M3 q_to_mat(Q4 q) {
  double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  M3 r;
  r.m[0][0] = 1 - (tyy + tzz);
  r.m[0][1] = txy - twz;
  r.m[0][2] = txz + twy;
  r.m[1][0] = txy + twz;
  r.m[1][1] = 1 - (txx + tzz);
  r.m[1][2] = tyz - twx;
  r.m[2][0] = txz - twy;
  r.m[2][1] = tyz + twx;
  r.m[2][2] = 1 - (txx + tyy);
  return r;
}
FD Q4 mat_to_q(const M3& m) {
  Q4 q;
  double t = m.m[0][0] + m.m[1][1] + m.m[2][2];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (m.m[2][1] - m.m[1][2]) * t;
    q.y = (m.m[0][2] - m.m[2][0]) * t;
    q.z = (m.m[1][0] - m.m[0][1]) * t;
  } else {
    int i = 0;
    if (m.m[1][1] > m.m[0][0]) i = 1;
    if (m.m[2][2] > m.m[i][i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(m.m[i][i] - m.m[j][j] - m.m[k][k] + 1.0);
    double v[3];
    v[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (m.m[k][j] - m.m[j][k]) * t;
    v[j] = (m.m[j][i] + m.m[i][j]) * t;
    v[k] = (m.m[k][i] + m.m[i][k]) * t;
    q.x = v[0];
    q.y = v[1];
    q.z = v[2];
  }
  return q;
}
FD V3 q_rotate(Q4 q, V3 v) {
  V3 qv{q.x, q.y, q.z};
  V3 uv = cross(qv, v);
  uv = uv + uv;
  return v + q.w * uv + cross(qv, uv);
}

// Sophus SE3 semantics
FD SE3d se3_identity() { return SE3d{q_identity(), V3{0, 0, 0}}; }
FD SE3d se3_from_quat(Q4 q, V3 t) { return SE3d{q_normalized(q), t}; }
FD SE3d se3_from_mat(const M3& R, V3 t) { return SE3d{mat_to_q(R), t}; }
FD SE3d se3_mul(const SE3d& a, const SE3d& b) {
  SE3d r;
  r.t = a.t + q_rotate(a.q, b.t);
  r.q = q_normalized(q_mul(a.q, b.q));
  return r;
}
FD SE3d se3_inverse(const SE3d& a) {
  SE3d r;
  r.q = q_normalized(q_conj(a.q));
  r.t = q_rotate(r.q, -1.0 * a.t);
  return r;
}
FD V3 se3_act(const SE3d& a, V3 p) { return q_rotate(a.q, p) + a.t; }
FD V3 so3_log(Q4 q) {
  const double SMALL_EPS = 1e-10;
  double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  double w = q.w;
  double squared_w = w * w;
  double f;
  if (n < SMALL_EPS) {
    f = 2. / w - 2. * (n * n) / (w * squared_w);
  } else {
    f = 2 * detm::det_atan(n / w) / n;
  }
  return V3{f * q.x, f * q.y, f * q.z};
}

// rotation vector -> unit quaternion and the inverse right Jacobian of SO(3) (Forster et al., "On-Manifold Preintegration",
// eq. 8): the gyro preintegration between keyframes and the IMU rotation factor of the window BA
FD Q4 q_exp(V3 phi) {
  const double th = norm(phi);
  if (th < 1e-8) return q_normalized(Q4{1.0, 0.5 * phi.x, 0.5 * phi.y, 0.5 * phi.z});
  const double s = detm::det_sin(0.5 * th) / th;
  return Q4{detm::det_cos(0.5 * th), s * phi.x, s * phi.y, s * phi.z};
}
FD M3 so3_jr_inv(V3 phi) {
  const double th = norm(phi);
  const M3 S = skew(phi);
  double c = 1.0 / 12.0;
  if (th > 1e-5) c = 1.0 / (th * th) - (1.0 + detm::det_cos(th)) / (2.0 * th * detm::det_sin(th));
  return m3_add(m3_add(m3_identity(), S, 0.5), S * S, c);
}

// g2o SE3Quat semantics
FD void g2o_normalize_rotation(Q4& q) {
  if (q.w < 0) {
    q.w = -q.w;
    q.x = -q.x;
    q.y = -q.y;
    q.z = -q.z;
  }
  q = q_normalized(q);
}
FD SE3d g2o_from_mat(const M3& R, V3 t) {
  SE3d r{mat_to_q(R), t};
  g2o_normalize_rotation(r.q);
  return r;
}
FD SE3d g2o_mul(const SE3d& a, const SE3d& b) {
  SE3d r;
  r.t = a.t + q_rotate(a.q, b.t);
  r.q = q_mul(a.q, b.q);
  g2o_normalize_rotation(r.q);
  return r;
}
FD V3 g2o_map(const SE3d& T, V3 p) { return q_rotate(T.q, p) + T.t; }
FD SE3d g2o_exp(const double* upd) {
  V3 omega{upd[0], upd[1], upd[2]}, upsilon{upd[3], upd[4], upd[5]};
  double theta = norm(omega);
  M3 Omega = skew(omega);
  M3 Omega2 = Omega * Omega;
  M3 R, V;
  M3 I = m3_identity();
  if (theta < 0.00001) {
    R = m3_add(m3_add(I, Omega, 1.0), Omega2, 0.5);
    V = m3_add(m3_add(I, Omega, 0.5), Omega2, 1.0 / 6.0);
  } else {
    double st = detm::det_sin(theta), ct = detm::det_cos(theta);
    R = m3_add(m3_add(I, Omega, st / theta), Omega2, (1 - ct) / (theta * theta));
    V = m3_add(m3_add(I, Omega, (1 - ct) / (theta * theta)), Omega2, (theta - st) / (theta * theta * theta));
  }
  SE3d r{mat_to_q(R), V * upsilon};
  g2o_normalize_rotation(r.q);
  return r;
}

// kinetic_math.h rpy helpers
FD M3 rpy2R(V3 rpy) {
  double r = rpy.x, p = rpy.y, y = rpy.z;
  double cy = detm::det_cos(y), sy = detm::det_sin(y), cp = detm::det_cos(p), sp = detm::det_sin(p), cr = detm::det_cos(r), sr = detm::det_sin(r);
  M3 R;
  R.m[0][0] = cy * cp;
  R.m[0][1] = cy * sp * sr - sy * cr;
  R.m[0][2] = cy * sp * cr + sy * sr;
  R.m[1][0] = sy * cp;
  R.m[1][1] = sy * sp * sr + cy * cr;
  R.m[1][2] = sy * sp * cr - cy * sr;
  R.m[2][0] = -sp;
  R.m[2][1] = cp * sr;
  R.m[2][2] = cp * cr;
  return R;
}
FD V3 R2rpy(const M3& R) {
  return V3{detm::det_atan2(R.m[2][1], R.m[2][2]), detm::det_atan2(-R.m[2][0], sqrt(R.m[2][1] * R.m[2][1] + R.m[2][2] * R.m[2][2])),
            detm::det_atan2(R.m[1][0], R.m[0][0])};
}
FD Q4 rpy2Q(V3 rpy) { return mat_to_q(rpy2R(rpy)); }
FD V3 Q2rpy(Q4 q) { return R2rpy(q_to_mat(q)); }

// pose7 storage: tx ty tz qx qy qz qw
FD SE3d load_pose7(const double* p) { return SE3d{Q4{p[6], p[3], p[4], p[5]}, V3{p[0], p[1], p[2]}}; }
FD void store_pose7(double* p, const SE3d& T) {
  p[0] = T.t.x;
  p[1] = T.t.y;
  p[2] = T.t.z;
  p[3] = T.q.x;
  p[4] = T.q.y;
  p[5] = T.q.z;
  p[6] = T.q.w;
}

// cv::RNG (core/operations.hpp): multiply-with-carry, output = low word of the new state.  Every RANSAC / LMedS run of OpenCV draws
// from its own `RNG rng((uint64)-1)` (calib3d/src/ptsetreg.cpp): the samples of a call depend on the point count alone.
struct CvRng {
  uint64_t state;
};
FD CvRng cv_rng_init() { return CvRng{0xffffffffffffffffull}; }
FD uint32_t cv_rng_next(CvRng& r) {
  r.state = (uint64_t)(uint32_t)r.state * 4164903690u + (uint32_t)(r.state >> 32);
  return (uint32_t)r.state;
}
// x % c for a divisor fixed per call: the quotient estimate mulhi(x, floor((2^32 - 1) / c)) is at most 2 too small
struct ModC {
  uint32_t c, inv;
};
FD ModC mod_c_make(uint32_t c) { return ModC{c, 0xffffffffu / c}; }
FD uint32_t mod_c(uint32_t x, ModC m) {
  const uint32_t q = (uint32_t)(((uint64_t)x * m.inv) >> 32);
  uint32_t r = x - q * m.c;
  if (r >= m.c) r -= m.c;
  if (r >= m.c) r -= m.c;
  return r;
}
// RANSACPointSetRegistrator::getSubset (ptsetreg.cpp, checkPartialSubsets false): per slot rng.uniform(0, count), redrawn while it
// repeats an earlier slot; the complete subset is redrawn (one more attempt) while `check` refuses it.  Control flow and draws are
// wave-uniform: a whole wave may run it redundantly and let `check` spread its work over the lanes.
template <int MAXM, class Check>
FD bool cv_get_subset(CvRng& rng, ModC mc, int m, int maxAttempts, int* idx, Check check, int iters0 = 0) {
#pragma unroll
  for (int j = 0; j < MAXM; j++) idx[j] = -1;
  int iters = iters0, i = 0;  // (iters0: attempts of this slot already made by the caller -- a tabulated candidate that `check` refused)
  for (; iters < maxAttempts; iters++) {
    for (i = 0; i < m;) {
      int idx_i;
      for (;;) {
        idx_i = (int)mod_c(cv_rng_next(rng), mc);
        bool dup = false;
#pragma unroll
        for (int j = 0; j < MAXM; j++) dup = dup || (j < i && idx[j] == idx_i);
        if (!dup) break;
      }
#pragma unroll
      for (int j = 0; j < MAXM; j++)
        if (j == i) idx[j] = idx_i;  // (static indices: the subset stays in registers)
      i++;
    }
    if (!check(idx)) continue;
    break;
  }
  return iters < maxAttempts;
}
// cv::RANSACUpdateNumIters
FD int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
  p = fmax(p, 0.);
  p = fmin(p, 1.);
  ep = fmax(ep, 0.);
  ep = fmin(ep, 1.);
  double num = fmax(1. - p, 2.2250738585072014e-308);
  double denom = 1. - detm::det_powi(1. - ep, modelPoints);
  if (denom < 2.2250738585072014e-308) return 0;
  num = detm::det_log(num);
  denom = detm::det_log(denom);
  return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : (int)rint(num / denom);
}

// ---- polynomial real roots (deg <= 4), ascending; bracketing between derivative roots + bisection
FD double poly_eval(const double* a, int deg, double x) {
  double r = a[deg];
  for (int i = deg - 1; i >= 0; i--) r = r * x + a[i];
  return r;
}
__device__ inline int poly_roots_quadratic(const double* a, double* roots) {
  double disc = a[1] * a[1] - 4 * a[2] * a[0];
  if (disc < 0) return 0;
  double sq = sqrt(disc);
  double q = -0.5 * (a[1] + (a[1] >= 0 ? sq : -sq));
  double r0 = q / a[2];
  double r1 = (q != 0) ? a[0] / q : r0;
  if (disc == 0) {
    roots[0] = r0;
    return 1;
  }
  roots[0] = fmin(r0, r1);
  roots[1] = fmax(r0, r1);
  return 2;
}
// Horner on five register-resident coefficients; coefficients above the degree are 0, which leaves every intermediate
// bit-identical to the degree-limited Horner (0 * x + a = a exactly for finite x)
FD double poly_eval5(double a0, double a1, double a2, double a3, double a4, double x) {
  double r = a4;
  r = r * x + a3;
  r = r * x + a2;
  r = r * x + a1;
  r = r * x + a0;
  return r;
}
// one bracketing level: roots of `a` (degree deg >= 3) given the real roots `crit` of its derivative.
// The coefficients and the knots live in registers (selects instead of indexed local arrays: indexed arrays end up in
// scratch memory, and the ~60-step bisections of every root then run at memory latency).
// The level comes in three pieces -- knots, the bisection of ONE interval, the emission of the roots -- so that the (up to four)
// bisections of a polynomial can also run on different waves (the RANSAC kernels: one hypothesis per lane, one interval per wave);
// every interval performs exactly the sequential arithmetic of the CPU restatement wherever it runs.
struct PolyBracket {
  double a0, a1, a2, a3, a4;  // coefficients (a4 = 0 for a cubic)
  double k0, k1, k2, k3, k4;  // knots: -B, the critical points inside (-B, B) in order, +B
  int nk;                     // number of knots (nk - 1 intervals)
};
FD double poly_bracket_lo(const PolyBracket& t, int i) { return i == 0 ? t.k0 : (i == 1 ? t.k1 : (i == 2 ? t.k2 : t.k3)); }
FD double poly_bracket_hi(const PolyBracket& t, int i) { return i == 0 ? t.k1 : (i == 1 ? t.k2 : (i == 2 ? t.k3 : t.k4)); }
__device__ inline PolyBracket poly_bracket_knots(const double* a, int deg, const double* crit, int nc) {
  PolyBracket t;
  t.a0 = a[0], t.a1 = a[1], t.a2 = a[2], t.a3 = a[3], t.a4 = deg >= 4 ? a[4] : 0.0;
  const double alead = deg >= 4 ? t.a4 : t.a3;
  double B = 0;
  B = fmax(B, fabs(t.a0 / alead));
  B = fmax(B, fabs(t.a1 / alead));
  B = fmax(B, fabs(t.a2 / alead));
  if (deg >= 4) B = fmax(B, fabs(t.a3 / alead));
  B += 1.0;
  t.k0 = -B, t.k1 = B, t.k2 = B, t.k3 = B, t.k4 = B;
  int nk = 1;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    if (i < nc) {
      const double c = crit[i];
      if (c > -B && c < B) {
        if (nk == 1) t.k1 = c;
        else if (nk == 2) t.k2 = c;
        else t.k3 = c;
        nk++;
      }
    }
  }
  if (nk == 1) t.k1 = B;
  else if (nk == 2) t.k2 = B;
  else if (nk == 3) t.k3 = B;
  else t.k4 = B;
  nk++;
  t.nk = nk;
  return t;
}
// interval i of the level: is there a sign change, and if so the bisected root 0.5 (lo + hi)
__device__ inline void poly_bracket_bisect(const PolyBracket& t, int i, bool& bis, double& root) {
  double lo = poly_bracket_lo(t, i), hi = poly_bracket_hi(t, i);
  const bool on = i + 1 < t.nk;
  double flo = on ? poly_eval5(t.a0, t.a1, t.a2, t.a3, t.a4, lo) : 1.0;
  const double fhi = on ? poly_eval5(t.a0, t.a1, t.a2, t.a3, t.a4, hi) : 1.0;
  bis = on && flo != 0 && fhi != 0 && ((flo < 0) != (fhi < 0));
  if (bis) {
    for (int it = 0; it < 200; it++) {
      const double mid = 0.5 * (lo + hi);
      if (mid == lo || mid == hi) break;
      const double fm = poly_eval5(t.a0, t.a1, t.a2, t.a3, t.a4, mid);
      if (fm == 0) {
        lo = hi = mid;
        break;
      } else if ((fm < 0) == (flo < 0)) {
        lo = mid;
        flo = fm;
      } else {
        hi = mid;
      }
    }
  }
  root = 0.5 * (lo + hi);
}
// the roots of the level in ascending order from the intervals' results (a knot that is an exact root is reported once)
__device__ inline int poly_bracket_emit(const PolyBracket& t, const bool* bis, const double* mid, double* roots) {
  const double a0 = t.a0, a1 = t.a1, a2 = t.a2, a3 = t.a3, a4 = t.a4;
  const int nk = t.nk;
  int nr = 0;
  double rprev = 0;  // roots[nr - 1]
#pragma unroll
  for (int i = 0; i < 4; i++) {
    if (i + 1 < nk) {
      const double klo = poly_bracket_lo(t, i), khi = poly_bracket_hi(t, i);
      bool emit = false;
      double rv = 0;
      if (bis[i]) {
        emit = true;
        rv = mid[i];
      } else {
        const double f0 = poly_eval5(a0, a1, a2, a3, a4, klo), f1 = poly_eval5(a0, a1, a2, a3, a4, khi);
        if (f0 == 0) {
          if (nr == 0 || rprev != klo) {
            emit = true;
            rv = klo;
          }
        } else if (f1 == 0) {
          if (i + 2 == nk) {
            emit = true;
            rv = khi;
          }
        }
      }
      if (emit) {
        if (nr == 0) roots[0] = rv;
        else if (nr == 1) roots[1] = rv;
        else if (nr == 2) roots[2] = rv;
        else roots[3] = rv;
        rprev = rv;
        nr++;
      }
    }
  }
  return nr;
}
// the whole level inside one lane: the (up to 4) sign-change intervals are bisected TOGETHER -- the chains are independent, so
// interleaving them hides the fp64 dependency latency of one chain behind the others
__device__ inline int poly_roots_bracket(const double* a, int deg, const double* crit, int nc, double* roots) {
  const PolyBracket t = poly_bracket_knots(a, deg, crit, nc);
  const double a0 = t.a0, a1 = t.a1, a2 = t.a2, a3 = t.a3, a4 = t.a4;
  const int nk = t.nk;
  double lo[4], hi[4], flo[4], fhi[4];
  bool bis[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    lo[i] = poly_bracket_lo(t, i);
    hi[i] = poly_bracket_hi(t, i);
    const bool on = i + 1 < nk;
    flo[i] = on ? poly_eval5(a0, a1, a2, a3, a4, lo[i]) : 1.0;
    fhi[i] = on ? poly_eval5(a0, a1, a2, a3, a4, hi[i]) : 1.0;
    bis[i] = on && flo[i] != 0 && fhi[i] != 0 && ((flo[i] < 0) != (fhi[i] < 0));
  }
  {
    bool run[4] = {bis[0], bis[1], bis[2], bis[3]};
    for (int it = 0; it < 200 && (run[0] || run[1] || run[2] || run[3]); it++) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (run[i]) {
          const double mid = 0.5 * (lo[i] + hi[i]);
          if (mid == lo[i] || mid == hi[i]) {
            run[i] = false;
          } else {
            const double fm = poly_eval5(a0, a1, a2, a3, a4, mid);
            if (fm == 0) {
              lo[i] = hi[i] = mid;
              run[i] = false;
            } else if ((fm < 0) == (flo[i] < 0)) {
              lo[i] = mid;
              flo[i] = fm;
            } else {
              hi[i] = mid;
            }
          }
        }
      }
    }
  }
  double mid[4];
#pragma unroll
  for (int i = 0; i < 4; i++) mid[i] = 0.5 * (lo[i] + hi[i]);
  return poly_bracket_emit(t, bis, mid, roots);
}
// generic entry (no recursion on device: explicit cascade 4 -> 3 -> 2)
__device__ inline int poly_real_roots(const double* a_in, int deg, double* roots) {
  double a[5];
  double amax = 0;
  for (int i = 0; i <= deg; i++) {
    a[i] = a_in[i];
    amax = fmax(amax, fabs(a[i]));
  }
  if (amax == 0) return 0;
  while (deg > 0 && fabs(a[deg]) <= 1e-14 * amax) deg--;
  if (deg == 0) return 0;
  if (deg == 1) {
    roots[0] = -a[0] / a[1];
    return 1;
  }
  if (deg == 2) return poly_roots_quadratic(a, roots);
  // derivative chain with the same leading-coefficient trimming rule at each level
  double d1[4], d2[3];
  int n1 = deg - 1;
  for (int i = 1; i <= deg; i++) d1[i - 1] = a[i] * i;
  double c1[4];
  int nc1;
  {
    double m1 = 0;
    for (int i = 0; i <= n1; i++) m1 = fmax(m1, fabs(d1[i]));
    int e1 = n1;
    while (e1 > 0 && fabs(d1[e1]) <= 1e-14 * m1) e1--;
    if (m1 == 0 || e1 == 0) {
      nc1 = 0;
    } else if (e1 == 1) {
      c1[0] = -d1[0] / d1[1];
      nc1 = 1;
    } else if (e1 == 2) {
      nc1 = poly_roots_quadratic(d1, c1);
    } else {  // e1 == 3 (deg == 4): need the roots of d1' first
      for (int i = 1; i <= 3; i++) d2[i - 1] = d1[i] * i;
      double c2[2];
      int nc2;
      double m2 = fmax(fmax(fabs(d2[0]), fabs(d2[1])), fabs(d2[2]));
      int e2 = 2;
      while (e2 > 0 && fabs(d2[e2]) <= 1e-14 * m2) e2--;
      if (m2 == 0 || e2 == 0)
        nc2 = 0;
      else if (e2 == 1) {
        c2[0] = -d2[0] / d2[1];
        nc2 = 1;
      } else
        nc2 = poly_roots_quadratic(d2, c2);
      nc1 = poly_roots_bracket(d1, 3, c2, nc2, c1);
    }
  }
  return poly_roots_bracket(a, deg, c1, nc1, roots);
}

// poly_real_roots(a_in, 3, roots) up to its bracketing level, for callers that bisect the level's intervals elsewhere (another
// wave): returns 1 when the level `t` has to be bisected and emitted, 0 when roots / nr are final (degenerate leading coefficients)
__device__ inline int poly_cubic_prepare(const double* a_in, PolyBracket& t, double* roots, int& nr) {
  double a[5];
  double amax = 0;
  int deg = 3;
  for (int i = 0; i <= deg; i++) {
    a[i] = a_in[i];
    amax = fmax(amax, fabs(a[i]));
  }
  a[4] = 0;
  nr = 0;
  if (amax == 0) return 0;
  while (deg > 0 && fabs(a[deg]) <= 1e-14 * amax) deg--;
  if (deg == 0) return 0;
  if (deg == 1) {
    roots[0] = -a[0] / a[1];
    nr = 1;
    return 0;
  }
  if (deg == 2) {
    nr = poly_roots_quadratic(a, roots);
    return 0;
  }
  double d1[4];
  for (int i = 1; i <= 3; i++) d1[i - 1] = a[i] * i;
  double c1[4];
  int nc1;
  double m1 = 0;
  for (int i = 0; i <= 2; i++) m1 = fmax(m1, fabs(d1[i]));
  int e1 = 2;
  while (e1 > 0 && fabs(d1[e1]) <= 1e-14 * m1) e1--;
  if (m1 == 0 || e1 == 0) {
    nc1 = 0;
  } else if (e1 == 1) {
    c1[0] = -d1[0] / d1[1];
    nc1 = 1;
  } else {
    nc1 = poly_roots_quadratic(d1, c1);
  }
  t = poly_bracket_knots(a, 3, c1, nc1);
  return 1;
}

// poly_real_roots(a_in, 4, roots) in stages, for callers that bisect the bracketing levels' intervals elsewhere.
//   stage 1: returns 0: roots / nr are final (degenerate leading coefficients);
//            returns 1: the derivative is a true cubic -- bisect and emit its level t1 (<= 3 intervals), then call stage 2 with its roots;
//            returns 2: the derivative's roots c1 / nc1 came in closed form -- call stage 2 with them.
//   stage 2: the quartic's own level (<= 4 intervals) from the derivative's roots.
// `a` receives the trimmed coefficients (needed again by stage 2).
__device__ inline int poly_quartic_stage1(const double* a_in, double* a, PolyBracket& t1, double* c1, int& nc1, double* roots, int& nr) {
  double amax = 0;
  int deg = 4;
  for (int i = 0; i <= deg; i++) {
    a[i] = a_in[i];
    amax = fmax(amax, fabs(a[i]));
  }
  nr = 0;
  nc1 = 0;
  if (amax == 0) return 0;
  while (deg > 0 && fabs(a[deg]) <= 1e-14 * amax) deg--;
  if (deg < 4) {  // (not a quartic after trimming: the generic solver, inside this lane)
    nr = poly_real_roots(a_in, 4, roots);
    return 0;
  }
  double d1[4], d2[3];
  for (int i = 1; i <= 4; i++) d1[i - 1] = a[i] * i;
  double m1 = 0;
  for (int i = 0; i <= 3; i++) m1 = fmax(m1, fabs(d1[i]));
  int e1 = 3;
  while (e1 > 0 && fabs(d1[e1]) <= 1e-14 * m1) e1--;
  if (m1 == 0 || e1 == 0) {
    nc1 = 0;
    return 2;
  }
  if (e1 == 1) {
    c1[0] = -d1[0] / d1[1];
    nc1 = 1;
    return 2;
  }
  if (e1 == 2) {
    nc1 = poly_roots_quadratic(d1, c1);
    return 2;
  }
  for (int i = 1; i <= 3; i++) d2[i - 1] = d1[i] * i;
  double c2[2];
  int nc2;
  const double m2 = fmax(fmax(fabs(d2[0]), fabs(d2[1])), fabs(d2[2]));
  int e2 = 2;
  while (e2 > 0 && fabs(d2[e2]) <= 1e-14 * m2) e2--;
  if (m2 == 0 || e2 == 0)
    nc2 = 0;
  else if (e2 == 1) {
    c2[0] = -d2[0] / d2[1];
    nc2 = 1;
  } else
    nc2 = poly_roots_quadratic(d2, c2);
  t1 = poly_bracket_knots(d1, 3, c2, nc2);
  return 1;
}
__device__ inline PolyBracket poly_quartic_stage2(const double* a, const double* c1, int nc1) { return poly_bracket_knots(a, 4, c1, nc1); }

FD double det3(const double* r0, const double* r1, const double* r2) {
  return r0[0] * (r1[1] * r2[2] - r1[2] * r2[1]) - r0[1] * (r1[0] * r2[2] - r1[2] * r2[0]) +
         r0[2] * (r1[0] * r2[1] - r1[1] * r2[0]);
}

}  // namespace flvis
