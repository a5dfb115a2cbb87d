// ORACLE (test infrastructure, NOT product code).
// CPU restatement of the small-matrix / Lie-group arithmetic the FLVIS hot path relies on.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Follows (semantics, not text):
//   Sophus (old, non-templated)  3rdPartLib/Sophus/sophus/so3.cpp:64-90,118-165  se3.cpp:59-95
//   Eigen quaternion <-> matrix conversions (external dep, published algorithm)
//   g2o SE3Quat                   3rdPartLib/g2o/g2o/types/slam3d/se3quat.h:99-123,173-285
//   kinetic_math.h                src/utils/include/kinetic_math.h:17-141
// parity unpinned: the reference ships no golden vectors for any of this.
#pragma once
// sin / cos / atan / atan2 / log / small integer powers are the SHARED deterministic definitions of the HIP kernels
// (flvis_amd/csrc/det_math.hpp, fdlibm algorithms, < 1 ulp from libm): one arithmetic on both sides makes the closed-loop
// front-end comparable bit for bit.  Where the reference calls std::pow(x, 3) / pow(1 - ep, n) this oracle multiplies.
#include "../flvis_amd/csrc/det_math.hpp"
#include <cmath>
#include <cstdint>
#include <cstring>

namespace ref {

struct Vec2 { double x, y; };
struct Vec3 {
  double x, y, z;
  double& operator[](int i) { return (&x)[i]; }
  double operator[](int i) const { return (&x)[i]; }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(double s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline Vec3 operator*(Vec3 a, double s) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double norm(Vec3 a) { return std::sqrt(dot(a, a)); }

struct Mat3 {
  double m[3][3];
  double& operator()(int r, int c) { return m[r][c]; }
  double operator()(int r, int c) const { return m[r][c]; }
};
inline Mat3 mat3_identity() { return {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}}; }
inline Mat3 operator*(const Mat3& a, const Mat3& b) {
  Mat3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
  return r;
}
inline Vec3 operator*(const Mat3& a, Vec3 v) {
  return {a.m[0][0] * v.x + a.m[0][1] * v.y + a.m[0][2] * v.z, a.m[1][0] * v.x + a.m[1][1] * v.y + a.m[1][2] * v.z,
          a.m[2][0] * v.x + a.m[2][1] * v.y + a.m[2][2] * v.z};
}
inline Mat3 transpose(const Mat3& a) {
  Mat3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[j][i];
  return r;
}
inline Mat3 skew(Vec3 v) { return {{{0, -v.z, v.y}, {v.z, 0, -v.x}, {-v.y, v.x, 0}}}; }
inline Mat3 mat3_add(const Mat3& a, const Mat3& b, double sb = 1.0) {
  Mat3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = a.m[i][j] + sb * b.m[i][j];
  return r;
}
// 3x3 inverse by cofactors / determinant (what Eigen's fixed-size inverse() does for 3x3).
inline bool mat3_inverse(const Mat3& a, Mat3& inv) {
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
  return det != 0.0 && std::isfinite(id);
}

// Quaternion, Eigen storage semantics (w,x,y,z accessors).
struct Quat {
  double w, x, y, z;
};
inline Quat quat_identity() { return {1, 0, 0, 0}; }
inline Quat quat_mul(Quat a, Quat b) {  // Hamilton product a*b (Eigen operator*)
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
inline Quat quat_conj(Quat q) { return {q.w, -q.x, -q.y, -q.z}; }
inline double quat_sqnorm(Quat q) { return q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z; }
inline Quat quat_normalized(Quat q) {
  double n = std::sqrt(quat_sqnorm(q));
  return {q.w / n, q.x / n, q.y / n, q.z / n};
}
// Eigen QuaternionBase::toRotationMatrix
inline Mat3 quat_to_mat(Quat q) {
  double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  Mat3 r;
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
// Eigen quaternion-from-matrix (Shoemake branch selection).
inline Quat mat_to_quat(const Mat3& m) {
  Quat q;
  double t = m.m[0][0] + m.m[1][1] + m.m[2][2];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
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
    t = std::sqrt(m.m[i][i] - m.m[j][j] - m.m[k][k] + 1.0);
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
// Eigen QuaternionBase::_transformVector
inline Vec3 quat_rotate(Quat q, Vec3 v) {
  Vec3 qv{q.x, q.y, q.z};
  Vec3 uv = cross(qv, v);
  uv = uv + uv;
  return v + q.w * uv + cross(qv, uv);
}

// SE3 with Sophus (old) semantics: unit quaternion + translation; products renormalise.
struct SE3 {
  Quat q;
  Vec3 t;
};
inline SE3 se3_identity() { return {quat_identity(), {0, 0, 0}}; }
inline SE3 se3_from_quat(Quat q, Vec3 t) { return {quat_normalized(q), t}; }       // SO3(Quaterniond) normalises
inline SE3 se3_from_mat(const Mat3& R, Vec3 t) { return {mat_to_quat(R), t}; }     // SO3(Matrix3d) does not
inline SE3 se3_mul(const SE3& a, const SE3& b) {                                   // se3.cpp:59-66
  SE3 r;
  r.t = a.t + quat_rotate(a.q, b.t);
  r.q = quat_normalized(quat_mul(a.q, b.q));
  return r;
}
inline SE3 se3_inverse(const SE3& a) {  // se3.cpp:76-83 ; SO3::inverse -> SO3(conjugate) normalises
  SE3 r;
  r.q = quat_normalized(quat_conj(a.q));
  r.t = quat_rotate(r.q, -1.0 * a.t);
  return r;
}
inline Vec3 se3_act(const SE3& a, Vec3 p) { return quat_rotate(a.q, p) + a.t; }
// SO3::logAndTheta so3.cpp:118-165
inline Vec3 so3_log(Quat q) {
  const double SMALL_EPS = 1e-10;
  double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  double w = q.w;
  double squared_w = w * w;
  double two_atan_nbyw_by_n;
  if (n < SMALL_EPS) {
    two_atan_nbyw_by_n = 2. / w - 2. * (n * n) / (w * squared_w);
  } else {
    if (std::fabs(w) < SMALL_EPS) {
      if (w > 0)
        two_atan_nbyw_by_n = M_PI / n;
      else
        two_atan_nbyw_by_n = -M_PI / n;
    }
    two_atan_nbyw_by_n = 2 * detm::det_atan(n / w) / n;
  }
  return {two_atan_nbyw_by_n * q.x, two_atan_nbyw_by_n * q.y, two_atan_nbyw_by_n * q.z};
}

// rotation vector -> unit quaternion (the increment of a gyro preintegration step) and the inverse right Jacobian of SO(3)
// (Forster et al., "On-Manifold Preintegration", eq. 8): used by the optional IMU rotation factor of the window BA
inline Quat quat_exp(Vec3 phi) {
  const double th = norm(phi);
  if (th < 1e-8) return quat_normalized({1.0, 0.5 * phi.x, 0.5 * phi.y, 0.5 * phi.z});
  const double s = detm::det_sin(0.5 * th) / th;
  return {detm::det_cos(0.5 * th), s * phi.x, s * phi.y, s * phi.z};
}
inline Mat3 so3_jr_inv(Vec3 phi) {
  const double th = norm(phi);
  const Mat3 S = skew(phi);
  double c = 1.0 / 12.0;
  if (th > 1e-5) c = 1.0 / (th * th) - (1.0 + detm::det_cos(th)) / (2.0 * th * detm::det_sin(th));
  return mat3_add(mat3_add(mat3_identity(), S, 0.5), S * S, c);
}

// g2o::SE3Quat (rotation first in the tangent): se3quat.h
inline void g2o_normalize_rotation(Quat& q) {
  if (q.w < 0) {
    q.w = -q.w;
    q.x = -q.x;
    q.y = -q.y;
    q.z = -q.z;
  }
  q = quat_normalized(q);
}
inline SE3 g2o_from_mat(const Mat3& R, Vec3 t) {
  SE3 r{mat_to_quat(R), t};
  g2o_normalize_rotation(r.q);
  return r;
}
inline SE3 g2o_mul(const SE3& a, const SE3& b) {  // se3quat.h:99-105
  SE3 r;
  r.t = a.t + quat_rotate(a.q, b.t);
  r.q = quat_mul(a.q, b.q);
  g2o_normalize_rotation(r.q);
  return r;
}
inline Vec3 g2o_map(const SE3& T, Vec3 p) { return quat_rotate(T.q, p) + T.t; }
// SE3Quat::exp se3quat.h:208-248 ; update = (omega, upsilon)
inline SE3 g2o_exp(const double* upd) {
  Vec3 omega{upd[0], upd[1], upd[2]}, upsilon{upd[3], upd[4], upd[5]};
  double theta = norm(omega);
  Mat3 Omega = skew(omega);
  Mat3 Omega2 = Omega * Omega;
  Mat3 R, V;
  Mat3 I = mat3_identity();
  if (theta < 0.00001) {
    R = mat3_add(mat3_add(I, Omega), Omega2, 0.5);
    V = mat3_add(mat3_add(I, Omega, 0.5), Omega2, 1.0 / 6.0);
  } else {
    const double st = detm::det_sin(theta), ct = detm::det_cos(theta);
    R = mat3_add(mat3_add(I, Omega, st / theta), Omega2, (1 - ct) / (theta * theta));
    V = mat3_add(mat3_add(I, Omega, (1 - ct) / (theta * theta)), Omega2, (theta - st) / (theta * theta * theta));
  }
  SE3 r{mat_to_quat(R), V * upsilon};
  g2o_normalize_rotation(r.q);
  return r;
}

// kinetic_math.h:17-91 roll/pitch/yaw helpers (R = Rz*Ry*Rx)
inline Mat3 rpy2R(Vec3 rpy) {
  double r = rpy.x, p = rpy.y, y = rpy.z;
  double cy = detm::det_cos(y), sy = detm::det_sin(y), cp = detm::det_cos(p), sp = detm::det_sin(p), cr = detm::det_cos(r), sr = detm::det_sin(r);
  Mat3 R;
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
inline Vec3 R2rpy(const Mat3& R) {
  Vec3 rpy;
  rpy.x = detm::det_atan2(R.m[2][1], R.m[2][2]);
  rpy.y = detm::det_atan2(-R.m[2][0], std::sqrt(R.m[2][1] * R.m[2][1] + R.m[2][2] * R.m[2][2]));
  rpy.z = detm::det_atan2(R.m[1][0], R.m[0][0]);
  return rpy;
}
inline Quat rpy2Q(Vec3 rpy) { return mat_to_quat(rpy2R(rpy)); }
inline Vec3 Q2rpy(Quat q) { return R2rpy(quat_to_mat(q)); }

// Rodrigues (cv::Rodrigues as used by common.h:151-167): rotation matrix <-> rotation vector.
inline Vec3 rodrigues_from_mat(const Mat3& R) {
  // angle-axis via quaternion (numerically stable everywhere except theta==pi, never hit on this path)
  Quat q = mat_to_quat(R);
  if (q.w < 0) q = {-q.w, -q.x, -q.y, -q.z};
  double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  if (n < 1e-15) return {2 * q.x, 2 * q.y, 2 * q.z};
  double theta = 2.0 * std::atan2(n, q.w);
  double s = theta / n;
  return {s * q.x, s * q.y, s * q.z};
}
inline Mat3 rodrigues_to_mat(Vec3 r) {
  double theta = norm(r);
  if (theta < 1e-15) return mat3_identity();
  Vec3 k = (1.0 / theta) * r;
  Mat3 K = skew(k);
  return mat3_add(mat3_add(mat3_identity(), K, std::sin(theta)), K * K, 1 - std::cos(theta));
}

// ---------------------------------------------------------------------------------------------
// cv::RNG (core/operations.hpp): multiply-with-carry, `state = (uint64)(unsigned)state * 4164903690U + (unsigned)(state >> 32)`, output =
// the low word.  Every RANSAC / LMedS run of OpenCV constructs its own `RNG rng((uint64)-1)` (ptsetreg.cpp), so the sample sequence of a
// call depends on nothing but the point count.
struct CvRNG {
  uint64_t state;
  explicit CvRNG(uint64_t s = 0xffffffffffffffffull) : state(s ? s : 0xffffffffull) {}
  uint32_t next() {
    state = (uint64_t)(uint32_t)state * 4164903690u + (uint32_t)(state >> 32);
    return (uint32_t)state;
  }
  int uniform(int a, int b) { return a == b ? a : (int)(next() % (uint32_t)(b - a) + a); }
};

}  // namespace ref
