// flvis_amd: device-side VIMOTION (Madgwick attitude filter + dead reckoning + vision feedback), one thread per stream.
// Mirrors src/processing/vi_motion.cpp:34-464 including its quirks (A16-A19): s *= s.norm(), 10*beta during init,
// float-typed scalar in scalar_multi_q (kinetic_math.h:123), gyro-bias saturation tested on the acc norm.
#pragma once
#include "dev_math.hpp"
#include "pipeline.hpp"

namespace flvis {

struct ViRing {
  MotionState* base;  // [VI_QUEUE] of this stream
  StreamState* st;
  FD int size() const { return st->vi_count; }
  FD MotionState& at(int i) const { return base[(st->vi_head + i) % VI_QUEUE]; }
  FD MotionState& back() const { return at(st->vi_count - 1); }
  FD void clear() const {
    st->vi_head = 0;
    st->vi_count = 0;
  }
  FD void push_back(const MotionState& m) const {  // followed by: if(size()>=STATES_QUEUE_SIZE) pop_front()
    base[(st->vi_head + st->vi_count) % VI_QUEUE] = m;
    st->vi_count++;
    if (st->vi_count >= VI_QUEUE) {
      st->vi_head = (st->vi_head + 1) % VI_QUEUE;
      st->vi_count--;
    }
  }
};

FD Q4 ms_q(const MotionState& m) { return Q4{m.q[0], m.q[1], m.q[2], m.q[3]}; }
FD void ms_set_q(MotionState& m, Q4 q) {
  m.q[0] = q.w;
  m.q[1] = q.x;
  m.q[2] = q.y;
  m.q[3] = q.z;
}
FD V3 ld3(const double* p) { return V3{p[0], p[1], p[2]}; }
FD void st3(double* p, V3 v) {
  p[0] = v.x;
  p[1] = v.y;
  p[2] = v.z;
}

FD Q4 scalar_multi_q(float a, Q4 b) { return Q4{a * b.w, a * b.x, a * b.y, a * b.z}; }
FD Q4 q1_multi_q2(Q4 q1, Q4 q2) {
  Q4 q;
  q.w = q2.w * q1.w - q2.x * q1.x - q2.y * q1.y - q2.z * q1.z;
  q.x = q2.x * q1.w + q2.w * q1.x + q2.z * q1.y - q2.y * q1.z;
  q.y = q2.y * q1.w - q2.z * q1.x + q2.w * q1.y + q2.x * q1.z;
  q.z = q2.z * q1.w + q2.y * q1.x - q2.x * q1.y + q2.w * q1.z;
  return q;
}
FD Q4 q_plus_q(Q4 a, Q4 b) { return Q4{a.w + b.w, a.x + b.x, a.y + b.y, a.z + b.z}; }

FD void madgwick_feedback(Q4 q_prev, V3 acc, double acc_norm, double gain, Q4& qdot) {
  double ax = acc.x / acc_norm, ay = acc.y / acc_norm, az = acc.z / acc_norm;
  double qw = q_prev.w, qx = q_prev.x, qy = q_prev.y, qz = q_prev.z;
  double s0 = 2 * qx * (ay + 2 * qw * qx + 2 * qy * qz) - 2 * qy * (ax - 2 * qw * qy + 2 * qx * qz);
  double s1 = 2 * qw * (ay + 2 * qw * qx + 2 * qy * qz) + 2 * qz * (ax - 2 * qw * qy + 2 * qx * qz) -
              4 * qx * (-2 * qx * qx - 2 * qy * qy + az + 1);
  double s2 = 2 * qz * (ay + 2 * qw * qx + 2 * qy * qz) - 2 * qw * (ax - 2 * qw * qy + 2 * qx * qz) -
              4 * qy * (-2 * qx * qx - 2 * qy * qy + az + 1);
  double s3 = 2 * qx * (ax - 2 * qw * qy + 2 * qx * qz) + 2 * qy * (ay + 2 * qw * qx + 2 * qy * qz);
  double sn = sqrt(s0 * s0 + s1 * s1 + s2 * s2 + s3 * s3);
  s0 *= sn;
  s1 *= sn;
  s2 *= sn;
  s3 *= sn;
  qdot.w -= gain * s0;
  qdot.x -= gain * s1;
  qdot.y -= gain * s2;
  qdot.z -= gain * s3;
}

// VIMOTION::viIMUPropagation (vi_motion.cpp:78-100) on values: the state after one sample from the state before it, plus the gyro
// rotation preintegration since the last keyframe (an addition, see KeyFrameDev::imu_dq)
__device__ inline void vi_propagate(const CamParams& cam, const MotionState& s_prev, double t, V3 acc_raw, V3 gyro_raw, V3 acc_bias,
                                    V3 gyro_bias, MotionState& s_new, Q4& kf_dq, double& kf_dt, V3& kf_dp, V3& kf_dv) {
  const double g = 9.81;
  const V3 acc = acc_raw - acc_bias, gyro = gyro_raw - gyro_bias;
  const double dt = t - s_prev.t;
  const Q4 q_prev = ms_q(s_prev);
  const M3 R_prev = q_to_mat(q_prev);
  const Q4 omega{0, gyro.x, gyro.y, gyro.z};
  Q4 qdot = scalar_multi_q(0.5f, q1_multi_q2(q_prev, omega));
  const double acc_norm = norm(acc);
  if ((acc_norm - g) < 0.3) madgwick_feedback(q_prev, acc, acc_norm, cam.vi_para[0], qdot);
  ms_set_q(s_new, q_normalized(q_plus_q(q_prev, scalar_multi_q((float)dt, qdot))));
  st3(s_new.pos, ld3(s_prev.pos) + ld3(s_prev.vel) * dt);
  st3(s_new.vel, ld3(s_prev.vel) + ((R_prev * acc) - V3{0, 0, -g}) * dt);
  st3(s_new.acc, acc_raw);
  st3(s_new.gyro, gyro_raw);
  s_new.t = t;
  {  // position / velocity preintegration in the body frame of the last keyframe (the attitude increment BEFORE this sample's rotation)
    const V3 a = q_to_mat(kf_dq) * acc;
    kf_dp = (kf_dp + kf_dv * dt) + a * ((0.5 * dt) * dt);
    kf_dv = kf_dv + a * dt;
  }
  kf_dq = q_normalized(q_mul(kf_dq, q_exp(gyro * dt)));
  kf_dt += dt;
}

// what F2FTracking::imu_feed hands back per sample (q_w_i, pos_w_i, vel_w_i; f2f_tracking.cpp:46-57): a row of the stream's
// IMU-state output ring, (t, qw qx qy qz, px py pz, vx vy vz)
constexpr int IMU_ROW = 11;
FD void imu_row_store(double* row, double t, Q4 q, V3 p, V3 v) {
  row[0] = t;
  row[1] = q.w, row[2] = q.x, row[3] = q.y, row[4] = q.z;
  row[5] = p.x, row[6] = p.y, row[7] = p.z;
  row[8] = v.x, row[9] = v.y, row[10] = v.z;
}

// F2FTracking::imu_feed for one sample (f2f_tracking.cpp:46-57); `row` receives the outputs of the call: during the attitude
// initialisation viIMUinitialization returns the identity / zeros (vi_motion.cpp:39-40) except for the sample that sets the first
// attitude (:60), afterwards viIMUPropagation returns the new state (:206-208)
__device__ inline void vi_imu_feed(const CamParams& cam, StreamState& st, const ViRing& ring, double t, V3 acc_raw,
                                   V3 gyro_raw, double* row) {
  const double g = 9.81;
  V3 acc = acc_raw - ld3(st.acc_bias), gyro = gyro_raw - ld3(st.gyro_bias);
  if (!st.vi_initialized) {
    st.has_imu = 1;
    Q4 q_out{1, 0, 0, 0};
    MotionState m;
    st3(m.pos, V3{0, 0, 0});
    st3(m.vel, V3{0, 0, 0});
    st3(m.acc, acc_raw);
    st3(m.gyro, gyro_raw);
    m.t = t;
    if (st.vi_first) {
      if ((norm(acc) - g) < 0.3) {
        V3 rpy{detm::det_atan2(-acc.y, -acc.z), detm::det_atan2(acc.x, -acc.z), 0};
        ms_set_q(m, rpy2Q(rpy));
        ring.push_back(m);
        st.vi_first = 0;
        q_out = ms_q(m);
      }
    } else {
      const MotionState& b = ring.back();
      double dt = t - b.t;
      Q4 q_prev = ms_q(b);
      Q4 omega{0, gyro.x, gyro.y, gyro.z};
      Q4 qdot = scalar_multi_q(0.5f, q1_multi_q2(q_prev, omega));
      double acc_norm = norm(acc);
      if ((acc_norm - g) < 0.3) madgwick_feedback(q_prev, acc, acc_norm, 10 * cam.vi_para[0], qdot);
      Q4 q_new = q_normalized(q_plus_q(q_prev, scalar_multi_q((float)dt, qdot)));
      ms_set_q(m, q_new);
      ring.push_back(m);
      if (ring.size() > 30) st.vi_initialized = 1;
    }
    imu_row_store(row, t, q_out, V3{0, 0, 0}, V3{0, 0, 0});
  } else {
    MotionState s_new;
    Q4 kdq{st.kf_dq[0], st.kf_dq[1], st.kf_dq[2], st.kf_dq[3]};
    double kdt = st.kf_dt;
    V3 kdp = ld3(st.kf_dp), kdv = ld3(st.kf_dv);
    vi_propagate(cam, ring.back(), t, acc_raw, gyro_raw, ld3(st.acc_bias), ld3(st.gyro_bias), s_new, kdq, kdt, kdp, kdv);
    st.kf_dq[0] = kdq.w, st.kf_dq[1] = kdq.x, st.kf_dq[2] = kdq.y, st.kf_dq[3] = kdq.z;
    st.kf_dt = kdt;
    st3(st.kf_dp, kdp);
    st3(st.kf_dv, kdv);
    ring.push_back(s_new);
    imu_row_store(row, t, ms_q(s_new), ld3(s_new.pos), ld3(s_new.vel));
  }
}

FD bool vi_find_state_idx(const ViRing& ring, double time, int& idx_out) {
  int idx = 9999;
  for (int i = ring.size() - 1; i >= 0; i--) {
    if ((ring.at(i).t - time) > 0) {
      idx = i;
    } else {
      idx = i;
      break;
    }
  }
  if (idx > 0 && idx != 9999) {
    idx_out = idx;
    return true;
  }
  return false;
}

__device__ inline void vi_vision_trigger(const ViRing& ring, Q4& init_q) {
  MotionState s = ring.back();
  st3(s.pos, V3{0, 0, 0});
  st3(s.vel, V3{0, 0, 0});
  V3 rpy = Q2rpy(ms_q(s));
  rpy.z = 0;
  Q4 q = q_normalized(rpy2Q(rpy));
  ms_set_q(s, q);
  ring.clear();
  ring.push_back(s);
  init_q = q;
}

__device__ inline bool vi_get_corr_frame_state(const CamParams& cam, const ViRing& ring, double time, SE3d& T_c_w) {
  int idx;
  if (!vi_find_state_idx(ring, time, idx)) return false;
  const MotionState& m = ring.at(idx);
  SE3d T_w_i = se3_from_quat(ms_q(m), ld3(m.pos));
  T_c_w = se3_inverse(se3_mul(T_w_i, load_pose7(cam.T_i_c)));
  return true;
}

__device__ inline void vi_vision_rp_compensation(const CamParams& cam, const ViRing& ring, double time, SE3d& T_c_w) {
  SE3d T_w_i_before = se3_mul(se3_inverse(T_c_w), load_pose7(cam.T_c_i));
  V3 rpy_before = Q2rpy(T_w_i_before.q);
  int idx;
  if (vi_find_state_idx(ring, time, idx)) {
    V3 rpy_imu = Q2rpy(q_normalized(ms_q(ring.at(idx))));
    V3 rpy_vimotion{rpy_imu.x, rpy_imu.y, rpy_before.z};
    double p2 = cam.vi_para[1];
    V3 after = rpy_before * (1 - p2) + rpy_vimotion * p2;
    SE3d T_w_i_after{q_normalized(rpy2Q(after)), T_w_i_before.t};
    T_c_w = se3_inverse(se3_mul(T_w_i_after, load_pose7(cam.T_i_c)));
  }
}

__device__ inline void vi_correction_from_vision(const CamParams& cam, StreamState& st, const ViRing& ring, double t_curr,
                                                 const SE3d& Tcw_curr, double t_last, const SE3d& Tcw_last) {
  const double ba_sat = 0.5, bw_sat = 0.1;  // vifusion_para5/6 never reach VIMOTION (quirk A19)
  int idx_curr, idx_last;
  if (!(vi_find_state_idx(ring, t_last, idx_last) && vi_find_state_idx(ring, t_curr, idx_curr))) return;
  if (idx_last == idx_curr) return;
  double dt = t_curr - t_last;
  int idx_mid = idx_last + (int)floor((double)((idx_curr - idx_last) / 2));
  SE3d T_c_i = load_pose7(cam.T_c_i);
  SE3d T_w_iA = se3_mul(se3_inverse(Tcw_last), T_c_i);
  SE3d T_w_iB = se3_mul(se3_inverse(Tcw_curr), T_c_i);
  SE3d T_w_ia = se3_from_quat(ms_q(ring.at(idx_last)), ld3(ring.at(idx_last).pos));
  SE3d T_w_ib = se3_from_quat(ms_q(ring.at(idx_curr)), ld3(ring.at(idx_curr).pos));
  SE3d T_w_im = se3_from_quat(ms_q(ring.at(idx_mid)), ld3(ring.at(idx_mid).pos));
  SE3d T_iB_iA = se3_mul(se3_inverse(T_w_iB), T_w_iA);
  SE3d T_ib_ia = se3_mul(se3_inverse(T_w_ib), T_w_ia);
  Q4 Q_B_A = T_iB_iA.q, Q_b_a = T_ib_ia.q;
  double n2 = q_sqnorm(Q_b_a);
  Q4 Q_b_a_inv{Q_b_a.w / n2, -Q_b_a.x / n2, -Q_b_a.y / n2, -Q_b_a.z / n2};
  Q4 Q_B_b = q_mul(Q_B_A, Q_b_a_inv);
  V3 gyro_bias_est{Q_B_b.x / dt, Q_B_b.y / dt, Q_B_b.z / dt};
  int cnt = idx_curr - idx_last + 1;
  V3 vel_imu{0, 0, 0};
  for (int i = idx_last; i <= idx_curr; i++) vel_imu = vel_imu + ld3(ring.at(i).vel);
  vel_imu = vel_imu * (1.0 / cnt);
  V3 vel_vision_world{(T_w_iB.t.x - T_w_iA.t.x) / dt, (T_w_iB.t.y - T_w_iA.t.y) / dt, (T_w_iB.t.z - T_w_iA.t.z) / dt};
  V3 diff_vel_world = vel_vision_world - vel_imu;
  Q4 qm = T_w_im.q;
  double nm2 = q_sqnorm(qm);
  Q4 qm_inv{qm.w / nm2, -qm.x / nm2, -qm.y / nm2, -qm.z / nm2};
  V3 diff_vel_local = q_to_mat(qm_inv) * diff_vel_world;
  V3 acc_bias_est{-diff_vel_local.x / dt, -diff_vel_local.y / dt, -diff_vel_local.z / dt};
  SE3d T_diff = se3_mul(T_w_iB, se3_inverse(T_w_ib));
  for (int i = idx_curr; i < ring.size(); i++) {
    MotionState& m = ring.at(i);
    SE3d newT = se3_mul(T_diff, se3_from_quat(ms_q(m), ld3(m.pos)));
    ms_set_q(m, newT.q);
    st3(m.pos, newT.t);
    st3(m.vel, ld3(m.vel) + diff_vel_world);
  }
  if (isnan(acc_bias_est.x)) acc_bias_est = V3{0, 0, 0};
  if (isnan(gyro_bias_est.x)) gyro_bias_est = V3{0, 0, 0};
  double ba_est_norm = norm(acc_bias_est);
  if (ba_est_norm > ba_sat) acc_bias_est = acc_bias_est * (ba_sat / ba_est_norm);
  double bw_est_norm = norm(gyro_bias_est);
  if (ba_est_norm > bw_sat) gyro_bias_est = gyro_bias_est * (bw_sat / bw_est_norm);
  if (dt < 0.1) {
    double p3 = cam.vi_para[2], p4 = cam.vi_para[3];
    st3(st.acc_bias, (1 - p3) * ld3(st.acc_bias) + (p3)*acc_bias_est);
    st3(st.gyro_bias, (1 - p3) * ld3(st.gyro_bias) + (p4)*gyro_bias_est);
  }
}

}  // namespace flvis
