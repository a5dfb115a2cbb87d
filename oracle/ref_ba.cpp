// ORACLE (test infrastructure, NOT product code) -- sliding-window bundle adjustment of the local-map nodelet.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Follows:
//   LocalMapNodeletClass::frame_callback   src/backend/vo_localmap.cpp:87-380   (graph bookkeeping incl. quirk A22/A23)
//   PoseLMBag                              src/backend/poselmbag.cpp:5-208
//   g2o pieces the callback exercises (3rdPartLib/g2o/g2o/...):
//     core/sparse_optimizer.cpp:168-272,366-431   initializeOptimization / index mapping / optimize loop
//     core/optimization_algorithm_levenberg.cpp:58-175   LM control (tau 1e-5, <=10 trials, rho test, lambda schedule)
//     core/block_solver.hpp:314-447,462-565        buildSystem / setLambda / Schur solve / back-substitution
//     core/base_binary_edge.hpp:61-134, core/robust_kernel_impl.cpp:65-78, core/base_edge.h:79-123   Huber, rho' weighting
//     types/sba/types_six_dof_expmap.cpp:389-433, types/slam3d/se3quat.h   projection edge, SE3 exp map
//   The reduced system is solved by dense Cholesky (reference: CHOLMOD sparse Cholesky, external, exact up to rounding).
//   Edges whose container order is unspecified in the reference (std::set of pointers, vo_localmap.cpp:254-260) are
//   kept in edge-id order here; outlier ids are therefore reported by descending edge id.
// parity unpinned: g2o's unit tests do not cover types/sba, the Schur path or Huber; tests/ pins this file
// against scipy.optimize.least_squares(loss='huber') optima and central-difference Jacobians.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "ref_api.h"

namespace ref {

// ------------------------------------------------------------------------------------------ PoseLMBag (poselmbag.cpp)
PoseLMBag::PoseLMBag(int n) : pose_buffer_size(n) { reset(); }
void PoseLMBag::reset() {
  lm_sub_bag.clear();
  pose_sub_bag.assign(pose_buffer_size, POSE_ITEM{0, 0, se3_identity()});
  wp_init = 0;
  pose_sub_bag_initialized = false;
  newest = oldest = 0;
}
bool PoseLMBag::hasTheLM(int64_t id, int& idx) const {
  idx = 0;
  for (size_t i = 0; i < lm_sub_bag.size(); i++)
    if (lm_sub_bag[i].id == id) {
      idx = (int)i;
      return true;
    }
  return false;
}
bool PoseLMBag::addLMObservationSlidingWindow(int64_t id, Vec3 p) {
  int idx;
  if (hasTheLM(id, idx)) {
    lm_sub_bag[idx].count++;
    return false;
  }
  lm_sub_bag.push_back({id, 1, p});
  return true;
}
bool PoseLMBag::addLMObservation(int64_t id, Vec3 p_in) {
  int idx;
  if (hasTheLM(id, idx)) {
    int cnt = lm_sub_bag[idx].count;
    Vec3 p = (double)cnt * lm_sub_bag[idx].p3d_w + p_in;
    cnt++;
    p = (1.0 / (double)cnt) * p;
    lm_sub_bag[idx].count = cnt;
    lm_sub_bag[idx].p3d_w = p;
    return false;
  }
  lm_sub_bag.push_back({id, 1, p_in});
  return true;
}
bool PoseLMBag::removeLMObservation(int64_t id) {
  int idx;
  if (hasTheLM(id, idx)) {
    lm_sub_bag[idx].count--;
    if (lm_sub_bag[idx].count == 0) {
      lm_sub_bag.erase(lm_sub_bag.begin() + idx);
      return true;
    }
  }
  return false;
}
void PoseLMBag::addPose(int64_t id, const SE3& pose) {
  if (pose_sub_bag_initialized) {
    newest = oldest;
    pose_sub_bag[newest].relevent_frame_id = id;
    pose_sub_bag[newest].pose = pose;
    oldest++;
    if (oldest == pose_buffer_size) oldest = 0;
  } else {
    pose_sub_bag[wp_init].relevent_frame_id = id;
    pose_sub_bag[wp_init].pose = pose;
    pose_sub_bag[wp_init].pose_id = wp_init;
    wp_init++;
    if (wp_init == pose_buffer_size) {
      pose_sub_bag_initialized = true;
      oldest = 0;
      newest = pose_buffer_size - 1;
    }
  }
}
int64_t PoseLMBag::getPoseIdByReleventFrameId(int64_t frame_id) const {
  for (int i = 0; i < pose_buffer_size; i++)
    if (pose_sub_bag[i].relevent_frame_id == frame_id) return i;
  return -1;
}

// ------------------------------------------------------------------------------------------ g2o graph (subset)
inline void edge_error(const SE3& T, Vec3 p, Vec2 z, const double K[4], double e[2]) {
  Vec3 X = g2o_map(T, p);
  e[0] = z.x - (X.x / X.z * K[0] + K[2]);
  e[1] = z.y - (X.y / X.z * K[1] + K[3]);
}

// EdgeSE3ProjectXYZ::linearizeOplus  (Ji: 2x3 wrt landmark, Jj: 2x6 wrt pose)
static inline void edge_jacobians(const SE3& T, Vec3 p, const double K[4], double Ji[2][3], double Jj[2][6]) {
  Vec3 X = g2o_map(T, p);
  double x = X.x, y = X.y, z = X.z, z2 = z * z, fx = K[0], fy = K[1];
  Mat3 R = quat_to_mat(T.q);
  double tmp[2][3] = {{fx, 0, -x / z * fx}, {0, fy, -y / z * fy}};
  for (int r = 0; r < 2; r++)
    for (int c = 0; c < 3; c++) {
      double s = tmp[r][0] * R.m[0][c] + tmp[r][1] * R.m[1][c] + tmp[r][2] * R.m[2][c];
      Ji[r][c] = -1. / z * s;
    }
  Jj[0][0] = x * y / z2 * fx;
  Jj[0][1] = -(1 + (x * x / z2)) * fx;
  Jj[0][2] = y / z * fx;
  Jj[0][3] = -1. / z * fx;
  Jj[0][4] = 0;
  Jj[0][5] = x / z2 * fx;
  Jj[1][0] = (1 + y * y / z2) * fy;
  Jj[1][1] = -x * y / z2 * fy;
  Jj[1][2] = -x / z * fy;
  Jj[1][3] = 0;
  Jj[1][4] = -1. / z * fy;
  Jj[1][5] = y / z2 * fy;
}

inline double huber_rho(double e) { return e <= 1.0 ? e : 2 * std::sqrt(e) - 1.0; }
static inline double huber_w(double e) { return e <= 1.0 ? 1.0 : 1.0 / std::sqrt(e); }

// dense Cholesky solve of an n x n SPD system (row-major); false if not positive definite
static bool chol_solve(std::vector<double>& A, int n, const double* b, double* x) {
  for (int j = 0; j < n; j++) {
    double s = A[(size_t)j * n + j];
    for (int k = 0; k < j; k++) s -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(s > 0) || !std::isfinite(s)) return false;
    double d = std::sqrt(s);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double v = A[(size_t)i * n + j];
      for (int k = 0; k < j; k++) v -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = v / d;
    }
  }
  std::vector<double> y(n);
  for (int i = 0; i < n; i++) {
    double v = b[i];
    for (int k = 0; k < i; k++) v -= A[(size_t)i * n + k] * y[k];
    y[i] = v / A[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double v = y[i];
    for (int k = i + 1; k < n; k++) v -= A[(size_t)k * n + i] * x[k];
    x[i] = v / A[(size_t)i * n + i];
  }
  return true;
}

// IMU rotation factor between pose slots a and b (T_c_w estimates as g2o SE3Quat, update T <- exp(dx) T with dx = (omega, upsilon)):
//   R_b = R_w_body = R_cw^T R_cb,   r = Log(dq^T R_b(a)^T R_b(b)) = Log(dq^T R_cb^T R_cw(a) R_cw(b)^T R_cb)
//   d r / d omega_a = Jr^-1(r) R_cb^T (R_cw(a) R_cw(b)^T)^T,   d r / d omega_b = -Jr^-1(r) R_cb^T        (translations: zero)
void imu_edge_linearize(const SE3& Ta, const SE3& Tb, Quat q_c_b, Quat dq, double r[3], double Ja[3][3], double Jb[3][3]) {
  Quat qr = quat_mul(quat_mul(quat_mul(quat_mul(quat_conj(dq), quat_conj(q_c_b)), Ta.q), quat_conj(Tb.q)), q_c_b);
  qr = quat_normalized(qr);
  if (qr.w < 0) qr = {-qr.w, -qr.x, -qr.y, -qr.z};
  const Vec3 rv = so3_log(qr);
  r[0] = rv.x;
  r[1] = rv.y;
  r[2] = rv.z;
  if (!Ja) return;
  const Mat3 Bt = transpose(quat_to_mat(q_c_b));
  const Mat3 M = quat_to_mat(Ta.q) * transpose(quat_to_mat(Tb.q));
  const Mat3 Ji = so3_jr_inv(rv);
  const Mat3 A = Ji * (Bt * transpose(M));
  const Mat3 Bm = Ji * Bt;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      Ja[i][j] = A.m[i][j];
      Jb[i][j] = -Bm.m[i][j];
    }
}

// Position rows of the IMU factor (an addition like the rotation rows: the reference's window has reprojection edges only).  Body
// position and attitude from the camera pose and the camera-from-body extrinsic: R_b = R_cw^T R_cb, p_b = R_cw^T (t_cb - t_cw).
// With the specific-force convention of VIMOTION (world acceleration = R f - g_w, g_w = (0, 0, -9.81); vi_motion.cpp:193-199) the
// preintegration dp = sum (dv dt + 1/2 dR f dt^2), dv = sum dR f dt over the samples between the keyframes gives
//   p_b(b) = p_b(a) + v_a dt - 1/2 g_w dt^2 + R_b(a) dp,       r = R_b(a)^T (p_b(b) - p_b(a) - v_a dt + 1/2 g_w dt^2) - dp,
// v_a = the tracker's filter velocity at keyframe a, a fixed quantity (no velocity vertex: the pose blocks stay 6-dimensional).
// g2o's update T <- exp((omega, upsilon)) T moves p_b by R_cw^T ([t_cb]x omega - upsilon) and R_b by Exp(-R_cb^T omega) on the right:
//   dr/d omega_b = R_b(a)^T R_cw(b)^T [t_cb]x          dr/d upsilon_b = -R_b(a)^T R_cw(b)^T
//   dr/d omega_a = -R_cb^T [t_cb]x - [R_b(a)^T d]x R_cb^T      dr/d upsilon_a = R_cb^T          (d = the bracket above)
void imu_edge_linearize_pos(const SE3& Ta, const SE3& Tb, Quat q_c_b, Vec3 t_c_b, Vec3 dp, Vec3 va, double dt, double r[3], double Ja[3][6],
                            double Jb[3][6]) {
  const Mat3 Rca = quat_to_mat(Ta.q), Rcb = quat_to_mat(Tb.q), Rci = quat_to_mat(q_c_b);
  const Mat3 RcaT = transpose(Rca), RcbT = transpose(Rcb), RciT = transpose(Rci);
  const Vec3 pa = RcaT * (t_c_b - Ta.t), pb = RcbT * (t_c_b - Tb.t);
  const Vec3 gw{0, 0, -9.81};
  const Vec3 d = ((pb - pa) - dt * va) + (0.5 * dt * dt) * gw;
  const Mat3 RbaT = RciT * Rca;  // R_b(a)^T = R_cb^T R_cw(a)
  const Vec3 rv = (RbaT * d) - dp;
  r[0] = rv.x;
  r[1] = rv.y;
  r[2] = rv.z;
  if (!Ja) return;
  const Mat3 Sx = skew(t_c_b);
  const Mat3 Job = (RbaT * RcbT) * Sx;  // omega_b
  const Mat3 Z = {{{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}};
  const Mat3 Jub = mat3_add(Z, RbaT * RcbT, -1.0);
  const Mat3 Joa = mat3_add(mat3_add(Z, RciT * Sx, -1.0), skew(RbaT * d) * RciT, -1.0);
  const Mat3 Jua = RciT;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      Ja[i][j] = Joa.m[i][j];
      Ja[i][3 + j] = Jua.m[i][j];
      Jb[i][j] = Job.m[i][j];
      Jb[i][3 + j] = Jub.m[i][j];
    }
}

// SparseOptimizer::initializeOptimization + optimize(iterations) on the current graph.
void BAGraph::optimize(int iterations) {
  // active edges in id order; active vertices = those touched by an active edge (all edges are active: landmarks are never fixed)
  std::vector<Edge*> act;
  for (auto& kv : edges) act.push_back(&kv.second);
  if (act.empty()) return;
  std::map<int, int> pose_index;      // slot -> hessian index (free poses only)
  std::map<int64_t, int> lm_index;    // lm id -> index
  for (Edge* e : act) {
    if (!poses[e->pose].fixed) pose_index[e->pose] = 0;
    lm_index[e->lm] = 0;
  }
  int P = 0, L = 0;
  for (auto& kv : pose_index) kv.second = P++;
  for (auto& kv : lm_index) kv.second = L++;
  std::vector<SE3*> pv(P);
  std::vector<Vec3*> lv(L);
  for (auto& kv : pose_index) pv[kv.second] = &poses[kv.first].est;
  for (auto& kv : lm_index) lv[kv.second] = &lms[kv.first];
  const int E = (int)act.size();
  std::vector<int> ep(E), el(E);
  for (int k = 0; k < E; k++) {
    ep[k] = poses[act[k]->pose].fixed ? -1 : pose_index[act[k]->pose];
    el[k] = lm_index[act[k]->lm];
  }
  const int sizePoses = 6 * P, sizeLms = 3 * L;
  std::vector<double> Hpp((size_t)P * 36), Hll((size_t)L * 9), Hpl((size_t)E * 18), b(sizePoses + sizeLms),
      x(sizePoses + sizeLms);

  // IMU rotation edges whose poses take part (a fixed pose contributes its estimate, not a block)
  auto hidx = [&](int slot) {
    auto it = pose_index.find(slot);
    return (it == pose_index.end() || poses[slot].fixed) ? -1 : it->second;
  };
  std::vector<double> Hoff((size_t)imu_edges.size() * 36, 0.0);  // Ja^T W Jb of every IMU edge (6x6 block (a, b) of Hpp)
  // residual (rotation rows, then position rows; the latter zero without them) and 6 x 6 Jacobians of an IMU edge
  auto imu_lin = [&](const ImuEdge& e, double r[6], double Ja[6][6], double Jb[6][6]) {
    double r3[3], A3[3][3], B3[3][3];
    imu_edge_linearize(poses[e.a].est, poses[e.b].est, q_c_b, e.dq, r3, Ja ? A3 : nullptr, Ja ? B3 : nullptr);
    for (int i = 0; i < 3; i++) r[i] = r3[i], r[3 + i] = 0;
    if (Ja)
      for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) Ja[i][j] = Jb[i][j] = 0;
    if (Ja)
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Ja[i][j] = A3[i][j], Jb[i][j] = B3[i][j];
    if (e.wp > 0) {
      double rp[3], Ap[3][6], Bp[3][6];
      imu_edge_linearize_pos(poses[e.a].est, poses[e.b].est, q_c_b, t_c_b, e.dp, e.va, e.dt, rp, Ja ? Ap : nullptr, Ja ? Bp : nullptr);
      for (int i = 0; i < 3; i++) r[3 + i] = rp[i];
      if (Ja)
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 6; j++) Ja[3 + i][j] = Ap[i][j], Jb[3 + i][j] = Bp[i][j];
    }
  };
  auto imu_chi = [&](const ImuEdge& e, const double r[6]) {
    double c = e.w * ((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
    if (e.wp > 0) c += e.wp * ((r[3] * r[3] + r[4] * r[4]) + r[5] * r[5]);
    return c;
  };
  auto robustChi2 = [&]() {
    double chi = 0;
    for (int k = 0; k < E; k++) {
      double er[2];
      edge_error(poses[act[k]->pose].est, lms[act[k]->lm], act[k]->z, K, er);
      chi += huber_rho(er[0] * er[0] + er[1] * er[1]);
    }
    for (const ImuEdge& e : imu_edges) {
      double r[6];
      imu_lin(e, r, nullptr, nullptr);
      chi += imu_chi(e, r);
    }
    return chi;
  };

  double lambda = -1, ni = 2;
  for (int iteration = 0; iteration < iterations; iteration++) {
    double currentChi = robustChi2();
    // ---- buildSystem
    std::fill(Hpp.begin(), Hpp.end(), 0.0);
    std::fill(Hll.begin(), Hll.end(), 0.0);
    std::fill(Hpl.begin(), Hpl.end(), 0.0);
    std::fill(b.begin(), b.end(), 0.0);
    for (int k = 0; k < E; k++) {
      const SE3& T = poses[act[k]->pose].est;
      Vec3 p = lms[act[k]->lm];
      double er[2], Ji[2][3], Jj[2][6];
      edge_error(T, p, act[k]->z, K, er);
      edge_jacobians(T, p, K, Ji, Jj);
      double w = huber_w(er[0] * er[0] + er[1] * er[1]);
      double o0 = -er[0] * w, o1 = -er[1] * w;
      double* hl = &Hll[(size_t)el[k] * 9];
      double* bl = &b[sizePoses + 3 * el[k]];
      for (int r = 0; r < 3; r++) {
        bl[r] += Ji[0][r] * o0 + Ji[1][r] * o1;
        for (int c = 0; c < 3; c++) hl[3 * r + c] += (Ji[0][r] * w) * Ji[0][c] + (Ji[1][r] * w) * Ji[1][c];
      }
      if (ep[k] >= 0) {
        double* hp = &Hpp[(size_t)ep[k] * 36];
        double* bp = &b[6 * ep[k]];
        double* hpl = &Hpl[(size_t)k * 18];  // 6x3 block  B^T w A
        for (int r = 0; r < 6; r++) {
          bp[r] += Jj[0][r] * o0 + Jj[1][r] * o1;
          for (int c = 0; c < 6; c++) hp[6 * r + c] += (Jj[0][r] * w) * Jj[0][c] + (Jj[1][r] * w) * Jj[1][c];
          for (int c = 0; c < 3; c++) hpl[3 * r + c] += (Jj[0][r] * w) * Ji[0][c] + (Jj[1][r] * w) * Ji[1][c];
        }
      }
    }
    for (size_t k = 0; k < imu_edges.size(); k++) {  // pose-pose edges: rows 0..2 rotation (weight w), rows 3..5 position (weight wp)
      const ImuEdge& e = imu_edges[k];
      double r[6], Ja[6][6], Jb[6][6];
      imu_lin(e, r, Ja, Jb);
      const int nrow = e.wp > 0 ? 6 : 3, ncol = e.wp > 0 ? 6 : 3;  // (rotation-only edges touch the rotation rows / columns alone)
      const double wr[6] = {e.w, e.w, e.w, e.wp, e.wp, e.wp};
      const int ia = hidx(e.a), ib = hidx(e.b);
      for (int i = 0; i < ncol; i++)
        for (int j = 0; j < ncol; j++) {
          double aa = 0, bb = 0, ab = 0;
          for (int m = 0; m < nrow; m++) {
            aa += (Ja[m][i] * wr[m]) * Ja[m][j];
            bb += (Jb[m][i] * wr[m]) * Jb[m][j];
            ab += (Ja[m][i] * wr[m]) * Jb[m][j];
          }
          if (ia >= 0) Hpp[(size_t)ia * 36 + 6 * i + j] += aa;
          if (ib >= 0) Hpp[(size_t)ib * 36 + 6 * i + j] += bb;
          Hoff[k * 36 + 6 * i + j] = ab;
        }
      for (int i = 0; i < ncol; i++) {
        double ga = 0, gb = 0;
        for (int m = 0; m < nrow; m++) {
          ga += (Ja[m][i] * wr[m]) * r[m];
          gb += (Jb[m][i] * wr[m]) * r[m];
        }
        if (ia >= 0) b[6 * ia + i] -= ga;
        if (ib >= 0) b[6 * ib + i] -= gb;
      }
    }
    if (iteration == 0) {
      double maxDiag = 0;
      for (int i = 0; i < P; i++)
        for (int j = 0; j < 6; j++) maxDiag = std::max(std::fabs(Hpp[(size_t)i * 36 + 7 * j]), maxDiag);
      for (int i = 0; i < L; i++)
        for (int j = 0; j < 3; j++) maxDiag = std::max(std::fabs(Hll[(size_t)i * 9 + 4 * j]), maxDiag);
      lambda = 1e-5 * maxDiag;
      ni = 2;
    }
    double rho = 0;
    int qmax = 0;
    bool lambda_bad = false;
    do {
      // push
      std::vector<SE3> bp(P);
      std::vector<Vec3> bl(L);
      for (int i = 0; i < P; i++) bp[i] = *pv[i];
      for (int i = 0; i < L; i++) bl[i] = *lv[i];
      // ---- Schur solve with lambda on both diagonals
      std::vector<double> Hs((size_t)sizePoses * sizePoses, 0.0), bs(sizePoses), coeff(sizePoses, 0.0);
      for (int i = 0; i < P; i++)
        for (int r = 0; r < 6; r++)
          for (int c = 0; c < 6; c++)
            Hs[(size_t)(6 * i + r) * sizePoses + 6 * i + c] = Hpp[(size_t)i * 36 + 6 * r + c] + (r == c ? lambda : 0.0);
      for (size_t k = 0; k < imu_edges.size(); k++) {
        const int ia = hidx(imu_edges[k].a), ib = hidx(imu_edges[k].b);
        if (ia < 0 || ib < 0) continue;
        const int nc = imu_edges[k].wp > 0 ? 6 : 3;
        for (int i = 0; i < nc; i++)
          for (int j = 0; j < nc; j++) {
            Hs[(size_t)(6 * ia + i) * sizePoses + 6 * ib + j] += Hoff[k * 36 + 6 * i + j];
            Hs[(size_t)(6 * ib + j) * sizePoses + 6 * ia + i] += Hoff[k * 36 + 6 * i + j];
          }
      }
      std::vector<Mat3> Dinv(L);
      std::vector<std::vector<int>> lm_edges(L);
      for (int k = 0; k < E; k++)
        if (ep[k] >= 0) lm_edges[el[k]].push_back(k);
      bool ok2 = true;
      for (int l = 0; l < L; l++) {
        Mat3 D;
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 3; c++) D.m[r][c] = Hll[(size_t)l * 9 + 3 * r + c] + (r == c ? lambda : 0.0);
        mat3_inverse(D, Dinv[l]);
        Vec3 db = Dinv[l] * Vec3{b[sizePoses + 3 * l], b[sizePoses + 3 * l + 1], b[sizePoses + 3 * l + 2]};
        std::vector<int>& le = lm_edges[l];
        std::sort(le.begin(), le.end(), [&](int a, int c) { return ep[a] < ep[c]; });
        for (size_t a = 0; a < le.size(); a++) {
          const double* Bi = &Hpl[(size_t)le[a] * 18];
          int i1 = ep[le[a]];
          double BD[18];
          for (int r = 0; r < 6; r++)
            for (int c = 0; c < 3; c++)
              BD[3 * r + c] = Bi[3 * r] * Dinv[l].m[0][c] + Bi[3 * r + 1] * Dinv[l].m[1][c] + Bi[3 * r + 2] * Dinv[l].m[2][c];
          for (int r = 0; r < 6; r++) coeff[6 * i1 + r] += Bi[3 * r] * db.x + Bi[3 * r + 1] * db.y + Bi[3 * r + 2] * db.z;
          for (size_t c2 = a; c2 < le.size(); c2++) {
            const double* Bj = &Hpl[(size_t)le[c2] * 18];
            int i2 = ep[le[c2]];
            for (int r = 0; r < 6; r++)
              for (int c = 0; c < 6; c++) {
                double v = BD[3 * r] * Bj[3 * c] + BD[3 * r + 1] * Bj[3 * c + 1] + BD[3 * r + 2] * Bj[3 * c + 2];
                Hs[(size_t)(6 * i1 + r) * sizePoses + 6 * i2 + c] -= v;
                if (i2 != i1) Hs[(size_t)(6 * i2 + c) * sizePoses + 6 * i1 + r] -= v;  // symmetric fill for the dense solver
              }
          }
        }
      }
      for (int i = 0; i < sizePoses; i++) bs[i] = b[i] - coeff[i];
      std::fill(x.begin(), x.end(), 0.0);
      if (sizePoses > 0) ok2 = chol_solve(Hs, sizePoses, bs.data(), x.data());
      if (ok2) {
        // landmarks: xl = Dinv (bl - B^T xp)
        for (int l = 0; l < L; l++) {
          Vec3 cl{b[sizePoses + 3 * l], b[sizePoses + 3 * l + 1], b[sizePoses + 3 * l + 2]};
          for (int k : lm_edges[l]) {
            const double* Bi = &Hpl[(size_t)k * 18];
            const double* xp = &x[6 * ep[k]];
            for (int c = 0; c < 3; c++) {
              double s = 0;
              for (int r = 0; r < 6; r++) s += Bi[3 * r + c] * xp[r];
              cl[c] -= s;
            }
          }
          Vec3 xl = Dinv[l] * cl;
          x[sizePoses + 3 * l] = xl.x;
          x[sizePoses + 3 * l + 1] = xl.y;
          x[sizePoses + 3 * l + 2] = xl.z;
        }
        // update (oplus)
        for (int i = 0; i < P; i++) *pv[i] = g2o_mul(g2o_exp(&x[6 * i]), *pv[i]);
        for (int i = 0; i < L; i++) *lv[i] = *lv[i] + Vec3{x[sizePoses + 3 * i], x[sizePoses + 3 * i + 1], x[sizePoses + 3 * i + 2]};
      }
      double tempChi = robustChi2();
      if (!ok2) tempChi = DBL_MAX;
      rho = currentChi - tempChi;
      double scale = 0;
      for (size_t j = 0; j < x.size(); j++) scale += x[j] * (lambda * x[j] + b[j]);
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
        for (int i = 0; i < P; i++) *pv[i] = bp[i];
        for (int i = 0; i < L; i++) *lv[i] = bl[i];
        if (!std::isfinite(lambda)) {
          lambda_bad = true;
          break;
        }
      }
      qmax++;
    } while (rho < 0 && qmax < 10);
    if (qmax == 10 || rho == 0 || lambda_bad) break;
  }
}

void BAGraph::remove_pose(int slot) {
  for (auto it = edges.begin(); it != edges.end();) it = (it->second.pose == slot) ? edges.erase(it) : std::next(it);
  poses[slot].present = false;
}
void BAGraph::remove_lm(int64_t id) {
  for (auto it = edges.begin(); it != edges.end();) it = (it->second.lm == id) ? edges.erase(it) : std::next(it);
  lms.erase(id);
}

// ------------------------------------------------------------------------------------------ LocalMap (vo_localmap.cpp)
LocalMap::LocalMap(int window, double fx, double fy, double cx, double cy) : bag(window), window_size(window) {
  graph.K[0] = fx;
  graph.K[1] = fy;
  graph.K[2] = cx;
  graph.K[3] = cy;
  graph.poses.assign(window, BAGraph::PoseV{se3_identity(), false, false});
  state = UN_INITIALIZED;
  edge_id = 0;
  slot_dq.assign(window, quat_identity());
  slot_dt.assign(window, 0.0);
  slot_has.assign(window, 0);
  slot_dp.assign(window, Vec3{0, 0, 0});
  slot_va.assign(window, Vec3{0, 0, 0});
}

void LocalMap::set_imu_factor(bool on, double sigma_g, Quat q_c_b) {
  imu_factor = on;
  imu_sigma_g = sigma_g;
  graph.q_c_b = q_c_b;
}
// position rows on top of the rotation rows: accelerometer noise density sigma_a (information I / (sigma_a^2 dt^3 / 3)), and the
// translation of the camera-from-body extrinsic
void LocalMap::set_imu_factor_pos(double sigma_a, Vec3 t_c_b) {
  imu_sigma_a = sigma_a;
  graph.t_c_b = t_c_b;
}
// the IMU edges of the current window: pose slot j is linked to its chronological predecessor (the previous ring slot) unless j
// is the oldest pose of the window
void LocalMap::rebuild_imu_edges() {
  graph.imu_edges.clear();
  if (!imu_factor) return;
  const int W = window_size;
  for (int j = 0; j < W; j++) {
    const int i = (j + W - 1) % W;
    if (j == bag.oldest || !slot_has[j] || !(slot_dt[j] > 0) || !graph.poses[i].present || !graph.poses[j].present) continue;
    BAGraph::ImuEdge e{i, j, slot_dq[j], 1.0 / (imu_sigma_g * imu_sigma_g * slot_dt[j])};
    if (imu_sigma_a > 0) {
      e.dp = slot_dp[j];
      e.va = slot_va[j];
      e.dt = slot_dt[j];
      e.wp = 1.0 / (imu_sigma_a * imu_sigma_a * ((slot_dt[j] * slot_dt[j]) * slot_dt[j]) / 3.0);
    }
    graph.imu_edges.push_back(e);
  }
}

void LocalMap::reset() {  // KFMSG_CMD_RESET_LM, vo_localmap.cpp:89-98
  state = UN_INITIALIZED;
  bag.reset();
  kfs.clear();
  graph.edges.clear();
  graph.lms.clear();
  for (auto& p : graph.poses) p = BAGraph::PoseV{se3_identity(), false, false};
}

static SE3 to_g2o(const SE3& T) { return g2o_from_mat(quat_to_mat(T.q), T.t); }

bool LocalMap::frame_callback(const KeyFrameStruct& kf, CorrectionInfStruct& out) {
  kfs.push_back(kf);
  switch (state) {
    case UN_INITIALIZED: {
      if ((int)kfs.size() < window_size) return false;  // :211-214 (no pop_front)
      for (int f = 0; f < window_size; f++) {
        bag.addPose(kfs[f].frame_id, kfs[f].T_c_w);
        for (int i = 0; i < kfs[f].lm_count; i++) bag.addLMObservation(kfs[f].lm_id[i], kfs[f].lm_3d[i]);
      }
      for (int f = 0; f < window_size; f++) {  // slot f holds keyframe f during initialisation
        slot_dq[f] = kfs[f].imu_dq;
        slot_dt[f] = kfs[f].imu_dt;
        slot_has[f] = (f > 0 && kfs[f].imu_valid) ? 1 : 0;
        slot_dp[f] = kfs[f].imu_dp;
        slot_va[f] = kfs[f].imu_va;
      }
      int oldest = bag.oldest;
      for (int i = 0; i < window_size; i++) {
        graph.poses[i].present = true;
        graph.poses[i].fixed = (bag.pose_sub_bag[i].pose_id == oldest);
        graph.poses[i].est = to_g2o(bag.pose_sub_bag[i].pose);
      }
      for (auto& lm : bag.lm_sub_bag) graph.lms[lm.id] = lm.p3d_w;
      edge_id = 0;
      graph.edges.clear();
      for (int f = 0; f < window_size; f++) {
        int slot = (int)bag.getPoseIdByReleventFrameId(kfs[f].frame_id);
        for (int i = 0; i < kfs[f].lm_count; i++) {
          graph.edges[edge_id] = BAGraph::Edge{edge_id, kfs[f].lm_id[i], slot, kfs[f].lm_2d[i]};
          edge_id++;
        }
      }
      state = OPTIMIZING;
      break;
    }
    case SLIDING_WINDOW: {
      graph.remove_pose(bag.oldest);
      for (int64_t id : kfs.at(0).lm_id)  // quirk A22: kfs[0] is k1 on the first slide
        if (bag.removeLMObservation(id)) graph.remove_lm(id);
      bag.addPose(kfs.back().frame_id, kfs.back().T_c_w);
      BAGraph::PoseV& nv = graph.poses[bag.newest];
      nv.present = true;
      nv.fixed = false;
      nv.est = to_g2o(kfs.back().T_c_w);
      slot_dq[bag.newest] = kfs.back().imu_dq;
      slot_dt[bag.newest] = kfs.back().imu_dt;
      slot_has[bag.newest] = kfs.back().imu_valid ? 1 : 0;
      slot_dp[bag.newest] = kfs.back().imu_dp;
      slot_va[bag.newest] = kfs.back().imu_va;
      graph.poses[bag.oldest].fixed = true;
      for (int i = 0; i < kfs.back().lm_count; i++)
        if (bag.addLMObservationSlidingWindow(kfs.back().lm_id[i], kfs.back().lm_3d[i]))
          graph.lms[kfs.back().lm_id[i]] = kfs.back().lm_3d[i];
      for (int i = 0; i < kfs.back().lm_count; i++) {
        int64_t lm = kfs.back().lm_id[i];
        // optimizer.vertex(lm) can be null in the reference when the bag and the graph disagree (quirk A22);
        // g2o's setVertex(nullptr) + addEdge then fails and the edge is not added.
        if (graph.lms.find(lm) != graph.lms.end())
          graph.edges[edge_id] = BAGraph::Edge{edge_id, lm, bag.newest, kfs.back().lm_2d[i]};
        edge_id++;
      }
      state = OPTIMIZING;
      break;
    }
    default:
      break;
  }
  bool produced = false;
  if (state == OPTIMIZING) {
    out = CorrectionInfStruct();
    rebuild_imu_edges();
    graph.optimize(12);
    std::vector<int64_t> ids;
    for (auto& kv : graph.edges) ids.push_back(kv.first);
    int outlier_cnt = 0;
    for (int i = (int)ids.size() - 1; i >= 0; i--) {
      BAGraph::Edge& e = graph.edges[ids[i]];
      double er[2];
      edge_error(graph.poses[e.pose].est, graph.lms[e.lm], e.z, graph.K, er);
      if (er[0] * er[0] + er[1] * er[1] > 3.0) {
        out.lm_outlier_id.push_back(e.lm);
        outlier_cnt++;
        graph.edges.erase(ids[i]);
      }
    }
    out.lm_outlier_count = outlier_cnt;
    graph.optimize(8);
    out.frame_id = kfs.back().frame_id;
    const SE3& Tn = graph.poses[bag.newest].est;
    out.T_c_w = se3_from_mat(quat_to_mat(Tn.q), Tn.t);
    for (auto& lm : bag.lm_sub_bag)
      if (lm.count >= 4) {  // getMultiViewLMs(lms, 4)
        out.lm_id.push_back(lm.id);
        auto it = graph.lms.find(lm.id);
        out.lm_3d.push_back(it != graph.lms.end() ? it->second : lm.p3d_w);
      }
    out.lm_count = (int)out.lm_id.size();
    state = SLIDING_WINDOW;
    produced = true;
  }
  kfs.pop_front();
  return produced;
}

}  // namespace ref

// ------------------------------------------------------------------------------------------ C entry points (ctypes)
extern "C" {
void* ref_localmap_create(int window, const double* K4) { return new ref::LocalMap(window, K4[0], K4[1], K4[2], K4[3]); }
void ref_localmap_destroy(void* h) { delete (ref::LocalMap*)h; }
// optional IMU rotation factor (off by default): q_c_b = rotation IMU body -> camera as (w, x, y, z)
void ref_localmap_set_imu_factor(void* h, int on, double sigma_g, const double* q_c_b_wxyz) {
  ((ref::LocalMap*)h)->set_imu_factor(on != 0, sigma_g, ref::Quat{q_c_b_wxyz[0], q_c_b_wxyz[1], q_c_b_wxyz[2], q_c_b_wxyz[3]});
}
void ref_localmap_set_imu_factor_pos(void* h, double sigma_a, const double* t_c_b3) {
  ((ref::LocalMap*)h)->set_imu_factor_pos(sigma_a, ref::Vec3{t_c_b3[0], t_c_b3[1], t_c_b3[2]});
}
static ref::Quat g_next_imu_dq = ref::quat_identity();
static double g_next_imu_dt = 0;
static bool g_next_imu_valid = false;
static ref::Vec3 g_next_imu_dp{0, 0, 0}, g_next_imu_va{0, 0, 0};
// the position preintegration that comes with the NEXT ref_localmap_push (after ref_localmap_next_imu): dp in the body frame of the
// previous keyframe, va = that keyframe's body velocity in the world frame
void ref_localmap_next_imu_pos(const double* dp3, const double* va3) {
  g_next_imu_dp = ref::Vec3{dp3[0], dp3[1], dp3[2]};
  g_next_imu_va = ref::Vec3{va3[0], va3[1], va3[2]};
}
// residual and Jacobians (3 x 6 each) of the position rows of one IMU edge (tests: central differences)
void ref_imu_edge_linearize_pos(const double* Ta7, const double* Tb7, const double* q_c_b_wxyz, const double* t_c_b3, const double* dp3,
                                const double* va3, double dt, double* r3, double* Ja18, double* Jb18) {
  auto se3 = [](const double* p) { return ref::SE3{{p[6], p[3], p[4], p[5]}, {p[0], p[1], p[2]}}; };
  double Ja[3][6], Jb[3][6];
  ref::imu_edge_linearize_pos(se3(Ta7), se3(Tb7), ref::Quat{q_c_b_wxyz[0], q_c_b_wxyz[1], q_c_b_wxyz[2], q_c_b_wxyz[3]},
                              ref::Vec3{t_c_b3[0], t_c_b3[1], t_c_b3[2]}, ref::Vec3{dp3[0], dp3[1], dp3[2]},
                              ref::Vec3{va3[0], va3[1], va3[2]}, dt, r3, Ja, Jb);
  for (int i = 0; i < 18; i++) {
    Ja18[i] = Ja[i / 6][i % 6];
    Jb18[i] = Jb[i / 6][i % 6];
  }
}
// the gyro preintegration that comes with the NEXT ref_localmap_push: dq (w, x, y, z) = R_b(previous keyframe)^T R_b(this), dt
void ref_localmap_next_imu(const double* dq_wxyz, double dt) {
  g_next_imu_dq = ref::Quat{dq_wxyz[0], dq_wxyz[1], dq_wxyz[2], dq_wxyz[3]};
  g_next_imu_dt = dt;
  g_next_imu_valid = true;
}
// residual and Jacobians of one IMU rotation edge (tests: central differences)
void ref_imu_edge_linearize(const double* Ta7, const double* Tb7, const double* q_c_b_wxyz, const double* dq_wxyz, double* r3, double* Ja9,
                            double* Jb9) {
  auto se3 = [](const double* p) { return ref::SE3{{p[6], p[3], p[4], p[5]}, {p[0], p[1], p[2]}}; };
  double Ja[3][3], Jb[3][3];
  ref::imu_edge_linearize(se3(Ta7), se3(Tb7), ref::Quat{q_c_b_wxyz[0], q_c_b_wxyz[1], q_c_b_wxyz[2], q_c_b_wxyz[3]},
                          ref::Quat{dq_wxyz[0], dq_wxyz[1], dq_wxyz[2], dq_wxyz[3]}, r3, Ja, Jb);
  for (int i = 0; i < 9; i++) {
    Ja9[i] = Ja[i / 3][i % 3];
    Jb9[i] = Jb[i / 3][i % 3];
  }
}
// T <- exp(dx) T with g2o's SE3Quat update (tests)
void ref_g2o_oplus(const double* T7, const double* dx6, double* out7) {
  ref::SE3 T{{T7[6], T7[3], T7[4], T7[5]}, {T7[0], T7[1], T7[2]}};
  T = ref::g2o_mul(ref::g2o_exp(dx6), T);
  const double o[7] = {T.t.x, T.t.y, T.t.z, T.q.x, T.q.y, T.q.z, T.q.w};
  memcpy(out7, o, sizeof(o));
}
// pose7 = tx ty tz qx qy qz qw.  Outputs: returns 1 if a CorrectionInf was produced.
int ref_localmap_push(void* h, int64_t frame_id, const double* pose7, int n, const int64_t* lm_id, const double* lm_2d,
                      const double* lm_3d, int64_t* out_frame_id, double* out_pose7, int* out_lm_count,
                      int64_t* out_lm_id, double* out_lm_3d, int out_cap, int* out_outlier_count,
                      int64_t* out_outlier_id, int outlier_cap) {
  ref::LocalMap* lm = (ref::LocalMap*)h;
  ref::KeyFrameStruct kf;
  if (g_next_imu_valid) {  // set by ref_localmap_next_imu for this keyframe
    kf.imu_dq = g_next_imu_dq;
    kf.imu_dt = g_next_imu_dt;
    kf.imu_valid = true;
    kf.imu_dp = g_next_imu_dp;
    kf.imu_va = g_next_imu_va;
    g_next_imu_valid = false;
    g_next_imu_dp = g_next_imu_va = ref::Vec3{0, 0, 0};
  }
  kf.frame_id = frame_id;
  kf.lm_count = n;
  kf.T_c_w = ref::SE3{{pose7[6], pose7[3], pose7[4], pose7[5]}, {pose7[0], pose7[1], pose7[2]}};
  for (int i = 0; i < n; i++) {
    kf.lm_id.push_back(lm_id[i]);
    kf.lm_2d.push_back({lm_2d[2 * i], lm_2d[2 * i + 1]});
    kf.lm_3d.push_back({lm_3d[3 * i], lm_3d[3 * i + 1], lm_3d[3 * i + 2]});
  }
  ref::CorrectionInfStruct c;
  if (!lm->frame_callback(kf, c)) return 0;
  *out_frame_id = c.frame_id;
  out_pose7[0] = c.T_c_w.t.x;
  out_pose7[1] = c.T_c_w.t.y;
  out_pose7[2] = c.T_c_w.t.z;
  out_pose7[3] = c.T_c_w.q.x;
  out_pose7[4] = c.T_c_w.q.y;
  out_pose7[5] = c.T_c_w.q.z;
  out_pose7[6] = c.T_c_w.q.w;
  *out_lm_count = c.lm_count;
  for (int i = 0; i < c.lm_count && i < out_cap; i++) {
    out_lm_id[i] = c.lm_id[i];
    out_lm_3d[3 * i] = c.lm_3d[i].x;
    out_lm_3d[3 * i + 1] = c.lm_3d[i].y;
    out_lm_3d[3 * i + 2] = c.lm_3d[i].z;
  }
  *out_outlier_count = c.lm_outlier_count;
  for (int i = 0; i < c.lm_outlier_count && i < outlier_cap; i++) out_outlier_id[i] = c.lm_outlier_id[i];
  return 1;
}
// exposes the window state for tests: all window poses (slot order) after the last optimisation
void ref_localmap_poses(void* h, double* pose7s, int* fixed, int* present) {
  ref::LocalMap* lm = (ref::LocalMap*)h;
  for (size_t i = 0; i < lm->graph.poses.size(); i++) {
    const ref::SE3& T = lm->graph.poses[i].est;
    double* p = pose7s + 7 * i;
    p[0] = T.t.x; p[1] = T.t.y; p[2] = T.t.z; p[3] = T.q.x; p[4] = T.q.y; p[5] = T.q.z; p[6] = T.q.w;
    fixed[i] = lm->graph.poses[i].fixed;
    present[i] = lm->graph.poses[i].present;
  }
}
}

// Direct BA problem entry (tests / parity of the HIP BA kernel): poses [P][7] (tx ty tz qx qy qz qw) in/out,
// fixed [P], landmarks [L][3] in/out with ids lm_ids [L], edges: e_pose [E], e_lm (index into landmarks) [E], e_uv [E][2].
// Runs optimize(it1); if cull: remove edges with chi2 > 3 (alive[] out); optimize(it2).  Returns final robust chi2.
extern "C" double ref_ba_solve(int P, double* poses7, const int* fixed, int L, double* lms3, int E, const int* e_pose,
                               const int* e_lm, const double* e_uv, const double* K4, int it1, int cull, int it2,
                               unsigned char* alive, double* chi2_trace3) {
  ref::BAGraph g;
  for (int i = 0; i < 4; i++) g.K[i] = K4[i];
  g.poses.resize(P);
  for (int i = 0; i < P; i++) {
    const double* p = poses7 + 7 * i;
    ref::SE3 T{{p[6], p[3], p[4], p[5]}, {p[0], p[1], p[2]}};
    g.poses[i] = {ref::g2o_from_mat(ref::quat_to_mat(T.q), T.t), fixed[i] != 0, true};
  }
  for (int l = 0; l < L; l++) g.lms[100 + l] = {lms3[3 * l], lms3[3 * l + 1], lms3[3 * l + 2]};
  for (int k = 0; k < E; k++) g.edges[k] = {k, 100 + e_lm[k], e_pose[k], {e_uv[2 * k], e_uv[2 * k + 1]}};
  auto chi = [&]() {
    double c = 0;
    for (auto& kv : g.edges) {
      double er[2];
      ref::edge_error(g.poses[kv.second.pose].est, g.lms[kv.second.lm], kv.second.z, g.K, er);
      c += ref::huber_rho(er[0] * er[0] + er[1] * er[1]);
    }
    return c;
  };
  if (chi2_trace3) chi2_trace3[0] = chi();
  g.optimize(it1);
  if (chi2_trace3) chi2_trace3[1] = chi();
  for (int k = 0; k < E; k++) alive[k] = 1;
  if (cull) {
    for (int k = E - 1; k >= 0; k--) {
      auto& e = g.edges[k];
      double er[2];
      ref::edge_error(g.poses[e.pose].est, g.lms[e.lm], e.z, g.K, er);
      if (er[0] * er[0] + er[1] * er[1] > 3.0) {
        alive[k] = 0;
        g.edges.erase(k);
      }
    }
  }
  g.optimize(it2);
  double c = chi();
  if (chi2_trace3) chi2_trace3[2] = c;
  for (int i = 0; i < P; i++) {
    const ref::SE3& T = g.poses[i].est;
    double* p = poses7 + 7 * i;
    p[0] = T.t.x; p[1] = T.t.y; p[2] = T.t.z; p[3] = T.q.x; p[4] = T.q.y; p[5] = T.q.z; p[6] = T.q.w;
  }
  for (int l = 0; l < L; l++) {
    ref::Vec3 v = g.lms[100 + l];
    lms3[3 * l] = v.x; lms3[3 * l + 1] = v.y; lms3[3 * l + 2] = v.z;
  }
  return c;
}
