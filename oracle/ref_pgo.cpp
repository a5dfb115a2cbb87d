// TEST INFRASTRUCTURE (CPU oracle, see oracle/README.md): restatement of the reference's pose-graph optimisation after a loop closure
// (SURVEY 8f-4):  src/backend/vo_loopclosing.cpp:742-944  loopClosureOnCovGraphG2ONew
//   graph        vertices kf_prev..kf_curr (VertexSE3, estimate T_w_c), edges i -> i+1..i+5 with the CURRENT relative poses as
//                measurements, one EdgeSE3 per recorded loop with the verified relative pose, information I, RobustKernelCauchy
//                (:779-875); fixed vertices: 0 and kf_prev unless the vertex is the later end of a loop (:786-826)
//   optimisation initializeOptimization, computeInitialGuess, optimize(100) with OptimizationAlgorithmLevenberg,
//                setUserLambdaInit(1e-10), BlockSolver_6_3 + LinearSolverCholmod on the pose blocks (:761-766,882-885)
//   write-back   T_c_w of every vertex, the drift Tw1_w2 of the last one (:889-915)
// g2o pieces (3rdPartLib/g2o/g2o):
//   types/slam3d/edge_se3.cpp:77-82              error = toVectorMQT(Z^-1 * Xi^-1 * Xj)
//   types/slam3d/isometry3d_mappings.cpp          toVectorMQT / fromVectorMQT (translation, then x y z of the unit quaternion with w >= 0)
//   types/slam3d/vertex_se3.h:105-114             oplus: X <- X * fromVectorMQT(update)
//   types/slam3d/isometry3d_gradients.h:150-222   computeEdgeSE3Gradient.  The rotation rows there go through a maxima-generated
//                                                 d(quaternion)/d(R) table; here the SAME derivative is written in quaternion algebra:
//                                                 q_e = q_a (x) conj(dq_i) (x) q_b  and  q_e (x) dq_j  (checked by central differences in
//                                                 tests/test_oracle_pgo.py)
//   core/robust_kernel_impl.cpp:91-99            Cauchy: rho = log(1 + e2), rho' = 1 / (1 + e2), delta = 1
//   core/base_binary_edge.hpp:77-142             constructQuadraticForm: first-order robust weight only
//   core/optimization_algorithm_levenberg.cpp    LM control (the same as the local map's, oracle/ref_ba.cpp)
//   core/sparse_optimizer.cpp:310-364, core/estimate_propagator.cpp:97-179   computeInitialGuess: every free vertex is re-initialised
//                                                 along a shortest-hop tree from the fixed vertices through edge measurements.  g2o breaks
//                                                 ties by POINTER order of the edges (std::set<Edge*>); here: breadth first, neighbours in
//                                                 ascending vertex index.
// Poses are kept as unit quaternion + translation (g2o: Isometry3, re-orthogonalised every 1000 updates -- never reached here).
// The linear system is block-banded (5 neighbours) plus one wide row per interior loop edge: solved by a profile (skyline) Cholesky.
// parity unpinned: no golden vectors exist for this path.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "ref_math.hpp"

namespace ref {

static inline Quat mqt_normalized(Quat q) {  // internal::normalize: unit norm and w >= 0
  q = quat_normalized(q);
  if (q.w < 0) q = {-q.w, -q.x, -q.y, -q.z};
  return q;
}
static inline SE3 iso_mul(const SE3& a, const SE3& b) { return {quat_mul(a.q, b.q), a.t + quat_rotate(a.q, b.t)}; }
static inline SE3 iso_inv(const SE3& a) {
  const Quat qi = quat_conj(a.q);
  return {qi, quat_rotate(qi, -1.0 * a.t)};
}
static inline SE3 from_mqt(const double* v) {  // fromVectorMQT
  const double n2 = v[3] * v[3] + v[4] * v[4] + v[5] * v[5];
  const double w = 1 - n2;
  Quat q = w < 0 ? quat_identity() : Quat{std::sqrt(w), v[3], v[4], v[5]};
  return {q, {v[0], v[1], v[2]}};
}

struct PgoEdge {
  int a, b;  // vertex (compact index)
  SE3 Z;     // measurement: X_a^-1 X_b
};

// EdgeSE3::computeError
static void pgo_error(const SE3& Xi, const SE3& Xj, const SE3& Z, double e[6]) {
  const SE3 E = iso_mul(iso_mul(iso_inv(Z), iso_inv(Xi)), Xj);
  const Quat q = mqt_normalized(E.q);
  e[0] = E.t.x, e[1] = E.t.y, e[2] = E.t.z, e[3] = q.x, e[4] = q.y, e[5] = q.z;
}

// EdgeSE3::linearizeOplus (computeEdgeSE3Gradient without offsets): rows = error, columns = (translation, compact quaternion)
static void pgo_linearize(const SE3& Xi, const SE3& Xj, const SE3& Z, double Ji[6][6], double Jj[6][6]) {
  const SE3 A = iso_inv(Z), B = iso_mul(iso_inv(Xi), Xj), E = iso_mul(A, B);
  const Mat3 Ra = quat_to_mat(A.q), Re = quat_to_mat(E.q);
  memset(Ji, 0, sizeof(double) * 36);
  memset(Jj, 0, sizeof(double) * 36);
  const Mat3 S = skew(B.t);  // d(R(v)^T t)/dv = 2 [t]x
  const Mat3 RaS = Ra * S;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) {
      Ji[r][c] = -Ra.m[r][c];
      Jj[r][c] = Re.m[r][c];
      Ji[r][3 + c] = 2.0 * RaS.m[r][c];
    }
  // rotation rows: q_e(v_i) = q_a (x) (1, -v_i) (x) q_b ,  q_e(v_j) = q_e (x) (1, v_j); sign of the w >= 0 normalisation
  Quat qe = quat_normalized(E.q);
  const double sgn = qe.w < 0 ? -1.0 : 1.0;
  const Quat qa = A.q, qb = B.q;
  for (int c = 0; c < 3; c++) {
    Quat u{0, 0, 0, 0};
    (&u.x)[c] = 1.0;
    const Quat di = quat_mul(quat_mul(qa, u), qb);  // d/dv_c of q_a (x) (0, v) (x) q_b
    const Quat dj = quat_mul(E.q, u);
    Ji[3][3 + c] = -sgn * di.x, Ji[4][3 + c] = -sgn * di.y, Ji[5][3 + c] = -sgn * di.z;
    Jj[3][3 + c] = sgn * dj.x, Jj[4][3 + c] = sgn * dj.y, Jj[5][3 + c] = sgn * dj.z;
  }
}

struct PoseGraph {
  std::vector<SE3> est;
  std::vector<uint8_t> fixed;
  std::vector<PgoEdge> edges;
  int iterations_done = 0;
  double chi2_initial = 0, chi2_final = 0;

  double robust_chi2() const {
    double chi = 0;
    for (const PgoEdge& e : edges) {
      double er[6];
      pgo_error(est[e.a], est[e.b], e.Z, er);
      double e2 = 0;
      for (int k = 0; k < 6; k++) e2 += er[k] * er[k];
      chi += detm::det_log(e2 + 1.0);
    }
    return chi;
  }

  // SparseOptimizer::computeInitialGuess with hop-count costs
  void initial_guess() {
    const int n = (int)est.size();
    std::vector<std::vector<int>> adj(n);  // edge indices per vertex
    for (int k = 0; k < (int)edges.size(); k++) {
      adj[edges[k].a].push_back(k);
      adj[edges[k].b].push_back(k);
    }
    std::vector<int> dist(n, -1), queue;
    for (int v = 0; v < n; v++)
      if (fixed[v]) {
        dist[v] = 0;
        queue.push_back(v);
      }
    for (size_t h = 0; h < queue.size(); h++) {
      const int u = queue[h];
      std::vector<std::pair<int, int>> nb;  // (neighbour, edge)
      for (int k : adj[u]) nb.push_back({edges[k].a == u ? edges[k].b : edges[k].a, k});
      std::sort(nb.begin(), nb.end());
      for (auto& p : nb) {
        const int z = p.first;
        if (dist[z] >= 0) continue;
        dist[z] = dist[u] + 1;
        const PgoEdge& e = edges[p.second];
        if (e.a == u)
          est[z] = iso_mul(est[u], e.Z);  // EdgeSE3::initialEstimate: to = from * Z
        else
          est[z] = iso_mul(est[u], iso_inv(e.Z));
        queue.push_back(z);
      }
    }
  }

  void optimize(int iterations, double lambda_init) {
    const int n = (int)est.size();
    std::vector<int> hidx(n, -1);
    int P = 0;
    for (int v = 0; v < n; v++)
      if (!fixed[v]) hidx[v] = P++;
    chi2_initial = chi2_final = robust_chi2();
    iterations_done = 0;
    if (P == 0 || edges.empty()) return;
    const int N = 6 * P;
    // profile of the lower triangle: first column of every block row
    std::vector<int> first(P);
    for (int i = 0; i < P; i++) first[i] = i;
    for (const PgoEdge& e : edges) {
      const int ia = hidx[e.a], ib = hidx[e.b];
      if (ia < 0 || ib < 0) continue;
      const int hi = std::max(ia, ib), lo = std::min(ia, ib);
      first[hi] = std::min(first[hi], lo);
    }
    std::vector<size_t> off(N + 1);
    std::vector<int> fcol(N);
    size_t tot = 0;
    for (int r = 0; r < N; r++) {
      fcol[r] = 6 * first[r / 6];
      off[r] = tot;
      tot += (size_t)(r - fcol[r] + 1);
    }
    off[N] = tot;
    std::vector<double> H(tot), L(tot), b(N), x(N);
    auto at = [&](std::vector<double>& M, int r, int c) -> double& { return M[off[r] + (size_t)(c - fcol[r])]; };

    double lambda = lambda_init, ni = 2;
    for (int iteration = 0; iteration < iterations; iteration++) {
      double currentChi = robust_chi2();
      std::fill(H.begin(), H.end(), 0.0);
      std::fill(b.begin(), b.end(), 0.0);
      for (const PgoEdge& e : edges) {
        double er[6], Ji[6][6], Jj[6][6];
        pgo_error(est[e.a], est[e.b], e.Z, er);
        pgo_linearize(est[e.a], est[e.b], e.Z, Ji, Jj);
        double e2 = 0;
        for (int k = 0; k < 6; k++) e2 += er[k] * er[k];
        const double w = 1.0 / (e2 + 1.0);
        const int ia = hidx[e.a], ib = hidx[e.b];
        for (int r = 0; r < 6; r++) {
          if (ia >= 0) {
            double g = 0;
            for (int k = 0; k < 6; k++) g += Ji[k][r] * (-er[k] * w);
            b[6 * ia + r] += g;
            for (int c = 0; c <= r; c++) {
              double h = 0;
              for (int k = 0; k < 6; k++) h += (Ji[k][r] * w) * Ji[k][c];
              at(H, 6 * ia + r, 6 * ia + c) += h;
            }
          }
          if (ib >= 0) {
            double g = 0;
            for (int k = 0; k < 6; k++) g += Jj[k][r] * (-er[k] * w);
            b[6 * ib + r] += g;
            for (int c = 0; c <= r; c++) {
              double h = 0;
              for (int k = 0; k < 6; k++) h += (Jj[k][r] * w) * Jj[k][c];
              at(H, 6 * ib + r, 6 * ib + c) += h;
            }
          }
        }
        if (ia >= 0 && ib >= 0) {  // off-diagonal block in the lower triangle
          for (int r = 0; r < 6; r++)
            for (int c = 0; c < 6; c++) {
              double h = 0;  // (Ji^T w Jj)(r, c)
              for (int k = 0; k < 6; k++) h += (Ji[k][r] * w) * Jj[k][c];
              if (ia > ib)
                at(H, 6 * ia + r, 6 * ib + c) += h;
              else
                at(H, 6 * ib + c, 6 * ia + r) += h;
            }
        }
      }
      double rho = 0;
      int qmax = 0;
      bool lambda_bad = false;
      do {
        const std::vector<SE3> backup = est;
        // profile Cholesky of H + lambda I, then the two triangular solves
        bool ok2 = true;
        for (int r = 0; r < N && ok2; r++) {
          for (int c = fcol[r]; c <= r; c++) {
            double s = at(H, r, c) + (r == c ? lambda : 0.0);
            for (int k = std::max(fcol[r], fcol[c]); k < c; k++) s -= at(L, r, k) * at(L, c, k);
            if (c < r) {
              at(L, r, c) = s / at(L, c, c);
            } else {
              if (!(s > 0) || !std::isfinite(s)) {
                ok2 = false;
                break;
              }
              at(L, r, r) = std::sqrt(s);
            }
          }
        }
        if (ok2) {
          for (int r = 0; r < N; r++) {
            double s = b[r];
            for (int k = fcol[r]; k < r; k++) s -= at(L, r, k) * x[k];
            x[r] = s / at(L, r, r);
          }
          for (int r = N - 1; r >= 0; r--) {
            x[r] /= at(L, r, r);
            for (int k = fcol[r]; k < r; k++) x[k] -= at(L, r, k) * x[r];
          }
          for (int v = 0; v < n; v++)
            if (hidx[v] >= 0) est[v] = iso_mul(est[v], from_mqt(&x[6 * hidx[v]]));
        }
        double tempChi = ok2 ? robust_chi2() : std::numeric_limits<double>::max();
        double scale = 0;
        if (ok2)
          for (int j = 0; j < N; j++) scale += x[j] * (lambda * x[j] + b[j]);
        scale += 1e-3;
        rho = (currentChi - tempChi) / scale;
        if (rho > 0 && std::isfinite(tempChi)) {
          double alpha = 1. - detm::det_powi(2 * rho - 1, 3);
          alpha = std::min(alpha, 2. / 3.);
          lambda *= std::max(1. / 3., alpha);
          ni = 2;
          currentChi = tempChi;
        } else {
          lambda *= ni;
          ni *= 2;
          est = backup;
          if (!std::isfinite(lambda)) {
            lambda_bad = true;
            break;
          }
        }
        qmax++;
      } while (rho < 0 && qmax < 10);
      iterations_done = iteration + 1;
      chi2_final = currentChi;
      if (qmax == 10 || rho == 0 || lambda_bad) break;
    }
    chi2_final = robust_chi2();
  }
};

}  // namespace ref

extern "C" {

static ref::SE3 se3_of(const double* p) { return {{p[6], p[3], p[4], p[5]}, {p[0], p[1], p[2]}}; }
static void se3_to(const ref::SE3& T, double* p) {
  p[0] = T.t.x, p[1] = T.t.y, p[2] = T.t.z, p[3] = T.q.x, p[4] = T.q.y, p[5] = T.q.z, p[6] = T.q.w;
}

// one edge: error (6) and the two 6x6 Jacobians (row-major); poses as (tx ty tz qx qy qz qw)
void ref_pgo_edge(const double* Xi7, const double* Xj7, const double* Z7, double* e6, double* Ji36, double* Jj36) {
  double Ji[6][6], Jj[6][6];
  ref::pgo_error(se3_of(Xi7), se3_of(Xj7), se3_of(Z7), e6);
  ref::pgo_linearize(se3_of(Xi7), se3_of(Xj7), se3_of(Z7), Ji, Jj);
  memcpy(Ji36, Ji, sizeof(Ji));
  memcpy(Jj36, Jj, sizeof(Jj));
}
void ref_pgo_oplus(const double* X7, const double* v6, double* out7) { se3_to(ref::iso_mul(se3_of(X7), ref::from_mqt(v6)), out7); }

// loopClosureOnCovGraphG2ONew: n_kf keyframes with T_c_w (pose7) and a presence flag, n_loops recorded loops (ids: earlier, later
// keyframe; pose: the verified T_later_earlier the reference stores in loop_poses).  T_c_w of the optimised keyframes is rewritten in
// place; drift7 = Tw1_w2 of the last optimised keyframe (vo_loopclosing.cpp:899-910).  stats: [0] iterations, [1] chi2 before (after
// the initial guess), [2] chi2 after, [3] vertices, [4] edges.  Returns 0 when there is nothing to optimise.
int ref_pgo_loop_closure(int n_kf, double* T_c_w7, const uint8_t* present, int n_loops, const int* loop_ids, const double* loop_pose7,
                         int iterations, int use_initial_guess, double* drift7, double* stats5) {
  if (n_loops <= 0) return 0;
  long long kf_prev = 2LL * n_kf, kf_curr = 0;
  for (int k = 0; k < n_loops; k++) {
    if (loop_ids[2 * k] < kf_prev) kf_prev = loop_ids[2 * k];
    if (loop_ids[2 * k + 1] > kf_curr) kf_curr = loop_ids[2 * k + 1];
  }
  if (kf_prev < 0 || kf_curr >= n_kf || kf_prev > kf_curr) return 0;
  ref::PoseGraph g;
  std::vector<int> vid(n_kf, -1), kf_list;
  std::vector<ref::SE3> Tcw(n_kf);
  for (int i = 0; i < n_kf; i++) Tcw[i] = se3_of(T_c_w7 + 7 * i);
  for (long long i = kf_prev; i <= kf_curr; i++) {
    if (!present[i]) continue;
    bool is_lc_j = false;
    for (int k = 0; k < n_loops; k++) {
      if (loop_ids[2 * k] == i) break;
      if (loop_ids[2 * k + 1] == i) {
        is_lc_j = true;
        break;
      }
    }
    vid[i] = (int)g.est.size();
    kf_list.push_back((int)i);
    g.est.push_back(ref::iso_inv(Tcw[i]));
    g.fixed.push_back((!is_lc_j && (i == 0 || i == kf_prev)) ? 1 : 0);
  }
  for (long long i = kf_prev; i <= kf_curr; i++)
    for (long long j = i + 1; j <= std::min(kf_curr, i + 5); j++)
      if (present[i] && present[j]) {
        const ref::SE3 sji = ref::iso_mul(Tcw[j], ref::iso_inv(Tcw[i]));
        g.edges.push_back({vid[i], vid[j], ref::iso_inv(sji)});
      }
  for (int k = 0; k < n_loops; k++) {
    const int a = loop_ids[2 * k], b = loop_ids[2 * k + 1];
    if (a < 0 || b < 0 || a >= n_kf || b >= n_kf || vid[a] < 0 || vid[b] < 0) return 0;  // (the reference dereferences a null vertex)
    g.edges.push_back({vid[a], vid[b], ref::iso_inv(se3_of(loop_pose7 + 7 * k))});
  }
  if (use_initial_guess) g.initial_guess();
  g.optimize(iterations, 1e-10);
  for (size_t v = 0; v < kf_list.size(); v++) {
    const int i = kf_list[v];
    const ref::SE3 Tw2c = g.est[v];
    if (v + 1 == kf_list.size()) se3_to(ref::iso_inv(ref::iso_mul(Tw2c, Tcw[i])), drift7);
    se3_to(ref::iso_inv(Tw2c), T_c_w7 + 7 * i);
  }
  if (stats5) {
    stats5[0] = g.iterations_done, stats5[1] = g.chi2_initial, stats5[2] = g.chi2_final, stats5[3] = (double)g.est.size(),
    stats5[4] = (double)g.edges.size();
  }
  return 1;
}
}
