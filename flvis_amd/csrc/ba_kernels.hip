// flvis_amd: batched sliding-window bundle adjustment for gfx950 (one workgroup per stream-window).
//
// Replaces the OPTIMIZING block of LocalMapNodeletClass::frame_callback (src/backend/vo_localmap.cpp:292-366) and the
// graph bookkeeping in front of it (:114-284, PoseLMBag src/backend/poselmbag.cpp), i.e. g2o's
//   SparseOptimizer::initializeOptimization/optimize    core/sparse_optimizer.cpp:208-272,366-431
//   OptimizationAlgorithmLevenberg::solve               core/optimization_algorithm_levenberg.cpp:58-175
//   BlockSolver<6,3>::buildSystem/setLambda/solve       core/block_solver.hpp:314-565   (Schur complement on the landmarks)
//   EdgeSE3ProjectXYZ + RobustKernelHuber               types/sba/types_six_dof_expmap.cpp:389-433, core/robust_kernel_impl.cpp:65-78
// as ONE kernel launch per keyframe: edge-parallel residuals/Jacobians (fp64), per-landmark 3x3 blocks and per-pose 6x6
// blocks assembled without atomics in a fixed order (landmark-major / pose-major edge lists) so results are reproducible
// run to run, Schur complement accumulated per (pose,pose) block element, reduced camera system (6P x 6P, P <= 15) factored
// by an in-LDS Cholesky, landmarks back-substituted, LM accept/reject with state backup -- all 12 + 8 iterations and the
// chi2 > 3 cull in between without leaving the kernel.  The reduced system is tiny (<= 90x90): MFMA is not the bound
// here, the critical path is the dependent LM trial chain.
#include "dev_common.hpp"
#include "dev_geom.hpp"
#include "track_kernels.hpp"

namespace flvis {

// ------------------------------------------------------------------------------------------------ bookkeeping
FD int bag_find(const WindowDev& w, long long id) {
  for (int i = 0; i < w.n_lm; i++)
    if (w.lm_id[i] == id) return i;
  return -1;
}

FD void bag_add_pose(WindowDev& w, int W, long long frame_id, const double* pose7) {  // poselmbag.cpp:110-136
  if (w.initialized) {
    w.newest = w.oldest;
    w.pose_frame_id[w.newest] = frame_id;
    for (int j = 0; j < 7; j++) w.bag_pose[w.newest][j] = pose7[j];
    w.oldest++;
    if (w.oldest == W) w.oldest = 0;
  } else {
    w.pose_frame_id[w.wp_init] = frame_id;
    for (int j = 0; j < 7; j++) w.bag_pose[w.wp_init][j] = pose7[j];
    w.wp_init++;
    if (w.wp_init == W) {
      w.initialized = 1;
      w.oldest = 0;
      w.newest = W - 1;
    }
  }
}

// removes edges flagged by pred (order preserving); one wave
template <typename Pred>
__device__ inline void edges_remove_if(WindowDev& w, Pred pred) {
  const int lane = threadIdx.x;
  const int n = w.n_edge;
  int kept = 0;
  for (int base = 0; base < n; base += 64) {
    int i = base + lane;
    bool keep = i < n && !pred(i);
    long long id = 0, lm = 0;
    int ps = 0;
    double u = 0, v = 0;
    if (keep) {
      id = w.e_id[i];
      lm = w.e_lm[i];
      ps = w.e_pose[i];
      u = w.e_uv[i][0];
      v = w.e_uv[i][1];
    }
    unsigned long long b = __ballot(keep);
    __syncthreads();
    if (keep) {
      int k = kept + lane_prefix(b);
      w.e_id[k] = id;
      w.e_lm[k] = lm;
      w.e_pose[k] = ps;
      w.e_uv[k][0] = u;
      w.e_uv[k][1] = v;
    }
    kept += __popcll(b);
    __syncthreads();
  }
  if (lane == 0) w.n_edge = kept;
  __syncthreads();
}

// adds the observations of one keyframe to the bag (init: running mean, sliding: count only) and, if add_edges, the
// projection edges to pose slot `slot`.  New landmarks are appended in keyframe order.
__device__ inline void bag_add_keyframe(WindowDev& w, const KeyFrameDev& kf, bool sliding, bool add_edges, int slot) {
  const int lane = threadIdx.x;
  const int n = kf.lm_count;
  for (int base = 0; base < n; base += 64) {
    int i = base + lane;
    int found = -1;
    bool isnew = false;
    if (i < n) {
      found = bag_find(w, kf.lm_id[i]);
      isnew = found < 0;
    }
    unsigned long long b = __ballot(isnew);
    __syncthreads();
    if (i < n) {
      if (isnew) {
        int k = w.n_lm + lane_prefix(b);
        if (k < BA_LMAX) {
          w.lm_id[k] = kf.lm_id[i];
          w.lm_count[k] = 1;
          for (int j = 0; j < 3; j++) {
            w.lm_p3d[k][j] = kf.lm_3d[i][j];
            w.lm_est[k][j] = kf.lm_3d[i][j];
          }
        }
      } else {
        int cnt = w.lm_count[found];
        if (!sliding) {  // PoseLMBag::addLMObservation: running mean (poselmbag.cpp:69-91)
          for (int j = 0; j < 3; j++) {
            double pj = (double)cnt * w.lm_p3d[found][j] + kf.lm_3d[i][j];
            w.lm_p3d[found][j] = (1.0 / (double)(cnt + 1)) * pj;
          }
        }
        w.lm_count[found] = cnt + 1;
      }
    }
    __syncthreads();
    if (lane == 0) {
      int nn = w.n_lm + __popcll(b);
      if (nn > BA_LMAX) {
        nn = BA_LMAX;
        w.overflow = 1;
      }
      w.n_lm = nn;
    }
    __syncthreads();
  }
  if (add_edges) {
    const int e0 = w.n_edge;
    for (int i = lane; i < n; i += 64) {
      int k = e0 + i;
      if (k < BA_EMAX) {
        w.e_id[k] = w.edge_next_id + i;
        w.e_lm[k] = kf.lm_id[i];
        w.e_pose[k] = slot;
        w.e_uv[k][0] = kf.lm_2d[i][0];
        w.e_uv[k][1] = kf.lm_2d[i][1];
      }
    }
    __syncthreads();
    if (lane == 0) {
      int ne = e0 + n;
      if (ne > BA_EMAX) {
        ne = BA_EMAX;
        w.overflow = 1;
      }
      w.n_edge = ne;
      w.edge_next_id += n;
    }
    __syncthreads();
  }
}

FD void pose_to_g2o(const double* pose7, double* out7) {
  SE3d T = load_pose7(pose7);
  store_pose7(out7, g2o_from_mat(q_to_mat(T.q), T.t));
}

// LocalMapNodeletClass::frame_callback up to (not including) the optimisation; one wave per stream
__global__ __launch_bounds__(64) void k_ba_update(Pipe p) {
  const int s = blockIdx.x;
  StreamState& st = p.st[s];
  WindowDev& w = p.win[s];
  const int lane = threadIdx.x;
  if (lane == 0) w.solve = 0;
  if (!st.kf_pending) return;
  __syncthreads();
  const int W = p.cam.window;
  KeyFrameDev* ring = p.kfs_ring + (size_t)s * BA_WMAX;
  // kfs.push_back(kf)
  {
    const KeyFrameDev& src = p.kf[s];
    KeyFrameDev& dst = ring[(w.kfs_head + w.kfs_size) % W];
    const int n = src.lm_count;
    for (int i = lane; i < n; i += 64) {
      dst.lm_id[i] = src.lm_id[i];
      dst.lm_2d[i][0] = src.lm_2d[i][0];
      dst.lm_2d[i][1] = src.lm_2d[i][1];
      dst.lm_3d[i][0] = src.lm_3d[i][0];
      dst.lm_3d[i][1] = src.lm_3d[i][1];
      dst.lm_3d[i][2] = src.lm_3d[i][2];
    }
    if (lane == 0) {
      dst.frame_id = src.frame_id;
      dst.lm_count = n;
      dst.valid = 1;
      for (int j = 0; j < 7; j++) dst.T_c_w[j] = src.T_c_w[j];
      w.kfs_size++;
      st.kf_pending = 0;
      if (p.counters) atomicAdd((unsigned long long*)&p.counters[1], 1ull);
    }
  }
  __syncthreads();
  if (w.overflow) return;
  if (st.lm_state == 0) {  // UN_INITIALIZED (vo_localmap.cpp:122-216)
    if (w.kfs_size < W) return;  // returns before pop_front (quirk A22)
    for (int f = 0; f < W; f++) {
      const KeyFrameDev& kf = ring[(w.kfs_head + f) % W];
      if (lane == 0) bag_add_pose(w, W, kf.frame_id, kf.T_c_w);
      __syncthreads();
      bag_add_keyframe(w, kf, false, false, 0);
    }
    if (lane < W) {
      w.pose_present[lane] = 1;
      w.pose_fixed[lane] = (lane == w.oldest) ? 1 : 0;
      pose_to_g2o(w.bag_pose[lane], w.pose_est[lane]);
    }
    for (int i = lane; i < w.n_lm; i += 64)
      for (int j = 0; j < 3; j++) w.lm_est[i][j] = w.lm_p3d[i][j];  // vertex estimate = running mean (quirk A23)
    if (lane == 0) {
      w.n_edge = 0;
      w.edge_next_id = 0;
    }
    __syncthreads();
    for (int f = 0; f < W; f++) {
      const KeyFrameDev& kf = ring[(w.kfs_head + f) % W];
      // pose vertex id = ring slot of the frame (getPoseIdByReleventFrameId): slot f during initialisation
      const int e0 = w.n_edge;
      const int n = kf.lm_count;
      for (int i = lane; i < n; i += 64) {
        int k = e0 + i;
        if (k < BA_EMAX) {
          w.e_id[k] = w.edge_next_id + i;
          w.e_lm[k] = kf.lm_id[i];
          w.e_pose[k] = f;
          w.e_uv[k][0] = kf.lm_2d[i][0];
          w.e_uv[k][1] = kf.lm_2d[i][1];
        }
      }
      __syncthreads();
      if (lane == 0) {
        int ne = e0 + n;
        if (ne > BA_EMAX) {
          ne = BA_EMAX;
          w.overflow = 1;
        }
        w.n_edge = ne;
        w.edge_next_id += n;
      }
      __syncthreads();
    }
  } else {  // SLIDING_WINDOW (vo_localmap.cpp:218-284)
    const int old = w.oldest;
    edges_remove_if(w, [&](int i) { return w.e_pose[i] == old; });
    if (lane == 0) w.pose_present[old] = 0;
    __syncthreads();
    // for(auto id : kfs.at(0).lm_id) if(bag->removeLMObservation(id)) optimizer.removeVertex(lm)
    {
      const KeyFrameDev& k0 = ring[w.kfs_head % W];
      for (int i = lane; i < k0.lm_count; i += 64) {
        int f = bag_find(w, k0.lm_id[i]);
        if (f >= 0) w.lm_count[f]--;
      }
      __syncthreads();
      // edges of landmarks whose count reached zero disappear with the vertex
      edges_remove_if(w, [&](int i) {
        int f = bag_find(w, w.e_lm[i]);
        return f >= 0 && w.lm_count[f] == 0;
      });
      // erase those landmarks from the bag (order preserving)
      const int n = w.n_lm;
      int kept = 0;
      for (int base = 0; base < n; base += 64) {
        int i = base + lane;
        bool keep = i < n && w.lm_count[i] != 0;
        long long id = 0;
        int cnt = 0;
        double a[3], b[3];
        if (keep) {
          id = w.lm_id[i];
          cnt = w.lm_count[i];
          for (int j = 0; j < 3; j++) {
            a[j] = w.lm_p3d[i][j];
            b[j] = w.lm_est[i][j];
          }
        }
        unsigned long long bal = __ballot(keep);
        __syncthreads();
        if (keep) {
          int k = kept + lane_prefix(bal);
          w.lm_id[k] = id;
          w.lm_count[k] = cnt;
          for (int j = 0; j < 3; j++) {
            w.lm_p3d[k][j] = a[j];
            w.lm_est[k][j] = b[j];
          }
        }
        kept += __popcll(bal);
        __syncthreads();
      }
      if (lane == 0) w.n_lm = kept;
      __syncthreads();
    }
    const KeyFrameDev& kn = ring[(w.kfs_head + w.kfs_size - 1) % W];
    if (lane == 0) {
      bag_add_pose(w, W, kn.frame_id, kn.T_c_w);
      w.pose_present[w.newest] = 1;
      w.pose_fixed[w.newest] = 0;
      pose_to_g2o(kn.T_c_w, w.pose_est[w.newest]);
      w.pose_fixed[w.oldest] = 1;
    }
    __syncthreads();
    bag_add_keyframe(w, kn, true, true, w.newest);
  }
  __syncthreads();
  if (lane == 0) {
    w.solve = w.overflow ? 0 : 1;
    // kfs.pop_front() happens after the optimisation in the reference; nothing reads kfs in between
    w.kfs_head = (w.kfs_head + 1) % W;
    w.kfs_size--;
  }
}

// ------------------------------------------------------------------------------------------------ solver
constexpr int BA_T = 256;
constexpr int BA_PMAX = BA_WMAX - 1;  // free poses
constexpr int BA_NRED = 6 * BA_PMAX;  // 90

struct BAScratch {  // carved out of Pipe::ba_scratch (doubles) per stream
  double* Hll;    // [L][9]
  double* bl;     // [L][3]
  double* Dinv;   // [L][9]
  double* db;     // [L][3]
  double* lm_bak; // [L][3]
  double* Hpl;    // [E][18]
  double* BD;     // [E][18]
  double* ebuf;   // [E][30]  per-edge: AtA(9) bA(3) -> consumed landmark-major; pose parts in pbuf
  double* pbuf;   // [E][27]  per-edge: BtB upper(21) bB(6)
  int* e_p;       // [E] hessian pose index or -1
  int* e_l;       // [E] landmark index
  int* e_alive;   // [E]
  int* lm_start;  // [L+1] CSR by landmark
  int* lm_list;   // [E]
  int* lm_pose_edge;  // [L][BA_WMAX] edge of landmark at hessian pose index, or -1
  int* ps_start;  // [BA_WMAX+1] CSR by hessian pose index (free poses only)
  int* ps_list;   // [E]
};

size_t ba_scratch_doubles() {
  size_t d = 0;
  d += (size_t)BA_LMAX * (9 + 3 + 9 + 3 + 3);
  d += (size_t)BA_EMAX * (18 + 18 + 30 + 27);
  size_t ints = (size_t)BA_EMAX * 5 + (BA_LMAX + 1) + (size_t)BA_LMAX * BA_WMAX + (BA_WMAX + 1) + 64;
  d += (ints + 1) / 2;
  return d + 64;
}

FD BAScratch carve(double* base) {
  BAScratch s;
  double* q = base;
  s.Hll = q; q += (size_t)BA_LMAX * 9;
  s.bl = q; q += (size_t)BA_LMAX * 3;
  s.Dinv = q; q += (size_t)BA_LMAX * 9;
  s.db = q; q += (size_t)BA_LMAX * 3;
  s.lm_bak = q; q += (size_t)BA_LMAX * 3;
  s.Hpl = q; q += (size_t)BA_EMAX * 18;
  s.BD = q; q += (size_t)BA_EMAX * 18;
  s.ebuf = q; q += (size_t)BA_EMAX * 30;
  s.pbuf = q; q += (size_t)BA_EMAX * 27;
  int* ii = reinterpret_cast<int*>(q);
  s.e_p = ii; ii += BA_EMAX;
  s.e_l = ii; ii += BA_EMAX;
  s.e_alive = ii; ii += BA_EMAX;
  s.lm_list = ii; ii += BA_EMAX;
  s.ps_list = ii; ii += BA_EMAX;
  s.lm_start = ii; ii += BA_LMAX + 1;
  s.lm_pose_edge = ii; ii += (size_t)BA_LMAX * BA_WMAX;
  s.ps_start = ii; ii += BA_WMAX + 1;
  return s;
}

struct BAShared {
  double pose[BA_WMAX][7];      // current estimates by ring slot
  double pose_bak[BA_WMAX][7];
  double Hpp[BA_PMAX][36];
  double b[BA_NRED];            // pose part of b
  double coeff[BA_NRED];
  double x[BA_NRED];
  double Hs[BA_NRED * BA_NRED];
  double red[BA_T];
  int slot_of[BA_PMAX];         // hessian pose index -> ring slot
  int hidx_of[BA_WMAX];         // ring slot -> hessian index or -1
  int P, L, E;
  int flag;
  double scal[4];
};

__device__ inline double block_sum(double v, double* red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int o = BA_T / 2; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  double r = red[0];
  __syncthreads();
  return r;
}
__device__ inline double block_max(double v, double* red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int o = BA_T / 2; o > 0; o >>= 1) {
    if (t < o) red[t] = fmax(red[t], red[t + o]);
    __syncthreads();
  }
  double r = red[0];
  __syncthreads();
  return r;
}

FD void ba_edge_error(const BAShared& sh, const WindowDev& w, const BAScratch& sc, int e, const double* K, double* er) {
  SE3d T = load_pose7(sh.pose[w.e_pose[e]]);
  const double* lp = w.lm_est[sc.e_l[e]];
  V3 X = g2o_map(T, V3{lp[0], lp[1], lp[2]});
  er[0] = w.e_uv[e][0] - (X.x / X.z * K[0] + K[2]);
  er[1] = w.e_uv[e][1] - (X.y / X.z * K[1] + K[3]);
}

__device__ inline double ba_robust_chi2(BAShared& sh, const WindowDev& w, const BAScratch& sc, const double* K) {
  double chi = 0;
  for (int e = threadIdx.x; e < sh.E; e += BA_T) {
    if (!sc.e_alive[e]) continue;
    double er[2];
    ba_edge_error(sh, w, sc, e, K, er);
    chi += huber_rho(er[0] * er[0] + er[1] * er[1]);
  }
  return block_sum(chi, sh.red);
}

// (re)builds index maps and edge lists for the currently alive edges
__device__ inline void ba_build_structure(BAShared& sh, WindowDev& w, const BAScratch& sc, int W) {
  const int t = threadIdx.x;
  const int E = w.n_edge, L = w.n_lm;
  if (t == 0) {
    sh.E = E;
    sh.L = L;
  }
  for (int e = t; e < E; e += BA_T) sc.e_l[e] = bag_find(w, w.e_lm[e]);
  if (t < BA_WMAX) sh.hidx_of[t] = -1;
  __syncthreads();
  if (t == 0) {
    // free poses that have at least one alive edge, ordered by slot id (vertex id order, sparse_optimizer.cpp:493-498)
    int P = 0;
    for (int slot = 0; slot < W; slot++) {
      if (!w.pose_present[slot] || w.pose_fixed[slot]) continue;
      bool has = false;
      for (int e = 0; e < E && !has; e++) has = sc.e_alive[e] && w.e_pose[e] == slot;
      if (has && P < BA_PMAX) {
        sh.slot_of[P] = slot;
        sh.hidx_of[slot] = P;
        P++;
      }
    }
    sh.P = P;
  }
  __syncthreads();
  for (int e = t; e < E; e += BA_T) sc.e_p[e] = sh.hidx_of[w.e_pose[e]];
  for (int i = t; i < L * BA_WMAX; i += BA_T) sc.lm_pose_edge[i] = -1;
  __syncthreads();
  // CSR by landmark (edges ascending) and by pose, built sequentially per owner for a deterministic order
  for (int l = t; l <= L; l += BA_T) sc.lm_start[l] = 0;
  __syncthreads();
  if (t == 0) {
    for (int e = 0; e < E; e++)
      if (sc.e_alive[e] && sc.e_l[e] >= 0) sc.lm_start[sc.e_l[e] + 1]++;
    for (int l = 0; l < L; l++) sc.lm_start[l + 1] += sc.lm_start[l];
  }
  if (t == 64) {
    for (int i = 0; i <= BA_WMAX; i++) sc.ps_start[i] = 0;
    for (int e = 0; e < E; e++)
      if (sc.e_alive[e] && sc.e_p[e] >= 0) sc.ps_start[sc.e_p[e] + 1]++;
    for (int i = 0; i < BA_WMAX; i++) sc.ps_start[i + 1] += sc.ps_start[i];
  }
  __syncthreads();
  // fill: each landmark scans... cheaper: one thread per pose / per landmark range using a running cursor
  if (t == 0) {
    // landmark lists
    // cursor array reuse: lm_list filled in edge order
    for (int e = 0; e < E; e++) {
      if (!sc.e_alive[e] || sc.e_l[e] < 0) continue;
      int l = sc.e_l[e];
      // position = start + number already placed: track with lm_pose_edge as a temp counter? use Hll[l*9] as cursor
      int pos = sc.lm_start[l] + (int)sc.Hll[(size_t)l * 9];
      sc.lm_list[pos] = e;
      sc.Hll[(size_t)l * 9] += 1.0;
    }
  }
  if (t == 64) {
    int cur[BA_WMAX];
    for (int i = 0; i < BA_WMAX; i++) cur[i] = 0;
    for (int e = 0; e < E; e++) {
      if (!sc.e_alive[e] || sc.e_p[e] < 0) continue;
      int pi = sc.e_p[e];
      sc.ps_list[sc.ps_start[pi] + cur[pi]] = e;
      cur[pi]++;
    }
  }
  __syncthreads();
  for (int e = t; e < E; e += BA_T)
    if (sc.e_alive[e] && sc.e_p[e] >= 0 && sc.e_l[e] >= 0) sc.lm_pose_edge[(size_t)sc.e_l[e] * BA_WMAX + sc.e_p[e]] = e;
  __syncthreads();
}

// one g2o optimize(iterations) call
__device__ void ba_optimize(BAShared& sh, WindowDev& w, const BAScratch& sc, const double* K, int W, int iterations) {
  const int t = threadIdx.x;
  // cursor temp for the landmark list fill
  for (int l = t; l < w.n_lm; l += BA_T) sc.Hll[(size_t)l * 9] = 0.0;
  __syncthreads();
  ba_build_structure(sh, w, sc, W);
  const int P = sh.P, L = sh.L, E = sh.E;
  const int NR = 6 * P;
  int nalive = 0;
  for (int e = t; e < E; e += BA_T) nalive += sc.e_alive[e] ? 1 : 0;
  if (block_sum((double)nalive, sh.red) == 0.0) return;
  double lambda = -1, ni = 2;
  for (int iteration = 0; iteration < iterations; iteration++) {
    double currentChi = ba_robust_chi2(sh, w, sc, K);
    // ---- buildSystem: per-edge blocks
    for (int e = t; e < E; e += BA_T) {
      if (!sc.e_alive[e]) continue;
      SE3d T = load_pose7(sh.pose[w.e_pose[e]]);
      const double* lp = w.lm_est[sc.e_l[e]];
      V3 pw{lp[0], lp[1], lp[2]};
      V3 X = g2o_map(T, pw);
      double x = X.x, y = X.y, z = X.z, z2 = z * z, fx = K[0], fy = K[1];
      double er0 = w.e_uv[e][0] - (x / z * fx + K[2]), er1 = w.e_uv[e][1] - (y / z * fy + K[3]);
      M3 R = q_to_mat(T.q);
      double tmp0[3] = {fx, 0, -x / z * fx}, tmp1[3] = {0, fy, -y / z * fy};
      double Ji[2][3], Jj[2][6];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        Ji[0][c] = -1. / z * (tmp0[0] * R.m[0][c] + tmp0[1] * R.m[1][c] + tmp0[2] * R.m[2][c]);
        Ji[1][c] = -1. / z * (tmp1[0] * R.m[0][c] + tmp1[1] * R.m[1][c] + tmp1[2] * R.m[2][c]);
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
      double wgt = huber_w(er0 * er0 + er1 * er1);
      double o0 = -er0 * wgt, o1 = -er1 * wgt;
      double* eb = sc.ebuf + (size_t)e * 30;
#pragma unroll
      for (int r = 0; r < 3; r++) {
        eb[9 + r] = Ji[0][r] * o0 + Ji[1][r] * o1;
#pragma unroll
        for (int c = 0; c < 3; c++) eb[3 * r + c] = (Ji[0][r] * wgt) * Ji[0][c] + (Ji[1][r] * wgt) * Ji[1][c];
      }
      if (sc.e_p[e] >= 0) {
        double* pb = sc.pbuf + (size_t)e * 27;
        double* hpl = sc.Hpl + (size_t)e * 18;
        int q = 0;
#pragma unroll
        for (int r = 0; r < 6; r++) {
          pb[21 + r] = Jj[0][r] * o0 + Jj[1][r] * o1;
#pragma unroll
          for (int c = r; c < 6; c++) pb[q++] = (Jj[0][r] * wgt) * Jj[0][c] + (Jj[1][r] * wgt) * Jj[1][c];
#pragma unroll
          for (int c = 0; c < 3; c++) hpl[3 * r + c] = (Jj[0][r] * wgt) * Ji[0][c] + (Jj[1][r] * wgt) * Ji[1][c];
        }
      }
    }
    __syncthreads();
    // landmark-major assembly (fixed edge order)
    for (int l = t; l < L; l += BA_T) {
      double h[9], bb[3];
#pragma unroll
      for (int k = 0; k < 9; k++) h[k] = 0;
      bb[0] = bb[1] = bb[2] = 0;
      for (int k = sc.lm_start[l]; k < sc.lm_start[l + 1]; k++) {
        const double* eb = sc.ebuf + (size_t)sc.lm_list[k] * 30;
#pragma unroll
        for (int j = 0; j < 9; j++) h[j] += eb[j];
        bb[0] += eb[9];
        bb[1] += eb[10];
        bb[2] += eb[11];
      }
#pragma unroll
      for (int j = 0; j < 9; j++) sc.Hll[(size_t)l * 9 + j] = h[j];
      sc.bl[3 * l] = bb[0];
      sc.bl[3 * l + 1] = bb[1];
      sc.bl[3 * l + 2] = bb[2];
    }
    // pose-major assembly: 27 values per pose, one thread per (pose, value)
    for (int i = t; i < P * 27; i += BA_T) {
      int pi = i / 27, k = i - pi * 27;
      double acc = 0;
      for (int q = sc.ps_start[pi]; q < sc.ps_start[pi + 1]; q++) acc += sc.pbuf[(size_t)sc.ps_list[q] * 27 + k];
      if (k < 21) {
        // unpack upper-triangular index k -> (r,c)
        int r = 0, rem = k;
        while (rem >= 6 - r) {
          rem -= 6 - r;
          r++;
        }
        int c = r + rem;
        sh.Hpp[pi][6 * r + c] = acc;
        sh.Hpp[pi][6 * c + r] = acc;
      } else {
        sh.b[6 * pi + (k - 21)] = acc;
      }
    }
    __syncthreads();
    if (iteration == 0) {
      double md = 0;
      for (int i = t; i < P * 6; i += BA_T) md = fmax(md, fabs(sh.Hpp[i / 6][7 * (i % 6)]));
      for (int i = t; i < L * 3; i += BA_T) {
        int l = i / 3;
        if (sc.lm_start[l + 1] > sc.lm_start[l]) md = fmax(md, fabs(sc.Hll[(size_t)l * 9 + 4 * (i % 3)]));
      }
      md = block_max(md, sh.red);
      lambda = 1e-5 * md;
      ni = 2;
    }
    double rho = 0;
    int qmax = 0;
    bool lambda_bad = false;
    do {
      // push
      for (int i = t; i < BA_WMAX * 7; i += BA_T) (&sh.pose_bak[0][0])[i] = (&sh.pose[0][0])[i];
      for (int i = t; i < L * 3; i += BA_T) sc.lm_bak[i] = (&w.lm_est[0][0])[i];
      // Dinv, db, BD per landmark / edge
      for (int l = t; l < L; l += BA_T) {
        if (sc.lm_start[l + 1] == sc.lm_start[l]) continue;
        M3 D, Di;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) D.m[r][c] = sc.Hll[(size_t)l * 9 + 3 * r + c] + (r == c ? lambda : 0.0);
        m3_inverse(D, Di);
        V3 dbv = Di * V3{sc.bl[3 * l], sc.bl[3 * l + 1], sc.bl[3 * l + 2]};
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) sc.Dinv[(size_t)l * 9 + 3 * r + c] = Di.m[r][c];
        sc.db[3 * l] = dbv.x;
        sc.db[3 * l + 1] = dbv.y;
        sc.db[3 * l + 2] = dbv.z;
      }
      __syncthreads();
      for (int e = t; e < E; e += BA_T) {
        if (!sc.e_alive[e] || sc.e_p[e] < 0) continue;
        const double* Bi = sc.Hpl + (size_t)e * 18;
        const double* Di = sc.Dinv + (size_t)sc.e_l[e] * 9;
        double* bd = sc.BD + (size_t)e * 18;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
          for (int c = 0; c < 3; c++) bd[3 * r + c] = Bi[3 * r] * Di[c] + Bi[3 * r + 1] * Di[3 + c] + Bi[3 * r + 2] * Di[6 + c];
      }
      __syncthreads();
      // reduced system: Hs = (Hpp + lambda I) - sum_l B_i Dinv B_j^T ; coeff_i = sum B_i db
      for (int i = t; i < NR * NR; i += BA_T) {
        int row = i / NR, col = i - row * NR;
        int i1 = row / 6, r = row - 6 * i1, i2 = col / 6, c = col - 6 * i2;
        double acc = 0;
        if (i1 == i2) acc = sh.Hpp[i1][6 * r + c] + (r == c ? lambda : 0.0);
        // loop over the edges of pose i1; partner edge of the same landmark at pose i2
        double sub = 0;
        for (int q = sc.ps_start[i1]; q < sc.ps_start[i1 + 1]; q++) {
          int e1 = sc.ps_list[q];
          int e2 = sc.lm_pose_edge[(size_t)sc.e_l[e1] * BA_WMAX + i2];
          if (e2 < 0) continue;
          const double* bd = sc.BD + (size_t)e1 * 18 + 3 * r;
          const double* Bj = sc.Hpl + (size_t)e2 * 18 + 3 * c;
          sub += bd[0] * Bj[0] + bd[1] * Bj[1] + bd[2] * Bj[2];
        }
        sh.Hs[i] = acc - sub;
      }
      for (int i = t; i < NR; i += BA_T) {
        int i1 = i / 6, r = i - 6 * i1;
        double acc = 0;
        for (int q = sc.ps_start[i1]; q < sc.ps_start[i1 + 1]; q++) {
          int e1 = sc.ps_list[q];
          const double* Bi = sc.Hpl + (size_t)e1 * 18 + 3 * r;
          const double* d = sc.db + 3 * sc.e_l[e1];
          acc += Bi[0] * d[0] + Bi[1] * d[1] + Bi[2] * d[2];
        }
        sh.coeff[i] = sh.b[i] - acc;  // bschur
      }
      __syncthreads();
      // Cholesky of Hs (lower), in LDS; one column at a time
      if (t == 0) sh.flag = 1;
      __syncthreads();
      for (int j = 0; j < NR; j++) {
        if (t == 0) {
          double sdiag = sh.Hs[j * NR + j];
          if (!(sdiag > 0) || !isfinite(sdiag)) {
            sh.flag = 0;
            sh.Hs[j * NR + j] = 1.0;
          } else {
            sh.Hs[j * NR + j] = sqrt(sdiag);
          }
        }
        __syncthreads();
        double d = sh.Hs[j * NR + j];
        for (int i = j + 1 + t; i < NR; i += BA_T) sh.Hs[i * NR + j] = sh.Hs[i * NR + j] / d;
        __syncthreads();
        // trailing update of the lower triangle: A[i][k] -= L[i][j] L[k][j] for j < k <= i
        const int m = NR - j - 1;
        for (int idx = t; idx < m * m; idx += BA_T) {
          int ii = j + 1 + idx / m, kk = j + 1 + idx % m;
          if (kk <= ii) sh.Hs[ii * NR + kk] -= sh.Hs[ii * NR + j] * sh.Hs[kk * NR + j];
        }
        __syncthreads();
      }
      bool ok2 = sh.flag != 0;
      if (t == 0) {
        if (ok2) {
          for (int i = 0; i < NR; i++) {  // forward
            double v = sh.coeff[i];
            for (int k = 0; k < i; k++) v -= sh.Hs[i * NR + k] * sh.x[k];
            sh.x[i] = v / sh.Hs[i * NR + i];
          }
          for (int i = NR - 1; i >= 0; i--) {  // backward
            double v = sh.x[i];
            for (int k = i + 1; k < NR; k++) v -= sh.Hs[k * NR + i] * sh.x[k];
            sh.x[i] = v / sh.Hs[i * NR + i];
          }
        } else {
          for (int i = 0; i < NR; i++) sh.x[i] = 0;
        }
      }
      __syncthreads();
      double scale_part = 0;
      if (ok2) {
        // landmarks: xl = Dinv (bl - B^T xp); update; accumulate x.(lambda x + b)
        for (int l = t; l < L; l += BA_T) {
          if (sc.lm_start[l + 1] == sc.lm_start[l]) continue;
          double cl[3] = {sc.bl[3 * l], sc.bl[3 * l + 1], sc.bl[3 * l + 2]};
          for (int k = sc.lm_start[l]; k < sc.lm_start[l + 1]; k++) {
            int e = sc.lm_list[k];
            if (sc.e_p[e] < 0) continue;
            const double* Bi = sc.Hpl + (size_t)e * 18;
            const double* xp = sh.x + 6 * sc.e_p[e];
#pragma unroll
            for (int c = 0; c < 3; c++) {
              double s2 = 0;
#pragma unroll
              for (int r = 0; r < 6; r++) s2 += Bi[3 * r + c] * xp[r];
              cl[c] -= s2;
            }
          }
          const double* Di = sc.Dinv + (size_t)l * 9;
          double xl[3];
#pragma unroll
          for (int r = 0; r < 3; r++) xl[r] = Di[3 * r] * cl[0] + Di[3 * r + 1] * cl[1] + Di[3 * r + 2] * cl[2];
#pragma unroll
          for (int r = 0; r < 3; r++) {
            w.lm_est[l][r] += xl[r];
            scale_part += xl[r] * (lambda * xl[r] + sc.bl[3 * l + r]);
          }
        }
        if (t < P) {
          SE3d T = load_pose7(sh.pose[sh.slot_of[t]]);
          T = g2o_mul(g2o_exp(sh.x + 6 * t), T);
          store_pose7(sh.pose[sh.slot_of[t]], T);
        }
        for (int i = t; i < NR; i += BA_T) scale_part += sh.x[i] * (lambda * sh.x[i] + sh.b[i]);
      }
      __syncthreads();
      double scale = block_sum(scale_part, sh.red) + 1e-3;
      double tempChi = ba_robust_chi2(sh, w, sc, K);
      if (!ok2) tempChi = 1.7976931348623157e308;
      rho = (currentChi - tempChi) / scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3.0);
        alpha = fmin(alpha, 2. / 3.);
        double scaleFactor = fmax(1. / 3., alpha);
        lambda *= scaleFactor;
        ni = 2;
        currentChi = tempChi;
      } else {
        lambda *= ni;
        ni *= 2;
        for (int i = t; i < BA_WMAX * 7; i += BA_T) (&sh.pose[0][0])[i] = (&sh.pose_bak[0][0])[i];
        for (int i = t; i < L * 3; i += BA_T) (&w.lm_est[0][0])[i] = sc.lm_bak[i];
        __syncthreads();
        if (!isfinite(lambda)) {
          lambda_bad = true;
          break;
        }
      }
      qmax++;
    } while (rho < 0 && qmax < 10);
    if (qmax == 10 || rho == 0 || lambda_bad) break;
  }
}

__global__ __launch_bounds__(BA_T) void k_ba_solve(Pipe p) {
  const int s = blockIdx.x;
  WindowDev& w = p.win[s];
  if (!w.solve) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  BAShared& sh = *reinterpret_cast<BAShared*>(smem);
  const BAScratch sc = carve(p.ba_scratch + (size_t)s * p.ba_scratch_stride);
  const int t = threadIdx.x;
  const int W = p.cam.window;
  const double K[4] = {p.cam.fx, p.cam.fy, p.cam.cx, p.cam.cy};
  for (int i = t; i < BA_WMAX * 7; i += BA_T) (&sh.pose[0][0])[i] = (&w.pose_est[0][0])[i];
  for (int e = t; e < w.n_edge; e += BA_T) sc.e_alive[e] = 1;
  __syncthreads();
  ba_optimize(sh, w, sc, K, W, 12);
  __syncthreads();
  // chi2 > 3 cull (vo_localmap.cpp:301-317): reverse edge order => outlier ids by descending edge id
  CorrectionDev& out = p.corr[s];
  {
    const int E = w.n_edge;
    for (int e = t; e < E; e += BA_T) {
      double er[2];
      ba_edge_error(sh, w, sc, e, K, er);
      sc.e_alive[e] = (er[0] * er[0] + er[1] * er[1] > 3.0) ? 0 : 1;
    }
    __syncthreads();
    if (t == 0) {
      int oc = 0;
      for (int e = E - 1; e >= 0; e--)
        if (!sc.e_alive[e]) {
          if (oc < BA_EMAX) out.lm_outlier_id[oc] = w.e_lm[e];
          oc++;
        }
      out.lm_outlier_count = oc;
    }
    __syncthreads();
  }
  ba_optimize(sh, w, sc, K, W, 8);
  __syncthreads();
  // write back estimates, drop culled edges for good (optimizer.removeEdge)
  for (int i = t; i < BA_WMAX * 7; i += BA_T) (&w.pose_est[0][0])[i] = (&sh.pose[0][0])[i];
  __syncthreads();
  if (t < 64) {
    // compaction by the first wave (order preserving)
    const int lane = t;
    const int n = w.n_edge;
    int kept = 0;
    for (int base = 0; base < n; base += 64) {
      int i = base + lane;
      bool keep = i < n && sc.e_alive[i];
      long long id = 0, lm = 0;
      int ps = 0;
      double u = 0, v = 0;
      if (keep) {
        id = w.e_id[i];
        lm = w.e_lm[i];
        ps = w.e_pose[i];
        u = w.e_uv[i][0];
        v = w.e_uv[i][1];
      }
      unsigned long long b = __ballot(keep);
      if (keep) {
        int k = kept + lane_prefix(b);
        w.e_id[k] = id;
        w.e_lm[k] = lm;
        w.e_pose[k] = ps;
        w.e_uv[k][0] = u;
        w.e_uv[k][1] = v;
      }
      kept += __popcll(b);
    }
    if (lane == 0) w.n_edge = kept;
  }
  __syncthreads();
  // CorrectionInf: newest pose, landmarks observed >= 4 times (getMultiViewLMs(lms,4)), in bag order
  if (t == 0) {
    const KeyFrameDev& kn = p.kf[s];
    out.frame_id = kn.frame_id;
    SE3d Tn = load_pose7(w.pose_est[w.newest]);
    store_pose7(out.T_c_w, se3_from_mat(q_to_mat(Tn.q), Tn.t));
    int c = 0;
    for (int i = 0; i < w.n_lm; i++)
      if (w.lm_count[i] >= 4) {
        out.lm_id[c] = w.lm_id[i];
        out.lm_3d[c][0] = w.lm_est[i][0];
        out.lm_3d[c][1] = w.lm_est[i][1];
        out.lm_3d[c][2] = w.lm_est[i][2];
        c++;
      }
    out.lm_count = c;
    out.valid = 1;
    p.st[s].lm_state = 1;
    w.solve = 0;
    w.ba_runs++;
    if (p.counters) atomicAdd((unsigned long long*)&p.counters[2], 1ull);
  }
}

void launch_ba_update(hipStream_t st, const Pipe& p) { hipLaunchKernelGGL(k_ba_update, dim3(p.S), dim3(64), 0, st, p); }
void launch_ba_solve(hipStream_t st, const Pipe& p) {
  hipLaunchKernelGGL(k_ba_solve, dim3(p.S), dim3(BA_T), sizeof(BAShared), st, p);
}
hipError_t ba_kernels_init() {
  return hipFuncSetAttribute((const void*)k_ba_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(BAShared));
}

}  // namespace flvis
