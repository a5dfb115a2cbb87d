// flvis_amd: batched sliding-window bundle adjustment for gfx950 (one workgroup per stream-window).
//
// Replaces the OPTIMIZING block of LocalMapNodeletClass::frame_callback (src/backend/vo_localmap.cpp:292-366) and the
// graph bookkeeping in front of it (:114-284, PoseLMBag src/backend/poselmbag.cpp), i.e. g2o's
//   SparseOptimizer::initializeOptimization/optimize    core/sparse_optimizer.cpp:208-272,366-431
//   OptimizationAlgorithmLevenberg::solve               core/optimization_algorithm_levenberg.cpp:58-175
//   BlockSolver<6,3>::buildSystem/setLambda/solve       core/block_solver.hpp:314-565   (Schur complement on the landmarks)
//   EdgeSE3ProjectXYZ + RobustKernelHuber               types/sba/types_six_dof_expmap.cpp:389-433, core/robust_kernel_impl.cpp:65-78
// as ONE kernel launch per keyframe (12 + 8 LM iterations and the chi2 > 3 cull in between, all in-kernel).
//
// Layout / mapping (1024 threads = 16 waves per window):
//   * edges are processed one thread each (fp64 residual, 2x3 / 2x6 Jacobians, Huber weight);
//   * per-landmark data is DENSE by (landmark, pose): Hpl[l][p] (6x3) and B*Dinv[l][p], plus a pose bit mask, so the Schur
//     complement streams contiguous landmark chunks through LDS and every reduced-system element (i1,r,i2,c) is owned by
//     ONE thread that sums its landmarks in index order -> no atomics, bit-reproducible run to run;
//   * per-pose 6x6 blocks / rhs: one wave per free pose, lanes stride that pose's (contiguous) edge range, butterfly sums;
//   * reduced camera system (6P x 6P, P <= 15) lives in LDS and is factored by an in-LDS Cholesky.
// The reduced system is tiny (<= 90x90): MFMA is not the bound here, the critical path is the dependent LM trial chain.
#include "dev_common.hpp"
#include "dev_geom.hpp"
#include "track_kernels.hpp"

namespace flvis {

// ------------------------------------------------------------------------------------------------ bookkeeping
// landmark id -> bag index; the id list of the bag is staged in LDS (sid) by k_ba_update and kept in sync with appends
FD int bag_find(const long long* sid, int n, long long id) {
  for (int i = 0; i < n; i++)
    if (sid[i] == id) return i;
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
    int ps = 0, li = 0;
    double u = 0, v = 0;
    if (keep) {
      id = w.e_id[i];
      lm = w.e_lm[i];
      ps = w.e_pose[i];
      li = w.e_lidx[i];
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
      w.e_lidx[k] = li;
      w.e_uv[k][0] = u;
      w.e_uv[k][1] = v;
    }
    kept += __popcll(b);
    __syncthreads();
  }
  if (lane == 0) w.n_edge = kept;
  __syncthreads();
}

// adds the observations of one keyframe to the bag (init: running mean, sliding: count only) and, if slot >= 0, the
// projection edges to that pose slot.  New landmarks are appended in keyframe order.
__device__ inline void bag_add_keyframe(WindowDev& w, long long* sid, const KeyFrameDev& kf, bool sliding, int slot) {
  const int lane = threadIdx.x;
  const int n = kf.lm_count;
  const int e0 = w.n_edge;
  for (int base = 0; base < n; base += 64) {
    int i = base + lane;
    int found = -1;
    bool isnew = false;
    if (i < n) {
      found = bag_find(sid, w.n_lm, kf.lm_id[i]);
      isnew = found < 0;
    }
    unsigned long long b = __ballot(isnew);
    __syncthreads();
    if (i < n) {
      int li;
      if (isnew) {
        int k = w.n_lm + lane_prefix(b);
        li = k;
        if (k < BA_LMAX) {
          w.lm_id[k] = kf.lm_id[i];
          sid[k] = kf.lm_id[i];
          w.lm_count[k] = 1;
          for (int j = 0; j < 3; j++) {
            w.lm_p3d[k][j] = kf.lm_3d[i][j];
            w.lm_est[k][j] = kf.lm_3d[i][j];
          }
        }
      } else {
        li = found;
        int cnt = w.lm_count[found];
        if (!sliding) {  // PoseLMBag::addLMObservation: running mean (poselmbag.cpp:69-91)
          for (int j = 0; j < 3; j++) {
            double pj = (double)cnt * w.lm_p3d[found][j] + kf.lm_3d[i][j];
            w.lm_p3d[found][j] = (1.0 / (double)(cnt + 1)) * pj;
          }
        }
        w.lm_count[found] = cnt + 1;
      }
      if (slot >= 0) {
        int k = e0 + i;
        if (k < BA_EMAX) {
          w.e_id[k] = w.edge_next_id + i;
          w.e_lm[k] = kf.lm_id[i];
          w.e_pose[k] = slot;
          w.e_lidx[k] = li < BA_LMAX ? li : 0;
          w.e_uv[k][0] = kf.lm_2d[i][0];
          w.e_uv[k][1] = kf.lm_2d[i][1];
        }
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
  if (slot >= 0 && lane == 0) {
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
  if (!p.kf[s].valid) return;
  __syncthreads();
  const int W = p.cam.window;
  KeyFrameDev* ring = p.kfs_ring + (size_t)s * BA_WMAX;
  __shared__ long long sid[BA_LMAX];  // bag landmark ids (32 KB): all id lookups of this callback scan LDS, not HBM
  for (int i = lane; i < w.n_lm; i += 64) sid[i] = w.lm_id[i];
  {  // kfs.push_back(kf)
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
      if (p.counters) atomicAdd((unsigned long long*)&p.counters[1], 1ull);
    }
  }
  __syncthreads();
  if (w.overflow) return;
  if (st.lm_state == 0) {  // UN_INITIALIZED (vo_localmap.cpp:122-216)
    if (w.kfs_size < W) return;  // returns before pop_front (quirk A22)
    if (lane == 0) {
      w.n_edge = 0;
      w.edge_next_id = 0;
    }
    __syncthreads();
    for (int f = 0; f < W; f++) {
      const KeyFrameDev& kf = ring[(w.kfs_head + f) % W];
      if (lane == 0) bag_add_pose(w, W, kf.frame_id, kf.T_c_w);
      __syncthreads();
      // pose vertex id = ring slot of the frame (getPoseIdByReleventFrameId): slot f during initialisation; edge ids
      // are assigned keyframe by keyframe in the reference (after all vertices exist), same order here
      bag_add_keyframe(w, sid, kf, false, f);
    }
    if (lane < W) {
      w.pose_present[lane] = 1;
      w.pose_fixed[lane] = (lane == w.oldest) ? 1 : 0;
      pose_to_g2o(w.bag_pose[lane], w.pose_est[lane]);
    }
    for (int i = lane; i < w.n_lm; i += 64)
      for (int j = 0; j < 3; j++) w.lm_est[i][j] = w.lm_p3d[i][j];  // vertex estimate = running mean (quirk A23)
  } else {  // SLIDING_WINDOW (vo_localmap.cpp:218-284)
    const int old = w.oldest;
    edges_remove_if(w, [&](int i) { return w.e_pose[i] == old; });
    if (lane == 0) w.pose_present[old] = 0;
    __syncthreads();
    {  // for(auto id : kfs.at(0).lm_id) if(bag->removeLMObservation(id)) optimizer.removeVertex(lm)
      const KeyFrameDev& k0 = ring[w.kfs_head % W];
      for (int i = lane; i < k0.lm_count; i += 64) {
        int f = bag_find(sid, w.n_lm, k0.lm_id[i]);
        if (f >= 0) w.lm_count[f]--;
      }
      __syncthreads();
      edges_remove_if(w, [&](int i) { return w.lm_count[w.e_lidx[i]] == 0; });  // edges vanish with the vertex
      // erase those landmarks from the bag (order preserving) and remap the edges' bag indices
      const int n = w.n_lm;
      int* remap = reinterpret_cast<int*>(p.ba_scratch + (size_t)s * p.ba_scratch_stride);
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
        if (i < n) remap[i] = keep ? kept + lane_prefix(bal) : 0;
        if (keep) {
          int k = kept + lane_prefix(bal);
          w.lm_id[k] = id;
          sid[k] = id;
          w.lm_count[k] = cnt;
          for (int j = 0; j < 3; j++) {
            w.lm_p3d[k][j] = a[j];
            w.lm_est[k][j] = b[j];
          }
        }
        kept += __popcll(bal);
        __syncthreads();
      }
      for (int e = lane; e < w.n_edge; e += 64) w.e_lidx[e] = remap[w.e_lidx[e]];
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
    bag_add_keyframe(w, sid, kn, true, w.newest);
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
constexpr int BA_T = 1024;
constexpr int BA_PMAX = BA_WMAX - 1;  // free poses
constexpr int BA_NRED = 6 * BA_PMAX;  // 90
constexpr int BA_EPT = (BA_PMAX * (BA_PMAX + 1) / 2 * 36 + BA_T - 1) / BA_T;  // reduced-system elements per thread
constexpr int BA_LDS_BUDGET = 150 * 1024;

struct BAScratch {  // carved out of Pipe::ba_scratch (doubles) per stream
  double* Hll;     // [L][9]
  double* bl;      // [L][3]
  double* Dinv;    // [L][9]
  double* db;      // [L][3]
  double* lm_bak;  // [L][3]
  double* HplD;    // [L][P][18]   w B^T A, dense by (landmark, free pose)
  double* ebuf;    // [E][12]  per edge: w A^T A (9), A^T(-w e) (3)
  double* pbuf;    // [E][27]  per edge: w B^T B upper (21), B^T(-w e) (6)
  int* e_alive;    // [E]
  int* lm_edge;    // [L][BA_WMAX] edge of landmark l at ring slot, or -1
  unsigned* lmask; // [L] bit p set: landmark has an alive edge at FREE pose (hessian index) p
};

size_t ba_scratch_doubles() {
  size_t d = (size_t)BA_LMAX * (9 + 3 + 9 + 3 + 3) + (size_t)BA_LMAX * BA_PMAX * 18 + (size_t)BA_EMAX * (12 + 27);
  size_t ints = (size_t)BA_EMAX + (size_t)BA_LMAX * BA_WMAX + BA_LMAX + 64;
  return ((d + (ints + 1) / 2 + 64) + 1) & ~(size_t)1;  // even: 16-byte alignment of every stream's slice
}

FD BAScratch carve(double* base, int P) {
  BAScratch s;
  double* q = base;
  s.Hll = q; q += (size_t)BA_LMAX * 9;
  s.bl = q; q += (size_t)BA_LMAX * 3;
  s.Dinv = q; q += (size_t)BA_LMAX * 9;
  s.db = q; q += (size_t)BA_LMAX * 3;
  s.lm_bak = q; q += (size_t)BA_LMAX * 3;
  s.HplD = q; q += (size_t)BA_LMAX * BA_PMAX * 18;
  s.ebuf = q; q += (size_t)BA_EMAX * 12;
  s.pbuf = q; q += (size_t)BA_EMAX * 27;
  int* ii = reinterpret_cast<int*>(q);
  s.e_alive = ii; ii += BA_EMAX;
  s.lm_edge = ii; ii += (size_t)BA_LMAX * BA_WMAX;
  s.lmask = reinterpret_cast<unsigned*>(ii);
  (void)P;
  return s;
}

struct BAShared {
  double pose[BA_WMAX][7];  // current estimates by ring slot
  double pose_bak[BA_WMAX][7];
  double Hpp[BA_PMAX][36];
  double b[BA_NRED];
  double coeff[BA_NRED];
  double x[BA_NRED];
  double red[BA_T / 64];
  int slot_of[BA_PMAX];
  int hidx_of[BA_WMAX];
  int slot_cnt[BA_WMAX], slot_first[BA_WMAX], slot_last[BA_WMAX];
  int P, L, E, flag;
  // followed in dynamic LDS by: Hs[NR*NR], then the landmark chunk staging area
};

__device__ inline double block_sum(double v, double* red) {
  v = wave_sum_f64(v);
  const int t = threadIdx.x;
  __syncthreads();
  if ((t & 63) == 0) red[t >> 6] = v;
  __syncthreads();
  double r = 0;
#pragma unroll
  for (int i = 0; i < BA_T / 64; i++) r += red[i];
  __syncthreads();
  return r;
}
__device__ inline double block_max(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  const int t = threadIdx.x;
  __syncthreads();
  if ((t & 63) == 0) red[t >> 6] = v;
  __syncthreads();
  double r = red[0];
#pragma unroll
  for (int i = 1; i < BA_T / 64; i++) r = fmax(r, red[i]);
  __syncthreads();
  return r;
}

FD void ba_edge_error(const BAShared& sh, const WindowDev& w, int e, const double* K, double* er) {
  SE3d T = load_pose7(sh.pose[w.e_pose[e]]);
  const double* lp = w.lm_est[w.e_lidx[e]];
  V3 X = g2o_map(T, V3{lp[0], lp[1], lp[2]});
  er[0] = w.e_uv[e][0] - (X.x / X.z * K[0] + K[2]);
  er[1] = w.e_uv[e][1] - (X.y / X.z * K[1] + K[3]);
}

__device__ inline double ba_robust_chi2(BAShared& sh, const WindowDev& w, const BAScratch& sc, const double* K) {
  double chi = 0;
  for (int e = threadIdx.x; e < sh.E; e += BA_T) {
    if (!sc.e_alive[e]) continue;
    double er[2];
    ba_edge_error(sh, w, e, K, er);
    chi += huber_rho(er[0] * er[0] + er[1] * er[1]);
  }
  return block_sum(chi, sh.red);
}

// index maps for the currently alive edges: free poses with edges (hessian order = slot order), dense landmark tables
__device__ inline void ba_build_structure(BAShared& sh, WindowDev& w, const BAScratch& sc, int W) {
  const int t = threadIdx.x;
  const int E = w.n_edge, L = w.n_lm;
  if (t == 0) {
    sh.E = E;
    sh.L = L;
  }
  if (t < BA_WMAX) {
    sh.hidx_of[t] = -1;
    sh.slot_cnt[t] = 0;
    sh.slot_first[t] = 0x7fffffff;
    sh.slot_last[t] = -1;
  }
  for (int i = t; i < L * BA_WMAX; i += BA_T) sc.lm_edge[i] = -1;
  for (int i = t; i < L; i += BA_T) sc.lmask[i] = 0u;
  __syncthreads();
  for (int e = t; e < E; e += BA_T) {
    if (!sc.e_alive[e]) continue;
    int slot = w.e_pose[e];
    atomicAdd(&sh.slot_cnt[slot], 1);
    atomicMin(&sh.slot_first[slot], e);
    atomicMax(&sh.slot_last[slot], e);
    sc.lm_edge[(size_t)w.e_lidx[e] * BA_WMAX + slot] = e;
  }
  __syncthreads();
  if (t == 0) {
    int P = 0;
    for (int slot = 0; slot < W; slot++) {
      if (!w.pose_present[slot] || w.pose_fixed[slot] || sh.slot_cnt[slot] == 0) continue;
      if (P < BA_PMAX) {
        sh.slot_of[P] = slot;
        sh.hidx_of[slot] = P;
        P++;
      }
    }
    sh.P = P;
  }
  __syncthreads();
  for (int l = t; l < L; l += BA_T) {
    unsigned m = 0;
    for (int slot = 0; slot < W; slot++)
      if (sc.lm_edge[(size_t)l * BA_WMAX + slot] >= 0) {
        int h = sh.hidx_of[slot];
        m |= (h >= 0) ? (1u << h) : (1u << 31);  // bit 31: observed by the fixed pose only -> still an active vertex
      }
    sc.lmask[l] = m;
  }
  __syncthreads();
}

// one g2o optimize(iterations) call
__device__ void ba_optimize(BAShared& sh, double* Hs, double* stage, int stage_doubles, WindowDev& w, const BAScratch& sc,
                            const double* K, int W, int iterations) {
  const int t = threadIdx.x;
  ba_build_structure(sh, w, sc, W);
  const int P = sh.P, L = sh.L, E = sh.E;
  const int NR = 6 * P;
  {
    int nalive = 0;
    for (int e = t; e < E; e += BA_T) nalive += sc.e_alive[e] ? 1 : 0;
    if (block_sum((double)nalive, sh.red) == 0.0) return;
  }
  // landmarks per LDS chunk: per landmark P*18 (Hpl) + P*18 (BD) + 3 (db) doubles + mask
  int CH = P > 0 ? stage_doubles / (P * 36 + 4) : 64;
  if (CH > 128) CH = 128;
  if (CH < 1) CH = 1;
  double* sH = stage;
  double* sB = sH + (size_t)CH * P * 18;
  double* sdb = sB + (size_t)CH * P * 18;
  unsigned* smask = reinterpret_cast<unsigned*>(sdb + (size_t)CH * 3);
  const int npairs = P * (P + 1) / 2;
  double lambda = -1, ni = 2;
  for (int iteration = 0; iteration < iterations; iteration++) {
    double currentChi = ba_robust_chi2(sh, w, sc, K);
    // ---- buildSystem: per-edge blocks
    for (int e = t; e < E; e += BA_T) {
      if (!sc.e_alive[e]) continue;
      const int slot = w.e_pose[e], l = w.e_lidx[e];
      SE3d T = load_pose7(sh.pose[slot]);
      const double* lp = w.lm_est[l];
      V3 X = g2o_map(T, V3{lp[0], lp[1], lp[2]});
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
      double* eb = sc.ebuf + (size_t)e * 12;
#pragma unroll
      for (int r = 0; r < 3; r++) {
        eb[9 + r] = Ji[0][r] * o0 + Ji[1][r] * o1;
#pragma unroll
        for (int c = 0; c < 3; c++) eb[3 * r + c] = (Ji[0][r] * wgt) * Ji[0][c] + (Ji[1][r] * wgt) * Ji[1][c];
      }
      const int h = sh.hidx_of[slot];
      if (h >= 0) {
        double* pb = sc.pbuf + (size_t)e * 27;
        double* hpl = sc.HplD + ((size_t)l * P + h) * 18;
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
    // landmark-major assembly: slots in ascending order
    for (int l = t; l < L; l += BA_T) {
      if (!sc.lmask[l]) continue;
      double h[9], bb[3];
#pragma unroll
      for (int k = 0; k < 9; k++) h[k] = 0;
      bb[0] = bb[1] = bb[2] = 0;
      for (int slot = 0; slot < W; slot++) {
        int e = sc.lm_edge[(size_t)l * BA_WMAX + slot];
        if (e < 0) continue;
        const double* eb = sc.ebuf + (size_t)e * 12;
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
    // pose-major assembly: one wave per free pose over its contiguous edge range
    {
      const int wv = t >> 6, lane = t & 63;
      for (int pi = wv; pi < P; pi += BA_T / 64) {
        const int slot = sh.slot_of[pi];
        double acc[27];
#pragma unroll
        for (int k = 0; k < 27; k++) acc[k] = 0;
        for (int e = sh.slot_first[slot] + lane; e <= sh.slot_last[slot]; e += 64) {
          if (!sc.e_alive[e] || w.e_pose[e] != slot) continue;
          const double* pb = sc.pbuf + (size_t)e * 27;
#pragma unroll
          for (int k = 0; k < 27; k++) acc[k] += pb[k];
        }
#pragma unroll
        for (int k = 0; k < 27; k++) acc[k] = wave_sum_f64(acc[k]);
        if (lane == 0) {
          int q = 0;
#pragma unroll
          for (int r = 0; r < 6; r++) {
            sh.b[6 * pi + r] = acc[21 + r];
#pragma unroll
            for (int c = r; c < 6; c++) {
              sh.Hpp[pi][6 * r + c] = acc[q];
              sh.Hpp[pi][6 * c + r] = acc[q];
              q++;
            }
          }
        }
      }
    }
    __syncthreads();
    if (iteration == 0) {
      double md = 0;
      for (int i = t; i < P * 6; i += BA_T) md = fmax(md, fabs(sh.Hpp[i / 6][7 * (i % 6)]));
      for (int i = t; i < L * 3; i += BA_T) {
        int l = i / 3;
        if (sc.lmask[l]) md = fmax(md, fabs(sc.Hll[(size_t)l * 9 + 4 * (i % 3)]));
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
      // Dinv, db, B*Dinv per landmark
      for (int l = t; l < L; l += BA_T) {
        const unsigned m = sc.lmask[l];
        if (!m) continue;
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
      // reduced system through LDS chunks: element (pair(i1<=i2), r, c) owned by one thread, landmarks in index order
      const int nelem = npairs * 36;
      double acc_e[BA_EPT];  // reduced-system elements owned by this thread
      double acc_c = 0;
#pragma unroll
      for (int k = 0; k < BA_EPT; k++) acc_e[k] = 0;
      for (int l0 = 0; l0 < L; l0 += CH) {
        const int nl = (L - l0) < CH ? (L - l0) : CH;
        // stage one (landmark, pose) 6x3 block per thread with 16-byte loads and form B*Dinv on the way into LDS
        for (int i = t; i < nl * P; i += BA_T) {
          const int ll = i / P, h = i - ll * P, l = l0 + ll;
          if (!(sc.lmask[l] & (1u << h))) continue;
          const double2* src = reinterpret_cast<const double2*>(sc.HplD + ((size_t)l * P + h) * 18);
          double Bi[18];
#pragma unroll
          for (int q = 0; q < 9; q++) {
            double2 v = src[q];
            Bi[2 * q] = v.x;
            Bi[2 * q + 1] = v.y;
          }
          const double* Di = sc.Dinv + (size_t)l * 9;
          double d[9];
#pragma unroll
          for (int q = 0; q < 9; q++) d[q] = Di[q];
          double* dh = sH + (size_t)i * 18;
          double* db2 = sB + (size_t)i * 18;
#pragma unroll
          for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
              dh[3 * r + c] = Bi[3 * r + c];
              db2[3 * r + c] = Bi[3 * r] * d[c] + Bi[3 * r + 1] * d[3 + c] + Bi[3 * r + 2] * d[6 + c];
            }
        }
        for (int i = t; i < nl * 3; i += BA_T) sdb[i] = sc.db[(size_t)l0 * 3 + i];
        for (int i = t; i < nl; i += BA_T) smask[i] = sc.lmask[l0 + i] & 0x7fffffffu;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BA_EPT; k++) {
          const int el = t + k * BA_T;
          if (el >= nelem) break;
          const int pr = el / 36, rc = el - pr * 36, r = rc / 6, c = rc - 6 * r;
          // unpack pair index pr -> (i1 <= i2)
          int i1 = 0, rem = pr;
          while (rem >= P - i1) {
            rem -= P - i1;
            i1++;
          }
          const int i2 = i1 + rem;
          const unsigned need = (1u << i1) | (1u << i2);
          double a = acc_e[k];
          for (int l = 0; l < nl; l++) {
            if ((smask[l] & need) != need) continue;
            const double* bd = sB + ((size_t)l * P + i1) * 18 + 3 * r;
            const double* Bj = sH + ((size_t)l * P + i2) * 18 + 3 * c;
            a += bd[0] * Bj[0] + bd[1] * Bj[1] + bd[2] * Bj[2];
          }
          acc_e[k] = a;
        }
        if (t < NR) {
          const int i1 = t / 6, r = t - 6 * i1;
          double a = acc_c;
          for (int l = 0; l < nl; l++) {
            if (!(smask[l] & (1u << i1))) continue;
            const double* Bi = sH + ((size_t)l * P + i1) * 18 + 3 * r;
            a += Bi[0] * sdb[3 * l] + Bi[1] * sdb[3 * l + 1] + Bi[2] * sdb[3 * l + 2];
          }
          acc_c = a;
        }
        __syncthreads();
      }
#pragma unroll
      for (int k = 0; k < BA_EPT; k++) {
        const int el = t + k * BA_T;
        if (el >= nelem) break;
        const int pr = el / 36, rc = el - pr * 36, r = rc / 6, c = rc - 6 * r;
        int i1 = 0, rem = pr;
        while (rem >= P - i1) {
          rem -= P - i1;
          i1++;
        }
        const int i2 = i1 + rem;
        double v = -acc_e[k];
        if (i1 == i2) v += sh.Hpp[i1][6 * r + c] + (r == c ? lambda : 0.0);
        Hs[(6 * i1 + r) * NR + 6 * i2 + c] = v;
        if (i1 != i2) Hs[(6 * i2 + c) * NR + 6 * i1 + r] = v;
      }
      if (t < NR) sh.coeff[t] = sh.b[t] - acc_c;  // bschur
      if (t == 0) sh.flag = 1;
      __syncthreads();
      // in-LDS left-looking Cholesky + both triangular solves by ONE wave (rows lane and lane+64): the column chain is
      // sequential anyway, so it runs without any workgroup barrier; the diagonal / solution values travel by shuffles
      if (t < 64) {
        const int lane = t, i0 = lane, i1 = lane + 64;
        bool okc = true;
        for (int j = 0; j < NR; j++) {
          double v0 = 0, v1 = 0;
          if (i0 >= j && i0 < NR) {
            double a = Hs[i0 * NR + j];
            for (int k = 0; k < j; k++) a -= Hs[i0 * NR + k] * Hs[j * NR + k];
            v0 = a;
          }
          if (i1 >= j && i1 < NR) {
            double a = Hs[i1 * NR + j];
            for (int k = 0; k < j; k++) a -= Hs[i1 * NR + k] * Hs[j * NR + k];
            v1 = a;
          }
          double vj = (j < 64) ? __shfl(v0, j, 64) : __shfl(v1, j - 64, 64);
          if (!(vj > 0) || !isfinite(vj)) {
            okc = false;
            vj = 1.0;
          }
          const double d = sqrt(vj);
          if (i0 >= j && i0 < NR) Hs[i0 * NR + j] = (i0 == j) ? d : v0 / d;
          if (i1 >= j && i1 < NR) Hs[i1 * NR + j] = (i1 == j) ? d : v1 / d;
          __threadfence_block();  // this wave's LDS writes are visible to its other lanes before the next column
        }
        double y0 = (i0 < NR) ? sh.coeff[i0] : 0.0, y1 = (i1 < NR) ? sh.coeff[i1] : 0.0;
        for (int j = 0; j < NR; j++) {  // forward substitution
          const double yj = (j < 64) ? __shfl(y0, j, 64) : __shfl(y1, j - 64, 64);
          const double xj = yj / Hs[j * NR + j];
          if (i0 == j) y0 = xj;
          if (i1 == j) y1 = xj;
          if (i0 > j && i0 < NR) y0 -= Hs[i0 * NR + j] * xj;
          if (i1 > j && i1 < NR) y1 -= Hs[i1 * NR + j] * xj;
        }
        for (int j = NR - 1; j >= 0; j--) {  // backward substitution with L^T
          const double yj = (j < 64) ? __shfl(y0, j, 64) : __shfl(y1, j - 64, 64);
          const double xj = yj / Hs[j * NR + j];
          if (i0 == j) y0 = xj;
          if (i1 == j) y1 = xj;
          if (i0 < j) y0 -= Hs[j * NR + i0] * xj;
          if (i1 < j) y1 -= Hs[j * NR + i1] * xj;
        }
        if (i0 < NR) sh.x[i0] = okc ? y0 : 0.0;
        if (i1 < NR) sh.x[i1] = okc ? y1 : 0.0;
        if (lane == 0) sh.flag = okc ? 1 : 0;
      }
      __syncthreads();
      const bool ok2 = sh.flag != 0;
      double scale_part = 0;
      if (ok2) {
        for (int l = t; l < L; l += BA_T) {
          const unsigned m = sc.lmask[l];
          if (!m) continue;
          double cl[3] = {sc.bl[3 * l], sc.bl[3 * l + 1], sc.bl[3 * l + 2]};
          for (int h = 0; h < P; h++) {
            if (!(m & (1u << h))) continue;
            const double* Bi = sc.HplD + ((size_t)l * P + h) * 18;
            const double* xp = sh.x + 6 * h;
#pragma unroll
            for (int c = 0; c < 3; c++) {
              double s2 = 0;
#pragma unroll
              for (int r = 0; r < 6; r++) s2 += Bi[3 * r + c] * xp[r];
              cl[c] -= s2;
            }
          }
          const double* Di = sc.Dinv + (size_t)l * 9;
#pragma unroll
          for (int r = 0; r < 3; r++) {
            double xl = Di[3 * r] * cl[0] + Di[3 * r + 1] * cl[1] + Di[3 * r + 2] * cl[2];
            w.lm_est[l][r] += xl;
            scale_part += xl * (lambda * xl + sc.bl[3 * l + r]);
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
      const double scale = block_sum(scale_part, sh.red) + 1e-3;
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
  const int W = p.cam.window;
  const int NRmax = 6 * (W - 1);
  double* Hs = reinterpret_cast<double*>(smem + ((sizeof(BAShared) + 15) / 16) * 16);
  double* stage = Hs + (size_t)NRmax * NRmax;
  const int stage_doubles = (int)((BA_LDS_BUDGET - ((sizeof(BAShared) + 15) / 16) * 16) / 8) - NRmax * NRmax;
  const BAScratch sc = carve(p.ba_scratch + (size_t)s * p.ba_scratch_stride, W - 1);
  const int t = threadIdx.x;
  const double K[4] = {p.cam.fx, p.cam.fy, p.cam.cx, p.cam.cy};
  for (int i = t; i < BA_WMAX * 7; i += BA_T) (&sh.pose[0][0])[i] = (&w.pose_est[0][0])[i];
  for (int e = t; e < w.n_edge; e += BA_T) sc.e_alive[e] = 1;
  __syncthreads();
  ba_optimize(sh, Hs, stage, stage_doubles, w, sc, K, W, 12);
  __syncthreads();
  // chi2 > 3 cull (vo_localmap.cpp:301-317): reverse edge order => outlier ids by descending edge id
  CorrectionDev& out = p.corr[s];
  {
    const int E = w.n_edge;
    for (int e = t; e < E; e += BA_T) {
      double er[2];
      ba_edge_error(sh, w, e, K, er);
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
  ba_optimize(sh, Hs, stage, stage_doubles, w, sc, K, W, 8);
  __syncthreads();
  for (int i = t; i < BA_WMAX * 7; i += BA_T) (&w.pose_est[0][0])[i] = (&sh.pose[0][0])[i];
  __syncthreads();
  if (t < 64) {  // optimizer.removeEdge for the culled edges: order-preserving compaction by the first wave
    const int lane = t;
    const int n = w.n_edge;
    int kept = 0;
    for (int base = 0; base < n; base += 64) {
      int i = base + lane;
      bool keep = i < n && sc.e_alive[i];
      long long id = 0, lm = 0;
      int ps = 0, li = 0;
      double u = 0, v = 0;
      if (keep) {
        id = w.e_id[i];
        lm = w.e_lm[i];
        ps = w.e_pose[i];
        li = w.e_lidx[i];
        u = w.e_uv[i][0];
        v = w.e_uv[i][1];
      }
      unsigned long long b = __ballot(keep);
      if (keep) {
        int k = kept + lane_prefix(b);
        w.e_id[k] = id;
        w.e_lm[k] = lm;
        w.e_pose[k] = ps;
        w.e_lidx[k] = li;
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
  hipLaunchKernelGGL(k_ba_solve, dim3(p.S), dim3(BA_T), BA_LDS_BUDGET, st, p);
}
hipError_t ba_kernels_init() {
  return hipFuncSetAttribute((const void*)k_ba_solve, hipFuncAttributeMaxDynamicSharedMemorySize, BA_LDS_BUDGET);
}

}  // namespace flvis
