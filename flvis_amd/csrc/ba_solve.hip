// flvis_amd: batched sliding-window bundle adjustment for gfx950 -- the optimiser (one workgroup per stream-window).
//
// Replaces the OPTIMIZING block of LocalMapNodeletClass::frame_callback (src/backend/vo_localmap.cpp:292-366), i.e. g2o's
//   SparseOptimizer::initializeOptimization/optimize    core/sparse_optimizer.cpp:208-272,366-431
//   OptimizationAlgorithmLevenberg::solve               core/optimization_algorithm_levenberg.cpp:58-175
//   BlockSolver<6,3>::buildSystem/setLambda/solve       core/block_solver.hpp:314-565   (Schur complement on the landmarks)
//   EdgeSE3ProjectXYZ + RobustKernelHuber               types/sba/types_six_dof_expmap.cpp:389-433, core/robust_kernel_impl.cpp:65-78
// as ONE kernel launch per keyframe (12 + 8 LM iterations and the chi2 > 3 cull in between, all in-kernel).
//
// Mapping (BA_T threads per window, everything landmark-major and structure-of-arrays in HBM so that thread l touches
// element l of every array -> coalesced):
//   * observation table: omask[l] (bit per ring slot), uv[slot][l], edge index [slot][l] -- rebuilt when edges change;
//   * linearisation: thread per landmark walks its <= W observations (fp64 residual, 2x3 / 2x6 Jacobians, Huber weight),
//     keeps Hll / bl in registers and stores the 6x3 blocks B = w Jp^T Jl to Bd[18][pose][l]; pose blocks Hpp / bp are
//     RE-computed from the observations by one wave per free pose (butterfly sums) instead of being staged per edge;
//   * LM trial: (Hll + lambda I) = G G^T per landmark (3x3 Cholesky in registers), Z = B G^-T and c = G^-1 bl are formed
//     on the way into LDS (double-buffered landmark chunks, one barrier per chunk, next chunk's global loads in flight
//     during the current chunk's arithmetic).  The reduced system S = Hpp + lambda I - sum_l Z Z^T is accumulated as 6x6
//     register tiles: thread = (pose pair, landmark slice), fixed slice partition + fixed butterfly order => bit-
//     reproducible run to run, no atomics.  rhs = bp - sum_l Z c likewise;
//   * the reduced camera system (6P x 6P, P <= 15) lives in LDS and is factored by ONE wave: left-looking Cholesky over
//     6x6 blocks (diagonal blocks factored + inverted in registers), block forward / backward substitution;
//   * back-substitution + trial chi2: thread per landmark, the trial landmark stays in registers between the two.
// The accepted / trial landmark sets are two SoA buffers whose roles swap on acceptance (no backup copies).
// This is latency-bound fp64 on a tiny problem (S is at most 90x90): MFMA is deliberately not used.
#include "dev_common.hpp"
#include "dev_geom.hpp"
#include "track_kernels.hpp"

namespace flvis {

constexpr int BA_T = 512;
constexpr int BA_NW = BA_T / 64;
constexpr int BA_PMAX = BA_WMAX - 1;   // free poses
constexpr int BA_NRMAX = 6 * BA_PMAX;  // 90
constexpr int BA_LDS_BUDGET = 160 * 1024;

// HBM scratch is addressed through explicit global-address-space pointers: inside the non-inlined phase functions the
// compiler then emits global_load/global_store (vmcnt only) instead of flat_* -- flat loads also tick lgkmcnt and would
// serialise the prefetched chunk loads behind every LDS wait of the Schur accumulation.
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) int gint;
typedef __attribute__((address_space(1))) unsigned guint;

struct BAScratch {  // carved out of Pipe::ba_scratch (doubles) per stream; Lc = landmark stride (multiple of 64)
  gdouble* lmA;     // [3][Lc]  accepted landmark estimates
  gdouble* lmB;     // [3][Lc]  trial estimates (roles swap on acceptance)
  gdouble* Hll;     // [6][Lc]  xx xy xz yy yz zz
  gdouble* bl;      // [3][Lc]
  gdouble* Bd;      // [18][P][Lc]  w Jp^T Jl (6x3, row-major) by free pose
  gdouble* uv;      // [W][2][Lc]
  gint* eid;        // [W][Lc]  edge index or -1
  guint* omask;     // [Lc]  bit slot: landmark has an alive edge to the pose in ring slot `slot`
  gint* e_alive;    // [E]
  int Lc;
};

size_t ba_scratch_doubles() {
  size_t d = (size_t)BA_LMAX * (3 + 3 + 6 + 3) + (size_t)BA_LMAX * BA_PMAX * 18 + (size_t)BA_LMAX * BA_WMAX * 2;
  size_t ints = (size_t)BA_LMAX * BA_WMAX + BA_LMAX + BA_EMAX + 64;
  return ((d + (ints + 1) / 2 + 64) + 1) & ~(size_t)1;  // even: 16-byte alignment of every stream's slice
}

FD BAScratch carve(double* base, int L, int W) {
  BAScratch s;
  const int Lc = ((L > 0 ? L : 1) + 63) & ~63;
  s.Lc = Lc;
  gdouble* q = (gdouble*)base;
  s.lmA = q; q += (size_t)3 * Lc;
  s.lmB = q; q += (size_t)3 * Lc;
  s.Hll = q; q += (size_t)6 * Lc;
  s.bl = q; q += (size_t)3 * Lc;
  s.Bd = q; q += (size_t)18 * (W - 1) * Lc;
  s.uv = q; q += (size_t)2 * W * Lc;
  gint* ii = (gint*)q;
  s.eid = ii; ii += (size_t)W * Lc;
  s.omask = (guint*)ii; ii += Lc;
  s.e_alive = ii;
  return s;
}

struct BAShared {
  double pose[BA_WMAX][7];   // accepted estimates by ring slot (g2o SE3Quat: t, q)
  double poseT[BA_WMAX][7];  // trial
  double RT[BA_WMAX][12];    // rotation matrix (row-major) + translation of pose / poseT
  double RTt[BA_WMAX][12];
  double Hpp[BA_PMAX][36];
  double b[BA_NRMAX];
  double x[BA_NRMAX];
  double red[BA_NW];
  int slot_of[BA_PMAX];
  int hidx_of[BA_WMAX];
  int slot_cnt[BA_WMAX];
  int P, L, E, flag, cnt, W;
  int NR, LD, off_linv, off_stage;       // reduced system geometry: Hs[NR][LD], Linv, chunk buffers (double offsets)
  int CH, bufd, npairs, slices, rs;      // landmark chunking / role partition of the Schur phase
  double K[4];
  BAScratch sc;
  long long* prof;  // optional phase timers (FLVIS_BA_PROF builds)
  long long tlast;
  // followed in dynamic LDS by: Hs[NR][NR+1], Linv[P][36], then the two landmark chunk buffers
};

// all per-window working state lives in dynamic LDS; the phase functions re-derive it from this symbol so that the
// compiler keeps LDS addressing (ds_* instructions) inside non-inlined functions
extern __shared__ __attribute__((aligned(16))) unsigned char ba_smem[];
constexpr size_t BA_SH_BYTES = ((sizeof(BAShared) + 15) / 16) * 16;
FD BAShared& ba_sh() { return *reinterpret_cast<BAShared*>(ba_smem); }
FD double* ba_dyn() { return reinterpret_cast<double*>(ba_smem + BA_SH_BYTES); }

#ifdef FLVIS_BA_PROF
#define BAPROF(i)                                                                          \
  do {                                                                                     \
    if (threadIdx.x == 0 && sh.prof) {                                                     \
      long long now_ = (long long)wall_clock64();                                          \
      atomicAdd((unsigned long long*)&sh.prof[i], (unsigned long long)(now_ - sh.tlast)); \
      sh.tlast = now_;                                                                     \
    }                                                                                      \
  } while (0)
#else
#define BAPROF(i) \
  do {            \
  } while (0)
#endif

__device__ inline double block_sum(double v, double* red) {
  v = wave_sum_f64(v);
  const int t = threadIdx.x;
  __syncthreads();
  if ((t & 63) == 0) red[t >> 6] = v;
  __syncthreads();
  double r = 0;
#pragma unroll
  for (int i = 0; i < BA_NW; i++) r += red[i];
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
  for (int i = 1; i < BA_NW; i++) r = fmax(r, red[i]);
  __syncthreads();
  return r;
}

// Orders this wave's LDS traffic in single-wave phases (no workgroup barrier).  The LDS unit executes one wave's DS
// instructions in issue order, so only the COMPILER has to be kept from reordering / caching across this point; a real
// fence would also drain vmcnt, i.e. wait for every outstanding global store of the phase.
FD void wave_lds_fence() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

FD void pose_to_rt(const double* pose7, double* rt) {
  SE3d T = load_pose7(pose7);
  M3 R = q_to_mat(T.q);
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) rt[3 * r + c] = R.m[r][c];
  rt[9] = T.t.x;
  rt[10] = T.t.y;
  rt[11] = T.t.z;
}

// EdgeSE3ProjectXYZ::computeError: squared reprojection error of landmark p in the camera rt = (R | t)
FD double ba_err2(const double* rt, double px, double py, double pz, double u, double v, const double* K) {
  const double x = rt[0] * px + rt[1] * py + rt[2] * pz + rt[9];
  const double y = rt[3] * px + rt[4] * py + rt[5] * pz + rt[10];
  const double z = rt[6] * px + rt[7] * py + rt[8] * pz + rt[11];
  const double e0 = u - (x / z * K[0] + K[2]), e1 = v - (y / z * K[1] + K[3]);
  return e0 * e0 + e1 * e1;
}

// EdgeSE3ProjectXYZ::linearizeOplus (types_six_dof_expmap.cpp:389-433): residual, Jl (2x3), Jp (2x6)
FD void ba_linearize(const double* rt, double px, double py, double pz, double u, double v, const double* K, double* er,
                     double (*Ji)[3], double (*Jj)[6]) {
  const double x = rt[0] * px + rt[1] * py + rt[2] * pz + rt[9];
  const double y = rt[3] * px + rt[4] * py + rt[5] * pz + rt[10];
  const double z = rt[6] * px + rt[7] * py + rt[8] * pz + rt[11];
  const double z2 = z * z, fx = K[0], fy = K[1];
  er[0] = u - (x / z * fx + K[2]);
  er[1] = v - (y / z * fy + K[3]);
  const double tmp0[3] = {fx, 0, -x / z * fx}, tmp1[3] = {0, fy, -y / z * fy};
  if (Ji) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      Ji[0][c] = -1. / z * (tmp0[0] * rt[c] + tmp0[1] * rt[3 + c] + tmp0[2] * rt[6 + c]);
      Ji[1][c] = -1. / z * (tmp1[0] * rt[c] + tmp1[1] * rt[3 + c] + tmp1[2] * rt[6 + c]);
    }
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

// 1/sqrt(s) for s > 0: hardware estimate + two Newton steps (~1 ulp).  Shorter dependent chain than sqrt + divide,
// which is what the serial factorisations below are bound by.
FD double rsqrt_nr(double s) {
  double r = __builtin_amdgcn_rsq(s);
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const double e = fma(-0.5 * r, s * r, 0.5);
    r = fma(r, e, r);
  }
  return r;
}

// lower Cholesky factor of the 3x3 (H + lambda I), H = xx xy xz yy yz zz; returns the factor with INVERTED diagonal
struct Chol3 {
  double i00, g10, g20, i11, g21, i22;
};
FD Chol3 chol3(const double* H, double lambda) {
  Chol3 g;
  g.i00 = rsqrt_nr(H[0] + lambda);
  g.g10 = H[1] * g.i00;
  g.g20 = H[2] * g.i00;
  g.i11 = rsqrt_nr(H[3] + lambda - g.g10 * g.g10);
  g.g21 = (H[4] - g.g20 * g.g10) * g.i11;
  g.i22 = rsqrt_nr(H[5] + lambda - g.g20 * g.g20 - g.g21 * g.g21);
  return g;
}

// sum over the 64 lanes of 32 values per lane, scattered: returns the total of value `idx` (idx as returned, < 32) in
// every lane; each exchange step halves the values a lane carries (63 shuffles instead of 32 x 6).  Fixed order.
template <int N, int O>
FD void wave_rs_step(const double (&in)[2 * N], double (&out)[N], int lane, int& base) {
  const bool up = (lane & O) != 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    const double send = up ? in[i] : in[i + N];
    const double keep = up ? in[i + N] : in[i];
    out[i] = keep + __shfl_xor(send, O, 64);
  }
  base += up ? N : 0;
}
FD double wave_reduce_scatter32(const double (&v)[32], int& idx) {
  const int lane = threadIdx.x & 63;
  int base = 0;
  double a16[16], a8[8], a4[4], a2[2], a1[1];
  wave_rs_step<16, 32>(v, a16, lane, base);
  wave_rs_step<8, 16>(a16, a8, lane, base);
  wave_rs_step<4, 8>(a8, a4, lane, base);
  wave_rs_step<2, 4>(a4, a2, lane, base);
  wave_rs_step<1, 2>(a2, a1, lane, base);
  idx = base;
  return a1[0] + __shfl_xor(a1[0], 1, 64);
}

// rebuilds the observation table from the alive edges and the free-pose numbering (hessian order = slot order)
__device__ __noinline__ void ba_build_structure(BAShared& sh, const WindowDev& w, const BAScratch& sc, int W) {
  const int t = threadIdx.x;
  const int E = w.n_edge, L = w.n_lm, Lc = sc.Lc;
  if (t == 0) {
    sh.E = E;
    sh.L = L;
    sh.cnt = 0;
  }
  if (t < BA_WMAX) {
    sh.hidx_of[t] = -1;
    sh.slot_cnt[t] = 0;
  }
  for (int i = t; i < W * Lc; i += BA_T) sc.eid[i] = -1;
  __syncthreads();
  int na = 0;
  for (int e = t; e < E; e += BA_T) {
    if (!sc.e_alive[e]) continue;
    const int slot = w.e_pose[e], l = w.e_lidx[e];
    na++;
    atomicAdd(&sh.slot_cnt[slot], 1);
    sc.eid[slot * Lc + l] = e;
    sc.uv[(size_t)(2 * slot) * Lc + l] = w.e_uv[e][0];
    sc.uv[(size_t)(2 * slot + 1) * Lc + l] = w.e_uv[e][1];
  }
  if (na) atomicAdd(&sh.cnt, na);
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
  for (int l = t; l < Lc; l += BA_T) {
    unsigned m = 0;
    if (l < L)
      for (int slot = 0; slot < W; slot++)
        if (sc.eid[slot * Lc + l] >= 0) m |= 1u << slot;
    sc.omask[l] = m;
  }
  __syncthreads();
}

// blocked (6x6) left-looking Cholesky of the lower triangle of Hs (leading dimension LD) by ONE wave, then the two
// triangular solves on sh.x; Linv receives the inverted diagonal blocks.  Returns false on a non-positive pivot.
__device__ __noinline__ bool ba_chol_solve() {
  BAShared& sh = ba_sh();
  double* Hs = ba_dyn();
  double* Linv = Hs + sh.off_linv;
  double* xs = sh.x;
  const int P = sh.P, LD = sh.LD;
  const int lane = threadIdx.x & 63;
  const int NR = 6 * P;
  bool okc = true;
  for (int jb = 0; jb < P; jb++) {
    const int c0 = 6 * jb;
    // panel rows (including the diagonal block's rows): subtract the contributions of the finished block columns
    for (int rr = lane; rr < NR; rr += 64) {
      if (rr < c0) continue;
      double a[6];
#pragma unroll
      for (int c = 0; c < 6; c++) a[c] = Hs[rr * LD + c0 + c];
      for (int kb = 0; kb < jb; kb++) {
        double lr[6];
#pragma unroll
        for (int k = 0; k < 6; k++) lr[k] = Hs[rr * LD + 6 * kb + k];
#pragma unroll
        for (int c = 0; c < 6; c++)
#pragma unroll
          for (int k = 0; k < 6; k++) a[c] = fma(-lr[k], Hs[(c0 + c) * LD + 6 * kb + k], a[c]);
      }
#pragma unroll
      for (int c = 0; c < 6; c++) Hs[rr * LD + c0 + c] = a[c];
    }
    wave_lds_fence();
    // diagonal block: every lane factors it redundantly in registers and inverts the factor
    double d[6][6], li[6][6];
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = 0; c <= r; c++) d[r][c] = Hs[(c0 + r) * LD + c0 + c];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      double s = d[j][j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= d[j][k] * d[j][k];
      if (!(s > 0) || !isfinite(s)) {
        okc = false;
        s = 1.0;
      }
      const double inv = rsqrt_nr(s), dj = s * inv;
      d[j][j] = dj;
      li[j][j] = inv;
#pragma unroll
      for (int i = j + 1; i < 6; i++) {
        double v = d[i][j];
#pragma unroll
        for (int k = 0; k < j; k++) v -= d[i][k] * d[j][k];
        d[i][j] = v * inv;
      }
    }
#pragma unroll
    for (int c = 0; c < 6; c++)  // li = d^-1 (lower): column c by forward substitution
#pragma unroll
      for (int r = c + 1; r < 6; r++) {
        double v = 0;
#pragma unroll
        for (int k = c; k < r; k++) v -= d[r][k] * li[k][c];
        li[r][c] = v * li[r][r];
      }
    wave_lds_fence();
    if (lane < 36) {
      const int r = lane / 6, c = lane - 6 * r;
      double dv = 0, lv = 0;
#pragma unroll
      for (int rr = 0; rr < 6; rr++)
#pragma unroll
        for (int cc = 0; cc <= rr; cc++)
          if (rr == r && cc == c) {
            dv = d[rr][cc];
            lv = li[rr][cc];
          }
      Linv[jb * 36 + lane] = lv;  // zero above the diagonal
      if (r >= c) Hs[(c0 + r) * LD + c0 + c] = dv;
    }
    // rows below the block: L_row = a_row * d^-T
    for (int rr = lane; rr < NR; rr += 64) {
      if (rr < c0 + 6) continue;
      double a[6], xr[6];
#pragma unroll
      for (int c = 0; c < 6; c++) a[c] = Hs[rr * LD + c0 + c];
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double v = 0;
#pragma unroll
        for (int k = 0; k <= c; k++) v = fma(a[k], li[c][k], v);
        xr[c] = v;
      }
#pragma unroll
      for (int c = 0; c < 6; c++) Hs[rr * LD + c0 + c] = xr[c];
    }
    wave_lds_fence();
  }
  // forward substitution L y = rhs (block-wise)
  for (int jb = 0; jb < P; jb++) {
    const int c0 = 6 * jb;
    double y[6];
#pragma unroll
    for (int r = 0; r < 6; r++) {
      double v = 0;
#pragma unroll
      for (int k = 0; k <= r; k++) v = fma(Linv[jb * 36 + 6 * r + k], xs[c0 + k], v);
      y[r] = v;
    }
    wave_lds_fence();
    if (lane < 6) {
      double v = 0;
#pragma unroll
      for (int r = 0; r < 6; r++)
        if (r == lane) v = y[r];
      xs[c0 + lane] = v;
    }
    for (int rr = lane; rr < NR; rr += 64) {
      if (rr < c0 + 6) continue;
      double v = xs[rr];
#pragma unroll
      for (int k = 0; k < 6; k++) v = fma(-Hs[rr * LD + c0 + k], y[k], v);
      xs[rr] = v;
    }
    wave_lds_fence();
  }
  // backward substitution L^T x = y
  for (int jb = P - 1; jb >= 0; jb--) {
    const int c0 = 6 * jb;
    double xb[6];
#pragma unroll
    for (int r = 0; r < 6; r++) {
      double v = 0;
#pragma unroll
      for (int k = r; k < 6; k++) v = fma(Linv[jb * 36 + 6 * k + r], xs[c0 + k], v);
      xb[r] = v;
    }
    wave_lds_fence();
    if (lane < 6) {
      double v = 0;
#pragma unroll
      for (int r = 0; r < 6; r++)
        if (r == lane) v = xb[r];
      xs[c0 + lane] = v;
    }
    for (int rr = lane; rr < c0; rr += 64) {
      double v = xs[rr];
#pragma unroll
      for (int k = 0; k < 6; k++) v = fma(-Hs[(c0 + k) * LD + rr], xb[k], v);
      xs[rr] = v;
    }
    wave_lds_fence();
  }
  return okc;
}

// ---- phases of one LM iteration.  Each is a separate (non-inlined) function so that its register allocation is its own:
// the phases share state only through LDS (BAShared) and the HBM scratch.

// computeActiveErrors + buildSystem in one pass over the observations (thread per landmark, waves walk the ring slots
// together): Hll / bl / the B blocks per landmark, and per free pose the 21 + 6 entries of Hpp / bp, reduced over the
// wave with a scattered butterfly and accumulated per wave in LDS (fixed order -> reproducible).  Returns this thread's
// share of the robust chi2.  ba_phase_finish_poses() folds the per-wave partials afterwards.
__device__ __noinline__ double ba_phase_linearize() {
  BAShared& sh = ba_sh();
  const BAScratch sc = sh.sc;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, L = sh.L, Lc = sc.Lc, W = sh.W, P = sh.P;
  const double K[4] = {sh.K[0], sh.K[1], sh.K[2], sh.K[3]};
  double* wacc = ba_dyn() + sh.off_stage + (size_t)wv * P * 27;  // [P][27] of this wave (chunk buffers are idle here)
  for (int i = lane; i < P * 27; i += 64) wacc[i] = 0.0;
  wave_lds_fence();
  double chi = 0;
  const size_t ks = (size_t)P * Lc;
  for (int l0 = wv * 64; l0 < L; l0 += BA_T) {
    const int l = l0 + lane;
    const unsigned m = l < L ? sc.omask[l] : 0u;
    double px = 0, py = 0, pz = 1;
    if (m) {
      px = sc.lmA[l];
      py = sc.lmA[Lc + l];
      pz = sc.lmA[2 * Lc + l];
    }
    double h[6] = {0, 0, 0, 0, 0, 0}, bb[3] = {0, 0, 0};
    for (int slot = 0; slot < W; slot++) {
      const bool has = (m >> slot) & 1u;
      if (__ballot(has) == 0ull) continue;
      const int hi = sh.hidx_of[slot];
      double pv[32];
#pragma unroll
      for (int k = 0; k < 32; k++) pv[k] = 0;
      if (has) {
        const double u = sc.uv[(size_t)(2 * slot) * Lc + l], v = sc.uv[(size_t)(2 * slot + 1) * Lc + l];
        double er[2], Ji[2][3], Jj[2][6];
        ba_linearize(sh.RT[slot], px, py, pz, u, v, K, er, Ji, Jj);
        const double e2 = er[0] * er[0] + er[1] * er[1];
        chi += huber_rho(e2);
        const double wgt = huber_w(e2);
        const double o0 = -er[0] * wgt, o1 = -er[1] * wgt;
        int q = 0;
#pragma unroll
        for (int r = 0; r < 3; r++) {
          bb[r] += Ji[0][r] * o0 + Ji[1][r] * o1;
#pragma unroll
          for (int c = r; c < 3; c++) h[q++] += (Ji[0][r] * wgt) * Ji[0][c] + (Ji[1][r] * wgt) * Ji[1][c];
        }
        if (hi >= 0) {
          gdouble* dst = sc.Bd + (size_t)hi * Lc + l;
          q = 0;
#pragma unroll
          for (int r = 0; r < 6; r++) {
            pv[21 + r] = Jj[0][r] * o0 + Jj[1][r] * o1;
#pragma unroll
            for (int c = r; c < 6; c++) pv[q++] = (Jj[0][r] * wgt) * Jj[0][c] + (Jj[1][r] * wgt) * Jj[1][c];
#pragma unroll
            for (int c = 0; c < 3; c++) dst[(3 * r + c) * ks] = (Jj[0][r] * wgt) * Ji[0][c] + (Jj[1][r] * wgt) * Ji[1][c];
          }
        }
      }
      if (hi >= 0) {
        int idx;
        const double tot = wave_reduce_scatter32(pv, idx);
        if (!(lane & 1) && idx < 27) wacc[hi * 27 + idx] += tot;
        wave_lds_fence();
      }
    }
    if (m) {
#pragma unroll
      for (int j = 0; j < 6; j++) sc.Hll[(size_t)j * Lc + l] = h[j];
#pragma unroll
      for (int j = 0; j < 3; j++) sc.bl[(size_t)j * Lc + l] = bb[j];
    }
  }
  return chi;
}

// Hpp / bp = sum of the per-wave partials in wave order (call after a barrier)
__device__ __noinline__ void ba_phase_finish_poses() {
  BAShared& sh = ba_sh();
  const int t = threadIdx.x, P = sh.P;
  const double* wacc = ba_dyn() + sh.off_stage;
  for (int i = t; i < P * 27; i += BA_T) {
    double a = 0;
#pragma unroll
    for (int wv = 0; wv < BA_NW; wv++) a += wacc[(size_t)wv * P * 27 + i];
    const int pi = i / 27, k = i - 27 * pi;
    if (k >= 21) {
      sh.b[6 * pi + (k - 21)] = a;
    } else {
      int r = 0, rem = k;
      while (rem >= 6 - r) {
        rem -= 6 - r;
        r++;
      }
      const int c = r + rem;
      sh.Hpp[pi][6 * r + c] = a;
      sh.Hpp[pi][6 * c + r] = a;
    }
  }
}

// largest diagonal entry of the (unreduced) hessian -> initial lambda (computeLambdaInit); this thread's share
__device__ __noinline__ double ba_phase_max_diag() {
  BAShared& sh = ba_sh();
  const BAScratch sc = sh.sc;
  const int t = threadIdx.x, L = sh.L, Lc = sc.Lc, P = sh.P;
  double md = 0;
  for (int i = t; i < P * 6; i += BA_T) md = fmax(md, fabs(sh.Hpp[i / 6][7 * (i % 6)]));
  for (int l = t; l < L; l += BA_T)
    if (sc.omask[l])
      md = fmax(md, fmax(fabs(sc.Hll[l]), fmax(fabs(sc.Hll[(size_t)3 * Lc + l]), fabs(sc.Hll[(size_t)5 * Lc + l]))));
  return md;
}

// reduced camera system S = Hpp + lambda I - sum_l Z Z^T (lower triangle into Hs) and rhs = bp - sum_l Z c (into sh.x),
// streaming the landmarks through double-buffered LDS chunks
__device__ __noinline__ void ba_phase_schur(double lambda) {
  BAShared& sh = ba_sh();
  const BAScratch sc = sh.sc;
  const int t = threadIdx.x, L = sh.L, Lc = sc.Lc, P = sh.P;
  const int CH = sh.CH, bufd = sh.bufd, slices = sh.slices, rs = sh.rs, npairs = sh.npairs, LD = sh.LD;
  double* Hs = ba_dyn();
  double* stage = Hs + sh.off_stage;
  // this thread's role in the accumulation
  int my_i1 = -1, my_i2 = -1, my_sl = 0, my_rp = -1, my_rsl = 0;
  if (t < npairs * slices) {
    const int pr = t / slices;
    my_sl = t - pr * slices;
    int i1 = 0, rem = pr;
    while (rem >= P - i1) {
      rem -= P - i1;
      i1++;
    }
    my_i1 = i1;
    my_i2 = i1 + rem;
  } else if (t - npairs * slices < P * rs) {
    const int tr = t - npairs * slices;
    my_rp = tr / rs;
    my_rsl = tr - my_rp * rs;
  }
  double acc[36], accr[6];
#pragma unroll
  for (int k = 0; k < 36; k++) acc[k] = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) accr[k] = 0;
  const int nchunk = (L + CH - 1) / CH;
  // staging item of this thread: (h, ll); h == P is the landmark's c vector + mask word
  const int st_h = t / CH, st_ll = t - st_h * CH;
  const bool st_on = t < CH * (P + 1);
  const int st_slot = (st_on && st_h < P) ? sh.slot_of[st_h] : 0;
  const size_t ks = (size_t)P * Lc;
  double rB[18], rH[6], rb[3];
  int rkind = 0;
  unsigned rmask = 0;
  auto prefetch = [&](int c) {
    rkind = 0;
    rmask = 0;
    if (!st_on) return;
    const int l = c * CH + st_ll;
    if (l >= L) return;
    const unsigned m = sc.omask[l];
    if (st_h < P) {
      if (!((m >> st_slot) & 1u)) return;
      rkind = 1;
      const gdouble* src = sc.Bd + (size_t)st_h * Lc + l;
#pragma unroll
      for (int k = 0; k < 18; k++) rB[k] = src[k * ks];
    } else {
      if (!m) return;
      rkind = 2;
      for (int h = 0; h < P; h++) rmask |= ((m >> sh.slot_of[h]) & 1u) << h;
#pragma unroll
      for (int k = 0; k < 3; k++) rb[k] = sc.bl[(size_t)k * Lc + l];
    }
#pragma unroll
    for (int k = 0; k < 6; k++) rH[k] = sc.Hll[(size_t)k * Lc + l];
  };
  auto commit = [&](int buf) {
    if (!st_on) return;
    double* zb = stage + (size_t)buf * bufd;
    double* cb = zb + (size_t)CH * P * 18;
    unsigned* mb = reinterpret_cast<unsigned*>(cb + (size_t)CH * 3);
    if (rkind == 0) {  // blocks of unobserved (landmark, pose) pairs are never read: the mask word gates them
      if (st_h == P) mb[st_ll] = 0u;
      return;
    }
    const Chol3 g = chol3(rH, lambda);
    if (rkind == 1) {
      double zz[18];
#pragma unroll
      for (int r = 0; r < 6; r++) {  // Z G^T = B, row by row
        zz[3 * r] = rB[3 * r] * g.i00;
        zz[3 * r + 1] = (rB[3 * r + 1] - zz[3 * r] * g.g10) * g.i11;
        zz[3 * r + 2] = (rB[3 * r + 2] - zz[3 * r] * g.g20 - zz[3 * r + 1] * g.g21) * g.i22;
      }
      double2* z = reinterpret_cast<double2*>(zb) + (size_t)st_h * 9 * CH + st_ll;  // element pairs: 16-byte LDS accesses
#pragma unroll
      for (int kp = 0; kp < 9; kp++) z[kp * CH] = double2{zz[2 * kp], zz[2 * kp + 1]};
    } else {  // c = G^-1 bl
      const double c0 = rb[0] * g.i00;
      const double c1 = (rb[1] - g.g10 * c0) * g.i11;
      const double c2 = (rb[2] - g.g20 * c0 - g.g21 * c1) * g.i22;
      cb[st_ll] = c0;
      cb[CH + st_ll] = c1;
      cb[2 * CH + st_ll] = c2;
      mb[st_ll] = rmask;
    }
  };
  prefetch(0);
  for (int c = 0; c < nchunk; c++) {
    const int buf = c & 1;
    commit(buf);
    __syncthreads();
    if (c + 1 < nchunk) prefetch(c + 1);
    const double* zb = stage + (size_t)buf * bufd;
    const double* cb = zb + (size_t)CH * P * 18;
    const unsigned* mb = reinterpret_cast<const unsigned*>(cb + (size_t)CH * 3);
    if (my_i1 >= 0) {
      const unsigned need = (1u << my_i1) | (1u << my_i2);
      const double2* zi = reinterpret_cast<const double2*>(zb) + (size_t)my_i1 * 9 * CH;
      const double2* zj = reinterpret_cast<const double2*>(zb) + (size_t)my_i2 * 9 * CH;
      for (int ll = my_sl; ll < CH; ll += slices) {
        if ((mb[ll] & need) != need) continue;
        double a[18];
#pragma unroll
        for (int kp = 0; kp < 9; kp++) {
          const double2 v2 = zi[kp * CH + ll];
          a[2 * kp] = v2.x;
          a[2 * kp + 1] = v2.y;
        }
#pragma unroll
        for (int cp = 0; cp < 3; cp++) {  // two columns of the tile (6 elements of Z_i2) per step
          const double2 q0 = zj[(3 * cp) * CH + ll], q1 = zj[(3 * cp + 1) * CH + ll], q2 = zj[(3 * cp + 2) * CH + ll];
          const double bq[6] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y};
#pragma unroll
          for (int h2 = 0; h2 < 2; h2++) {
            const int cc = 2 * cp + h2;
#pragma unroll
            for (int r = 0; r < 6; r++)
              acc[6 * r + cc] = fma(a[3 * r + 2], bq[3 * h2 + 2], fma(a[3 * r + 1], bq[3 * h2 + 1], fma(a[3 * r], bq[3 * h2], acc[6 * r + cc])));
          }
        }
      }
    } else if (my_rp >= 0) {
      const double2* zi = reinterpret_cast<const double2*>(zb) + (size_t)my_rp * 9 * CH;
      for (int ll = my_rsl; ll < CH; ll += rs) {
        if (!((mb[ll] >> my_rp) & 1u)) continue;
        const double c0 = cb[ll], c1 = cb[CH + ll], c2 = cb[2 * CH + ll];
        double a[18];
#pragma unroll
        for (int kp = 0; kp < 9; kp++) {
          const double2 v2 = zi[kp * CH + ll];
          a[2 * kp] = v2.x;
          a[2 * kp + 1] = v2.y;
        }
#pragma unroll
        for (int r = 0; r < 6; r++) accr[r] = fma(a[3 * r + 2], c2, fma(a[3 * r + 1], c1, fma(a[3 * r], c0, accr[r])));
      }
    }
  }
  // combine the slices (fixed butterfly order) and write S / rhs
  for (int off = slices >> 1; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < 36; k++) acc[k] += __shfl_xor(acc[k], off, 64);
  }
  if (my_i1 >= 0 && my_sl == 0) {
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int cc = 0; cc < 6; cc++) {
        double v = -acc[6 * r + cc];
        if (my_i1 == my_i2) v += sh.Hpp[my_i1][6 * r + cc] + (r == cc ? lambda : 0.0);
        Hs[(6 * my_i2 + cc) * LD + 6 * my_i1 + r] = v;  // lower triangle (and the full diagonal blocks)
      }
  }
  for (int off = rs >> 1; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < 6; k++) accr[k] += __shfl_xor(accr[k], off, 64);
  }
  if (my_rp >= 0 && my_rsl == 0) {
#pragma unroll
    for (int r = 0; r < 6; r++) sh.x[6 * my_rp + r] = sh.b[6 * my_rp + r] - accr[r];  // bschur
  }
}

// trial update x (+) : poses by threads < W, landmarks by back-substitution (thread per landmark) into the trial buffer;
// returns this thread's share of the gain-ratio denominator
__device__ __noinline__ double ba_phase_update(double lambda, int ok2) {
  BAShared& sh = ba_sh();
  const BAScratch sc = sh.sc;
  const int t = threadIdx.x, L = sh.L, Lc = sc.Lc, P = sh.P, W = sh.W;
  if (t < W) {
#pragma unroll
    for (int j = 0; j < 7; j++) sh.poseT[t][j] = sh.pose[t][j];
    const int hi = sh.hidx_of[t];
    if (ok2 && hi >= 0) {
      SE3d T = load_pose7(sh.pose[t]);
      T = g2o_mul(g2o_exp(sh.x + 6 * hi), T);
      store_pose7(sh.poseT[t], T);
    }
    pose_to_rt(sh.poseT[t], sh.RTt[t]);
  }
  double scale_part = 0;
  const size_t ks = (size_t)P * Lc;
  for (int l = t; l < L; l += BA_T) {
    const unsigned m = sc.omask[l];
    double p[3] = {sc.lmA[l], sc.lmA[Lc + l], sc.lmA[2 * Lc + l]};
    if (m && ok2) {
      const double bl[3] = {sc.bl[l], sc.bl[Lc + l], sc.bl[2 * Lc + l]};
      double v[3] = {bl[0], bl[1], bl[2]};
      double H[6];
#pragma unroll
      for (int k = 0; k < 6; k++) H[k] = sc.Hll[(size_t)k * Lc + l];
      for (int h = 0; h < P; h++) {
        if (!((m >> sh.slot_of[h]) & 1u)) continue;
        const gdouble* src = sc.Bd + (size_t)h * Lc + l;
        const double* xp = sh.x + 6 * h;
#pragma unroll
        for (int r = 0; r < 6; r++) {
          const double xr = xp[r];
#pragma unroll
          for (int c = 0; c < 3; c++) v[c] -= src[(3 * r + c) * ks] * xr;
        }
      }
      const Chol3 g = chol3(H, lambda);
      const double y0 = v[0] * g.i00;
      const double y1 = (v[1] - g.g10 * y0) * g.i11;
      const double y2 = (v[2] - g.g20 * y0 - g.g21 * y1) * g.i22;
      const double d2 = y2 * g.i22;
      const double d1 = (y1 - g.g21 * d2) * g.i11;
      const double d0 = (y0 - g.g10 * d1 - g.g20 * d2) * g.i00;
      p[0] += d0;
      p[1] += d1;
      p[2] += d2;
      scale_part += d0 * (lambda * d0 + bl[0]) + d1 * (lambda * d1 + bl[1]) + d2 * (lambda * d2 + bl[2]);
    }
    sc.lmB[l] = p[0];
    sc.lmB[Lc + l] = p[1];
    sc.lmB[2 * Lc + l] = p[2];
  }
  if (ok2)
    for (int i = t; i < 6 * P; i += BA_T) scale_part += sh.x[i] * (lambda * sh.x[i] + sh.b[i]);
  return scale_part;
}

// robust chi2 of the trial state (lmB, RTt); the same thread owns a landmark here and in ba_phase_update
__device__ __noinline__ double ba_phase_trial_chi2() {
  BAShared& sh = ba_sh();
  const BAScratch sc = sh.sc;
  const int t = threadIdx.x, L = sh.L, Lc = sc.Lc, W = sh.W;
  const double K[4] = {sh.K[0], sh.K[1], sh.K[2], sh.K[3]};
  double chit = 0;
  for (int l = t; l < L; l += BA_T) {
    const unsigned m = sc.omask[l];
    if (!m) continue;
    const double px = sc.lmB[l], py = sc.lmB[Lc + l], pz = sc.lmB[2 * Lc + l];
    for (int slot = 0; slot < W; slot++) {
      if (!((m >> slot) & 1u)) continue;
      const double u = sc.uv[(size_t)(2 * slot) * Lc + l], v = sc.uv[(size_t)(2 * slot + 1) * Lc + l];
      chit += huber_rho(ba_err2(sh.RTt[slot], px, py, pz, u, v, K));
    }
  }
  return chit;
}

// one g2o optimize(iterations) call
__device__ __noinline__ void ba_optimize(const WindowDev& w, int iterations) {
  BAShared& sh = ba_sh();
  const int t = threadIdx.x;
  BAPROF(0);
  ba_build_structure(sh, w, sh.sc, sh.W);
  BAPROF(1);
  if (sh.cnt == 0) return;
  if (t == 0) {
    const int P = sh.P;
    const int NR = 6 * P, LD = NR + 1;
    sh.NR = NR;
    sh.LD = LD;
    sh.off_linv = NR * LD;
    sh.off_stage = NR * LD + P * 36;
    const int stage_doubles = (int)((BA_LDS_BUDGET - BA_SH_BYTES) / 8) - sh.off_stage;
    // landmark chunk: per landmark P*18 (Z) + 3 (c) doubles + one mask word, two buffers
    int CH = (stage_doubles / 2 - 8) / (P * 18 + 4);
    if (CH > 128) CH = 128;
    CH &= ~15;
    if (CH < 16) CH = 16;
    if (CH * (P + 1) > BA_T) CH = (BA_T / (P + 1)) & ~15;
    sh.CH = CH;
    sh.bufd = CH * (P * 18 + 4);
    const int npairs = P * (P + 1) / 2;
    int slices = 64;
    while (slices > 1 && slices * npairs > BA_T) slices >>= 1;
    int rs = slices;  // <= slices keeps the rhs groups aligned to their butterfly width
    while (rs > 1 && rs * P > BA_T - slices * npairs) rs >>= 1;
    sh.npairs = npairs;
    sh.slices = slices;
    sh.rs = rs;
  }
  __syncthreads();
  double lambda = -1, ni = 2;
  for (int iteration = 0; iteration < iterations; iteration++) {
    BAPROF(0);
    const double chi = ba_phase_linearize();
    BAPROF(3);
    double currentChi = block_sum(chi, sh.red);  // (its barriers also publish the per-wave partials / Hll / bl / Bd)
    ba_phase_finish_poses();
    __syncthreads();
    BAPROF(4);
    if (iteration == 0) {
      lambda = 1e-5 * block_max(ba_phase_max_diag(), sh.red);
      ni = 2;
    }
    double rho = 0;
    int qmax = 0;
    bool lambda_bad = false;
    do {
      BAPROF(0);
      ba_phase_schur(lambda);
      __syncthreads();
      BAPROF(7);
      if (t < 64) {
        const bool okc = ba_chol_solve();
        if (t == 0) sh.flag = okc ? 1 : 0;
      }
      __syncthreads();
      BAPROF(8);
      const int ok2 = sh.flag;
      const double scale = block_sum(ba_phase_update(lambda, ok2), sh.red) + 1e-3;  // (barriers publish poseT / RTt)
      BAPROF(9);
      double tempChi = block_sum(ba_phase_trial_chi2(), sh.red);
      BAPROF(10);
#ifdef FLVIS_BA_PROF
      if (t == 0 && sh.prof) atomicAdd((unsigned long long*)&sh.prof[14], 1ull);
#endif
      if (!ok2) tempChi = 1.7976931348623157e308;
      rho = (currentChi - tempChi) / scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3.0);
        alpha = fmin(alpha, 2. / 3.);
        double scaleFactor = fmax(1. / 3., alpha);
        lambda *= scaleFactor;
        ni = 2;
        currentChi = tempChi;
        // accept: the trial buffers become the estimates
        if (t == 0) {
          gdouble* tmp = sh.sc.lmA;
          sh.sc.lmA = sh.sc.lmB;
          sh.sc.lmB = tmp;
        }
        if (t < sh.W) {
#pragma unroll
          for (int j = 0; j < 7; j++) sh.pose[t][j] = sh.poseT[t][j];
#pragma unroll
          for (int j = 0; j < 12; j++) sh.RT[t][j] = sh.RTt[t][j];
        }
        __syncthreads();
      } else {
        lambda *= ni;
        ni *= 2;
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
  BAShared& sh = ba_sh();
  const int W = p.cam.window;
  const int L = w.n_lm, E = w.n_edge;
  const int t = threadIdx.x, lane = t & 63;
  if (t == 0) {
    sh.sc = carve(p.ba_scratch + (size_t)s * p.ba_scratch_stride, L, W);
    sh.W = W;
    sh.K[0] = p.cam.fx;
    sh.K[1] = p.cam.fy;
    sh.K[2] = p.cam.cx;
    sh.K[3] = p.cam.cy;
    sh.prof = nullptr;
#ifdef FLVIS_BA_PROF
    sh.prof = p.counters ? p.counters + 8 : nullptr;
    sh.tlast = (long long)wall_clock64();
    if (sh.prof) {
      atomicAdd((unsigned long long*)&sh.prof[15], 1ull);
      atomicAdd((unsigned long long*)&sh.prof[16], (unsigned long long)E);
      atomicAdd((unsigned long long*)&sh.prof[17], (unsigned long long)L);
    }
#endif
  }
  if (t < BA_WMAX) {
#pragma unroll
    for (int j = 0; j < 7; j++) sh.pose[t][j] = t < W ? w.pose_est[t][j] : (j == 6 ? 1.0 : 0.0);
    pose_to_rt(sh.pose[t], sh.RT[t]);
  }
  __syncthreads();
  {
    const BAScratch sc = sh.sc;
    const int Lc = sc.Lc;
    for (int l = t; l < Lc; l += BA_T) {
      const bool in = l < L;
      sc.lmA[l] = in ? w.lm_est[l][0] : 0.0;
      sc.lmA[Lc + l] = in ? w.lm_est[l][1] : 0.0;
      sc.lmA[2 * Lc + l] = in ? w.lm_est[l][2] : 1.0;
    }
    for (int e = t; e < E; e += BA_T) sc.e_alive[e] = 1;
  }
  __syncthreads();
  ba_optimize(w, 12);
  __syncthreads();
  BAPROF(0);
  // chi2 > 3 cull (vo_localmap.cpp:301-317): reverse edge order => outlier ids by descending edge id
  CorrectionDev& out = p.corr[s];
  {
    const BAScratch sc = sh.sc;
    const int Lc = sc.Lc;
    const double K[4] = {sh.K[0], sh.K[1], sh.K[2], sh.K[3]};
    for (int l = t; l < L; l += BA_T) {
      const unsigned m = sc.omask[l];
      if (!m) continue;
      const double px = sc.lmA[l], py = sc.lmA[Lc + l], pz = sc.lmA[2 * Lc + l];
      for (int slot = 0; slot < W; slot++) {
        if (!((m >> slot) & 1u)) continue;
        const double u = sc.uv[(size_t)(2 * slot) * Lc + l], v = sc.uv[(size_t)(2 * slot + 1) * Lc + l];
        if (ba_err2(sh.RT[slot], px, py, pz, u, v, K) > 3.0) sc.e_alive[sc.eid[slot * Lc + l]] = 0;
      }
    }
    __syncthreads();
    if (t < 64) {
      int oc = 0;
      for (int base = 0; base < E; base += 64) {
        const int e = E - 1 - (base + lane);
        const bool dead = e >= 0 && !sc.e_alive[e];
        const unsigned long long bal = __ballot(dead);
        if (dead) {
          const int k = oc + lane_prefix(bal);
          if (k < BA_EMAX) out.lm_outlier_id[k] = w.e_lm[e];
        }
        oc += __popcll(bal);
      }
      if (lane == 0) out.lm_outlier_count = oc;
    }
  }
  __syncthreads();
  BAPROF(11);
  ba_optimize(w, 8);
  __syncthreads();
  BAPROF(0);
  const BAScratch sc = sh.sc;
  const int Lc = sc.Lc;
  if (t < W) {
#pragma unroll
    for (int j = 0; j < 7; j++) w.pose_est[t][j] = sh.pose[t][j];
  }
  for (int l = t; l < L; l += BA_T) {
    w.lm_est[l][0] = sc.lmA[l];
    w.lm_est[l][1] = sc.lmA[Lc + l];
    w.lm_est[l][2] = sc.lmA[2 * Lc + l];
  }
  if (t < 64) {
    // optimizer.removeEdge for the culled edges: order-preserving compaction
    const int n = E;
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
    // CorrectionInf: newest pose, landmarks observed >= 4 times (getMultiViewLMs(lms,4)), in bag order
    int c = 0;
    for (int base = 0; base < L; base += 64) {
      const int i = base + lane;
      const bool take = i < L && w.lm_count[i] >= 4;
      const unsigned long long bal = __ballot(take);
      if (take) {
        const int k = c + lane_prefix(bal);
        out.lm_id[k] = w.lm_id[i];
        out.lm_3d[k][0] = sc.lmA[i];
        out.lm_3d[k][1] = sc.lmA[Lc + i];
        out.lm_3d[k][2] = sc.lmA[2 * Lc + i];
      }
      c += __popcll(bal);
    }
    if (lane == 0) {
      w.n_edge = kept;
      out.frame_id = p.kf[s].frame_id;
      SE3d Tn = load_pose7(sh.pose[w.newest]);
      store_pose7(out.T_c_w, se3_from_mat(q_to_mat(Tn.q), Tn.t));
      out.lm_count = c;
      out.valid = 1;
      p.st[s].lm_state = 1;
      w.solve = 0;
      w.ba_runs++;
      if (p.counters) atomicAdd((unsigned long long*)&p.counters[2], 1ull);
    }
  }
  BAPROF(11);
}

void launch_ba_solve(hipStream_t st, const Pipe& p) {
  hipLaunchKernelGGL(k_ba_solve, dim3(p.S), dim3(BA_T), BA_LDS_BUDGET, st, p);
}
hipError_t ba_kernels_init() {
  return hipFuncSetAttribute((const void*)k_ba_solve, hipFuncAttributeMaxDynamicSharedMemorySize, BA_LDS_BUDGET);
}

}  // namespace flvis
