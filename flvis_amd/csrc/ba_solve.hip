// flvis_amd: batched sliding-window bundle adjustment for gfx950 -- the optimiser (one workgroup per stream-window).
//
// Replaces the OPTIMIZING block of LocalMapNodeletClass::frame_callback (src/backend/vo_localmap.cpp:292-366), i.e. g2o's
//   SparseOptimizer::initializeOptimization/optimize    core/sparse_optimizer.cpp:208-272,366-431
//   OptimizationAlgorithmLevenberg::solve               core/optimization_algorithm_levenberg.cpp:58-175
//   BlockSolver<6,3>::buildSystem/setLambda/solve       core/block_solver.hpp:314-565   (Schur complement on the landmarks)
//   EdgeSE3ProjectXYZ + RobustKernelHuber               types/sba/types_six_dof_expmap.cpp:389-433, core/robust_kernel_impl.cpp:65-78
// as ONE kernel launch per keyframe (12 + 8 LM iterations and the chi2 > 3 cull in between, all in-kernel).
//
// Mapping (BA_T = 512 threads per window; round 5).  Everything a trial touches per OBSERVATION lives in LDS:
//   * observation table (HBM, structure of arrays, rebuilt when edges change): omask[l] (bit per ring slot), uv[slot][l], edge index
//     [slot][l]; the observations by FREE poses are numbered landmark-major as "items" (ibase[l] = exclusive prefix of their count);
//   * records: per item three 16-byte pairs in LDS -- (x/z, y/z) (1/z, w) (u, v): the point in the camera at the linearisation point,
//     the Huber weight, the pixel.  The 6x3 block of the classical formulation, B = w Jp^T Jl, has rank 2 and is never formed:
//     Z = B G^-T = Jp^T M with M = w Jl G^-T (2x3), Jp a function of (x/z, y/z, 1/z) and Jl of those and the pose's rotation rows;
//     Z_i Z_j^T = Jp_i^T (M_i M_j^T) Jp_j.  A window whose items fit one buffer (~1900 items at 550 landmarks: the D435 windows)
//     keeps its records RESIDENT for the whole optimisation; larger windows rebuild them chunk by chunk through two buffers;
//   * per landmark in LDS: G (the 3x3 Cholesky factor of Hll + lambda I, inverted diagonal), c = G^-1 bl, its free-pose mask and
//     item base; in HBM (coalesced, one latency per phase): the estimate (accepted / trial buffers), Hll, bl (accepted / trial);
//   * linearisation: thread per landmark walking ITS observations; Hll / bl stay in registers.  Hpp / bp are needed as such only once
//     per optimize() call (lambda's initial value: a wave walks the poses together and reduce-scatters the 27 entries); afterwards
//     the Schur phase sums them inside its own walk over the records (the diagonal pair of a pose visits exactly its observations);
//   * LM trial: [staging: G, c per landmark] -> Schur accumulate: thread = (pose pair, landmark slice of 16), 6x6 register tiles from
//     the factored products (diagonal pairs: the lower triangle + the pose's right-hand side), fixed slice partition, slices summed
//     with DPP row shifts => bit-reproducible run to run, no atomics -> the reduced system (6P x 6P, P <= 15) in LDS, factored by
//     ONE wave: left-looking Cholesky over 6x6 blocks (16-byte aligned rows, 128-bit reads), block substitutions, trial poses ->
//     update: thread per landmark, the step from the records, and the trial state's evaluation AS a linearisation (chi2, Hll / bl
//     into the trial buffers, the records overwritten): an accepted trial -- the rule -- leaves the next iteration nothing to linearise.
// The phase functions are kept small enough to live in the caller-saved registers: a called function that needs more stores and
// reloads every callee-saved register it touches through scratch memory, per call.
// This is latency-bound fp64 on a tiny problem (S is at most 90x90); the matrix cores were measured for the Schur complement (slower:
// FLVIS_BA_MFMA=1, ba_phase_schur_mfma) and are not used.
#include "dev_common.hpp"
#include "dev_geom.hpp"
#include "track_kernels.hpp"

namespace flvis {

// g2o optimize(12) / optimize(8) of vo_localmap.cpp:296,345 (overridable for timing experiments only)
#ifndef FLVIS_BA_IT1
#define FLVIS_BA_IT1 12
#define FLVIS_BA_IT2 8
#endif
#ifndef FLVIS_BA_T
#define FLVIS_BA_T 512  // (build-variant knob: threads of a local-map workgroup)
#endif
constexpr int BA_T = FLVIS_BA_T;
// (build-variant knob: waves per SIMD the kernel's register allocation aims at -- the backend hands the budget on to the out-of-line
// phases.  By default (2: the workgroup's own two waves per SIMD) a local-map workgroup may take the whole register file of its CU, so
// that no wave of another kernel fits beside it)
#ifdef FLVIS_BA_WAVES
#define BA_ATTR __attribute__((amdgpu_waves_per_eu(FLVIS_BA_WAVES, FLVIS_BA_WAVES)))
#else
#define BA_ATTR
#endif
// (build-variant knob: the per-trial phases as calls -- each with a register allocation of its own, and ~110 callee-saved VGPRs stored and
// reloaded around every call -- or inlined into ba_optimize)
#ifndef FLVIS_BA_PHASE_FN
#define FLVIS_BA_PHASE_FN __noinline__
#endif
constexpr int BA_NW = BA_T / 64;
constexpr int BA_PMAX = BA_WMAX - 1;   // free poses
constexpr int BA_NRMAX = 6 * BA_PMAX;  // 90
constexpr int BA_LDS_BUDGET = 159 * 1024;  // dynamic LDS of the worker (1 KB left for its static words)
constexpr int BA_MAXCHUNK = 160;

// HBM scratch is addressed through explicit global-address-space pointers: inside the non-inlined phase functions the
// compiler then emits global_load/global_store (vmcnt only) instead of flat_* -- flat loads also tick lgkmcnt and would
// serialise the prefetched chunk loads behind every LDS wait of the Schur accumulation.
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) int gint;
typedef __attribute__((address_space(1))) unsigned guint;

struct BAScratch {  // carved out of Pipe::ba_scratch per stream; Lc = landmark stride (a multiple of 64)
  gdouble* lmA;     // [3][Lc]  accepted landmark estimates
  gdouble* lmB;     // [3][Lc]  trial estimates (roles swap on acceptance)
  gdouble* Hll;     // [6][Lc]  xx xy xz yy yz zz (at the accepted estimates)
  gdouble* bl;      // [3][Lc]
  gdouble* Hll2;    // ... at the trial estimates, written by the trial's own evaluation (resident records); swapped in on acceptance
  gdouble* bl2;
  gdouble* uv;      // [W][2][Lc]
  gint* eid;        // [W][Lc]  edge index or -1
  guint* omask;     // [Lc]  bit slot: landmark has an alive edge to the pose in ring slot `slot`
  guint* fmask;     // [Lc]  bit h: ... to FREE pose h (hessian index)
  gint* ibase;      // [Lc + 64]  first item of landmark l (exclusive prefix of popc(fmask)), ibase[L] = item count
  gint* e_alive;    // [E]
  int Lc;
};
// (Rounds 1-4 also kept, per observation by a free pose, the 6x3 block B = w Jp^T Jl and a copy of its landmark's Hll in HBM: written by
// every linearisation, read back by every trial -- 164 MB per launch, ~100 x the keyframes' own bytes (round-4 counters).  B has rank 2:
// Z = B G^-T = Jp^T (w Jl G^-T) = Jp^T M with Jp a function of three numbers (x/z, y/z, 1/z).  The Schur phase now rebuilds (x/z, y/z,
// 1/z, M) per observation from the landmark, the pose and the pixel -- 9 doubles in LDS instead of 24 through HBM.)

size_t ba_scratch_doubles() {
  size_t d = (size_t)BA_LMAX * (3 + 3 + 6 + 3 + 6 + 3) + (size_t)BA_LMAX * BA_WMAX * 2;
  size_t ints = (size_t)BA_LMAX * BA_WMAX + (size_t)BA_LMAX * 3 + 64 + BA_EMAX + 64;
  return ((d + (ints + 1) / 2 + 64) + 1) & ~(size_t)1;  // even: 16-byte alignment of every stream's slice
}

FD BAScratch carve(double* base, int L, int E, int W) {
  BAScratch s;
  const int Lc = ((L > 0 ? L : 1) + 63) & ~63;
  s.Lc = Lc;
  gdouble* q = (gdouble*)base;
  s.lmA = q; q += (size_t)3 * Lc;
  s.lmB = q; q += (size_t)3 * Lc;
  s.Hll = q; q += (size_t)6 * Lc;
  s.bl = q; q += (size_t)3 * Lc;
  s.Hll2 = q; q += (size_t)6 * Lc;
  s.bl2 = q; q += (size_t)3 * Lc;
  s.uv = q; q += (size_t)2 * W * Lc;
  gint* ii = (gint*)q;
  s.eid = ii; ii += (size_t)W * Lc;
  s.omask = (guint*)ii; ii += Lc;
  s.fmask = (guint*)ii; ii += Lc;
  s.ibase = ii; ii += Lc + 64;
  s.e_alive = ii;
  (void)E;
  return s;
}

struct BAShared {
  double pose[BA_WMAX][7];   // accepted estimates by ring slot (g2o SE3Quat: t, q)
  double poseT[BA_WMAX][7];  // trial
  double RT[BA_WMAX][12];    // rotation matrix (row-major) + translation of pose / poseT
  double RTt[BA_WMAX][12];
  double Hpp[BA_PMAX][36];
  double b[BA_NRMAX];      // pose gradient (-J^T W r) of the linearised system; "inline" iterations: its IMU part only (see ba_phase_schur)
  double bfull[BA_NRMAX];  // ... always the whole gradient (written by the Schur phase): what the gain ratio's scale is formed with
  double x[BA_NRMAX];
  double red[2][BA_NW];
  double K[4];
  BAScratch sc;
  int slot_of[BA_PMAX];
  int hidx_of[BA_WMAX];
  int slot_cnt[BA_WMAX];
  int P, L, E, flag, cnt, W, nitems;
  int lds_budget;  // dynamic LDS of this launch (Pipe::ba_lds_bytes): sizes the Schur chunk buffers
  int n_trials;  // LM trials (reduced-system solves) of this optimisation: flop accounting of bench.py
  long long t_begin;  // wall_clock64 (100 MHz) at the start of the optimisation
  int NR, LD, off_linv, off_imu, off_stage;  // reduced system geometry: Hs[NR][LD], Linv, IMU blocks, chunk buffers (double offsets)
  int fused;       // the window's items fit ONE chunk: their records stay in LDS for the whole optimisation (see ba_phase_linearize)
  int fixed_slot;  // ring slot of the fixed pose if it has observations (its pixels are the only ones read from HBM per phase then), else -1
  int CI, CL, bufd, nchunk;         // chunk capacities (items, landmarks), doubles per buffer, chunk count
  int npairs, slices, rs;           // role partition of the Schur phase
  // ... balanced (round 6): lanes per pose pair in proportion to the landmarks the pair shares (powers of two, 4 .. 64), the diagonal
  // pairs in whole waves of their own.  pair_tab[pair] = first thread | log2(lanes) << 12 | i1 << 16 | i2 << 20; role_pair[thread] = its
  // pair (255: none); pair_cnt = landmarks seen by both poses of the pair
  int balance, balanced, max_slices;
  int pair_cnt[64];
  unsigned pair_tab[64];
  unsigned char role_pair[BA_T];
  int use_mfma, NRp, CLm, nchunk_m;  // MFMA variant of the Schur phase: padded system size, landmarks per dense chunk
  // IMU rotation edges (optional factor): edge k links ring slots imu_a[k] -> imu_b[k]
  int n_imu, imu_a[BA_WMAX], imu_b[BA_WMAX];
  double imu_w[BA_WMAX], imu_dq[BA_WMAX][4], q_c_b[4];
  // position rows (imu_wp[k] > 0): preintegrated displacement, velocity of keyframe a, interval; t_c_b = body origin in the camera frame
  double imu_wp[BA_WMAX], imu_dp[BA_WMAX][3], imu_va[BA_WMAX][3], imu_dtk[BA_WMAX], t_c_b[3];
  // (the edges' linearised blocks -- Ja^T W Ja, Jb^T W Jb, Ja^T W Jb, the two gradients: 120 doubles per edge -- live in dynamic LDS at
  // off_imu, only when the factor is on: 17 KB of static LDS otherwise taken from the Schur buffers of every window)
  double imu_chi[BA_WMAX], imu_chit[BA_WMAX];  // w |r|^2 at the accepted / the trial poses
  int wscan[BA_NW];
  int chunk_l0[BA_MAXCHUNK + 1];    // first landmark / first item of every chunk
  int chunk_i0[BA_MAXCHUNK + 1];
  long long* prof;  // optional phase timers (FLVIS_BA_PROF builds)
  long long tlast;
  long long prof_acc[16];
  // followed in dynamic LDS by: Hs[NR][NR+1], Linv[P][36], the IMU blocks, then the chunk buffer(s); the per-wave partial sums of the
  // linearisation lie over Hs (dead until the Schur phase writes it)
};

// all per-window working state lives in dynamic LDS; the phase functions re-derive it from this symbol so that the
// compiler keeps LDS addressing (ds_* instructions) inside non-inlined functions
extern __shared__ __attribute__((aligned(16))) unsigned char ba_smem[];
constexpr size_t BA_SH_BYTES = ((sizeof(BAShared) + 15) / 16) * 16;
FD BAShared& ba_sh() { return *reinterpret_cast<BAShared*>(ba_smem); }
FD double* ba_dyn() { return reinterpret_cast<double*>(ba_smem + BA_SH_BYTES); }
// IMU edge k's blocks: which = 0 Ja^T W Ja, 1 Jb^T W Jb, 2 Ja^T W Jb (36 doubles each), 3 / 4 the gradients Ja^T W r / Jb^T W r (6 each)
// Chunk buffer `buf` of the Schur phase.  Per item (an observation by a free pose) a record of three 16-byte pairs [3][CI]:
//   (xn, yn) (iz, w) (u, v)  -- the point in the camera (x/z, y/z, 1/z) at the linearisation point, the Huber weight, the pixel;
// per landmark [CL]: G (the factor of Hll + lambda I: i00 g10 g20 i11 g21 i22), c = G^-1 bl, its free-pose mask, its first local item.
struct SchurBuf {
  double2* z;
  double *gb, *cb;
  unsigned* mb;
  int* lb;
};
FD SchurBuf ba_schur_buf(int buf);
FD double* ba_imu_blk(int k, int which) { return ba_dyn() + ba_sh().off_imu + k * 120 + (which < 3 ? 36 * which : 108 + 6 * (which - 3)); }

FD SchurBuf ba_schur_buf(int buf) {
  BAShared& sh = ba_sh();
  SchurBuf b;
  double* zb = ba_dyn() + sh.off_stage + (size_t)buf * sh.bufd;
  b.z = reinterpret_cast<double2*>(zb);
  b.gb = zb + (size_t)6 * sh.CI;
  b.cb = b.gb + (size_t)6 * sh.CL;
  b.mb = reinterpret_cast<unsigned*>(b.cb + (size_t)3 * sh.CL);
  b.lb = reinterpret_cast<int*>(b.mb + sh.CL);
  return b;
}

#ifdef FLVIS_BA_PROF
// (the phase timers are summed in LDS and written out once per optimisation: a global atomic per mark had to drain before the next
// barrier and added a microsecond or two to whatever phase followed it)
#define BAPROF(i)                                  \
  do {                                             \
    if (threadIdx.x == 0 && sh.prof) {             \
      long long now_ = (long long)wall_clock64();  \
      sh.prof_acc[i] += now_ - sh.tlast;           \
      sh.tlast = now_;                             \
    }                                              \
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

FD double rsqrt_fwd(double s);

// 1/z: hardware estimate + two Newton steps (~1 ulp).  The projection Jacobians below use ONE reciprocal per observation
// instead of g2o's fourteen divisions (fp64 division is a ~15-instruction sequence and dominated these phases).
FD double rcp_nr(double z) {
  double r = __builtin_amdgcn_rcp(z);
#pragma unroll
  for (int it = 0; it < 2; it++) r = fma(r, fma(-z, r, 1.0), r);
  return r;
}

// EdgeSE3ProjectXYZ::computeError: squared reprojection error of landmark p in the camera rt = (R | t)
FD double ba_err2(const double* rt, double px, double py, double pz, double u, double v, const double* K) {
  const double x = rt[0] * px + rt[1] * py + rt[2] * pz + rt[9];
  const double y = rt[3] * px + rt[4] * py + rt[5] * pz + rt[10];
  const double z = rt[6] * px + rt[7] * py + rt[8] * pz + rt[11];
  const double iz = rcp_nr(z);
  const double e0 = u - (x * iz * K[0] + K[2]), e1 = v - (y * iz * K[1] + K[3]);
  return e0 * e0 + e1 * e1;
}

// EdgeSE3ProjectXYZ::linearizeOplus (types_six_dof_expmap.cpp:389-433): residual, Jl (2x3), Jp (2x6)
FD void ba_linearize(const double* rt, double px, double py, double pz, double u, double v, const double* K, double* er,
                     double (*Ji)[3], double (*Jj)[6]) {
  const double x = rt[0] * px + rt[1] * py + rt[2] * pz + rt[9];
  const double y = rt[3] * px + rt[4] * py + rt[5] * pz + rt[10];
  const double z = rt[6] * px + rt[7] * py + rt[8] * pz + rt[11];
  const double iz = rcp_nr(z), fx = K[0], fy = K[1];
  const double xn = x * iz, yn = y * iz;
  er[0] = u - (xn * fx + K[2]);
  er[1] = v - (yn * fy + K[3]);
  if (Ji) {
    const double ax = -iz * fx, ay = -iz * fy;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      Ji[0][c] = ax * (rt[c] - xn * rt[6 + c]);
      Ji[1][c] = ay * (rt[3 + c] - yn * rt[6 + c]);
    }
  }
  Jj[0][0] = xn * yn * fx;
  Jj[0][1] = -(1 + xn * xn) * fx;
  Jj[0][2] = yn * fx;
  Jj[0][3] = -iz * fx;
  Jj[0][4] = 0;
  Jj[0][5] = xn * iz * fx;
  Jj[1][0] = (1 + yn * yn) * fy;
  Jj[1][1] = -xn * yn * fy;
  Jj[1][2] = -xn * fy;
  Jj[1][3] = 0;
  Jj[1][4] = -iz * fy;
  Jj[1][5] = yn * iz * fy;
}

// RobustKernelHuber (delta = 1) on the squared error e: rho(e) and the weight rho'(e)
FD double ba_huber_rho(double e) { return e <= 1.0 ? e : 2.0 * (e * rsqrt_fwd(e)) - 1.0; }
FD double ba_huber_w(double e) { return e <= 1.0 ? 1.0 : rsqrt_fwd(e); }

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
FD double rsqrt_fwd(double s) { return rsqrt_nr(s); }

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

// sum over the wave as a wave-uniform value
FD int ba_wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
FD int ba_wave_max_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int u = __shfl_xor(v, o, 64);
    v = u > v ? u : v;
  }
  return v;
}

// Lanes of the Schur accumulate per pose pair (round 6; ONE wave, lane = pair, pairs numbered as ba_schur_role numbers them: the P
// diagonal pairs, then i1 < i2 row by row).  A pair's lanes split the landmarks both of its poses see among them; with 16 lanes for every
// pair a window's neighbouring keyframes (~150-190 shared landmarks) kept their waves busy four to eight times as long as the pairs six
// keyframes apart (~20), and the phase ended with the slowest wave (profiles/r05_ba_phases.md: 108 us of waiting per optimisation).
// Here a pair gets a power of two of lanes (4 .. 64) in proportion to its landmark count (a diagonal pair's product, which carries the
// right-hand side, counts 4 : 3), the largest counts that fit the workgroup: the diagonal pairs in whole waves of their own (a wave
// holding both kinds would run the two product bodies one after the other), each region sorted by lane count so that a group never
// straddles a DPP row it does not fill.  A function of the window's structure only: the same partition, the same order of summation, run
// after run.
FD void ba_balance_pairs() {
  BAShared& sh = ba_sh();
  const int p = threadIdx.x & 63, P = sh.P, NP = P + P * (P - 1) / 2;
  const bool diag = p < P, on = p < NP;
  // work of the pair per LM trial in units of a ninth of an off-diagonal block product (two records, a 6x6 tile: measured ~1.8 times the
  // diagonal product with its lower triangle and right-hand side)
  const int load = on ? sh.pair_cnt[p] * (diag ? 5 : 9) : 0;
  const int total = ba_wave_sum_i(load);
  int sl = on ? 4 : 0, dsum = 0, osum = 0;
  if (total > 0) {
    const float base = (float)load * (float)BA_T / (float)total;  // lanes at the even share of work per lane
    for (int k = 0; k < 40; k++) {                                // ... at 2^(k/4) times that share: the first that fits
      const float need = base * exp2f(-0.25f * (float)k);
      const int c = !on ? 0 : need <= 4.f ? 4 : need <= 8.f ? 8 : need <= 16.f ? 16 : need <= 32.f ? 32 : 64;
      dsum = ba_wave_sum_i(diag ? c : 0);
      osum = ba_wave_sum_i(diag ? 0 : c);
      if (((dsum + 63) & ~63) + osum <= BA_T) {
        sl = c;
        break;
      }
    }
  }
  dsum = ba_wave_sum_i(diag ? sl : 0);
  osum = ba_wave_sum_i(diag ? 0 : sl);
  // what is left over goes to the busiest lanes, one doubling at a time
  for (int it = 0; it < 8; it++) {
    const int q = on && sl < 64 ? (load << 8) / sl : -1;  // load per lane (sl is a power of two: a shift)
    const int qm = ba_wave_max_i(q);
    if (qm <= 0) break;
    const unsigned long long who = __ballot(q == qm);
    const int w0 = __builtin_ctzll(who);  // (one pair per step: the first of the busiest)
    const int wsl = __shfl(sl, w0, 64);
    const int nd = dsum + (w0 < P ? wsl : 0), no = osum + (w0 < P ? 0 : wsl);
    if (((nd + 63) & ~63) + no > BA_T) break;
    if (p == w0) sl *= 2;
    dsum = nd;
    osum = no;
  }
  // first thread of every pair: its region's pairs with more lanes in front of it, equal ones in pair order
  int off = 0;
  for (int q = 0; q < NP; q++) {
    const int sq = __shfl(sl, q, 64);
    if ((q < P) == diag && (sq > sl || (sq == sl && q < p))) off += sq;
  }
  if (!diag) off += (dsum + 63) & ~63;
  if (on) {
    int i1 = p, i2 = p;
    if (!diag) {
      int rem = p - P;
      i1 = 0;
      while (rem >= P - 1 - i1) {
        rem -= P - 1 - i1;
        i1++;
      }
      i2 = i1 + 1 + rem;
    }
    sh.pair_tab[p] = (unsigned)off | ((unsigned)(31 - __builtin_clz(sl)) << 12) | ((unsigned)i1 << 16) | ((unsigned)i2 << 20);
  }
  // role_pair: the lanes write the table side by side, pair after pair
  for (int q = 0; q < NP; q++) {
    const int sq = __shfl(sl, q, 64), oq = __shfl(off, q, 64);
    if (p < sq) sh.role_pair[oq + p] = (unsigned char)q;
  }
  const int ms = ba_wave_max_i(sl);
  if (p == 0) sh.max_slices = ms;
#ifdef FLVIS_BA_PROF
  // (the partition by keyframe distance: debug counters 48 + 2 d: lanes, 49 + 2 d: landmarks, d = i2 - i1 of the pair)
  if (on && sh.prof) {
    const unsigned tab = sh.pair_tab[p];
    const int d = (int)((tab >> 20) & 15u) - (int)((tab >> 16) & 15u);
    if (d < 7) {
      atomicAdd((unsigned long long*)&sh.prof[40 + 2 * d], (unsigned long long)sl);
      atomicAdd((unsigned long long*)&sh.prof[41 + 2 * d], (unsigned long long)sh.pair_cnt[p]);
    }
  }
#endif
}

// rebuilds the observation table from the alive edges: free-pose numbering (hessian order = slot order), per-landmark
// masks, the item numbering and the chunk table of the Schur phase
__device__ __noinline__ void ba_build_structure(const WindowDev& w) {
  BAShared& sh = ba_sh();
  const BAScratch sc = sh.sc;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int E = w.n_edge, L = w.n_lm, Lc = sc.Lc, W = sh.W;
  if (t == 0) {
    sh.E = E;
    sh.L = L;
    sh.cnt = 0;
  }
  if (t < BA_WMAX) {
    sh.hidx_of[t] = -1;
    sh.slot_cnt[t] = 0;
  }
  if (t < 64) sh.pair_cnt[t] = 0;
  sh.role_pair[t] = 255;
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
    // the poses that are not free but observed: the fixed (oldest) one.  More than one (more than BA_PMAX free poses) rules out the
    // resident records below, whose phases read one such pose's pixels per landmark
    int fs = -1, nfs = 0;
    for (int slot = 0; slot < W; slot++)
      if (sh.hidx_of[slot] < 0 && sh.slot_cnt[slot] > 0) {
        fs = slot;
        nfs++;
      }
    sh.fixed_slot = nfs == 1 ? fs : -1;
    sh.fused = nfs <= 1 ? 1 : 0;  // (and one chunk, not the matrix-core variant: decided with the chunk table below)
    // leading dimension of the reduced system: rows 16-byte aligned (128-bit LDS reads of a row's 6-column blocks) and LD / 2 odd
    // (the rows a wave's lanes read together fall into different bank groups)
    const int NR = 6 * P, LD = (P & 1) ? NR + 4 : NR + 2;
    sh.NR = NR;
    sh.LD = LD;
    sh.off_linv = (NR + 1) * LD;  // (row NR: the right-hand side rides through the factorisation, ba_chol_factor_wg)
    // the per-wave partials of the linearisation (BA_NW x P x 27) lie over Hs / Linv: nothing behind them may start inside
    const int wacc_end = BA_NW * 27 * P;
    sh.off_imu = (NR + 1) * LD + P * 36 > wacc_end ? (NR + 1) * LD + P * 36 : wacc_end;
    sh.off_stage = sh.off_imu + (sh.n_imu > 0 ? 120 * W : 0);
    // pair groups of the Schur accumulate: the P diagonal pairs first, padded to whole waves -- a wave that held both kinds of pairs
    // would run the two product bodies one after the other --, then the P (P - 1) / 2 pairs i1 < i2
    int slices = 64, npairs = 0;
    for (;; slices >>= 1) {
      const int ppw = 64 / slices;
      npairs = ((P + ppw - 1) / ppw) * ppw + P * (P - 1) / 2;
      if (slices == 1 || slices * npairs <= BA_T) break;
    }
    sh.npairs = npairs;
    sh.slices = slices;
    sh.rs = 1;
    sh.balanced = sh.balance && P >= 2 && P + P * (P - 1) / 2 <= 64 ? 1 : 0;
    sh.max_slices = slices;
    // MFMA variant: the chunk is the DENSE (NR + 1 rows padded to 16) x (3 columns per landmark) slice of Z' = [Z; c^T]
    const int NRp = (NR + 1 + 15) & ~15;
    const int avail_m = ((int)((sh.lds_budget - BA_SH_BYTES) / 8) - sh.off_stage) & ~1;
    int CLm = (avail_m / (3 * NRp)) & ~3;  // 3 * CLm columns, a multiple of the MFMA's K = 4
    if (CLm > 252) CLm = 252;
    sh.NRp = NRp;
    sh.CLm = CLm;
    sh.nchunk_m = CLm > 0 ? (L + CLm - 1) / CLm : 0;
  }
  __syncthreads();
  for (int l = t; l < Lc; l += BA_T) {
    unsigned m = 0, fm = 0;
    if (l < L)
      for (int slot = 0; slot < W; slot++)
        if (sc.eid[slot * Lc + l] >= 0) {
          m |= 1u << slot;
          const int hi = sh.hidx_of[slot];
          if (hi >= 0) fm |= 1u << hi;
        }
    sc.omask[l] = m;
    sc.fmask[l] = fm;
  }
  if (sh.balanced) {  // landmarks per pose pair (a, b), a <= b: the work of the pair's lanes in the Schur accumulate (one vote per pair and wave)
    const int P = sh.P;
    for (int l0 = 0; l0 < L; l0 += BA_T) {
      const unsigned fm = l0 + t < L ? sc.fmask[l0 + t] : 0u;  // (this thread wrote it above)
      int idx = 0;
      for (int a = 0; a < P; a++) {
        const int n = __popcll(__ballot((fm >> a) & 1u));
        if (lane == 0 && n) atomicAdd(&sh.pair_cnt[a], n);
      }
      idx = P;
      for (int a = 0; a < P; a++)
        for (int b = a + 1; b < P; b++, idx++) {
          const unsigned need = (1u << a) | (1u << b);
          const int n = __popcll(__ballot((fm & need) == need));
          if (lane == 0 && n) atomicAdd(&sh.pair_cnt[idx], n);
        }
    }
  }
  __syncthreads();
  if (sh.balanced && wv == 0) ba_balance_pairs();
  // item numbering: exclusive scan of popc(fmask) over the landmarks (thread = contiguous run of landmarks)
  int* lbase = reinterpret_cast<int*>(ba_dyn() + sh.off_stage);  // LDS copy of ibase for the chunk search below
  {
    const int per = (L + BA_T - 1) / BA_T;
    const int la = t * per, lb = (la + per < L) ? la + per : L;
    int c = 0;
    for (int l = la; l < lb; l++) c += __popc(sc.fmask[l]);
    int inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(inc, o, 64);
      if (lane >= o) inc += v;
    }
    if (lane == 63) sh.wscan[wv] = inc;
    __syncthreads();
    int off = 0;
    for (int k = 0; k < wv; k++) off += sh.wscan[k];
    int run = off + inc - c;
    for (int l = la; l < lb; l++) {
      sc.ibase[l] = run;
      lbase[l] = run;
      run += __popc(sc.fmask[l]);
    }
    if (t == BA_T - 1) {
      sc.ibase[L] = run;  // (the last thread's run ends at L or is empty: run == total either way)
      lbase[L] = run;
      sh.nitems = run;
    }
  }
  __syncthreads();
  if (t == 0) {
    // Chunk buffers of the Schur phase (ba_schur_buf): 6 doubles per item, 10 per landmark.  A window whose items all fit (~1900 at
    // 159 KB, 7 free poses and 550 landmarks: the D435 windows) is ONE chunk in one buffer -- its records stay resident, no barrier
    // inside the phase; larger windows stream through two buffers.
    const int avail = ((int)((sh.lds_budget - BA_SH_BYTES) / 8) - sh.off_stage) & ~1;
    const int nit = lbase[L];
    int CI, CL, bufd;
    if (6 * ((nit + 1) & ~1) + 10 * ((L + 1) & ~1) <= avail) {
      CI = nit > 0 ? ((nit + 1) & ~1) : 2;
      CL = L > 0 ? ((L + 1) & ~1) : 2;
      bufd = avail;
    } else {
      bufd = (avail / 2) & ~1;
      CL = 256;
      while (CL > 32 && CL * 10 > bufd / 4) CL >>= 1;  // (the landmark arrays take at most a quarter of a buffer)
      CI = ((bufd - 10 * CL) / 6) & ~1;
    }
    sh.CL = CL;
    sh.CI = CI;
    sh.bufd = bufd;
    // greedy chunking: as many landmarks as fit both capacities
    int c = 0, l0 = 0;
    while (l0 < L && c < BA_MAXCHUNK) {
      int lo = l0 + 1, hi = (l0 + CL < L) ? l0 + CL : L;  // invariant: [l0, lo) always fits (one landmark has <= P items)
      const int i0 = lbase[l0];
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (lbase[mid] - i0 <= CI) lo = mid;
        else hi = mid - 1;
      }
      sh.chunk_l0[c] = l0;
      sh.chunk_i0[c] = i0;
      c++;
      l0 = lo;
    }
    sh.chunk_l0[c] = L;
    sh.chunk_i0[c] = lbase[L];
    sh.nchunk = c;
    if (c != 1 || sh.use_mfma) sh.fused = 0;
  }
  __syncthreads();
  if (sh.fused) {
    // resident records: the pixels of the items are written once per optimize() call (nothing rewrites the (u, v) pair); the phases then
    // read no per-observation data from HBM any more.  (lbase, at the start of the same buffer, is dead now.)
    double2* z = reinterpret_cast<double2*>(ba_dyn() + sh.off_stage);
    const int CI = sh.CI;
    for (int e = t; e < E; e += BA_T) {
      if (!sc.e_alive[e]) continue;
      const int hi = sh.hidx_of[w.e_pose[e]];
      if (hi < 0) continue;
      const int l = w.e_lidx[e];
      const int idx = sc.ibase[l] + __popc(sc.fmask[l] & ((1u << hi) - 1u));
      z[2 * CI + idx] = double2{w.e_uv[e][0], w.e_uv[e][1]};
    }
    __syncthreads();
  }
}

// blocked (6x6) left-looking Cholesky of the lower triangle of Hs (leading dimension LD) by ONE wave, then the two
// triangular solves on sh.x; Linv receives the inverted diagonal blocks.  Returns false on a non-positive pivot.
FD void ba_chol_subst(bool fwd);
__device__ FLVIS_BA_PHASE_FN bool ba_chol_solve() {
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
#ifdef FLVIS_BA_SOLVE_MFMA
    // (build variant, round 5: this update -- panel -= L[rows, 0 : c0] L[c0 : c0 + 6, 0 : c0]^T -- on the matrix cores: 16-row tiles x the 6
    // columns padded to 16 x c0 / 4 k-steps of v_mfma_f64_16x16x4_f64; north_star names "an MFMA dense solve only for the reduced
    // camera block", this is the A/B that settles it.  profiles/r05_ba_phases.md)
    if (jb > 0) {
      typedef double mf_d4 __attribute__((ext_vector_type(4)));
      const int ar = lane & 15, ak = lane >> 4, ksteps = (c0 + 3) >> 2;
      for (int ti = 0; 16 * ti < NR; ti++) {
        if (16 * ti + 15 < c0) continue;
        const int row = 16 * ti + ar;
        mf_d4 acc = {0, 0, 0, 0};
        for (int ks = 0; ks < ksteps; ks++) {
          const int k = 4 * ks + ak;
          const double a = (row < NR && k < c0) ? Hs[row * LD + k] : 0.0;
          const double b = (ar < 6 && k < c0) ? Hs[(c0 + ar) * LD + k] : 0.0;
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; v++) {  // D[(lane >> 4) + 4 v][lane & 15]
          const int r = 16 * ti + (lane >> 4) + 4 * v, c = lane & 15;
          if (r >= c0 && r < NR && c < 6) Hs[r * LD + c0 + c] -= acc[v];
        }
      }
    }
    for (int rr = lane; false && rr < NR; rr += 64) {
#else
    for (int rr = lane; rr < NR; rr += 64) {
#endif
      if (rr < c0) continue;
      double2* own = reinterpret_cast<double2*>(Hs + rr * LD);  // (rows are 16-byte aligned: three 128-bit accesses per 6-column block)
      double2 a0 = own[c0 / 2], a1 = own[c0 / 2 + 1], a2 = own[c0 / 2 + 2];
      double a[6] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y};
      for (int kb = 0; kb < jb; kb++) {
        const double2 l0 = own[3 * kb], l1 = own[3 * kb + 1], l2 = own[3 * kb + 2];
#pragma unroll
        for (int c = 0; c < 6; c++) {
          const double2* dr = reinterpret_cast<const double2*>(Hs + (c0 + c) * LD) + 3 * kb;  // (the same address in every lane: a broadcast)
          const double2 d0 = dr[0], d1 = dr[1], d2 = dr[2];
          a[c] = fma(-l2.y, d2.y, fma(-l2.x, d2.x, fma(-l1.y, d1.y, fma(-l1.x, d1.x, fma(-l0.y, d0.y, fma(-l0.x, d0.x, a[c]))))));
        }
      }
      own[c0 / 2] = double2{a[0], a[1]};
      own[c0 / 2 + 1] = double2{a[2], a[3]};
      own[c0 / 2 + 2] = double2{a[4], a[5]};
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
  BAPROF(5);
  ba_chol_subst(true);
  return okc;
}

// the two triangular solves on sh.x with the factor in Hs / Linv, by ONE wave.  fwd = false: the forward substitution happened inside
// the factorisation (ba_chol_factor_wg), y is row NR of Hs.
FD void ba_chol_subst(bool fwd) {
  BAShared& sh = ba_sh();
  double* Hs = ba_dyn();
  double* Linv = Hs + sh.off_linv;
  double* xs = sh.x;
  const int P = sh.P, LD = sh.LD;
  const int lane = threadIdx.x & 63;
  const int NR = 6 * P;
  if (!fwd) {
    for (int i = lane; i < NR; i += 64) xs[i] = Hs[NR * LD + i];
    wave_lds_fence();
  }
  // forward substitution L y = rhs (block-wise)
  for (int jb = 0; fwd && jb < P; jb++) {
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
      const double2* own = reinterpret_cast<const double2*>(Hs + rr * LD) + c0 / 2;
      const double2 l0 = own[0], l1 = own[1], l2 = own[2];
      xs[rr] = fma(-l2.y, y[5], fma(-l2.x, y[4], fma(-l1.y, y[3], fma(-l1.x, y[2], fma(-l0.y, y[1], fma(-l0.x, y[0], xs[rr]))))));
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
}

#ifndef FLVIS_BA_CHOL_WG
#ifdef FLVIS_BA_SOLVE_MFMA
#define FLVIS_BA_CHOL_WG 0
#else
#define FLVIS_BA_CHOL_WG 1
#endif
#endif
// The factorisation of the reduced system by the WHOLE workgroup (round 6), right-looking.  Wave 0 factors the diagonal block of block
// column jb and scales the rows below it -- the chain of six pivots nobody can help with; then every wave takes one block column kb > jb
// of the trailing matrix, lane = row: the very products, in the very order, that the one-wave left-looking form (ba_chol_solve) subtracts
// when it reaches column kb, so the factor is the same bit for bit.  The one-wave form walked all finished block columns per panel with
// seven waves waiting at the phase's barrier (12.5 us per LM trial, 20 trials per optimisation); here a block column costs its pivot chain,
// one 36-FMA update and two barriers.  -DFLVIS_BA_CHOL_WG=0 keeps the one-wave form.
__device__ FLVIS_BA_PHASE_FN bool ba_chol_factor_wg() {
  BAShared& sh = ba_sh();
  double* Hs = ba_dyn();
  double* Linv = Hs + sh.off_linv;
  const int P = sh.P, LD = sh.LD;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int NR = 6 * P;
  bool okc = true;
  // the right-hand side as row NR of the matrix: scaled and updated like every other row below the diagonal, it leaves the loop as
  // y = L^-1 rhs -- the forward substitution (14 dependent LDS round trips on one wave before) for nothing, the same products in the same
  // order
  if (wv == 0)
    for (int i = lane; i < NR; i += 64) Hs[NR * LD + i] = sh.x[i];
  for (int jb = 0; jb < P; jb++) {
    const int c0 = 6 * jb;
    if (wv == 0) {
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
      for (int rr = c0 + 6 + lane; rr <= NR; rr += 64) {
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
    if (jb + 1 == P) break;
    __syncthreads();
    // trailing matrix: block column kb, rows from its diagonal block down, minus L[row, jb] L[kb rows, jb]^T
    for (int kb = jb + 1 + wv; kb < P; kb += BA_NW) {
      const int k0 = 6 * kb;
      for (int rr = k0 + lane; rr <= NR; rr += 64) {
        double2* own = reinterpret_cast<double2*>(Hs + rr * LD);  // (rows are 16-byte aligned: three 128-bit accesses per 6-column block)
        const double2 a0 = own[k0 / 2], a1 = own[k0 / 2 + 1], a2 = own[k0 / 2 + 2];
        double a[6] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y};
        const double2 l0 = own[3 * jb], l1 = own[3 * jb + 1], l2 = own[3 * jb + 2];
#pragma unroll
        for (int c = 0; c < 6; c++) {
          const double2* dr = reinterpret_cast<const double2*>(Hs + (k0 + c) * LD) + 3 * jb;  // (the same address in every lane: a broadcast)
          const double2 d0 = dr[0], d1 = dr[1], d2 = dr[2];
          a[c] = fma(-l2.y, d2.y, fma(-l2.x, d2.x, fma(-l1.y, d1.y, fma(-l1.x, d1.x, fma(-l0.y, d0.y, fma(-l0.x, d0.x, a[c]))))));
        }
        own[k0 / 2] = double2{a[0], a[1]};
        own[k0 / 2 + 1] = double2{a[2], a[3]};
        own[k0 / 2 + 2] = double2{a[4], a[5]};
      }
    }
    __syncthreads();
  }
  BAPROF(5);
  return okc;
}

// ---- phases of one LM iteration.  Each is a separate (non-inlined) function so that its register allocation is its own:
// the phases share state only through LDS (BAShared) and the HBM scratch.

// ---- the factored observation blocks of the Schur phase.  For an observation of landmark l by free pose h, with the projection
// Jacobians Jp (2x6) and Jl (2x3), the Huber weight w and (Hll + lambda I) = G G^T:
//   Z = w Jp^T Jl G^-T = Jp^T M,  M = w Jl G^-T (2x3),   Z c = Jp^T (M c) = Jp^T r,   Z_i Z_j^T = Jp_i^T (M_i M_j^T) Jp_j,
// and Jp is a function of (x/z, y/z, 1/z) of the point in the camera (types_six_dof_expmap.cpp:389-433): the record of an item is
// (xn, yn, iz, w) and the pixel (u, v): three 16-byte pairs (ba_schur_buf); M is rebuilt per product from the record, the pose's
// rotation rows and the landmark's factor G (ba_record_M); c = G^-1 bl is kept per landmark.
struct JpRows {
  double a0, a1, a2, a3, a5;  // row 0 (entry 4 is zero)
  double b0, b1, b2, b4, b5;  // row 1 (entry 3 is zero)
};
FD JpRows jp_rows(double xn, double yn, double iz, double fx, double fy) {
  JpRows J;
  const double xy = xn * yn;
  J.a0 = xy * fx;
  J.a1 = -(1 + xn * xn) * fx;
  J.a2 = yn * fx;
  J.a3 = -iz * fx;
  J.a5 = xn * iz * fx;
  J.b0 = (1 + yn * yn) * fy;
  J.b1 = -xy * fy;
  J.b2 = -xn * fy;
  J.b4 = -iz * fy;
  J.b5 = yn * iz * fy;
  return J;
}
// M = (w Jl) G^-T row by row (G lower, inverted diagonal)
FD void ba_item_scale(const double (&wJl)[6], const Chol3& g, double (&M)[6]) {
#pragma unroll
  for (int a = 0; a < 2; a++) {
    M[3 * a] = wJl[3 * a] * g.i00;
    M[3 * a + 1] = (wJl[3 * a + 1] - M[3 * a] * g.g10) * g.i11;
    M[3 * a + 2] = (wJl[3 * a + 2] - M[3 * a] * g.g20 - M[3 * a + 1] * g.g21) * g.i22;
  }
}

// One observation at the linearisation point (EdgeSE3ProjectXYZ::computeError + linearizeOplus, types_six_dof_expmap.cpp:389-433, and
// RobustKernelHuber): normalised point (xn, yn, 1/z), residual, Huber weight, Jl (2x3, wrt the landmark) and w Jl.  The pose Jacobian
// Jp follows from (xn, yn, iz) alone (jp_rows).
struct BAObs {
  double xn, yn, iz, e0, e1, wgt;
  double Jl[6], wJl[6];
};
FD BAObs ba_obs(const double* rt, double px, double py, double pz, double u, double v, const double* K) {
  BAObs o;
  const double x = rt[0] * px + rt[1] * py + rt[2] * pz + rt[9];
  const double y = rt[3] * px + rt[4] * py + rt[5] * pz + rt[10];
  const double z = rt[6] * px + rt[7] * py + rt[8] * pz + rt[11];
  o.iz = rcp_nr(z);
  o.xn = x * o.iz;
  o.yn = y * o.iz;
  const double fx = K[0], fy = K[1];
  o.e0 = u - (o.xn * fx + K[2]);
  o.e1 = v - (o.yn * fy + K[3]);
  o.wgt = ba_huber_w(o.e0 * o.e0 + o.e1 * o.e1);
  const double ax = -o.iz * fx, ay = -o.iz * fy;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    o.Jl[c] = ax * (rt[c] - o.xn * rt[6 + c]);
    o.Jl[3 + c] = ay * (rt[3 + c] - o.yn * rt[6 + c]);
    o.wJl[c] = o.Jl[c] * o.wgt;
    o.wJl[3 + c] = o.Jl[3 + c] * o.wgt;
  }
  return o;
}
// the observation's share of its landmark's Hll (xx xy xz yy yz zz) / bl and of the robust chi2
FD double ba_obs_landmark(const BAObs& o, double (&h)[6], double (&bb)[3]) {
  const double o0 = -o.e0 * o.wgt, o1 = -o.e1 * o.wgt;
  int q = 0;
#pragma unroll
  for (int r = 0; r < 3; r++) {
    bb[r] += o.Jl[r] * o0 + o.Jl[3 + r] * o1;
#pragma unroll
    for (int c = r; c < 3; c++) h[q++] += o.wJl[r] * o.Jl[c] + o.wJl[3 + r] * o.Jl[3 + c];
  }
  return ba_huber_rho(o.e0 * o.e0 + o.e1 * o.e1);
}
// ... and of its pose's Hpp (21 upper entries, row-major) / bp (6): pv[0 .. 26]
FD void ba_obs_pose(const BAObs& o, const double* K, double (&pv)[32]) {
  const JpRows J = jp_rows(o.xn, o.yn, o.iz, K[0], K[1]);
  const double ja[6] = {J.a0, J.a1, J.a2, J.a3, 0.0, J.a5}, jb[6] = {J.b0, J.b1, J.b2, 0.0, J.b4, J.b5};
  const double o0 = -o.e0 * o.wgt, o1 = -o.e1 * o.wgt;
  int q = 0;
#pragma unroll
  for (int r = 0; r < 6; r++) {
    pv[21 + r] = ja[r] * o0 + jb[r] * o1;
#pragma unroll
    for (int c = r; c < 6; c++) pv[q++] = (ja[r] * o.wgt) * ja[c] + (jb[r] * o.wgt) * jb[c];
  }
}

// computeActiveErrors + buildSystem in one pass over the observations (thread per landmark, a wave walks the poses
// together): Hll / bl per landmark and per free pose the 21 + 6 entries of Hpp / bp, reduced over
// the wave with a scattered butterfly and accumulated per wave in LDS (fixed order -> reproducible).  Returns this
// thread's share of the robust chi2.  ba_phase_finish_poses() folds the per-wave partials afterwards.
// Resident records (sh.fused): the pixels come from the items' records in LDS (only the fixed pose's from HBM), and with lam > 0 --
// the lambda of the trial that follows is known from the second iteration on -- the landmark's thread also leaves the SCALED records
// and c = G^-1 bl behind, i.e. the staging of the Schur phase: that phase then starts with its products.
// the linearisation over resident records (sh.fused).  POSE_SUMS: also Hpp / bp, a wave walking the poses together (reduce-scatter
// per pose); without them every lane walks its own observations and nothing crosses lanes.  The records get the point in the camera
// and the Huber weight; with lam > 0 (the lambda of the trial that follows is known from the second iteration on) the landmark's thread
// also leaves G = chol(Hll + lam I) and c = G^-1 bl behind, i.e. the whole staging of the Schur phase.
template <bool POSE_SUMS>
FD double ba_linearize_resident(double lam) {
  BAShared& sh = ba_sh();
  const BAScratch sc = sh.sc;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, L = sh.L, Lc = sc.Lc, P = sh.P;
  const double K[4] = {sh.K[0], sh.K[1], sh.K[2], sh.K[3]};
  double* wacc = ba_dyn() + (size_t)wv * P * 27;  // [P][27] of this wave, over Hs (dead until the Schur phase's combine)
  if (POSE_SUMS) {
    for (int i = lane; i < P * 27; i += 64) wacc[i] = 0.0;
    wave_lds_fence();
  }
  double chi = 0;
  const int CI = sh.CI, CL = sh.CL, fs = sh.fixed_slot;
  const SchurBuf B = ba_schur_buf(0);
  double2* z = B.z;
  for (int l0 = wv * 64; l0 < L; l0 += BA_T) {
    const int l = l0 + lane;
    const unsigned m = l < L ? sc.omask[l] : 0u;
    double px = 0, py = 0, pz = 1, uf = 0, vf = 0;
    unsigned fm = 0;
    int ib = 0;
    const bool hf = fs >= 0 && ((m >> fs) & 1u);
    if (m) {
      px = sc.lmA[l];
      py = sc.lmA[Lc + l];
      pz = sc.lmA[2 * Lc + l];
      fm = sc.fmask[l];
      ib = sc.ibase[l];
      if (hf) {
        uf = sc.uv[(size_t)(2 * fs) * Lc + l];
        vf = sc.uv[(size_t)(2 * fs + 1) * Lc + l];
      }
    }
    double h[6] = {0, 0, 0, 0, 0, 0}, bb[3] = {0, 0, 0};
    if (hf) {
      const BAObs o = ba_obs(sh.RT[fs], px, py, pz, uf, vf, K);
      chi += ba_obs_landmark(o, h, bb);
    }
    int idx = ib;
    if (!POSE_SUMS) {
      unsigned rem = fm;
#pragma unroll 1
      while (rem) {
        const int hi = __builtin_ctz(rem);
        rem &= rem - 1u;
        const double2 uv = z[2 * CI + idx];
        const BAObs o = ba_obs(sh.RT[sh.slot_of[hi]], px, py, pz, uv.x, uv.y, K);
        chi += ba_obs_landmark(o, h, bb);
        z[idx] = double2{o.xn, o.yn};
        z[CI + idx] = double2{o.iz, o.wgt};
        idx++;
      }
    } else {
#pragma unroll 1
      for (int hi = 0; hi < P; hi++) {  // (not unrolled: instruction-cache footprint)
        const bool has = (fm >> hi) & 1u;
        if (__ballot(has) == 0ull) continue;
        double pv[32];
#pragma unroll
        for (int k = 0; k < 32; k++) pv[k] = 0;
        if (has) {
          const double2 uv = z[2 * CI + idx];
          const BAObs o = ba_obs(sh.RT[sh.slot_of[hi]], px, py, pz, uv.x, uv.y, K);
          chi += ba_obs_landmark(o, h, bb);
          ba_obs_pose(o, K, pv);
          z[idx] = double2{o.xn, o.yn};
          z[CI + idx] = double2{o.iz, o.wgt};
          idx++;
        }
        int k;
        const double tot = wave_reduce_scatter32(pv, k);
        if (!(lane & 1) && k < 27) wacc[hi * 27 + k] += tot;
        wave_lds_fence();
      }
    }
    if (m) {
#pragma unroll
      for (int j = 0; j < 6; j++) sc.Hll[(size_t)j * Lc + l] = h[j];
#pragma unroll
      for (int j = 0; j < 3; j++) sc.bl[(size_t)j * Lc + l] = bb[j];
    }
    if (lam > 0 && fm) {  // the Schur staging for the trial at `lam`
      const Chol3 g = chol3(h, lam);
      const double c0 = bb[0] * g.i00;
      const double c1 = (bb[1] - g.g10 * c0) * g.i11;
      const double c2 = (bb[2] - g.g20 * c0 - g.g21 * c1) * g.i22;
      B.cb[l] = c0, B.cb[CL + l] = c1, B.cb[2 * CL + l] = c2;
      B.gb[l] = g.i00, B.gb[CL + l] = g.g10, B.gb[2 * CL + l] = g.g20, B.gb[3 * CL + l] = g.i11, B.gb[4 * CL + l] = g.g21, B.gb[5 * CL + l] = g.i22;
    }
  }
  return chi;
}

// pose_sums == 0 (resident records only): Hpp / bp are not formed here at all -- the Schur phase sums them inside its own walk over the
// records (ba_schur_accumulate, inl).  A function of its own: few registers, no callee-saved ones to store and reload per call.
__device__ FLVIS_BA_PHASE_FN double ba_phase_linearize_rec(double lam) { return ba_linearize_resident<false>(lam); }

__device__ FLVIS_BA_PHASE_FN double ba_phase_linearize(double lam) {
  BAShared& sh = ba_sh();
  if (sh.fused) return ba_linearize_resident<true>(lam);
  const BAScratch sc = sh.sc;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, L = sh.L, Lc = sc.Lc, W = sh.W, P = sh.P;
  const double K[4] = {sh.K[0], sh.K[1], sh.K[2], sh.K[3]};
  double* wacc = ba_dyn() + (size_t)wv * P * 27;  // [P][27] of this wave, over Hs (dead until the Schur phase's combine)
  for (int i = lane; i < P * 27; i += 64) wacc[i] = 0.0;
  wave_lds_fence();
  double chi = 0;
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
    // the loop body is deliberately NOT unrolled (instruction-cache footprint); the next slot's observation is loaded
    // while the current one is processed
    double un = (m & 1u) ? sc.uv[l] : 0.0, vn = (m & 1u) ? sc.uv[(size_t)Lc + l] : 0.0;
#pragma unroll 1
    for (int slot = 0; slot < W; slot++) {
      const double u = un, v = vn;
      const bool has = (m >> slot) & 1u;
      if (slot + 1 < W) {
        const bool hn = (m >> (slot + 1)) & 1u;
        un = hn ? sc.uv[(size_t)(2 * slot + 2) * Lc + l] : 0.0;
        vn = hn ? sc.uv[(size_t)(2 * slot + 3) * Lc + l] : 0.0;
      }
      if (__ballot(has) == 0ull) continue;
      const int hi = sh.hidx_of[slot];
      double pv[32];
#pragma unroll
      for (int k = 0; k < 32; k++) pv[k] = 0;
      if (has) {
        const BAObs o = ba_obs(sh.RT[slot], px, py, pz, u, v, K);
        chi += ba_obs_landmark(o, h, bb);
        if (hi >= 0) ba_obs_pose(o, K, pv);
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
  const double* wacc = ba_dyn();
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

// "inline" iterations: Hpp / b hold the IMU edges' share only (ba_phase_imu_gather adds it); the observations' share is summed by the
// Schur phase
__device__ __noinline__ void ba_phase_zero_poses() {
  BAShared& sh = ba_sh();
  const int t = threadIdx.x, P = sh.P;
  for (int i = t; i < P * 36; i += BA_T) sh.Hpp[i / 36][i % 36] = 0.0;
  for (int i = t; i < P * 6; i += BA_T) sh.b[i] = 0.0;
}

// ---- optional IMU rotation factor between consecutive keyframes (north_star: "reprojection + IMU-preintegration factors"; the
// reference's window holds reprojection edges only).  With R_b = R_cw^T R_cb the body orientation of a keyframe and dq the gyro
// preintegration between keyframes a and b:   r = Log(dq^T R_b(a)^T R_b(b)) = Log(dq^T R_cb^T R_cw(a) R_cw(b)^T R_cb),
//   dr/domega_a = Jr^-1(r) R_cb^T (R_cw(a) R_cw(b)^T)^T,   dr/domega_b = -Jr^-1(r) R_cb^T     (g2o update T <- exp(dx) T),
// information w I3 with w = 1 / (sigma_g^2 dt).  The edges touch the rotation rows / columns (0..2) of the 6x6 pose blocks and
// add off-diagonal pose-pose blocks to the reduced system.
FD V3 imu_edge_residual(const double* Ta7, const double* Tb7, Q4 qcb, Q4 dq) {
  const Q4 qa = load_pose7(Ta7).q, qb = load_pose7(Tb7).q;
  Q4 qr = q_normalized(q_mul(q_mul(q_mul(q_mul(q_conj(dq), q_conj(qcb)), qa), q_conj(qb)), qcb));
  if (qr.w < 0) qr = Q4{-qr.w, -qr.x, -qr.y, -qr.z};
  return so3_log(qr);
}
// position rows of the factor:
//   r = R_b(a)^T (p_b(b) - p_b(a) - v_a dt + 1/2 g_w dt^2) - dp,   R_b = R_cw^T R_cb,  p_b = R_cw^T (t_cb - t_cw),  g_w = (0, 0, -9.81)
// (body attitude and position from the camera pose and the camera-from-body extrinsic; dp, v_a: see KeyFrameDev::imu_dp / imu_va).
// g2o's update T <- exp((omega, upsilon)) T moves p_b by R_cw^T ([t_cb]x omega - upsilon) and R_b by Exp(-R_cb^T omega) on the right:
//   dr/d omega_b = R_b(a)^T R_cw(b)^T [t_cb]x          dr/d upsilon_b = -R_b(a)^T R_cw(b)^T
//   dr/d omega_a = -R_cb^T [t_cb]x - [R_b(a)^T d]x R_cb^T      dr/d upsilon_a = R_cb^T          (d = the bracket of r)
FD V3 imu_edge_residual_pos(const double* Ta7, const double* Tb7, Q4 qcb, V3 tcb, V3 dp, V3 va, double dt, M3* RbaT_out, V3* d_out) {
  const SE3d Ta = load_pose7(Ta7), Tb = load_pose7(Tb7);
  const M3 Rca = q_to_mat(Ta.q), RcaT = transpose(Rca), RcbT = transpose(q_to_mat(Tb.q)), RciT = transpose(q_to_mat(qcb));
  const V3 pa = RcaT * (tcb - Ta.t), pb = RcbT * (tcb - Tb.t);
  const V3 gw{0, 0, -9.81};
  const V3 d = ((pb - pa) - dt * va) + (0.5 * dt * dt) * gw;
  const M3 RbaT = RciT * Rca;
  if (RbaT_out) *RbaT_out = RbaT;
  if (d_out) *d_out = d;
  return (RbaT * d) - dp;
}
// lanes of wave 0, one per edge: linearise at the accepted poses.  Rows 0..2: rotation (weight w), rows 3..5: position (weight wp,
// present when wp > 0); a rotation-only edge touches the rotation rows / columns of the pose blocks alone
__device__ __noinline__ void ba_phase_imu_linearize() {
  BAShared& sh = ba_sh();
  const int k = threadIdx.x;
  if (k >= sh.n_imu) return;
  const double* Ta = sh.pose[sh.imu_a[k]];
  const double* Tb = sh.pose[sh.imu_b[k]];
  const Q4 qcb{sh.q_c_b[0], sh.q_c_b[1], sh.q_c_b[2], sh.q_c_b[3]};
  const Q4 dq{sh.imu_dq[k][0], sh.imu_dq[k][1], sh.imu_dq[k][2], sh.imu_dq[k][3]};
  const V3 rv = imu_edge_residual(Ta, Tb, qcb, dq);
  const double w = sh.imu_w[k], wp = sh.imu_wp[k];
  const bool pos = wp > 0;
  const M3 Bt = transpose(q_to_mat(qcb));
  const M3 M = q_to_mat(load_pose7(Ta).q) * transpose(q_to_mat(load_pose7(Tb).q));
  const M3 Ji = so3_jr_inv(rv);
  const M3 A = Ji * (Bt * transpose(M));
  const M3 Bm = Ji * Bt;  // Jb (rotation rows) = -Bm
  double r[6] = {rv.x, rv.y, rv.z, 0, 0, 0};
  double Ja[6][6], Jb[6][6];
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j < 6; j++) Ja[i][j] = Jb[i][j] = 0;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      Ja[i][j] = A.m[i][j];
      Jb[i][j] = -Bm.m[i][j];
    }
  if (pos) {
    const V3 tcb{sh.t_c_b[0], sh.t_c_b[1], sh.t_c_b[2]};
    M3 RbaT;
    V3 d;
    const V3 rp = imu_edge_residual_pos(Ta, Tb, qcb, tcb, V3{sh.imu_dp[k][0], sh.imu_dp[k][1], sh.imu_dp[k][2]},
                                        V3{sh.imu_va[k][0], sh.imu_va[k][1], sh.imu_va[k][2]}, sh.imu_dtk[k], &RbaT, &d);
    r[3] = rp.x, r[4] = rp.y, r[5] = rp.z;
    const M3 RcbT = transpose(q_to_mat(load_pose7(Tb).q));
    const M3 Sx = skew(tcb);
    const M3 RR = RbaT * RcbT;
    const M3 Job = RR * Sx;
    const M3 t1 = Bt * Sx, t2 = skew(RbaT * d) * Bt;  // (Bt = R_cb^T)
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        Ja[3 + i][j] = (0.0 - t1.m[i][j]) - t2.m[i][j];
        Ja[3 + i][3 + j] = Bt.m[i][j];
        Jb[3 + i][j] = Job.m[i][j];
        Jb[3 + i][3 + j] = 0.0 - RR.m[i][j];
      }
  }
  const int nr = pos ? 6 : 3;
  const double wr[6] = {w, w, w, wp, wp, wp};
#pragma unroll
  for (int i = 0; i < 6; i++) {
#pragma unroll
    for (int j = 0; j < 6; j++) {
      double aa = 0, bb = 0, ab = 0;
#pragma unroll
      for (int m = 0; m < 6; m++) {
        if (m < nr && i < nr && j < nr) {
          aa += (Ja[m][i] * wr[m]) * Ja[m][j];
          bb += (Jb[m][i] * wr[m]) * Jb[m][j];
          ab += (Ja[m][i] * wr[m]) * Jb[m][j];
        }
      }
      ba_imu_blk(k, 0)[6 * i + j] = aa;
      ba_imu_blk(k, 1)[6 * i + j] = bb;
      ba_imu_blk(k, 2)[6 * i + j] = ab;
    }
    double ga = 0, gb = 0;
#pragma unroll
    for (int m = 0; m < 6; m++) {
      if (m < nr && i < nr) {
        ga += (Ja[m][i] * wr[m]) * r[m];
        gb += (Jb[m][i] * wr[m]) * r[m];
      }
    }
    ba_imu_blk(k, 3)[i] = ga;
    ba_imu_blk(k, 4)[i] = gb;
  }
  double chi = w * ((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
  if (pos) chi += wp * ((r[3] * r[3] + r[4] * r[4]) + r[5] * r[5]);
  sh.imu_chi[k] = chi;
}
// one thread per free pose: gather the blocks of its (at most two) edges into Hpp / b, in edge order (call after a barrier)
__device__ __noinline__ void ba_phase_imu_gather() {
  BAShared& sh = ba_sh();
  const int hi = threadIdx.x;
  if (hi >= sh.P) return;
  const int slot = sh.slot_of[hi];
  for (int k = 0; k < sh.n_imu; k++) {
    const bool isa = sh.imu_a[k] == slot, isb = sh.imu_b[k] == slot;
    if (!isa && !isb) continue;
    const double* blk = ba_imu_blk(k, isa ? 0 : 1);
    const double* g = ba_imu_blk(k, isa ? 3 : 4);
    const int nc = sh.imu_wp[k] > 0 ? 6 : 3;
    for (int i = 0; i < nc; i++) {
      for (int j = 0; j < nc; j++) sh.Hpp[hi][6 * i + j] += blk[6 * i + j];
      sh.b[6 * hi + i] -= g[i];
    }
  }
}
// off-diagonal pose-pose blocks Ja^T W Jb into the lower triangle of the reduced system (after the Schur phase and a barrier)
__device__ __noinline__ void ba_phase_imu_offdiag() {
  BAShared& sh = ba_sh();
  // one item per (edge, entry of its 6x6 block), strided: window_size 16 has up to 15 edges = 540 items > BA_T.  Every item owns
  // its entry of the reduced system (an edge joins one pair of poses), so the order of the items does not matter
  double* Hs = ba_dyn();
  const int LD = sh.LD;
  for (int t = threadIdx.x; t < 36 * sh.n_imu; t += BA_T) {
    const int k = t / 36, e = t - 36 * k, i = e / 6, j = e - 6 * i;
    const int nc = sh.imu_wp[k] > 0 ? 6 : 3;
    if (i >= nc || j >= nc) continue;
    const int ia = sh.hidx_of[sh.imu_a[k]], ib = sh.hidx_of[sh.imu_b[k]];
    if (ia < 0 || ib < 0) continue;
    if (ia < ib)
      Hs[(6 * ib + j) * LD + 6 * ia + i] += ba_imu_blk(k, 2)[e];
    else
      Hs[(6 * ia + i) * LD + 6 * ib + j] += ba_imu_blk(k, 2)[e];
  }
}
// lanes of wave 0: chi2 of the edges at the trial poses
FD void ba_phase_imu_trial() {
  BAShared& sh = ba_sh();
  const int k = threadIdx.x;
  if (k >= sh.n_imu) return;
  const Q4 qcb{sh.q_c_b[0], sh.q_c_b[1], sh.q_c_b[2], sh.q_c_b[3]};
  const Q4 dq{sh.imu_dq[k][0], sh.imu_dq[k][1], sh.imu_dq[k][2], sh.imu_dq[k][3]};
  const V3 r = imu_edge_residual(sh.poseT[sh.imu_a[k]], sh.poseT[sh.imu_b[k]], qcb, dq);
  double chi = sh.imu_w[k] * ((r.x * r.x + r.y * r.y) + r.z * r.z);
  if (sh.imu_wp[k] > 0) {
    const V3 rp = imu_edge_residual_pos(sh.poseT[sh.imu_a[k]], sh.poseT[sh.imu_b[k]], qcb, V3{sh.t_c_b[0], sh.t_c_b[1], sh.t_c_b[2]},
                                        V3{sh.imu_dp[k][0], sh.imu_dp[k][1], sh.imu_dp[k][2]},
                                        V3{sh.imu_va[k][0], sh.imu_va[k][1], sh.imu_va[k][2]}, sh.imu_dtk[k], nullptr, nullptr);
    chi += sh.imu_wp[k] * ((rp.x * rp.x + rp.y * rp.y) + rp.z * rp.z);
  }
  sh.imu_chit[k] = chi;
}

// largest diagonal entry of the (unreduced) hessian -> initial lambda (computeLambdaInit); this thread's share
__device__ FLVIS_BA_PHASE_FN double ba_phase_max_diag() {
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

// reduced camera system S = Hpp + lambda I - sum_l Z Z^T (lower triangle into Hs) and rhs = bp - sum_l Z c (into sh.x).
// Per chunk of landmarks (the whole window when its items fit one buffer):
//   stage       thread per landmark: G = chol(Hll + lambda I), c = G^-1 bl, then one record per observation by a free pose (re-linearised
//               at the accepted state from the landmark, the pose table in LDS and the pixel) straight into LDS -- no HBM scratch;
//   accumulate  thread = (pose pair, landmark slice): 6x6 register tiles from the factored products, fixed slice partition and fixed
//               butterfly order => bit-reproducible run to run, no atomics; the rhs threads sum Jp^T r likewise.
// ---- The Schur phase is split into functions with small register footprints: a called function that needs more than the ~140
// caller-saved VGPRs saves and reloads every callee-saved one it touches through scratch memory on each call, and the stores have to
// drain before its first barrier (measured in round 5: 65 registers, ~9 us per call).  The accumulation alone stays below that.

// this thread's role in the accumulation: pose pair (i1, i2) and landmark slice sl -- the diagonal pairs first (pr < P: (pr, pr)), then
// the pairs i1 < i2 row by row.  A diagonal pair walks every observation of its pose: it also sums the pose's right-hand side.
struct SchurRole {
  int i1, i2, sl, slices;
};
FD SchurRole ba_schur_role(int t, int P, int npairs, int slices) {
  SchurRole r{-1, -1, 0, slices};
  if (ba_sh().balanced) {  // (ba_balance_pairs)
    const BAShared& sh = ba_sh();
    const int pr = sh.role_pair[t];
    if (pr != 255) {
      const unsigned tab = sh.pair_tab[pr];
      r.sl = t - (int)(tab & 0xfffu);
      r.slices = 1 << ((tab >> 12) & 15u);
      r.i1 = (int)((tab >> 16) & 15u);
      r.i2 = (int)((tab >> 20) & 15u);
    }
    return r;
  }
  if (t < npairs * slices) {
    const int pr = t / slices, ppw = 64 / slices, dpad = ((P + ppw - 1) / ppw) * ppw;  // (diagonal pairs padded to whole waves)
    r.sl = t - pr * slices;
    if (pr < P) {
      r.i1 = r.i2 = pr;
    } else if (pr >= dpad) {
      int i1 = 0, rem = pr - dpad;
      while (rem >= P - 1 - i1) {
        rem -= P - 1 - i1;
        i1++;
      }
      r.i1 = i1;
      r.i2 = i1 + 1 + rem;
    }
  }
  return r;
}

// staging of one chunk, thread per landmark: G = chol(Hll + lambda I) and c = G^-1 bl.  Resident records (fused) are current from the
// linearisation: nothing per observation is left to do.  Otherwise the chunk's records are rebuilt here -- every observation by a free
// pose re-linearised at the accepted state from the landmark, the pose table in LDS and the pixel in the observation table (HBM), the
// next pose's in flight while this one is processed.
FD void ba_stage_chunk(int c, int buf, double lambda) {
  BAShared& sh = ba_sh();
  const BAScratch sc = sh.sc;
  const int t = threadIdx.x, Lc = sc.Lc, P = sh.P, CI = sh.CI, CL = sh.CL;
  const bool fused = sh.fused != 0;
  const double K[4] = {sh.K[0], sh.K[1], sh.K[2], sh.K[3]};
  const int l0 = sh.chunk_l0[c], nl = sh.chunk_l0[c + 1] - l0, i0 = sh.chunk_i0[c];
  const SchurBuf B = ba_schur_buf(buf);
  double2* z = B.z;
  for (int b0 = 0; b0 < nl; b0 += BA_T) {  // (wave-uniform trip count: the pose walk below votes)
    const int ll = b0 + t, l = l0 + ll;
    const bool in = ll < nl;
    const unsigned fm = in ? sc.fmask[l] : 0u;
    int base = 0;
    if (in) {
      base = sc.ibase[l] - i0;
      B.mb[ll] = fm;
      B.lb[ll] = base;
    }
    double px = 0, py = 0, pz = 1;
    if (fm) {
      double H[6];
#pragma unroll
      for (int k = 0; k < 6; k++) H[k] = sc.Hll[(size_t)k * Lc + l];
      const double b0l = sc.bl[l], b1l = sc.bl[(size_t)Lc + l], b2l = sc.bl[(size_t)2 * Lc + l];
      if (!fused) {
        px = sc.lmA[l];
        py = sc.lmA[Lc + l];
        pz = sc.lmA[2 * Lc + l];
      }
      const Chol3 g = chol3(H, lambda);
      const double c0 = b0l * g.i00;  // c = G^-1 bl
      const double c1 = (b1l - g.g10 * c0) * g.i11;
      const double c2 = (b2l - g.g20 * c0 - g.g21 * c1) * g.i22;
      B.cb[ll] = c0, B.cb[CL + ll] = c1, B.cb[2 * CL + ll] = c2;
      B.gb[ll] = g.i00, B.gb[CL + ll] = g.g10, B.gb[2 * CL + ll] = g.g20, B.gb[3 * CL + ll] = g.i11, B.gb[4 * CL + ll] = g.g21, B.gb[5 * CL + ll] = g.i22;
    }
    if (fused) continue;
    int idx = base;
    double un = 0, vn = 0;
    if (fm & 1u) {
      const int s0 = sh.slot_of[0];
      un = sc.uv[(size_t)(2 * s0) * Lc + l];
      vn = sc.uv[(size_t)(2 * s0 + 1) * Lc + l];
    }
#pragma unroll 1
    for (int h = 0; h < P; h++) {
      const double u = un, v = vn;
      const bool has = (fm >> h) & 1u;
      if (h + 1 < P && ((fm >> (h + 1)) & 1u)) {
        const int sn = sh.slot_of[h + 1];
        un = sc.uv[(size_t)(2 * sn) * Lc + l];
        vn = sc.uv[(size_t)(2 * sn + 1) * Lc + l];
      }
      if (__ballot(has) == 0ull) continue;
      if (has) {
        const BAObs o = ba_obs(sh.RT[sh.slot_of[h]], px, py, pz, u, v, K);
        z[idx] = double2{o.xn, o.yn};
        z[CI + idx] = double2{o.iz, o.wgt};
        z[2 * CI + idx] = double2{u, v};
        idx++;
      }
    }
  }
}

// M = w Jl G^-T (2x3) of a record for the pose whose rotation rows are R (row-major 3x3): Jl = -1/z [fx (r0 - xn r2); fy (r1 - yn r2)]
// (types_six_dof_expmap.cpp:389-433), rebuilt from four numbers instead of being stored
FD void ba_record_M(const double2 p0, const double2 p1, const double* R, const double* g6, double fx, double fy, double (&M)[6]) {
  const double s = -p1.x * p1.y, sx = s * fx, sy = s * fy;
  double A[6];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    A[c] = sx * (R[c] - p0.x * R[6 + c]);
    A[3 + c] = sy * (R[3 + c] - p0.y * R[6 + c]);
  }
#pragma unroll
  for (int a = 0; a < 2; a++) {
    M[3 * a] = A[3 * a] * g6[0];
    M[3 * a + 1] = (A[3 * a + 1] - M[3 * a] * g6[1]) * g6[3];
    M[3 * a + 2] = (A[3 * a + 2] - M[3 * a] * g6[2] - M[3 * a + 1] * g6[4]) * g6[5];
  }
}

// sum over one DPP row (16 lanes) into its lane 0, in a fixed order; the other lanes end with partial sums
template <int CTRL>
FD double ba_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int ROWS>
FD double ba_dpp_bcast(double v) {  // (rows outside ROWS read 0)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWS, 0xf, false);
  return __hiloint2double(hi, lo);
}
FD double ba_row16_sum_to_lane0(double v) {
  v += ba_dpp<0x108>(v);  // row_shl:8  (lane i += lane i + 8; out-of-row sources read as 0)
  v += ba_dpp<0x104>(v);  // row_shl:4
  v += ba_dpp<0x102>(v);  // row_shl:2
  v += ba_dpp<0x101>(v);  // row_shl:1
  return v;
}

// accumulation over one staged chunk, acc = this thread's 36 accumulators:
//   pair i1 < i2   acc[6 r + c] = - sum Z_i1 Z_i2^T, the 6x6 tile;
//   pair (h, h)    acc[r (r + 1) / 2 + c], c <= r: the lower triangle of (Hpp share) - Z_h Z_h^T (21), acc[21 .. 26] = sum Z_h c,
//                  acc[27 .. 32] = the bp share -- the diagonal pair walks every observation of pose h, so the right-hand side rides along.
// inl: Hpp and bp are summed here too: the tile accumulates Jp^T (w I - M M^T) Jp and Jp^T (-w e); sh.Hpp / sh.b then hold the IMU
// edges' share only.
FD void ba_schur_accumulate(const SchurRole& ro, int buf, int nl, int inl, double (&acc)[36]) {
  BAShared& sh = ba_sh();
  if (ro.i1 < 0) return;
  const int CI = sh.CI, CL = sh.CL, slices = ro.slices;
  const double fx = sh.K[0], fy = sh.K[1], cx = sh.K[2], cy = sh.K[3];
  const SchurBuf B = ba_schur_buf(buf);
  const double2* z2 = B.z;
  const unsigned* mb = B.mb;
  const int* lb = B.lb;
  const bool diag = ro.i1 == ro.i2;
  const unsigned need = (1u << ro.i1) | (1u << ro.i2);
  const unsigned lt1 = (1u << ro.i1) - 1u, lt2 = (1u << ro.i2) - 1u;
  double Ri[9], Rj[9];  // the two poses' rotations: fixed per thread
#pragma unroll
  for (int k = 0; k < 9; k++) {
    Ri[k] = sh.RT[sh.slot_of[ro.i1]][k];
    Rj[k] = sh.RT[sh.slot_of[ro.i2]][k];
  }
  // Two passes over this thread's landmarks of the chunk (sl, sl + slices, ...).  First the masks only, 8 LDS reads in flight at
  // a time: which landmarks are seen by both poses -> one bit each.  Then the block products, every lane walking the set bits of
  // ITS word: a wave runs the product body as often as its busiest lane has landmarks, not once per landmark ANY of its 64 lanes has.
  const int nmine = nl > ro.sl ? (nl - ro.sl + slices - 1) / slices : 0;
  for (int jb = 0; jb < nmine; jb += 32) {
    unsigned hit = 0;
    const int jn = nmine - jb < 32 ? nmine - jb : 32;
    for (int j0 = 0; j0 < jn; j0 += 8) {
      unsigned mk[8];
#pragma unroll
      for (int q = 0; q < 8; q++) mk[q] = j0 + q < jn ? mb[ro.sl + (jb + j0 + q) * slices] : 0u;
#pragma unroll
      for (int q = 0; q < 8; q++)
        if ((mk[q] & need) == need) hit |= 1u << (j0 + q);
    }
#pragma unroll 1
    while (hit) {
      const int j = __builtin_ctz(hit);
      hit &= hit - 1u;
      const int ll = ro.sl + (jb + j) * slices;
      const unsigned m = mb[ll];
      const int base = lb[ll];
      const double g6[6] = {B.gb[ll], B.gb[CL + ll], B.gb[2 * CL + ll], B.gb[3 * CL + ll], B.gb[4 * CL + ll], B.gb[5 * CL + ll]};
      const double2* zi = z2 + base + __popc(m & lt1);
      const double2 p0 = zi[0], p1 = zi[CI];
      double Mi[6];
      ba_record_M(p0, p1, Ri, g6, fx, fy, Mi);
      const JpRows Ji = jp_rows(p0.x, p0.y, p1.x, fx, fy);
      const double ia[6] = {Ji.a0, Ji.a1, Ji.a2, Ji.a3, 0.0, Ji.a5}, ib[6] = {Ji.b0, Ji.b1, Ji.b2, 0.0, Ji.b4, Ji.b5};
      if (diag) {
        // N = -M M^T (+ w I on an inline iteration), symmetric; T = Jp^T N; lower triangle += T Jp
        double n00 = -fma(Mi[2], Mi[2], fma(Mi[1], Mi[1], Mi[0] * Mi[0])), n01 = -fma(Mi[2], Mi[5], fma(Mi[1], Mi[4], Mi[0] * Mi[3]));
        double n11 = -fma(Mi[5], Mi[5], fma(Mi[4], Mi[4], Mi[3] * Mi[3]));
        if (inl) {
          n00 += p1.y;
          n11 += p1.y;
        }
#pragma unroll
        for (int r = 0; r < 6; r++) {
          double t0, t1;
          if (r == 3) {
            t0 = ia[3] * n00, t1 = ia[3] * n01;
          } else if (r == 4) {
            t0 = ib[4] * n01, t1 = ib[4] * n11;
          } else {
            t0 = fma(ib[r], n01, ia[r] * n00), t1 = fma(ib[r], n11, ia[r] * n01);
          }
#pragma unroll
          for (int c = 0; c <= r; c++) {
            const int k = r * (r + 1) / 2 + c;
            if (c == 3)
              acc[k] = fma(t0, ia[3], acc[k]);
            else if (c == 4)
              acc[k] = fma(t1, ib[4], acc[k]);
            else
              acc[k] = fma(t1, ib[c], fma(t0, ia[c], acc[k]));
          }
        }
        // right-hand side: Z c = Jp^T (M c), and on an inline iteration the bp share Jp^T (-w e)
        const double c0 = B.cb[ll], c1 = B.cb[CL + ll], c2 = B.cb[2 * CL + ll];
        const double r0 = fma(Mi[2], c2, fma(Mi[1], c1, Mi[0] * c0)), r1 = fma(Mi[5], c2, fma(Mi[4], c1, Mi[3] * c0));
        acc[21] = fma(Ji.b0, r1, fma(Ji.a0, r0, acc[21]));
        acc[22] = fma(Ji.b1, r1, fma(Ji.a1, r0, acc[22]));
        acc[23] = fma(Ji.b2, r1, fma(Ji.a2, r0, acc[23]));
        acc[24] = fma(Ji.a3, r0, acc[24]);
        acc[25] = fma(Ji.b4, r1, acc[25]);
        acc[26] = fma(Ji.b5, r1, fma(Ji.a5, r0, acc[26]));
        if (inl) {
          const double2 uv = zi[2 * CI];
          const double e0 = uv.x - (p0.x * fx + cx), e1 = uv.y - (p0.y * fy + cy);
          const double o0 = -e0 * p1.y, o1 = -e1 * p1.y;
          acc[27] = fma(Ji.b0, o1, fma(Ji.a0, o0, acc[27]));
          acc[28] = fma(Ji.b1, o1, fma(Ji.a1, o0, acc[28]));
          acc[29] = fma(Ji.b2, o1, fma(Ji.a2, o0, acc[29]));
          acc[30] = fma(Ji.a3, o0, acc[30]);
          acc[31] = fma(Ji.b4, o1, acc[31]);
          acc[32] = fma(Ji.b5, o1, fma(Ji.a5, o0, acc[32]));
        }
      } else {
        const double2* zj = z2 + base + __popc(m & lt2);
        const double2 q0 = zj[0], q1 = zj[CI];
        double Mj[6];
        ba_record_M(q0, q1, Rj, g6, fx, fy, Mj);
        const JpRows Jj = jp_rows(q0.x, q0.y, q1.x, fx, fy);
        // N = -M_i M_j^T; T = Jp_i^T N (6x2); tile += T Jp_j
        const double n00 = -fma(Mi[2], Mj[2], fma(Mi[1], Mj[1], Mi[0] * Mj[0])), n01 = -fma(Mi[2], Mj[5], fma(Mi[1], Mj[4], Mi[0] * Mj[3]));
        const double n10 = -fma(Mi[5], Mj[2], fma(Mi[4], Mj[1], Mi[3] * Mj[0])), n11 = -fma(Mi[5], Mj[5], fma(Mi[4], Mj[4], Mi[3] * Mj[3]));
#pragma unroll
        for (int r = 0; r < 6; r++) {
          double t0, t1;
          if (r == 3) {
            t0 = ia[3] * n00, t1 = ia[3] * n01;
          } else if (r == 4) {
            t0 = ib[4] * n10, t1 = ib[4] * n11;
          } else {
            t0 = fma(ib[r], n10, ia[r] * n00), t1 = fma(ib[r], n11, ia[r] * n01);
          }
          acc[6 * r + 0] = fma(t1, Jj.b0, fma(t0, Jj.a0, acc[6 * r + 0]));
          acc[6 * r + 1] = fma(t1, Jj.b1, fma(t0, Jj.a1, acc[6 * r + 1]));
          acc[6 * r + 2] = fma(t1, Jj.b2, fma(t0, Jj.a2, acc[6 * r + 2]));
          acc[6 * r + 3] = fma(t0, Jj.a3, acc[6 * r + 3]);
          acc[6 * r + 4] = fma(t1, Jj.b4, acc[6 * r + 4]);
          acc[6 * r + 5] = fma(t1, Jj.b5, fma(t0, Jj.a5, acc[6 * r + 5]));
        }
      }
    }
  }
}

// combine the slices (fixed order) and write S (lower triangle + the full diagonal blocks) / rhs
FD void ba_schur_combine(const SchurRole& ro, double lambda, double (&acc)[36]) {
  BAShared& sh = ba_sh();
  double* Hs = ba_dyn();
  const int LD = sh.LD, slices = sh.slices;
  if (sh.balanced) {
    // a pair's lanes are contiguous and aligned to their (power of two) count: summed into the group's LAST lane, inside a DPP row by
    // row shifts (a step as wide as the group or wider is switched off by a factor 0: the lane it would read belongs to another pair),
    // across rows by the two row broadcasts -- no LDS
    const double k4 = ro.slices >= 8 ? 1.0 : 0.0, k8 = ro.slices >= 16 ? 1.0 : 0.0, k16 = ro.slices >= 32 ? 1.0 : 0.0,
                 k32 = ro.slices >= 64 ? 1.0 : 0.0;
    const bool wide = sh.max_slices > 16;
#pragma unroll
    for (int k = 0; k < 36; k++) {
      double v = acc[k];
      v += ba_dpp<0x111>(v);  // row_shr:1 (lane i += lane i - 1; out-of-row sources read as 0)
      v += ba_dpp<0x112>(v);  // row_shr:2
      v = fma(ba_dpp<0x114>(v), k4, v);
      v = fma(ba_dpp<0x118>(v), k8, v);
      if (wide) {
        v = fma(ba_dpp_bcast<0x142, 0xa>(v), k16, v);  // row_bcast:15 into rows 1 and 3
        v = fma(ba_dpp_bcast<0x143, 0xc>(v), k32, v);  // row_bcast:31 into rows 2 and 3
      }
      acc[k] = v;
    }
    if (ro.i1 < 0 || ro.sl != ro.slices - 1) return;
  } else {
  if (slices == 16) {
#pragma unroll
    for (int k = 0; k < 36; k++) acc[k] = ba_row16_sum_to_lane0(acc[k]);
  } else {
    for (int off = slices >> 1; off > 0; off >>= 1) {
#pragma unroll
      for (int k = 0; k < 36; k++) acc[k] += __shfl_xor(acc[k], off, 64);
    }
  }
  if (ro.i1 < 0 || ro.sl != 0) return;
  }
  if (ro.i1 != ro.i2) {
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int cc = 0; cc < 6; cc++) Hs[(6 * ro.i2 + cc) * LD + 6 * ro.i1 + r] = acc[6 * r + cc];
    return;
  }
  const int h = ro.i1;
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int c = 0; c <= r; c++) {
      const double v = acc[r * (r + 1) / 2 + c] + sh.Hpp[h][6 * r + c] + (r == c ? lambda : 0.0);
      Hs[(6 * h + r) * LD + 6 * h + c] = v;
      Hs[(6 * h + c) * LD + 6 * h + r] = v;
    }
#pragma unroll
  for (int r = 0; r < 6; r++) {
    const double bf = sh.b[6 * h + r] + acc[27 + r];  // (the bp share is zero unless inl)
    sh.bfull[6 * h + r] = bf;
    sh.x[6 * h + r] = bf - acc[21 + r];  // bschur
  }
}

// reduced camera system S = Hpp + lambda I - sum_l Z Z^T (lower triangle into Hs) and rhs = bp - sum_l Z c (into sh.x):
// accumulate -- thread = (pose pair, landmark slice): 6x6 register tiles from the factored products, fixed slice partition and fixed
// butterfly order => bit-reproducible run to run, no atomics; the rhs threads sum Jp^T r likewise.

// resident records, staged by the caller (ba_phase_stage) or by the linearisation: one chunk, no barrier inside
__device__ FLVIS_BA_PHASE_FN void ba_phase_schur_resident(double lambda, int inl) {
  BAShared& sh = ba_sh();
  const SchurRole ro = ba_schur_role(threadIdx.x, sh.P, sh.npairs, sh.slices);
  double acc[36];
#pragma unroll
  for (int k = 0; k < 36; k++) acc[k] = 0;
#ifdef FLVIS_BA_PROF
  const long long t0_ = (long long)wall_clock64();
#endif
  ba_schur_accumulate(ro, 0, sh.chunk_l0[1] - sh.chunk_l0[0], inl, acc);
#ifdef FLVIS_BA_PROF
  // (how long every wave's accumulation takes: debug counters 40 .. 47)
  if ((threadIdx.x & 63) == 0 && sh.prof) atomicAdd((unsigned long long*)&sh.prof[32 + (threadIdx.x >> 6)], (unsigned long long)((long long)wall_clock64() - t0_));
#endif
  BAPROF(12);
  ba_schur_combine(ro, lambda, acc);
  BAPROF(13);
}
// the staging of the resident chunk on its own (the first trial of an optimize() call, and a trial that follows a rejected one)
__device__ FLVIS_BA_PHASE_FN void ba_phase_stage(double lambda) { ba_stage_chunk(0, 0, lambda); }

// windows whose items do not fit one buffer: staged chunk by chunk through two buffers, one barrier per chunk
__device__ FLVIS_BA_PHASE_FN void ba_phase_schur_chunked(double lambda) {
  BAShared& sh = ba_sh();
  const int nchunk = sh.nchunk;
  const SchurRole ro = ba_schur_role(threadIdx.x, sh.P, sh.npairs, sh.slices);
  double acc[36];
#pragma unroll
  for (int k = 0; k < 36; k++) acc[k] = 0;
  if (nchunk > 0) ba_stage_chunk(0, 0, lambda);
  for (int c = 0; c < nchunk; c++) {
    __syncthreads();
    BAPROF(2);
    ba_schur_accumulate(ro, c & 1, sh.chunk_l0[c + 1] - sh.chunk_l0[c], 0, acc);
    BAPROF(12);
    if (c + 1 < nchunk) ba_stage_chunk(c + 1, (c + 1) & 1, lambda);
  }
  ba_schur_combine(ro, lambda, acc);
  BAPROF(13);
}

// The same reduced system through the matrix cores (north_star: "an MFMA dense solve only for the reduced camera block").
// Z' = [Z; c^T] is formed DENSE, (NR + 1) rows padded to NRp (a multiple of 16) by 3 columns per landmark, in LDS chunks of CLm
// landmarks, k-major (Zt[column][NRp]); unobserved (landmark, pose) blocks are written as zeros.  S' = Z' Z'^T is a SYRK:
// every 16 x 16 tile of its lower triangle is accumulated by one wave with v_mfma_f64_16x16x4_f64 over all columns (A = rows
// of tile ti, B = rows of tile tj, both read from the same Zt).  S' holds sum Z Z^T in its leading NR x NR block and sum Z c in
// row NR.  More arithmetic than the sparse register-tile version (zeros are multiplied too), on units that are otherwise idle.
typedef double ba_d4 __attribute__((ext_vector_type(4)));
__device__ __noinline__ void ba_phase_schur_mfma(double lambda) {
  BAShared& sh = ba_sh();
  const BAScratch sc = sh.sc;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, Lc = sc.Lc, P = sh.P, L = sh.L;
  const int NR = sh.NR, NRp = sh.NRp, CLm = sh.CLm, LD = sh.LD;
  double* Hs = ba_dyn();
  double* Zt = Hs + sh.off_stage;
  const int nt = NRp >> 4, ntiles = nt * (nt + 1) / 2;
  // this wave's tiles (lower triangle, row-major enumeration): slot q handles tile index wv + q * BA_NW
  constexpr int MAXQ = 3;  // 21 tiles (W = 16) over 8 waves
  int ti[MAXQ], tj[MAXQ];
  ba_d4 acc[MAXQ];
#pragma unroll
  for (int q = 0; q < MAXQ; q++) {
    const int idx = wv + q * BA_NW;
    ti[q] = -1;
    tj[q] = 0;
    if (idx < ntiles) {
      int r = 0, rem = idx;
      while (rem > r) {
        rem -= r + 1;
        r++;
      }
      ti[q] = r;
      tj[q] = rem;
    }
    acc[q] = ba_d4{0, 0, 0, 0};
  }
  const int a_row = lane & 15, a_k = lane >> 4;
  for (int c = 0; c < sh.nchunk_m; c++) {
    const int l0 = c * CLm, nl = (L - l0 < CLm) ? L - l0 : CLm;
    const int ncol = (3 * nl + 3) & ~3;
    __syncthreads();  // the previous chunk's MFMAs are done with Zt
    // stage: one (landmark, free pose) block per thread step -- Z = B G^-T or zeros; then the c row and the padding
    for (int idx = t; idx < nl * P; idx += BA_T) {
      const int ll = idx / P, h = idx - ll * P, l = l0 + ll;
      const unsigned fm = sc.fmask[l];
      double zz[18];
      if ((fm >> h) & 1u) {  // Z = Jp^T M from the factors (see ba_phase_schur)
        const int slot = sh.slot_of[h];
        double rH[6], M[6];
#pragma unroll
        for (int k = 0; k < 6; k++) rH[k] = sc.Hll[(size_t)k * Lc + l];
        const Chol3 g = chol3(rH, lambda);
        const BAObs o = ba_obs(sh.RT[slot], sc.lmA[l], sc.lmA[Lc + l], sc.lmA[2 * Lc + l], sc.uv[(size_t)(2 * slot) * Lc + l],
                               sc.uv[(size_t)(2 * slot + 1) * Lc + l], sh.K);
        ba_item_scale(o.wJl, g, M);
        const JpRows J = jp_rows(o.xn, o.yn, o.iz, sh.K[0], sh.K[1]);
        const double ja[6] = {J.a0, J.a1, J.a2, J.a3, 0.0, J.a5}, jb[6] = {J.b0, J.b1, J.b2, 0.0, J.b4, J.b5};
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
          for (int j = 0; j < 3; j++) zz[3 * r + j] = ja[r] * M[j] + jb[r] * M[3 + j];
      } else {
#pragma unroll
        for (int k = 0; k < 18; k++) zz[k] = 0.0;
      }
#pragma unroll
      for (int j = 0; j < 3; j++) {
        double* col = Zt + (size_t)(3 * ll + j) * NRp + 6 * h;
#pragma unroll
        for (int r = 0; r < 6; r++) col[r] = zz[3 * r + j];
      }
    }
    for (int ll = t; ll < nl; ll += BA_T) {  // row NR: c = G^-1 bl; rows above it up to NRp: zero
      const int l = l0 + ll;
      double cv[3] = {0, 0, 0};
      if (sc.fmask[l]) {
        double rH2[6];
#pragma unroll
        for (int k = 0; k < 6; k++) rH2[k] = sc.Hll[(size_t)k * Lc + l];
        const Chol3 g = chol3(rH2, lambda);
        cv[0] = sc.bl[l] * g.i00;
        cv[1] = (sc.bl[(size_t)Lc + l] - g.g10 * cv[0]) * g.i11;
        cv[2] = (sc.bl[(size_t)2 * Lc + l] - g.g20 * cv[0] - g.g21 * cv[1]) * g.i22;
      }
#pragma unroll
      for (int j = 0; j < 3; j++) {
        double* col = Zt + (size_t)(3 * ll + j) * NRp;
        col[NR] = cv[j];
        for (int r = NR + 1; r < NRp; r++) col[r] = 0.0;
      }
    }
    for (int i = t; i < (ncol - 3 * nl) * NRp; i += BA_T) Zt[(size_t)3 * nl * NRp + i] = 0.0;  // padding columns
    __syncthreads();
    BAPROF(2);
    // SYRK on the matrix cores: operands of 8 k-steps are fetched from LDS before their 8 MFMAs are issued, two accumulators
    // per tile (even / odd k-steps) keep consecutive MFMAs independent
    const int ksteps = ncol >> 2;
#pragma unroll
    for (int q = 0; q < MAXQ; q++) {
      if (ti[q] < 0) continue;
      const double* pa = Zt + (size_t)a_k * NRp + 16 * ti[q] + a_row;
      const double* pb = Zt + (size_t)a_k * NRp + 16 * tj[q] + a_row;
      ba_d4 d0 = acc[q], d1 = ba_d4{0, 0, 0, 0};
      int ks = 0;
      for (; ks + 8 <= ksteps; ks += 8) {
        double a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          a[u] = pa[(size_t)4 * (ks + u) * NRp];
          b[u] = pb[(size_t)4 * (ks + u) * NRp];
        }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
          d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], d0, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u + 1], b[u + 1], d1, 0, 0, 0);
        }
      }
      for (; ks < ksteps; ks++) d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[(size_t)4 * ks * NRp], pb[(size_t)4 * ks * NRp], d0, 0, 0, 0);
      acc[q] = d0 + d1;
    }
    BAPROF(12);
  }
  // S (lower triangle + full diagonal blocks) and the rhs from the accumulated tiles.  C/D layout of v_mfma_f64_16x16x4_f64:
  // register v of lane l holds D[(l >> 4) + 4 v][l & 15] (NOT the f32 forms' 4 (l >> 4) + v)
#pragma unroll
  for (int q = 0; q < MAXQ; q++) {
    if (ti[q] < 0) continue;
#pragma unroll
    for (int v = 0; v < 4; v++) {
      const int row = 16 * ti[q] + (lane >> 4) + 4 * v, col = 16 * tj[q] + (lane & 15);
      const double a = acc[q][v];
      if (row == NR && col < NR) {
        sh.bfull[col] = sh.b[col];
        sh.x[col] = sh.b[col] - a;  // bschur = bp - sum Z c
      }
      if (row < NR && col <= row) {
        const bool same_block = (row / 6) == (col / 6);
        double val = -a;
        if (same_block) val += sh.Hpp[row / 6][6 * (col % 6) + (row % 6)] + (row == col ? lambda : 0.0);
        Hs[row * LD + col] = val;
        if (same_block && row != col) Hs[col * LD + row] = val;
      }
    }
  }
}

// wave 0: factor + solve the reduced system, then form the trial poses x (+) pose (unchanged poses if the factorisation
// failed) and their (R | t) tables
__device__ FLVIS_BA_PHASE_FN void ba_phase_solve_poses(bool okc_wg) {
  BAShared& sh = ba_sh();
  const int lane = threadIdx.x & 63;
#if FLVIS_BA_CHOL_WG
  const bool okc = okc_wg;
  ba_chol_subst(false);
#else
  const bool okc = ba_chol_solve();
#endif
  wave_lds_fence();
  BAPROF(6);
  if (lane == 0) sh.flag = okc ? 1 : 0;
  if (lane < sh.W) {
#pragma unroll
    for (int j = 0; j < 7; j++) sh.poseT[lane][j] = sh.pose[lane][j];
    const int hi = sh.hidx_of[lane];
    if (okc && hi >= 0) {
      SE3d T = load_pose7(sh.pose[lane]);
      T = g2o_mul(g2o_exp(sh.x + 6 * hi), T);
      store_pose7(sh.poseT[lane], T);
    }
    pose_to_rt(sh.poseT[lane], sh.RTt[lane]);
  }
  if (sh.n_imu) {
    wave_lds_fence();
    ba_phase_imu_trial();
  }
}

// landmark back-substitution and the robust chi2 of the trial state in one pass (thread per landmark): the observations
// are re-linearised at the accepted state (B^T dx_pose = w Jl^T (Jp dx_pose) needs no stored block), the step is solved
// with the landmark's 3x3 factor, the trial landmark goes to lmB and its reprojection errors against the trial poses
// are summed.  out[0] = share of the gain-ratio denominator, out[1] = share of the trial chi2.
__device__ FLVIS_BA_PHASE_FN void ba_phase_update_chi2(double lambda, int ok2, double* out) {
  BAShared& sh = ba_sh();
  const BAScratch sc = sh.sc;
  const int t = threadIdx.x, L = sh.L, Lc = sc.Lc, W = sh.W;
  const double K[4] = {sh.K[0], sh.K[1], sh.K[2], sh.K[3]};
  double scale_part = 0, chit = 0;
  for (int l = t; l < L; l += BA_T) {
    const unsigned m = sc.omask[l];
    double p[3] = {sc.lmA[l], sc.lmA[Lc + l], sc.lmA[2 * Lc + l]};
    if (m && ok2) {
      const double bl[3] = {sc.bl[l], sc.bl[Lc + l], sc.bl[2 * Lc + l]};
      double H[6];
#pragma unroll
      for (int k = 0; k < 6; k++) H[k] = sc.Hll[(size_t)k * Lc + l];
      double v[3] = {bl[0], bl[1], bl[2]};
      double un = (m & 1u) ? sc.uv[l] : 0.0, vn = (m & 1u) ? sc.uv[(size_t)Lc + l] : 0.0;
#pragma unroll 1
      for (int slot = 0; slot < W; slot++) {  // (not unrolled: instruction-cache footprint; next observation in flight)
        const double uu = un, vv = vn;
        if (slot + 1 < W) {
          const bool hn = (m >> (slot + 1)) & 1u;
          un = hn ? sc.uv[(size_t)(2 * slot + 2) * Lc + l] : 0.0;
          vn = hn ? sc.uv[(size_t)(2 * slot + 3) * Lc + l] : 0.0;
        }
        const int hi = sh.hidx_of[slot];
        if (!((m >> slot) & 1u) || hi < 0) continue;
        double er[2], Ji[2][3], Jj[2][6];
        ba_linearize(sh.RT[slot], p[0], p[1], p[2], uu, vv, K, er, Ji, Jj);
        const double wgt = ba_huber_w(er[0] * er[0] + er[1] * er[1]);
        const double* xp = sh.x + 6 * hi;
        double j0 = 0, j1 = 0;
#pragma unroll
        for (int r = 0; r < 6; r++) {
          j0 += Jj[0][r] * xp[r];
          j1 += Jj[1][r] * xp[r];
        }
        j0 *= wgt;
        j1 *= wgt;
#pragma unroll
        for (int c = 0; c < 3; c++) v[c] -= Ji[0][c] * j0 + Ji[1][c] * j1;
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
    if (m) {
      double un = (m & 1u) ? sc.uv[l] : 0.0, vn = (m & 1u) ? sc.uv[(size_t)Lc + l] : 0.0;
#pragma unroll 1
      for (int slot = 0; slot < W; slot++) {
        const double uu = un, vv = vn;
        if (slot + 1 < W) {
          const bool hn = (m >> (slot + 1)) & 1u;
          un = hn ? sc.uv[(size_t)(2 * slot + 2) * Lc + l] : 0.0;
          vn = hn ? sc.uv[(size_t)(2 * slot + 3) * Lc + l] : 0.0;
        }
        if (!((m >> slot) & 1u)) continue;
        chit += ba_huber_rho(ba_err2(sh.RTt[slot], p[0], p[1], p[2], uu, vv, K));
      }
    }
  }
  if (ok2)
    for (int i = t; i < 6 * sh.P; i += BA_T) scale_part += sh.x[i] * (lambda * sh.x[i] + sh.bfull[i]);
  out[0] = scale_part;
  out[1] = chit;
}

// The same with resident records (sh.fused): no per-observation HBM traffic.  The landmark's step is
// (Hll + lambda I)^-1 (bl - sum_h (w Jl_h)^T (Jp_h dx_h)) with Jp and w Jl rebuilt from the records; the pixels come from the records, the
// fixed pose's from the observation table.  The evaluation of the trial state IS the next iteration's linearisation when the trial is
// accepted (g2o: the errors computed for the gain ratio are the ones the next buildSystem starts from): every observation is linearised
// at the trial state here -- chi2, the landmark's Hll / bl (into Hll2 / bl2, swapped in on acceptance) and the record's (x/z, y/z, 1/z, w),
// overwritten in place (the landmark's own thread has consumed them).  A rejected trial (rare: none in the benchmark's windows) leaves
// records of a state that is not kept: the caller rebuilds them (ba_phase_linearize_rec).
__device__ FLVIS_BA_PHASE_FN void ba_phase_update_lin(double lambda, int ok2, double* out) {
  BAShared& sh = ba_sh();
  const BAScratch sc = sh.sc;
  const int t = threadIdx.x, L = sh.L, Lc = sc.Lc, CI = sh.CI, CL = sh.CL, fs = sh.fixed_slot;
  const double K[4] = {sh.K[0], sh.K[1], sh.K[2], sh.K[3]};
  const SchurBuf B = ba_schur_buf(0);
  double2* z = B.z;
  double scale_part = 0, chit = 0;
  for (int l = t; l < L; l += BA_T) {
    const unsigned m = sc.omask[l];
    double p[3] = {sc.lmA[l], sc.lmA[Lc + l], sc.lmA[2 * Lc + l]};
    unsigned fm = 0;
    int ib = 0;
    double uf = 0, vf = 0;
    const bool hf = fs >= 0 && ((m >> fs) & 1u);
    if (m) {
      fm = sc.fmask[l];
      ib = sc.ibase[l];
      if (hf) {
        uf = sc.uv[(size_t)(2 * fs) * Lc + l];
        vf = sc.uv[(size_t)(2 * fs + 1) * Lc + l];
      }
    }
    if (m && ok2) {
      const double bl[3] = {sc.bl[l], sc.bl[Lc + l], sc.bl[2 * Lc + l]};
      double v[3] = {bl[0], bl[1], bl[2]};
      int idx = ib;
      unsigned rem = fm;
#pragma unroll 1
      while (rem) {
        const int hi = __builtin_ctz(rem);
        rem &= rem - 1u;
        const double2 p0 = z[idx], p1 = z[CI + idx];
        idx++;
        const JpRows J = jp_rows(p0.x, p0.y, p1.x, K[0], K[1]);
        const double* xp = sh.x + 6 * hi;
        const double q0 = fma(J.a5, xp[5], fma(J.a3, xp[3], fma(J.a2, xp[2], fma(J.a1, xp[1], J.a0 * xp[0]))));
        const double q1 = fma(J.b5, xp[5], fma(J.b4, xp[4], fma(J.b2, xp[2], fma(J.b1, xp[1], J.b0 * xp[0]))));
        const double* R = sh.RT[sh.slot_of[hi]];
        const double s = -p1.x * p1.y, j0 = (s * K[0]) * q0, j1 = (s * K[1]) * q1;  // (w Jl)^T q, Jl = -1/z [fx (r0 - xn r2); fy (r1 - yn r2)]
#pragma unroll
        for (int c = 0; c < 3; c++) v[c] -= (R[c] - p0.x * R[6 + c]) * j0 + (R[3 + c] - p0.y * R[6 + c]) * j1;
      }
      Chol3 g;
      if (fm) {  // the factor of Hll + lambda I the staging left in LDS (landmarks seen by the fixed pose only have none)
        g = Chol3{B.gb[l], B.gb[CL + l], B.gb[2 * CL + l], B.gb[3 * CL + l], B.gb[4 * CL + l], B.gb[5 * CL + l]};
      } else {
        double H[6];
#pragma unroll
        for (int k = 0; k < 6; k++) H[k] = sc.Hll[(size_t)k * Lc + l];
        g = chol3(H, lambda);
      }
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
    if (m) {
      double h[6] = {0, 0, 0, 0, 0, 0}, bb[3] = {0, 0, 0};
      if (hf) {
        const BAObs o = ba_obs(sh.RTt[fs], p[0], p[1], p[2], uf, vf, K);
        chit += ba_obs_landmark(o, h, bb);
      }
      int idx = ib;
      unsigned rem = fm;
#pragma unroll 1
      while (rem) {
        const int hi = __builtin_ctz(rem);
        rem &= rem - 1u;
        const double2 uv = z[2 * CI + idx];
        const BAObs o = ba_obs(sh.RTt[sh.slot_of[hi]], p[0], p[1], p[2], uv.x, uv.y, K);
        chit += ba_obs_landmark(o, h, bb);
        z[idx] = double2{o.xn, o.yn};
        z[CI + idx] = double2{o.iz, o.wgt};
        idx++;
      }
#pragma unroll
      for (int j = 0; j < 6; j++) sc.Hll2[(size_t)j * Lc + l] = h[j];
#pragma unroll
      for (int j = 0; j < 3; j++) sc.bl2[(size_t)j * Lc + l] = bb[j];
    }
  }
  if (ok2)
    for (int i = t; i < 6 * sh.P; i += BA_T) scale_part += sh.x[i] * (lambda * sh.x[i] + sh.bfull[i]);
  out[0] = scale_part;
  out[1] = chit;
}

// two block sums with one set of barriers
__device__ inline void block_sum2(double& a, double& b, double (*red)[BA_NW]) {
  a = wave_sum_f64(a);
  b = wave_sum_f64(b);
  const int t = threadIdx.x;
  __syncthreads();
  if ((t & 63) == 0) {
    red[0][t >> 6] = a;
    red[1][t >> 6] = b;
  }
  __syncthreads();
  double ra = 0, rb = 0;
#pragma unroll
  for (int i = 0; i < BA_NW; i++) {
    ra += red[0][i];
    rb += red[1][i];
  }
  __syncthreads();
  a = ra;
  b = rb;
}

// one g2o optimize(iterations) call
__device__ __noinline__ void ba_optimize(const WindowDev& w, int iterations) {
  BAShared& sh = ba_sh();
  const int t = threadIdx.x;
  BAPROF(0);
  ba_build_structure(w);
  BAPROF(1);
  if (sh.cnt == 0) return;
  double lambda = -1, ni = 2;
  double currentChi = 0;
  bool have_lin = false;  // resident records: the accepted trial's evaluation left the linearisation of the state it produced behind
  for (int iteration = 0; iteration < iterations; iteration++) {
    BAPROF(0);
    // (resident records: from the second iteration on Hpp / bp are summed by the Schur phase, and the linearisation -- when one is
    // still needed -- leaves the Schur phase's staging for the known lambda behind)
    const int inl = (sh.fused && iteration > 0) ? 1 : 0;  // (the first iteration needs Hpp itself: lambda's initial value)
    int staged = 0;
    if (!have_lin) {
      const double chi = inl ? ba_phase_linearize_rec(lambda) : ba_phase_linearize(-1.0);
      staged = inl;
      BAPROF(3);
      currentChi = block_sum(chi, sh.red[0]);  // (its barriers also publish the per-wave partials / Hll / bl / the records)
    }
    if (inl)
      ba_phase_zero_poses();
    else
      ba_phase_finish_poses();
    if (sh.n_imu && t < 64) ba_phase_imu_linearize();
    __syncthreads();
    if (sh.n_imu) {
      ba_phase_imu_gather();
      if (!have_lin)  // (carried over from the accepted trial otherwise: its chi2 included the edges at the state it produced)
        for (int k = 0; k < sh.n_imu; k++) currentChi += sh.imu_chi[k];
      __syncthreads();
    }
    have_lin = false;
    BAPROF(4);
    if (iteration == 0) {
      lambda = 1e-5 * block_max(ba_phase_max_diag(), sh.red[0]);
      ni = 2;
    }
    double rho = 0;
    int qmax = 0;
    bool lambda_bad = false;
    do {
      BAPROF(0);
      if (sh.use_mfma) {
        ba_phase_schur_mfma(lambda);
      } else if (sh.fused) {
        if (!staged) {
          ba_phase_stage(lambda);
        }
        __syncthreads();
        BAPROF(2);
        ba_phase_schur_resident(lambda, inl);
      } else {
        ba_phase_schur_chunked(lambda);
      }
      staged = 0;  // a further trial of this iteration has another lambda: the records are rebuilt from the accepted state
      __syncthreads();
      if (sh.n_imu) {
        ba_phase_imu_offdiag();
        __syncthreads();
      }
      BAPROF(7);
#if FLVIS_BA_CHOL_WG
      const bool okc_wg = ba_chol_factor_wg();  // (ends behind wave 0's last diagonal block: the substitutions below are wave 0's too)
#else
      const bool okc_wg = true;
#endif
      if (t < 64) ba_phase_solve_poses(okc_wg);
      __syncthreads();
      BAPROF(8);
      const int ok2 = sh.flag;
      double parts[2];
      if (sh.fused)
        ba_phase_update_lin(lambda, ok2, parts);
      else
        ba_phase_update_chi2(lambda, ok2, parts);
      double scale = parts[0], tempChi = parts[1];
      block_sum2(scale, tempChi, sh.red);
      for (int k = 0; k < sh.n_imu; k++) tempChi += sh.imu_chit[k];
      scale += 1e-3;
      BAPROF(9);
#ifdef FLVIS_BA_PROF
      if (t == 0 && sh.prof) sh.prof_acc[14] += 1;
#endif
      if (!ok2) tempChi = 1.7976931348623157e308;
      rho = (currentChi - tempChi) / scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - detm::det_powi((2 * rho - 1), 3);
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
          if (sh.fused) {  // ... and the trial's evaluation the linearisation of the next iteration
            tmp = sh.sc.Hll, sh.sc.Hll = sh.sc.Hll2, sh.sc.Hll2 = tmp;
            tmp = sh.sc.bl, sh.sc.bl = sh.sc.bl2, sh.sc.bl2 = tmp;
          }
        }
        have_lin = sh.fused != 0;
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
        if (sh.fused) {  // the records hold the rejected state: rebuilt from the accepted one, staged for the next trial's lambda
          __syncthreads();
          (void)ba_phase_linearize_rec(lambda);
          staged = 1;
        }
      }
      qmax++;
      if (t == 0) sh.n_trials++;
    } while (rho < 0 && qmax < 10);
    if (qmax == 10 || rho == 0 || lambda_bad) break;
  }
}

#include "ba_update.hpp"

// the OPTIMIZING block for the window of stream s (w.solve set by ba_update_dev); frame_id = the keyframe that triggered it
__device__ __noinline__ void ba_solve_dev(const Pipe& p, int s, long long frame_id) {
  WindowDev& w = p.win[s];
  BAShared& sh = ba_sh();
  const int W = p.cam.window;
  const int L = w.n_lm, E = w.n_edge;
  const int t = threadIdx.x, lane = t & 63;
  if (t == 0) {
    sh.sc = carve(p.ba_scratch + (size_t)s * p.ba_scratch_stride, L, E, W);
    sh.lds_budget = p.ba_lds_bytes;
    sh.W = W;
    sh.use_mfma = p.ba_mfma;
    sh.balance = p.ba_balance && !p.ba_mfma;
    sh.K[0] = p.cam.fx;
    sh.K[1] = p.cam.fy;
    sh.K[2] = p.cam.cx;
    sh.K[3] = p.cam.cy;
    // IMU rotation edges of this window: slot j is linked to its chronological predecessor (the previous ring slot) unless j
    // is the oldest pose
    int ne = 0;
    if (p.imu_factor) {
      for (int j = 0; j < W; j++) {
        const int i = (j + W - 1) % W;
        if (j == w.oldest || !w.imu_has[j] || !(w.imu_dt[j] > 0) || !w.pose_present[i] || !w.pose_present[j]) continue;
        sh.imu_a[ne] = i;
        sh.imu_b[ne] = j;
        for (int q = 0; q < 4; q++) sh.imu_dq[ne][q] = w.imu_dq[j][q];
        sh.imu_w[ne] = 1.0 / (p.imu_sigma_g * p.imu_sigma_g * w.imu_dt[j]);
        sh.imu_wp[ne] = 0.0;
        if (p.imu_sigma_a > 0) {
          for (int q = 0; q < 3; q++) sh.imu_dp[ne][q] = w.imu_dp[j][q], sh.imu_va[ne][q] = w.imu_va[j][q];
          sh.imu_dtk[ne] = w.imu_dt[j];
          sh.imu_wp[ne] = 1.0 / (p.imu_sigma_a * p.imu_sigma_a * ((w.imu_dt[j] * w.imu_dt[j]) * w.imu_dt[j]) / 3.0);
        }
        ne++;
      }
      sh.q_c_b[0] = p.cam.T_c_i[6], sh.q_c_b[1] = p.cam.T_c_i[3], sh.q_c_b[2] = p.cam.T_c_i[4], sh.q_c_b[3] = p.cam.T_c_i[5];
      sh.t_c_b[0] = p.cam.T_c_i[0], sh.t_c_b[1] = p.cam.T_c_i[1], sh.t_c_b[2] = p.cam.T_c_i[2];
    }
    sh.n_imu = ne;
    sh.n_trials = 0;
    sh.t_begin = (long long)wall_clock64();
    sh.prof = nullptr;
#ifdef FLVIS_BA_PROF
    sh.prof = p.counters ? p.counters + 8 : nullptr;
    sh.tlast = (long long)wall_clock64();
    for (int i = 0; i < 16; i++) sh.prof_acc[i] = 0;
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
  ba_optimize(w, FLVIS_BA_IT1);
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
  ba_optimize(w, FLVIS_BA_IT2);
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
      out.frame_id = frame_id;
      SE3d Tn = load_pose7(sh.pose[w.newest]);
      store_pose7(out.T_c_w, se3_from_mat(q_to_mat(Tn.q), Tn.t));
      out.lm_count = c;
      out.valid = 1;
      p.st[s].lm_state = 1;
      w.solve = 0;
      w.ba_runs++;
      if (p.counters) {
        atomicAdd((unsigned long long*)&p.counters[2], 1ull);
        // flop accounting (per trial E (120 + 300) + sum_l k_l^2 / 2 * 324 + (6 P)^3 / 3): trials, trials x observations by
        // free poses, trials x landmarks, trials x free poses
        const unsigned long long nt = (unsigned long long)sh.n_trials;
        atomicAdd((unsigned long long*)&p.counters[4], nt);
        atomicAdd((unsigned long long*)&p.counters[5], nt * (unsigned long long)sh.nitems);
        atomicAdd((unsigned long long*)&p.counters[6], nt * (unsigned long long)sh.L);
        atomicAdd((unsigned long long*)&p.counters[7], nt * (unsigned long long)sh.P);
        const unsigned long long ticks = (unsigned long long)((long long)wall_clock64() - sh.t_begin);  // 10 ns ticks
        atomicAdd((unsigned long long*)&p.counters[60], ticks);
#ifndef FLVIS_BA_PROF
        if (!sh.fused) {  // (counters 28 / 29: windows whose records did not stay resident -- streamed in chunks -- and their ticks)
          atomicAdd((unsigned long long*)&p.counters[28], 1ull);
          atomicAdd((unsigned long long*)&p.counters[29], ticks);
        }
#endif
      }
    }
  }
  BAPROF(11);
#ifdef FLVIS_BA_PROF
  if (t == 0 && sh.prof)
    for (int i = 0; i < 15; i++) atomicAdd((unsigned long long*)&sh.prof[i], (unsigned long long)sh.prof_acc[i]);
#endif
}

// Local-map worker: one workgroup per stream drains the stream's keyframe queue (bookkeeping + optimisation per keyframe,
// strictly in order).  It is launched after every frame on one of the local-map HIP streams; a workgroup that finds the
// stream's window owned by a workgroup of an earlier launch leaves at once -- the owner re-checks the queue before and
// after releasing the window, so a keyframe is picked up at the latest by the launch that follows it.  An owner takes ba_drain
// keyframes (one) and leaves the rest to the next launch unless ba_backlog or more are waiting.  The tracker therefore never waits
// for the optimiser unless a stream has fallen behind by half a queue (KFQ keyframes).
//
// Which stream a workgroup serves (round 5).  A working workgroup takes its CU whole (512 threads at 253 registers, 159 KB of LDS), and
// the dispatcher deals the workgroups of a launch to the eight XCDs in turn -- so with workgroup s serving stream s, the streams that
// happen to have a keyframe decide how many CUs each XCD loses for the next 1.2 ms, while the tracker's kernels are dealt to the XCDs
// evenly and finish with the XCD that has the fewest CUs left.  With p.ba_remap the FIRST workgroup of a launch to arrive (an arrival
// counter per local-map HIP stream; launches on one HIP stream do not overlap) writes the list of the streams whose queue holds a
// keyframe, publishes it under the launch's tag, and workgroup r serves the r-th stream of the list: the working workgroups are the
// first n of the launch, n / 8 per XCD.  Every stream is still either found empty (when the list is made), found owned, or served: the
// argument above and the back-pressure bound (pipeline.cpp) hold as before.  The others wait for the list in a sleep loop: the
// workgroup they wait for is running by construction.  Off by default: measured, no effect on the LK launches (pipeline.cpp).
__global__ __launch_bounds__(BA_T) BA_ATTR void k_ba_worker(Pipe p, int plan_slot, unsigned launch_tag) {
  const int t = threadIdx.x;
  __shared__ int s_go, s_pick;
  __shared__ unsigned s_head, s_tail;
  __shared__ int s_cnt[BA_NW];
  if (p.ba_remap) {
    if (t < 64) {
      unsigned* plan = p.ba_plan + (size_t)plan_slot * (3 + p.S);
      unsigned ticket = 0;
      if (t == 0) {
        ticket = atomicAdd(&plan[0], 1u);
        // (the last arrival of a launch puts the counter back to 0 -- launches on one local-map HIP stream do not overlap -- so that it never
        // wraps: with a grid size that is not a power of two a wrapped counter would leave a launch without a list maker)
        if (ticket + 1u == gridDim.x) __hip_atomic_store(&plan[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      ticket = __shfl(ticket, 0);
      if (ticket == 0) {
        unsigned n = 0;
        for (int j0 = 0; j0 < p.S; j0 += 64) {
          const int j = j0 + t;
          const bool waiting = j < p.S && __hip_atomic_load(&p.kfq_tail[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) !=
                                              __hip_atomic_load(&p.kfq_head[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long m = __ballot(waiting);
          if (waiting) plan[3 + n + __popcll(m & ((1ull << t) - 1ull))] = (unsigned)j;
          n += (unsigned)__popcll(m);
        }
        if (t == 0) plan[2] = n;
        __threadfence();
        if (t == 0) __hip_atomic_store(&plan[1], launch_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (t == 0) {
        while (__hip_atomic_load(&plan[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != launch_tag) __builtin_amdgcn_s_sleep(4);
        const unsigned n = __hip_atomic_load(&plan[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_pick = blockIdx.x < n ? (int)__hip_atomic_load(&plan[3 + blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
      }
    }
    __syncthreads();
    if (s_pick < 0) return;
  }
  const int s = p.ba_remap ? s_pick : (int)blockIdx.x;
#ifndef FLVIS_BA_WG_FENCES
  // Cache maintenance of the hand-over (round 5).  Every XCD has its own L2: an acquire at agent scope is a `buffer_inv sc1` (the XCD's L2
  // and the CU's L1 drop their lines), a release a `buffer_wbl2 sc1` (the L2 writes its dirty lines back) -- for EVERY wave that executes
  // one, and the tracker's kernels on the same XCD lose their cached pyramid rows and templates with it.  Up to round 4 the eight waves
  // of the workgroup each executed four system-scope fences per keyframe beside thread 0's acquire loads and release stores (~40 L2
  // operations per keyframe, ~1200 per frame of 64 streams).  One wave is enough: thread 0 reads the queue counters with relaxed loads,
  // and only when it has taken the window does it execute ONE acquire fence; the other threads are ordered behind it by the workgroup
  // barrier (scoped happens-before is transitive), and they share its CU's L1 and its XCD's L2.  On the way out: barrier, ONE release
  // fence by thread 0, then the head counter, then (release store) the ownership flag.  -DFLVIS_BA_WG_FENCES: the old form (A/B).
  while (true) {
    if (t == 0) {
      s_go = 0;
      const unsigned tl = __hip_atomic_load(&p.kfq_tail[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned hd = __hip_atomic_load(&p.kfq_head[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tl != hd && atomicCAS(&p.ba_busy[s], 0, 1) == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the previous owner's window and head, the tracker's keyframes
        s_go = 1;
        s_head = __hip_atomic_load(&p.kfq_head[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // only the owner advances the head
        // the tail as read BEFORE the fence: the fence synchronises with the release stores of the keyframes up to that value only.  A
        // keyframe k_frame_end appends after it is seen by the loop below (tl2 != tl), which acquires again before its slot is read.
        s_tail = tl;
      }
    }
    __syncthreads();
    if (!s_go) return;
    int taken = 0;
    bool capped = false;
    while (true) {
      const unsigned hd = s_head, tl = s_tail;
      __syncthreads();  // (everybody has read the two words before thread 0 rewrites them below)
      if (hd == tl) break;
      const KeyFrameDev& kf = p.kfq[(size_t)s * KFQ + (hd % KFQ)];
      const long long frame_id = kf.frame_id;
#ifdef FLVIS_BA_PROF
      const long long tu0 = (long long)wall_clock64();
#endif
      ba_update_dev(p, s, kf, reinterpret_cast<long long*>(ba_dyn()), s_cnt);
      __syncthreads();
#ifdef FLVIS_BA_PROF
      if (t == 0 && p.counters) {  // (counter 26 / 27: ticks in the bookkeeping of a keyframe, keyframes)
        atomicAdd((unsigned long long*)&p.counters[26], (unsigned long long)((long long)wall_clock64() - tu0));
        atomicAdd((unsigned long long*)&p.counters[27], 1ull);
      }
#endif
      if (p.win[s].solve) ba_solve_dev(p, s, frame_id);
      __syncthreads();
      if (t == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the window, the landmarks, CorrectionInf: before the head moves
        __hip_atomic_store(&p.kfq_head[s], hd + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_head = hd + 1u;
        const unsigned tl2 = __hip_atomic_load(&p.kfq_tail[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tl2 != tl) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // keyframes appended since the last look: their slots are read afresh
        s_tail = tl2;
      }
      __syncthreads();
      if (p.ba_drain > 0 && ++taken >= p.ba_drain) {
        // the next launch goes on (one follows every frame, and flvis_hip_synchronize launches until the queues are empty) -- unless
        // the stream has fallen behind by ba_backlog keyframes: then the owner stays, so that a finished launch has left less than
        // ba_backlog keyframes of the frames before it in every queue (what the tracker's back-pressure counts on, pipeline.cpp)
        if (s_tail - s_head < (unsigned)p.ba_backlog) {
          capped = true;
          break;
        }
      }
    }
    __syncthreads();
    // (the head store above is ordered before this one by the release; nothing of the window was written since the last release fence)
    if (t == 0) __hip_atomic_store(&p.ba_busy[s], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (capped) return;
    // a keyframe may have arrived between the emptiness test and the release: look again
  }
#else
  while (true) {
    if (t == 0) {
      s_go = 0;
      const unsigned tl = __hip_atomic_load(&p.kfq_tail[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned hd = __hip_atomic_load(&p.kfq_head[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      if (tl != hd && atomicCAS(&p.ba_busy[s], 0, 1) == 0) s_go = 1;
    }
    __syncthreads();
    if (!s_go) return;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    // only the owner advances the head: it is read once (after the acquire above) and then kept in a register
    unsigned my_head = __hip_atomic_load(&p.kfq_head[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    int taken = 0;
    bool capped = false;
    while (true) {
      if (t == 0) {
        s_head = my_head;
        s_tail = __hip_atomic_load(&p.kfq_tail[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      const unsigned hd = s_head, tl = s_tail;
      __syncthreads();
      if (hd == tl) break;
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      const KeyFrameDev& kf = p.kfq[(size_t)s * KFQ + (hd % KFQ)];
      const long long frame_id = kf.frame_id;
#ifdef FLVIS_BA_PROF
      const long long tu0 = (long long)wall_clock64();
#endif
      ba_update_dev(p, s, kf, reinterpret_cast<long long*>(ba_dyn()), s_cnt);
      __syncthreads();
#ifdef FLVIS_BA_PROF
      if (t == 0 && p.counters) {  // (counter 26 / 27: ticks in the bookkeeping of a keyframe, keyframes)
        atomicAdd((unsigned long long*)&p.counters[26], (unsigned long long)((long long)wall_clock64() - tu0));
        atomicAdd((unsigned long long*)&p.counters[27], 1ull);
      }
#endif
      if (p.win[s].solve) ba_solve_dev(p, s, frame_id);
      __atomic_thread_fence(__ATOMIC_RELEASE);
      __syncthreads();
      if (t == 0) __hip_atomic_store(&p.kfq_head[s], hd + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      my_head = hd + 1u;
      if (p.ba_drain > 0 && ++taken >= p.ba_drain) {
        // the next launch goes on (one follows every frame, and flvis_hip_synchronize launches until the queues are empty) -- unless
        // the stream has fallen behind by ba_backlog keyframes: then the owner stays, so that a finished launch has left less than
        // ba_backlog keyframes of the frames before it in every queue (what the tracker's back-pressure counts on, pipeline.cpp)
        if (t == 0) s_tail = __hip_atomic_load(&p.kfq_tail[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned left = s_tail - my_head;
        __syncthreads();
        if (left < (unsigned)p.ba_backlog) {
          capped = true;
          break;
        }
      }
    }
    __atomic_thread_fence(__ATOMIC_RELEASE);
    __syncthreads();
    if (t == 0) __hip_atomic_store(&p.ba_busy[s], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (capped) return;
    // a keyframe may have arrived between the emptiness test and the release: look again
  }
#endif
}

void launch_ba_worker(hipStream_t st, const Pipe& p, int plan_slot, unsigned launch_tag) {
  hipLaunchKernelGGL(k_ba_worker, dim3(p.S), dim3(BA_T), p.ba_lds_bytes, st, p, plan_slot, launch_tag);
}
int ba_lds_budget_max() { return BA_LDS_BUDGET; }
hipError_t ba_kernels_init() {
  return hipFuncSetAttribute((const void*)k_ba_worker, hipFuncAttributeMaxDynamicSharedMemorySize, BA_LDS_BUDGET);
}

}  // namespace flvis
