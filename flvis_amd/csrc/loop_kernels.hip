// flvis_amd: place recognition + pose-graph optimisation of the reference's loop closing for gfx950, next to
// the ORB extraction / matching of orb_kernels.hip (§8f-1):
//   * the DBoW3 bag-of-words of a keyframe's ORB descriptors   voc.transform(kf.lm_descriptor, kf_bv)   vo_loopclosing.cpp:249-253
//   * one row of the similarity matrix                          voc.score(kf_bv, kf_lc_tmp[i]->kf_bv)    vo_loopclosing.cpp:417-437
//   * the loop-candidate selection on that row (host logic)     isLoopCandidate                          vo_loopclosing.cpp:520-590
// DBoW3 semantics (3rdPartLib/DBow3/src): the tree descent takes the FIRST child of minimal Hamming distance
// (Vocabulary.cpp:836-874), a word's value is its idf weight added once per occurrence (BowVector.cpp:34-46), the vector is
// L1-normalised with the norm summed in ascending word order (BowVector.cpp:62-84), and the L1 score is summed over the common
// words in ascending order (ScoringObject.cpp:23-68).  The fp64 sums are kept SEQUENTIAL in exactly that order, so ids, values
// and scores are bit-identical to the CPU restatement the tests check against: the candidate selection thresholds the scores.
//
// All byte / integer work plus short fp64 sums, keyframe rate (not on the per-frame path): one thread per descriptor for the
// descent (the tree, <= 1 MB for a 10^6-word vocabulary's upper levels, stays in L2), one 1024-thread workgroup per keyframe for
// the sort / unique / count in LDS, one wave per database vector for the scores.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

#include "../../include/flvis_hip.h"
#include "ctx.hpp"
#include "dev_common.hpp"
#include "dev_geom.hpp"
#include "dev_math.hpp"
#include "track_kernels.hpp"

namespace flvis {

constexpr int BOW_T = 1024;
constexpr int BOW_MAXF = 2048;  // descriptors per keyframe (the reference extracts 1000)

struct VocDev {
  const int* child_ptr;
  const int* child_idx;
  const uint8_t* desc;      // [n_nodes][32]
  const int* word_id;       // per node (leaves)
  const double* weight;     // per node (leaves: idf)
  const double* word_weight;  // per word id
  int n_nodes, n_words;
  int depth;  // levels below the root (bounds the descent of k_bow_words)
};

// exclusive prefix sum of one int per thread over the workgroup (NW waves); total = the sum
template <int NW>
__device__ inline int block_exclusive_scan(int v, int* s_part, int& total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int n = __shfl_up(inc, o, 64);
    if (lane >= o) inc += n;
  }
  __syncthreads();
  if (lane == 63) s_part[wv] = inc;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < NW; i++) {
    const int c = s_part[i];
    if (i < wv) off += c;
    tot += c;
  }
  total = tot;
  return off + inc - v;
}

__device__ inline int hamming256(const uint4 a0, const uint4 a1, const uint8_t* b) {
  const uint4* q = reinterpret_cast<const uint4*>(b);
  const uint4 b0 = q[0], b1 = q[1];
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
         __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// word of every descriptor: d_words[img][r] = word id, or INT_MAX when r >= count or the word is stopped (weight <= 0)
__global__ __launch_bounds__(256) void k_bow_words(VocDev v, const uint8_t* __restrict__ desc, const int* __restrict__ count, int dcap,
                                                   int* __restrict__ words) {
  const int img = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
  if (r >= dcap) return;
  int out = INT_MAX;
  if (r < min(count[img], dcap)) {
    const uint4* f = reinterpret_cast<const uint4*>(desc + ((size_t)img * dcap + r) * 32);
    const uint4 f0 = f[0], f1 = f[1];
    int node = 0;
    // (flvis_hip_bow_set_vocabulary only accepts trees, so the descent ends after at most `depth` levels; the bound keeps a
    // corrupted table from hanging the GPU: it then yields no word)
    bool leaf = false;
    for (int level = 0; level <= v.depth; level++) {
      const int c0 = v.child_ptr[node], c1 = v.child_ptr[node + 1];
      if (c0 == c1) {
        leaf = true;
        break;
      }
      int best_d = INT_MAX, best = node;
      for (int c = c0; c < c1; c++) {
        const int id = v.child_idx[c];
        const int d = hamming256(f0, f1, v.desc + (size_t)id * 32);
        if (d < best_d) {
          best_d = d;
          best = id;
        }
      }
      node = best;
    }
    if (leaf && v.weight[node] > 0) out = v.word_id[node];
  }
  words[(size_t)img * dcap + r] = out;
}

// one workgroup per keyframe: sort the words (bitonic, LDS), collapse equal ids, value = weight added count times, L1-normalise
__global__ __launch_bounds__(BOW_T) void k_bow_vector(VocDev v, const int* __restrict__ words, int dcap, int vcap, int* __restrict__ ids,
                                                      double* __restrict__ vals, int* __restrict__ nnz) {
  __shared__ int s_w[BOW_MAXF];
  __shared__ int s_scan[BOW_T / 64];
  __shared__ double s_norm;
  __shared__ int s_n;
  const int img = blockIdx.x, t = threadIdx.x;
  for (int i = t; i < BOW_MAXF; i += BOW_T) s_w[i] = i < dcap ? words[(size_t)img * dcap + i] : INT_MAX;
  __syncthreads();
  for (int k = 2; k <= BOW_MAXF; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = t; i < BOW_MAXF; i += BOW_T) {
        const int p = i ^ j;
        if (p > i) {
          const int a = s_w[i], b = s_w[p];
          const bool up = (i & k) == 0;
          if ((a > b) == up) {
            s_w[i] = b;
            s_w[p] = a;
          }
        }
      }
      __syncthreads();
    }
  // heads of runs of equal ids, in order: two elements per thread (2t, 2t + 1)
  int head[2], cnt = 0;
  for (int e = 0; e < 2; e++) {
    const int i = 2 * t + e;
    head[e] = (s_w[i] != INT_MAX && (i == 0 || s_w[i - 1] != s_w[i])) ? 1 : 0;
    cnt += head[e];
  }
  int tot;
  const int base = block_exclusive_scan<BOW_T / 64>(cnt, s_scan, tot);
  if (t == 0) s_n = tot;
  __syncthreads();
  double* ov = vals + (size_t)img * vcap;
  int* oi = ids + (size_t)img * vcap;
  int k = base;
  for (int e = 0; e < 2; e++) {
    if (!head[e]) continue;
    const int i = 2 * t + e, w = s_w[i];
    int n = 1;
    while (i + n < BOW_MAXF && s_w[i + n] == w) n++;
    const double wt = v.word_weight[w];
    double val = wt;
    for (int q = 1; q < n; q++) val += wt;  // BowVector::addWeight, once per occurrence
    if (k < vcap) {
      oi[k] = w;
      ov[k] = val;
    }
    k++;
  }
  __threadfence_block();
  __syncthreads();
  const int n_out = min(s_n, vcap);
  if (t == 0) {
    double norm = 0.0;  // BowVector::normalize(L1): ascending word id
    for (int i = 0; i < n_out; i++) norm += fabs(ov[i]);
    s_norm = norm;
    nnz[img] = n_out;
  }
  __syncthreads();
  const double norm = s_norm;
  if (norm > 0.0)
    for (int i = t; i < n_out; i += BOW_T) ov[i] /= norm;
}

// L1Scoring::score(query, db vector) by one wave: lanes binary-search the query's words in the database vector, the terms of a chunk of
// 64 query words go to LDS and lane 0 adds them in ascending word order (DBoW3's order)
__device__ inline double bow_score_wave(const int* __restrict__ q_ids, const double* __restrict__ q_vals, int nq, const int* __restrict__ di,
                                        const double* __restrict__ dv, int nd, double* s_term, int lane) {
  double score = 0;
  for (int base = 0; base < nq; base += 64) {
    const int i = base + lane;
    double term = 0.0;
    bool hit = false;
    if (i < nq) {
      const int id = q_ids[i];
      int lo = 0, hi = nd;  // lower_bound
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (di[mid] < id) lo = mid + 1;
        else hi = mid;
      }
      if (lo < nd && di[lo] == id) {
        const double vi = q_vals[i], wi = dv[lo];
        term = fabs(vi - wi) - fabs(vi) - fabs(wi);
        hit = true;
      }
    }
    const unsigned long long m = __ballot(hit);
    if (m == 0ull) continue;
    s_term[lane] = term;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      unsigned long long mm = m;
      while (mm) {  // common words in ascending order
        const int b = __ffsll((long long)mm) - 1;
        score += s_term[b];
        mm &= mm - 1;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  return -score / 2.0;
}

// one wave per database vector; db_nnz[j] < 0 marks an absent keyframe (score 0)
__global__ __launch_bounds__(256) void k_bow_score(const int* __restrict__ q_ids, const double* __restrict__ q_vals, const int* __restrict__ q_nnz,
                                                   const int* __restrict__ db_ids, const double* __restrict__ db_vals,
                                                   const int* __restrict__ db_nnz, int vcap, int n_db, double* __restrict__ scores) {
  __shared__ double s_term[4][64];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + wv;
  if (j >= n_db) return;
  const int nd = db_nnz[j], nq = q_nnz[0];
  double sc = 0.0;
  if (nd >= 0) sc = bow_score_wave(q_ids, q_vals, nq, db_ids + (size_t)j * vcap, db_vals + (size_t)j * vcap, nd, s_term[wv], lane);
  if (lane == 0) scores[j] = sc;
}

// the same for several (query, database range) jobs of one vector store in one launch: job = (query vector, first database vector,
// number of database vectors), all indices into the store; scores[first + j]
__global__ __launch_bounds__(256) void k_bow_score_jobs(const int* __restrict__ jobs, const int* __restrict__ ids, const double* __restrict__ vals,
                                                        const int* __restrict__ nnz, int vcap, double* __restrict__ scores) {
  __shared__ double s_term[4][64];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int q = jobs[3 * blockIdx.y], first = jobs[3 * blockIdx.y + 1], n_db = jobs[3 * blockIdx.y + 2];
  const int j = blockIdx.x * 4 + wv;
  if (j >= n_db) return;
  const size_t v = (size_t)first + j;
  const int nd = nnz[v], nq = nnz[q];
  double sc = 0.0;
  if (nd >= 0 && nq >= 0) sc = bow_score_wave(ids + (size_t)q * vcap, vals + (size_t)q * vcap, nq, ids + v * vcap, vals + v * vcap, nd, s_term[wv], lane);
  if (lane == 0) scores[v] = sc;
}


// ------------------------------------------------------------------------------------------------ pose-graph optimisation
// loopClosureOnCovGraphG2ONew (vo_loopclosing.cpp:742-944): vertices kf_prev..kf_curr (g2o VertexSE3, estimate T_w_c), EdgeSE3 to
// the next five keyframes with the current relative poses as measurements plus one edge per recorded loop, Cauchy kernel,
// Levenberg (lambda 1e-10, up to 100 iterations, up to 10 trials each), Cholesky on the pose blocks.  g2o semantics as cited in
// the test-side CPU restatement (error / update in the translation + compact-quaternion chart of types/slam3d/isometry3d_mappings.cpp).
//
// One workgroup per graph (a batch = the graphs of independent sequences).  The host only does the integer bookkeeping (which
// keyframes become vertices, which are fixed, the edge list, the block profile); measurements, initial guess, linearisation,
// factorisation and write-back run here.  The normal matrix is block-banded (5 neighbours) with one wide block row per loop edge
// between two free vertices: stored and factored as a block PROFILE (every block row from its first non-zero column to the
// diagonal; fill stays inside), row by row by one wave -- lane (r, c) owns one entry of the current 6x6 block.
constexpr int PGO_T = 256;

struct PgoGraph {
  int n, E, P, n_kf, iterations, use_guess;
  const int *vkf, *fixed, *hidx, *ea, *eb, *eloop, *adj_ptr, *adj_e, *first;
  const long long* boff;  // block offset of block row i's first block; boff[P] = number of blocks
  int* queue;             // [n] BFS scratch
  int* dist;              // [n]
  double *est, *bak, *Z, *H, *L, *b, *x;
  double* T_c_w;            // [n_kf][7] in / out
  const double* loop_pose;  // [n_loops][7]
  double* drift;            // [7]
  double* stats;            // [5]
};

FD SE3d iso_mul(const SE3d& a, const SE3d& b) { return SE3d{q_mul(a.q, b.q), a.t + q_rotate(a.q, b.t)}; }
FD SE3d iso_inv(const SE3d& a) {
  const Q4 qi = q_conj(a.q);
  return SE3d{qi, q_rotate(qi, -1.0 * a.t)};
}
FD SE3d pgo_from_mqt(const double* v) {
  const double w = 1 - (v[3] * v[3] + v[4] * v[4] + v[5] * v[5]);
  const Q4 q = w < 0 ? q_identity() : Q4{sqrt(w), v[3], v[4], v[5]};
  return SE3d{q, V3{v[0], v[1], v[2]}};
}
FD void pgo_error(const SE3d& Xi, const SE3d& Xj, const SE3d& Z, double* e) {
  const SE3d Ee = iso_mul(iso_mul(iso_inv(Z), iso_inv(Xi)), Xj);
  Q4 q = q_normalized(Ee.q);
  if (q.w < 0) q = Q4{-q.w, -q.x, -q.y, -q.z};
  e[0] = Ee.t.x, e[1] = Ee.t.y, e[2] = Ee.t.z, e[3] = q.x, e[4] = q.y, e[5] = q.z;
}
// Jacobians of the error w.r.t. the updates of Xi (Ji) and Xj (Jj): rows = error, columns = (translation, compact quaternion)
FD void pgo_linearize(const SE3d& Xi, const SE3d& Xj, const SE3d& Z, double Ji[6][6], double Jj[6][6]) {
  const SE3d A = iso_inv(Z), B = iso_mul(iso_inv(Xi), Xj), Ee = iso_mul(A, B);
  const M3 Ra = q_to_mat(A.q), Re = q_to_mat(Ee.q);
  const M3 RaS = Ra * skew(B.t);
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int c = 0; c < 6; c++) Ji[r][c] = Jj[r][c] = 0.0;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      Ji[r][c] = -Ra.m[r][c];
      Jj[r][c] = Re.m[r][c];
      Ji[r][3 + c] = 2.0 * RaS.m[r][c];
    }
  const Q4 qe = q_normalized(Ee.q);
  const double sgn = qe.w < 0 ? -1.0 : 1.0;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const Q4 u{0, c == 0 ? 1.0 : 0.0, c == 1 ? 1.0 : 0.0, c == 2 ? 1.0 : 0.0};
    const Q4 di = q_mul(q_mul(A.q, u), B.q), dj = q_mul(Ee.q, u);
    Ji[3][3 + c] = -sgn * di.x, Ji[4][3 + c] = -sgn * di.y, Ji[5][3 + c] = -sgn * di.z;
    Jj[3][3 + c] = sgn * dj.x, Jj[4][3 + c] = sgn * dj.y, Jj[5][3 + c] = sgn * dj.z;
  }
}

// sum over the workgroup in a fixed order (per-wave butterfly, waves in order): the same value in every thread
__device__ inline double pgo_block_sum(double v, double* s_red) {
  v = wave_sum_f64(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0;
#pragma unroll
  for (int i = 0; i < PGO_T / 64; i++) r += s_red[i];
  __syncthreads();
  return r;
}

__device__ inline double pgo_chi2(const PgoGraph& g, double* s_red) {
  double chi = 0;
  for (int k = threadIdx.x; k < g.E; k += PGO_T) {
    double e[6];
    pgo_error(load_pose7(g.est + 7 * g.ea[k]), load_pose7(g.est + 7 * g.eb[k]), load_pose7(g.Z + 7 * k), e);
    double e2 = 0;
#pragma unroll
    for (int q = 0; q < 6; q++) e2 += e[q] * e[q];
    chi += detm::det_log(e2 + 1.0);
  }
  return pgo_block_sum(chi, s_red);
}

// H (block profile, lower triangle) and b: one thread per free vertex gathers its incident edges in adjacency order
__device__ inline void pgo_build(const PgoGraph& g) {
  for (int v = threadIdx.x; v < g.n; v += PGO_T) {
    const int hi = g.hidx[v];
    if (hi < 0) continue;
    const int f = g.first[hi];
    double* Hrow = g.H + g.boff[hi] * 36;
    for (int q = 0; q < (hi - f + 1) * 36; q++) Hrow[q] = 0.0;
    double bb[6] = {0, 0, 0, 0, 0, 0};
    double* Hd = Hrow + (size_t)(hi - f) * 36;
    for (int a = g.adj_ptr[v]; a < g.adj_ptr[v + 1]; a++) {
      const int k = g.adj_e[a];
      const int va = g.ea[k], vb = g.eb[k];
      const SE3d Xi = load_pose7(g.est + 7 * va), Xj = load_pose7(g.est + 7 * vb), Z = load_pose7(g.Z + 7 * k);
      double e[6], Ji[6][6], Jj[6][6];
      pgo_error(Xi, Xj, Z, e);
      pgo_linearize(Xi, Xj, Z, Ji, Jj);
      double e2 = 0;
#pragma unroll
      for (int q = 0; q < 6; q++) e2 += e[q] * e[q];
      const double w = 1.0 / (e2 + 1.0);  // Cauchy rho'
      const bool is_a = va == v;
      const double(*Jm)[6] = is_a ? Ji : Jj;   // this vertex's Jacobian
      const double(*Jo)[6] = is_a ? Jj : Ji;   // the other end's
      for (int r = 0; r < 6; r++) {
        double gsum = 0;
        for (int q = 0; q < 6; q++) gsum += Jm[q][r] * (-e[q] * w);
        bb[r] += gsum;
        for (int c = 0; c <= r; c++) {
          double h = 0;
          for (int q = 0; q < 6; q++) h += (Jm[q][r] * w) * Jm[q][c];
          Hd[6 * r + c] += h;
        }
      }
      const int ho = g.hidx[is_a ? vb : va];
      if (ho >= 0 && ho < hi) {  // block (hi, ho) of the lower triangle: Jm^T w Jo
        double* Ho = Hrow + (size_t)(ho - f) * 36;
        for (int r = 0; r < 6; r++)
          for (int c = 0; c < 6; c++) {
            double h = 0;
            for (int q = 0; q < 6; q++) h += (Jm[q][r] * w) * Jo[q][c];
            Ho[6 * r + c] += h;
          }
      }
    }
    for (int r = 0; r < 6; r++) g.b[6 * hi + r] = bb[r];
  }
}

// wave 0: block-profile Cholesky of H + lambda I into L, then x = (L L^T)^-1 b.  Returns false on a non-positive pivot.
__device__ inline bool pgo_factor_solve(const PgoGraph& g, double lambda) {
  const int lane = threadIdx.x & 63;
  const int r = lane / 6, c = lane - 6 * r;  // lanes 0..35 own entry (r, c) of the current block
  const bool act = lane < 36;
  bool ok = true;
  for (int i = 0; i < g.P && ok; i++) {
    const int f = g.first[i];
    const double* Hrow = g.H + g.boff[i] * 36;
    double* Lrow = g.L + g.boff[i] * 36;
    for (int j = f; j <= i; j++) {
      const int fj = g.first[j];
      const double* Lj = g.L + g.boff[j] * 36;  // block row j: blocks fj..j
      double s = act ? Hrow[(size_t)(j - f) * 36 + 6 * r + c] : 0.0;
      if (j == i && act && r == c) s += lambda;
      if (act) {
        for (int k = max(f, fj); k < j; k++) {
          const double* A = Lrow + (size_t)(k - f) * 36 + 6 * r;   // L(i,k)[r][:]
          const double* Bk = Lj + (size_t)(k - fj) * 36 + 6 * c;   // L(j,k)[c][:]
#pragma unroll
          for (int m = 0; m < 6; m++) s -= A[m] * Bk[m];
        }
      }
      if (j < i) {  // X Ljj^T = S, column by column
        const double* D = Lj + (size_t)(j - fj) * 36;  // Ljj (lower)
#pragma unroll
        for (int cc = 0; cc < 6; cc++) {
          const double dcc = D[7 * cc];
          if (act && c == cc) s = s / dcc;
          const double xr = __shfl(s, r * 6 + cc, 64);  // X[r][cc]
          if (act && c > cc) s -= xr * D[6 * c + cc];
        }
        if (act) Lrow[(size_t)(j - f) * 36 + 6 * r + c] = s;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
      } else {  // 6x6 Cholesky of the diagonal block (lower triangle, zeros above)
#pragma unroll
        for (int cc = 0; cc < 6; cc++) {
          const double d = __shfl(s, cc * 6 + cc, 64);
          if (!(d > 0) || !isfinite(d)) ok = false;
          const double l = sqrt(ok ? d : 1.0);
          if (act && c == cc) s = (r >= cc) ? s / l : 0.0;
          if (act && c == cc && r == cc) s = l;
          const double lr = __shfl(s, r * 6 + cc, 64), lc = __shfl(s, c * 6 + cc, 64);  // L[r][cc], L[c][cc]
          if (act && c > cc && r >= c) s -= lr * lc;
        }
        if (act) Lrow[(size_t)(i - f) * 36 + 6 * r + c] = (c <= r) ? s : 0.0;
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
      }
    }
  }
  if (!ok) return false;
  // forward: y_i = Lii^-1 (b_i - sum_k L(i,k) y_k), kept in x
  for (int i = 0; i < g.P; i++) {
    const int f = g.first[i];
    const double* Lrow = g.L + g.boff[i] * 36;
    double s = 0;
    if (lane < 6) {
      s = g.b[6 * i + lane];
      for (int k = f; k < i; k++) {
        const double* A = Lrow + (size_t)(k - f) * 36 + 6 * lane;
#pragma unroll
        for (int m = 0; m < 6; m++) s -= A[m] * g.x[6 * k + m];
      }
    }
    const double* D = Lrow + (size_t)(i - f) * 36;
#pragma unroll
    for (int cc = 0; cc < 6; cc++) {
      if (lane == cc) s = s / D[7 * cc];
      const double yc = __shfl(s, cc, 64);
      if (lane > cc && lane < 6) s -= D[6 * lane + cc] * yc;
    }
    if (lane < 6) g.x[6 * i + lane] = s;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
  }
  // backward: x_i = Lii^-T y_i, then y_k -= L(i,k)^T x_i for the blocks of row i
  for (int i = g.P - 1; i >= 0; i--) {
    const int f = g.first[i];
    const double* Lrow = g.L + g.boff[i] * 36;
    const double* D = Lrow + (size_t)(i - f) * 36;
    double s = lane < 6 ? g.x[6 * i + lane] : 0.0;
#pragma unroll
    for (int cc = 5; cc >= 0; cc--) {
      if (lane == cc) s = s / D[7 * cc];
      const double xc = __shfl(s, cc, 64);
      if (lane < cc) s -= D[6 * cc + lane] * xc;
    }
    if (lane < 6) g.x[6 * i + lane] = s;
    double xi[6];
#pragma unroll
    for (int m = 0; m < 6; m++) xi[m] = __shfl(s, m, 64);
    // lanes = (block within a group of 10, column): y_k[c] -= sum_r L(i,k)[r][c] x_i[r]
    for (int k0 = f; k0 < i; k0 += 10) {
      const int k = k0 + lane / 6, cc = lane % 6;
      if (lane < 60 && k < i) {
        const double* A = Lrow + (size_t)(k - f) * 36;
        double acc = g.x[6 * k + cc];
#pragma unroll
        for (int m = 0; m < 6; m++) acc -= A[6 * m + cc] * xi[m];
        g.x[6 * k + cc] = acc;
      }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
  }
  return true;
}

__global__ __launch_bounds__(PGO_T) void k_pgo(const PgoGraph* graphs) {
  const PgoGraph g = graphs[blockIdx.x];
  __shared__ double s_red[PGO_T / 64];
  __shared__ int s_ok;
  const int t = threadIdx.x;
  // vertices: T_w_c of the keyframe; measurements: the current relative pose / the verified loop pose
  for (int v = t; v < g.n; v += PGO_T) store_pose7(g.est + 7 * v, iso_inv(load_pose7(g.T_c_w + 7 * g.vkf[v])));
  for (int k = t; k < g.E; k += PGO_T) {
    SE3d Z;
    if (g.eloop[k] < 0) {
      const SE3d Ti = load_pose7(g.T_c_w + 7 * g.vkf[g.ea[k]]), Tj = load_pose7(g.T_c_w + 7 * g.vkf[g.eb[k]]);
      Z = iso_inv(iso_mul(Tj, iso_inv(Ti)));  // sij = (T_c_w(j) T_c_w(i)^-1)^-1
    } else {
      Z = iso_inv(load_pose7(g.loop_pose + 7 * g.eloop[k]));
    }
    store_pose7(g.Z + 7 * k, Z);
  }
  __threadfence_block();
  __syncthreads();
  if (g.use_guess && t == 0) {
    // computeInitialGuess: breadth first from the fixed vertices, neighbours in ascending index (the adjacency lists are sorted)
    int qh = 0, qt = 0;
    for (int v = 0; v < g.n; v++) {
      g.dist[v] = g.fixed[v] ? 0 : -1;
      if (g.fixed[v]) g.queue[qt++] = v;
    }
    while (qh < qt) {
      const int u = g.queue[qh++];
      const SE3d Xu = load_pose7(g.est + 7 * u);
      for (int a = g.adj_ptr[u]; a < g.adj_ptr[u + 1]; a++) {
        const int k = g.adj_e[a];
        const int z = g.ea[k] == u ? g.eb[k] : g.ea[k];
        if (g.dist[z] >= 0) continue;
        g.dist[z] = g.dist[u] + 1;
        const SE3d Zk = load_pose7(g.Z + 7 * k);
        store_pose7(g.est + 7 * z, g.ea[k] == u ? iso_mul(Xu, Zk) : iso_mul(Xu, iso_inv(Zk)));
        g.queue[qt++] = z;
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  double currentChi = pgo_chi2(g, s_red);
  const double chi_initial = currentChi;
  int done = 0;
  if (g.P > 0 && g.E > 0) {
    double lambda = 1e-10, ni = 2;
    for (int iteration = 0; iteration < g.iterations; iteration++) {
      currentChi = pgo_chi2(g, s_red);
      pgo_build(g);
      __threadfence_block();
      __syncthreads();
      double rho = 0;
      int qmax = 0;
      bool lambda_bad = false;
      do {
        for (int q = t; q < 7 * g.n; q += PGO_T) g.bak[q] = g.est[q];
        if (t < 64) {
          const bool okf = pgo_factor_solve(g, lambda);
          if (t == 0) s_ok = okf ? 1 : 0;
        }
        __threadfence_block();
        __syncthreads();
        const int ok2 = s_ok;
        double scale = 0;
        if (ok2) {
          for (int v = t; v < g.n; v += PGO_T)
            if (g.hidx[v] >= 0) store_pose7(g.est + 7 * v, iso_mul(load_pose7(g.est + 7 * v), pgo_from_mqt(g.x + 6 * g.hidx[v])));
          for (int j = t; j < 6 * g.P; j += PGO_T) scale += g.x[j] * (lambda * g.x[j] + g.b[j]);
        }
        __threadfence_block();
        __syncthreads();
        scale = pgo_block_sum(scale, s_red) + 1e-3;
        double tempChi = pgo_chi2(g, s_red);
        if (!ok2) tempChi = 1.7976931348623157e308;
        rho = (currentChi - tempChi) / scale;
        if (rho > 0 && isfinite(tempChi)) {
          double alpha = 1. - detm::det_powi(2 * rho - 1, 3);
          alpha = fmin(alpha, 2. / 3.);
          lambda *= fmax(1. / 3., alpha);
          ni = 2;
          currentChi = tempChi;
        } else {
          lambda *= ni;
          ni *= 2;
          for (int q = t; q < 7 * g.n; q += PGO_T) g.est[q] = g.bak[q];
          __threadfence_block();
          __syncthreads();
          if (!isfinite(lambda)) {
            lambda_bad = true;
            break;
          }
        }
        qmax++;
      } while (rho < 0 && qmax < 10);
      done = iteration + 1;
      if (qmax == 10 || rho == 0 || lambda_bad) break;
    }
  }
  const double chi_final = pgo_chi2(g, s_red);
  // write-back: T_c_w = (T_w_c)^-1; drift of the last optimised keyframe Tw1_w2 = (Tw2c * Tcw1)^-1
  if (t == 0 && g.n > 0) {
    const int v = g.n - 1;
    store_pose7(g.drift, iso_inv(iso_mul(load_pose7(g.est + 7 * v), load_pose7(g.T_c_w + 7 * g.vkf[v]))));
    g.stats[0] = done, g.stats[1] = chi_initial, g.stats[2] = chi_final, g.stats[3] = g.n, g.stats[4] = g.E;
  }
  __syncthreads();
  for (int v = t; v < g.n; v += PGO_T) store_pose7(g.T_c_w + 7 * g.vkf[v], iso_inv(load_pose7(g.est + 7 * v)));
}


// ---- 3-D positions of a keyframe's ORB keypoints (vo_loopclosing.cpp:255-372) ---------------------------------------------------
struct LcCam {
  double P0[12], P1[12];  // STEREO_RECT: the rectified projection matrices (dc.P0_, dc.P1_)
  double fx, fy, cx, cy;  // DEPTH_D435
  int cam_type, w, h;
};
constexpr int LC_T = 1024;
constexpr int LC_MAXF = 2048;

// keypoint rows (x, y, size, angle, response, octave) -> the two point lists of calcOpticalFlowPyrLK (lm_img1 = lm_img0, :270)
__global__ __launch_bounds__(256) void k_lc_points(const float* __restrict__ kps, const int* __restrict__ count, int cap, float* __restrict__ p0,
                                                   float* __restrict__ p1) {
  const int img = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cap) return;
  const size_t o = (size_t)img * cap + i;
  float x = 0.f, y = 0.f;
  if (i < count[img]) x = kps[o * 6], y = kps[o * 6 + 1];
  p0[o * 2] = p1[o * 2] = x;
  p0[o * 2 + 1] = p1[o * 2 + 1] = y;
}

// one workgroup per keyframe, two keypoints per thread: the mask of :280-349, then the ordered removal of :362-371
__global__ __launch_bounds__(LC_T) void k_lc_landmarks(LcCam cam, const float* __restrict__ kps, const uint8_t* desc, const int* __restrict__ count,
                                                       int cap, const float* __restrict__ next_pts, const uint8_t* __restrict__ status,
                                                       const uint16_t* __restrict__ depth, float* lm_2d, double* lm_3d, uint8_t* lm_desc,
                                                       int* __restrict__ lm_count) {
  __shared__ int s_scan[LC_T / 64];
  const int img = blockIdx.x, t = threadIdx.x;
  const int n = min(count[img], cap);
  bool keep[2] = {false, false};
  float xy[2][2];
  V3 p3[2];
  uint4 d[2][2];
#pragma unroll
  for (int e = 0; e < 2; e++) {
    const int i = 2 * t + e;
    if (i >= n) continue;
    const size_t o = (size_t)img * cap + i;
    const float x = kps[o * 6], y = kps[o * 6 + 1];
    xy[e][0] = x, xy[e][1] = y;
    const uint4* q = reinterpret_cast<const uint4*>(desc + o * 32);
    d[e][0] = q[0], d[e][1] = q[1];
    if (cam.cam_type == 0) {
      if (status[o] == 1) {
        const V3 pc = triangulate_dlt((double)x, (double)y, (double)next_pts[o * 2], (double)next_pts[o * 2 + 1], cam.P0, cam.P1);
        if (!(pc.z < 0 || pc.z > (double)100.0f)) {  // trignaulationPtFromStereo, range = 100.0 (triangulation.h:24)
          keep[e] = true;
          p3[e] = pc;
        }
      }
    } else if (cam.cam_type == 2) {
      // img1.at<ushort>(Point2f): nearest-even rounding of the position; `ushort / 1000` is an integer division (:331)
      const int ix = min(max(__float2int_rn(x), 0), cam.w - 1), iy = min(max(__float2int_rn(y), 0), cam.h - 1);
      const double dm = (double)(depth[(size_t)img * cam.w * cam.h + (size_t)iy * cam.w + ix] / 1000);
      if (dm >= 0.3 && dm <= 10) {
        keep[e] = true;
        p3[e] = V3{((double)x - cam.cx) / cam.fx * dm, ((double)y - cam.cy) / cam.fy * dm, dm};
      }
    }
  }
  int tot;
  int k = block_exclusive_scan<LC_T / 64>((int)keep[0] + (int)keep[1], s_scan, tot);  // (its barriers also order the in-place case)
#pragma unroll
  for (int e = 0; e < 2; e++) {
    if (!keep[e]) continue;
    const size_t o = (size_t)img * cap + k;
    lm_2d[o * 2] = xy[e][0], lm_2d[o * 2 + 1] = xy[e][1];
    lm_3d[o * 3] = p3[e].x, lm_3d[o * 3 + 1] = p3[e].y, lm_3d[o * 3 + 2] = p3[e].z;
    uint4* q = reinterpret_cast<uint4*>(lm_desc + o * 32);
    q[0] = d[e][0], q[1] = d[e][1];
    k++;
  }
  if (t == 0) lm_count[img] = tot;
}

}  // namespace flvis

using namespace flvis;

#define CHECK_CTX(c) \
  if (!(c)) return FLVIS_ERR_INVALID_ARG
#define CHECK_LAUNCH(c, what)                           \
  do {                                                  \
    hipError_t e_ = hipGetLastError();                  \
    if (e_ != hipSuccess) return (c)->hip_fail(e_, what); \
  } while (0)

extern "C" {

int flvis_hip_bow_set_vocabulary(flvis_ctx* ctx, int n_nodes, const int* h_child_ptr, const int* h_child_idx, const uint8_t* h_desc,
                                 const double* h_weight, const int* h_word_id) {
  CHECK_CTX(ctx);
  if (n_nodes < 2 || !h_child_ptr || !h_child_idx || !h_desc || !h_weight || !h_word_id)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "bow_set_vocabulary: bad args");
  if (h_child_ptr[0] != 0 || h_child_ptr[1] == 0) return ctx->fail(FLVIS_ERR_INVALID_ARG, "bow_set_vocabulary: node 0 must be the root");
  int n_words = 0;
  for (int n = 0; n < n_nodes; n++) {
    if (h_child_ptr[n + 1] < h_child_ptr[n]) return ctx->fail(FLVIS_ERR_INVALID_ARG, "bow_set_vocabulary: child_ptr must ascend");
    if (h_child_ptr[n + 1] == h_child_ptr[n]) {
      if (h_word_id[n] < 0 || h_word_id[n] >= INT_MAX - 1) return ctx->fail(FLVIS_ERR_INVALID_ARG, "bow_set_vocabulary: bad word id");
      n_words = std::max(n_words, h_word_id[n] + 1);
    }
  }
  const int n_edges = h_child_ptr[n_nodes];
  if (n_edges != n_nodes - 1) return ctx->fail(FLVIS_ERR_INVALID_ARG, "bow_set_vocabulary: a tree of n nodes has n - 1 child links");
  for (int c = 0; c < n_edges; c++)
    if (h_child_idx[c] <= 0 || h_child_idx[c] >= n_nodes) return ctx->fail(FLVIS_ERR_INVALID_ARG, "bow_set_vocabulary: bad child index");
  // the links must form a tree below node 0: every non-root node listed exactly once and reached from the root (k_bow_words walks
  // down until it meets a leaf: a node that is its own descendant would never let it)
  int depth = 0;
  {
    std::vector<char> listed((size_t)n_nodes, 0);
    for (int c = 0; c < n_edges; c++) {
      if (listed[h_child_idx[c]]) return ctx->fail(FLVIS_ERR_INVALID_ARG, "bow_set_vocabulary: a node is the child of two nodes");
      listed[h_child_idx[c]] = 1;
    }
    std::vector<std::pair<int, int>> stack{{0, 0}};
    size_t reached = 0;
    while (!stack.empty()) {
      const std::pair<int, int> nd = stack.back();
      stack.pop_back();
      reached++;
      depth = std::max(depth, nd.second);
      for (int c = h_child_ptr[nd.first]; c < h_child_ptr[nd.first + 1]; c++) stack.push_back({h_child_idx[c], nd.second + 1});
    }
    if (reached != (size_t)n_nodes) return ctx->fail(FLVIS_ERR_INVALID_ARG, "bow_set_vocabulary: nodes that are not reachable from the root");
  }
  std::vector<double> ww((size_t)n_words, 0.0);
  for (int n = 0; n < n_nodes; n++)
    if (h_child_ptr[n + 1] == h_child_ptr[n]) ww[h_word_id[n]] = h_weight[n];
  hipSetDevice(ctx->device);
  int* cp = (int*)ctx->scratch("voc_child_ptr", sizeof(int) * (size_t)(n_nodes + 1));
  int* ci = (int*)ctx->scratch("voc_child_idx", sizeof(int) * (size_t)std::max(n_edges, 1));
  uint8_t* ds = (uint8_t*)ctx->scratch("voc_desc", (size_t)n_nodes * 32);
  int* wi = (int*)ctx->scratch("voc_word_id", sizeof(int) * (size_t)n_nodes);
  double* wt = (double*)ctx->scratch("voc_weight", sizeof(double) * (size_t)n_nodes);
  double* wwd = (double*)ctx->scratch("voc_word_weight", sizeof(double) * (size_t)std::max(n_words, 1));
  if (!cp || !ci || !ds || !wi || !wt || !wwd) return ctx->fail(FLVIS_ERR_HIP, "bow_set_vocabulary: device allocation failed");
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) e = hipMemcpy(cp, h_child_ptr, sizeof(int) * (size_t)(n_nodes + 1), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ci, h_child_idx, sizeof(int) * (size_t)n_edges, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ds, h_desc, (size_t)n_nodes * 32, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(wi, h_word_id, sizeof(int) * (size_t)n_nodes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(wt, h_weight, sizeof(double) * (size_t)n_nodes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(wwd, ww.data(), sizeof(double) * (size_t)n_words, hipMemcpyHostToDevice);
  if (e != hipSuccess) return ctx->hip_fail(e, "bow_set_vocabulary");
  ctx->voc_nodes = n_nodes;
  ctx->voc_words = n_words;
  ctx->voc_depth = depth;
  return FLVIS_OK;
}

// STEP 1.5 / 1.6 of the loop-closing keyframe (vo_loopclosing.cpp:255-372) for n_img keyframes: which ORB keypoints get a 3-D
// position (stereo LK into img1 + DLT triangulation, or the depth image), and the keypoint / descriptor lists without the others.
int flvis_hip_lc_keyframe_landmarks(flvis_ctx* ctx, const uint8_t* d_img0, const void* d_img1, int w, int h, int n_img, int cam_type,
                                    const double* h_P0, const double* h_P1, const double* h_K4, const float* d_kps, const uint8_t* d_desc,
                                    const int* d_count, int cap, float* d_lm_2d, double* d_lm_3d, uint8_t* d_lm_desc, int* d_lm_count) {
  CHECK_CTX(ctx);
  if (!d_kps || !d_desc || !d_count || !d_lm_2d || !d_lm_3d || !d_lm_desc || !d_lm_count || n_img <= 0 || cap <= 0 || w <= 0 || h <= 0)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "lc_keyframe_landmarks: bad args");
  if (cap > LC_MAXF) return ctx->fail(FLVIS_ERR_CAPACITY, "lc_keyframe_landmarks: at most 2048 keypoints per keyframe");
  if (cam_type < 0 || cam_type > 2) return ctx->fail(FLVIS_ERR_INVALID_ARG, "lc_keyframe_landmarks: cam_type must be 0 (stereo rectified), 1 (stereo unrectified) or 2 (depth)");
  // (d_lm_desc == d_desc is fine: every workgroup reads its keyframe's rows before the scan's barrier and writes after)
  LcCam cam{};
  cam.cam_type = cam_type, cam.w = w, cam.h = h;
  hipStream_t st = ctx->stream;
  if (cam_type == 1) {  // the reference's STEREO_UNRECT case is empty (:316-322): no keypoint gets a position
    hipError_t e = hipMemsetAsync(d_lm_count, 0, sizeof(int) * (size_t)n_img, st);
    if (e != hipSuccess) return ctx->hip_fail(e, "lc_keyframe_landmarks");
    return FLVIS_OK;
  }
  if (!d_img1) return ctx->fail(FLVIS_ERR_INVALID_ARG, "lc_keyframe_landmarks: no second image");
  const float* next = nullptr;
  const uint8_t* status = nullptr;
  if (cam_type == 0) {
    if (!d_img0 || !h_P0 || !h_P1) return ctx->fail(FLVIS_ERR_INVALID_ARG, "lc_keyframe_landmarks: stereo needs img0, P0 and P1");
    memcpy(cam.P0, h_P0, sizeof(cam.P0));
    memcpy(cam.P1, h_P1, sizeof(cam.P1));
    hipSetDevice(ctx->device);
    const size_t np = (size_t)n_img * cap;
    float* p0 = (float*)ctx->scratch("lc_pts0", sizeof(float) * 2 * np);
    float* p1 = (float*)ctx->scratch("lc_pts1", sizeof(float) * 2 * np);
    uint8_t* stt = (uint8_t*)ctx->scratch("lc_status", np);
    if (!p0 || !p1 || !stt) return ctx->fail(FLVIS_ERR_HIP, "lc_keyframe_landmarks: scratch allocation failed");
    k_lc_points<<<dim3((cap + 255) / 256, n_img), 256, 0, st>>>(d_kps, d_count, cap, p0, p1);
    // calcOpticalFlowPyrLK(img0, img1, lm_img0, lm_img1, ., ., Size(31,31), 5, (COUNT+EPS, 30, 0.001), OPTFLOW_USE_INITIAL_FLOW)  (:274-278)
    const int rc = flvis_hip_lk_track(ctx, d_img0, (const uint8_t*)d_img1, w, h, n_img, p0, p1, stt, d_count, cap, 5, 30, 0.001, 1);
    if (rc != FLVIS_OK) return rc;
    next = p1, status = stt;
  } else {
    if (!h_K4) return ctx->fail(FLVIS_ERR_INVALID_ARG, "lc_keyframe_landmarks: the depth camera needs fx, fy, cx, cy");
    cam.fx = h_K4[0], cam.fy = h_K4[1], cam.cx = h_K4[2], cam.cy = h_K4[3];
  }
  k_lc_landmarks<<<n_img, LC_T, 0, st>>>(cam, d_kps, d_desc, d_count, cap, next, status, (const uint16_t*)d_img1, d_lm_2d, d_lm_3d, d_lm_desc,
                                         d_lm_count);
  CHECK_LAUNCH(ctx, "lc_keyframe_landmarks");
  return FLVIS_OK;
}

int flvis_hip_bow_transform(flvis_ctx* ctx, const uint8_t* d_desc, const int* d_count, int dcap, int n_img, int vcap, int* d_ids,
                            double* d_vals, int* d_nnz) {
  CHECK_CTX(ctx);
  if (!d_desc || !d_count || !d_ids || !d_vals || !d_nnz || dcap <= 0 || n_img <= 0 || vcap <= 0)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "bow_transform: bad args");
  if (dcap > BOW_MAXF) return ctx->fail(FLVIS_ERR_CAPACITY, "bow_transform: at most 2048 descriptors per keyframe");
  if (ctx->voc_nodes < 2) return ctx->fail(FLVIS_ERR_CONFIG, "bow_transform: no vocabulary (flvis_hip_bow_set_vocabulary)");
  if (vcap < std::min(dcap, ctx->voc_words))
    return ctx->fail(FLVIS_ERR_CAPACITY, "bow_transform: vcap must hold min(dcap, number of words) entries (a vector is never truncated)");
  VocDev v{(const int*)ctx->scratch("voc_child_ptr", 0), (const int*)ctx->scratch("voc_child_idx", 0),
           (const uint8_t*)ctx->scratch("voc_desc", 0), (const int*)ctx->scratch("voc_word_id", 0),
           (const double*)ctx->scratch("voc_weight", 0), (const double*)ctx->scratch("voc_word_weight", 0), ctx->voc_nodes, ctx->voc_words, ctx->voc_depth};
  int* words = (int*)ctx->scratch("bow_words", sizeof(int) * (size_t)dcap * n_img);
  if (!words) return ctx->fail(FLVIS_ERR_HIP, "bow_transform: scratch allocation failed");
  hipStream_t st = ctx->stream;
  k_bow_words<<<dim3((dcap + 255) / 256, n_img), 256, 0, st>>>(v, d_desc, d_count, dcap, words);
  k_bow_vector<<<n_img, BOW_T, 0, st>>>(v, words, dcap, vcap, d_ids, d_vals, d_nnz);
  CHECK_LAUNCH(ctx, "bow_transform");
  return FLVIS_OK;
}

int flvis_hip_bow_score(flvis_ctx* ctx, const int* d_q_ids, const double* d_q_vals, const int* d_q_nnz, const int* d_db_ids,
                        const double* d_db_vals, const int* d_db_nnz, int vcap, int n_db, double* d_scores) {
  CHECK_CTX(ctx);
  if (!d_q_ids || !d_q_vals || !d_q_nnz || !d_db_ids || !d_db_vals || !d_db_nnz || !d_scores || vcap <= 0 || n_db <= 0)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "bow_score: bad args");
  k_bow_score<<<(n_db + 3) / 4, 256, 0, ctx->stream>>>(d_q_ids, d_q_vals, d_q_nnz, d_db_ids, d_db_vals, d_db_nnz, vcap, n_db, d_scores);
  CHECK_LAUNCH(ctx, "bow_score");
  return FLVIS_OK;
}

int flvis_hip_bow_score_jobs(flvis_ctx* ctx, int n_jobs, const int* h_jobs3, const int* d_ids, const double* d_vals, const int* d_nnz, int vcap,
                             double* d_scores) {
  CHECK_CTX(ctx);
  if (n_jobs <= 0 || !h_jobs3 || !d_ids || !d_vals || !d_nnz || !d_scores || vcap <= 0) return ctx->fail(FLVIS_ERR_INVALID_ARG, "bow_score_jobs: bad args");
  int max_n = 0;
  for (int i = 0; i < n_jobs; i++) {
    if (h_jobs3[3 * i] < 0 || h_jobs3[3 * i + 1] < 0 || h_jobs3[3 * i + 2] < 0) return ctx->fail(FLVIS_ERR_INVALID_ARG, "bow_score_jobs: negative index");
    max_n = std::max(max_n, h_jobs3[3 * i + 2]);
  }
  if (max_n == 0) return FLVIS_OK;
  hipSetDevice(ctx->device);
  int* jobs = (int*)ctx->scratch("bow_jobs", sizeof(int) * 3 * (size_t)n_jobs);
  if (!jobs) return ctx->fail(FLVIS_ERR_HIP, "bow_score_jobs: scratch allocation failed");
  hipError_t e = hipMemcpyAsync(jobs, h_jobs3, sizeof(int) * 3 * (size_t)n_jobs, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // h_jobs3 is the caller's: done with it before returning
  if (e != hipSuccess) return ctx->hip_fail(e, "bow_score_jobs");
  k_bow_score_jobs<<<dim3((max_n + 3) / 4, n_jobs), 256, 0, ctx->stream>>>(jobs, d_ids, d_vals, d_nnz, vcap, d_scores);
  CHECK_LAUNCH(ctx, "bow_score_jobs");
  return FLVIS_OK;
}

// isLoopClosureKF's geometric check (vo_loopclosing.cpp:660-686) for n_sets candidate pairs: solvePnPRansac on the matched
// (3-D point of the earlier keyframe, pixel in the current keyframe) correspondences -- the tracker's solver (track_kernels.hip),
// P3P hypotheses, on caller arrays.
int flvis_hip_pnp_ransac(flvis_ctx* ctx, const float* d_p3d, const float* d_p2d, const int* d_count, int cap, int n_sets, const double* h_K4,
                         int iterations, double reproj_px, double confidence, const uint64_t* h_seeds, double* d_pose7,
                         uint8_t* d_inlier_mask, int* d_n_inliers) {
  CHECK_CTX(ctx);
  if (!d_p3d || !d_p2d || !d_count || !h_K4 || !h_seeds || !d_pose7 || !d_inlier_mask || !d_n_inliers || cap <= 0 || n_sets <= 0 ||
      iterations <= 0 || !(reproj_px > 0) || !(confidence > 0 && confidence < 1))
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "pnp_ransac: bad args");
  if (cap > pnp_ransac_max_points()) return ctx->fail(FLVIS_ERR_CAPACITY, "pnp_ransac: at most 1024 correspondences per set");
  hipSetDevice(ctx->device);
  unsigned long long* seeds = (unsigned long long*)ctx->scratch("pnp_seeds", sizeof(unsigned long long) * (size_t)n_sets);
  if (!seeds) return ctx->fail(FLVIS_ERR_HIP, "pnp_ransac: scratch allocation failed");
  hipError_t e = hipMemcpyAsync(seeds, h_seeds, sizeof(unsigned long long) * (size_t)n_sets, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // (h_seeds is pageable caller memory)
  if (e != hipSuccess) return ctx->hip_fail(e, "pnp_ransac seeds");
  launch_pnp_ransac_sets(ctx->stream, d_p3d, d_p2d, d_count, cap, n_sets, h_K4, 0, nullptr, seeds, iterations, reproj_px, confidence,
                         d_pose7, d_inlier_mask, d_n_inliers);
  CHECK_LAUNCH(ctx, "pnp_ransac");
  return FLVIS_OK;
}

// cv::solvePnP(..., SOLVEPNP_EPNP) alone on n_sets correspondence sets (the solver inside flvis_hip_pnp_ransac and the tracker), with its
// intermediate values: what the tests compare with the CPU restatement value by value.
int flvis_hip_debug_epnp(flvis_ctx* ctx, const float* d_p3d, const float* d_p2d, const int* d_count, int cap, int n_sets, const double* h_K4,
                         double* d_out160) {
  CHECK_CTX(ctx);
  if (!d_p3d || !d_p2d || !d_count || !h_K4 || !d_out160 || cap <= 0 || n_sets <= 0) return ctx->fail(FLVIS_ERR_INVALID_ARG, "debug_epnp: bad args");
  if (cap > pnp_ransac_max_points()) return ctx->fail(FLVIS_ERR_CAPACITY, "debug_epnp: at most 1024 correspondences per set");
  hipSetDevice(ctx->device);
  launch_epnp_sets(ctx->stream, d_p3d, d_p2d, d_count, cap, n_sets, h_K4, d_out160);
  CHECK_LAUNCH(ctx, "debug_epnp");
  return FLVIS_OK;
}

// loopClosureOnCovGraphG2ONew for n_graphs independent sequences in one launch.  Host side: the integer bookkeeping of
// vo_loopclosing.cpp:747-875 (vertex range, fixed flags, edge list) plus the adjacency lists and the block profile of the normal
// matrix; everything numeric runs in k_pgo.
int flvis_hip_pgo_loop_closure(flvis_ctx* ctx, int n_graphs, const int* h_n_kf, double* d_T_c_w7, const uint8_t* h_present,
                               const int* h_n_loops, const int* h_loop_ids, const double* d_loop_pose7, int iterations,
                               int use_initial_guess, double* d_drift7, double* d_stats5, int* h_ran) {
  CHECK_CTX(ctx);
  if (n_graphs <= 0 || !h_n_kf || !d_T_c_w7 || !h_present || !h_n_loops || !h_loop_ids || !d_loop_pose7 || !d_drift7 || !d_stats5 ||
      !h_ran || iterations < 0)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "pgo_loop_closure: bad args");
  struct HostGraph {
    int n = 0, E = 0, P = 0, n_kf = 0;
    size_t int_off = 0, dbl_off = 0;  // offsets into the packed int / double scratch
    long long nblk = 0;
    size_t kf_base = 0, loop_base = 0;
    int slot = 0;
  };
  std::vector<HostGraph> hg;
  std::vector<int> ints;          // per graph: vkf n | fixed n | hidx n | ea E | eb E | eloop E | adj_ptr n+1 | adj_e 2E | first P | queue n | dist n
  std::vector<long long> boffs;   // per graph: P + 1
  std::vector<size_t> boff_off;
  size_t kf_base = 0, loop_base = 0, dbl_total = 0;
  for (int gi = 0; gi < n_graphs; gi++) {
    const int n_kf = h_n_kf[gi], n_loops = h_n_loops[gi];
    const uint8_t* present = h_present + kf_base;
    const int* loops = h_loop_ids + 2 * loop_base;
    h_ran[gi] = 0;
    HostGraph G;
    G.n_kf = n_kf;
    G.kf_base = kf_base;
    G.loop_base = loop_base;
    G.slot = gi;
    kf_base += (size_t)std::max(n_kf, 0);
    loop_base += (size_t)std::max(n_loops, 0);
    if (n_kf <= 0 || n_loops <= 0) continue;
    long long kf_prev = 2LL * n_kf, kf_curr = 0;  // vo_loopclosing.cpp:747-756
    bool valid = true;
    for (int k = 0; k < n_loops; k++) {
      const int a = loops[2 * k], b = loops[2 * k + 1];
      if (a < 0 || b < 0 || a >= n_kf || b >= n_kf || !present[a] || !present[b]) valid = false;
      kf_prev = std::min<long long>(kf_prev, a);
      kf_curr = std::max<long long>(kf_curr, b);
    }
    if (!valid || kf_prev > kf_curr) continue;  // (the reference would dereference a missing vertex)
    std::vector<int> vid((size_t)n_kf, -1), vkf, fixed;
    for (long long i = kf_prev; i <= kf_curr; i++) {
      if (!present[i]) continue;
      bool later_end = false;  // :786-800: the first loop that names i decides
      for (int k = 0; k < n_loops; k++) {
        if (loops[2 * k] == i) break;
        if (loops[2 * k + 1] == i) {
          later_end = true;
          break;
        }
      }
      vid[i] = (int)vkf.size();
      vkf.push_back((int)i);
      fixed.push_back((!later_end && (i == 0 || i == kf_prev)) ? 1 : 0);
    }
    std::vector<int> ea, eb, eloop;
    for (long long i = kf_prev; i <= kf_curr; i++)
      for (long long j = i + 1; j <= std::min(kf_curr, i + 5); j++)
        if (present[i] && present[j]) {
          ea.push_back(vid[i]);
          eb.push_back(vid[j]);
          eloop.push_back(-1);
        }
    for (int k = 0; k < n_loops; k++) {
      ea.push_back(vid[loops[2 * k]]);
      eb.push_back(vid[loops[2 * k + 1]]);
      eloop.push_back(k);
    }
    const int n = (int)vkf.size(), E = (int)ea.size();
    std::vector<int> hidx((size_t)n, -1);
    int P = 0;
    for (int v = 0; v < n; v++)
      if (!fixed[v]) hidx[v] = P++;
    std::vector<std::vector<std::pair<int, int>>> adj((size_t)n);  // (neighbour, edge)
    for (int k = 0; k < E; k++) {
      adj[ea[k]].push_back({eb[k], k});
      adj[eb[k]].push_back({ea[k], k});
    }
    std::vector<int> first((size_t)P);
    for (int i = 0; i < P; i++) first[i] = i;
    for (int k = 0; k < E; k++) {
      const int ia = hidx[ea[k]], ib = hidx[eb[k]];
      if (ia >= 0 && ib >= 0) first[std::max(ia, ib)] = std::min(first[std::max(ia, ib)], std::min(ia, ib));
    }
    G.n = n, G.E = E, G.P = P;
    G.int_off = ints.size();
    ints.insert(ints.end(), vkf.begin(), vkf.end());
    ints.insert(ints.end(), fixed.begin(), fixed.end());
    ints.insert(ints.end(), hidx.begin(), hidx.end());
    ints.insert(ints.end(), ea.begin(), ea.end());
    ints.insert(ints.end(), eb.begin(), eb.end());
    ints.insert(ints.end(), eloop.begin(), eloop.end());
    int run = 0;
    for (int v = 0; v < n; v++) {
      ints.push_back(run);
      run += (int)adj[v].size();
    }
    ints.push_back(run);
    for (int v = 0; v < n; v++) {
      std::sort(adj[v].begin(), adj[v].end());
      for (auto& pr : adj[v]) ints.push_back(pr.second);
    }
    ints.insert(ints.end(), first.begin(), first.end());
    ints.insert(ints.end(), (size_t)2 * n, 0);  // queue, dist
    boff_off.push_back(boffs.size());
    long long nb = 0;
    for (int i = 0; i < P; i++) {
      boffs.push_back(nb);
      nb += i - first[i] + 1;
    }
    boffs.push_back(nb);
    G.nblk = nb;
    G.dbl_off = dbl_total;
    dbl_total += (size_t)14 * n + (size_t)7 * E + (size_t)72 * nb + (size_t)12 * P + 16;
    h_ran[gi] = 1;
    hg.push_back(G);
  }
  if (hg.empty()) return FLVIS_OK;
  hipSetDevice(ctx->device);
  int* d_ints = (int*)ctx->scratch("pgo_ints", sizeof(int) * ints.size());
  long long* d_boff = (long long*)ctx->scratch("pgo_boff", sizeof(long long) * boffs.size());
  double* d_dbl = (double*)ctx->scratch("pgo_dbl", sizeof(double) * dbl_total);
  PgoGraph* d_graphs = (PgoGraph*)ctx->scratch("pgo_graphs", sizeof(PgoGraph) * hg.size());
  if (!d_ints || !d_boff || !d_dbl || !d_graphs) return ctx->fail(FLVIS_ERR_HIP, "pgo_loop_closure: device allocation failed");
  std::vector<PgoGraph> descs(hg.size());
  for (size_t q = 0; q < hg.size(); q++) {
    const HostGraph& G = hg[q];
    PgoGraph& d = descs[q];
    const int n = G.n, E = G.E, P = G.P;
    int* ip = d_ints + G.int_off;
    d.n = n, d.E = E, d.P = P, d.n_kf = G.n_kf, d.iterations = iterations, d.use_guess = use_initial_guess;
    d.vkf = ip, ip += n;
    d.fixed = ip, ip += n;
    d.hidx = ip, ip += n;
    d.ea = ip, ip += E;
    d.eb = ip, ip += E;
    d.eloop = ip, ip += E;
    d.adj_ptr = ip, ip += n + 1;
    d.adj_e = ip, ip += 2 * E;
    d.first = ip, ip += P;
    d.queue = ip, ip += n;
    d.dist = ip;
    d.boff = d_boff + boff_off[q];
    double* dp = d_dbl + G.dbl_off;
    d.est = dp, dp += (size_t)7 * n;
    d.bak = dp, dp += (size_t)7 * n;
    d.Z = dp, dp += (size_t)7 * E;
    d.H = dp, dp += (size_t)36 * G.nblk;
    d.L = dp, dp += (size_t)36 * G.nblk;
    d.b = dp, dp += (size_t)6 * P;
    d.x = dp;
    d.T_c_w = d_T_c_w7 + 7 * G.kf_base;
    d.loop_pose = d_loop_pose7 + 7 * G.loop_base;
    d.drift = d_drift7 + 7 * (size_t)G.slot;
    d.stats = d_stats5 + 5 * (size_t)G.slot;
  }
  hipStream_t st = ctx->stream;
  hipError_t e = hipStreamSynchronize(st);  // (pageable host vectors: plain synchronous copies)
  if (e == hipSuccess) e = hipMemcpy(d_ints, ints.data(), sizeof(int) * ints.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_boff, boffs.data(), sizeof(long long) * boffs.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_graphs, descs.data(), sizeof(PgoGraph) * descs.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) return ctx->hip_fail(e, "pgo_loop_closure upload");
  k_pgo<<<(int)hg.size(), PGO_T, 0, st>>>(d_graphs);
  CHECK_LAUNCH(ctx, "pgo_loop_closure");
  return FLVIS_OK;
}

// isLoopCandidate (vo_loopclosing.cpp:520-590): control logic on one row of the similarity matrix, on the host as in the reference
// (its pgoProcess thread).  h_row[i] = sim_matrix[i][g_size - 1]; h_present[i] = the i-th keyframe exists.
int flvis_loop_candidate(int g_size, const double* h_row, const uint8_t* h_present, int lcKFDist, int lcKFMaxDist, int lcNKFClosest,
                         double minScore, int64_t* kf_prev_idx) {
  if (!h_row || !h_present || !kf_prev_idx || g_size < 0) return FLVIS_ERR_INVALID_ARG;
  const long long recent0 = (long long)g_size - lcKFDist;  // the last lcKFDist keyframes are neighbours, not loop partners
  if (g_size < 40 || recent0 <= 0) return 0;
  const long long first = recent0 > 5000 ? recent0 - 5000 : 0;
  std::vector<std::pair<double, int>> older;
  older.reserve((size_t)(recent0 - first));
  for (long long i = first; i < recent0; i++)
    if (h_present[i]) older.emplace_back(h_row[i], (int)i);
  if (older.empty()) return 0;
  // descending score; equal scores keep the earlier keyframe first (std::sort leaves their order open in the reference)
  std::stable_sort(older.begin(), older.end(), [](const std::pair<double, int>& x, const std::pair<double, int>& y) { return x.first > y.first; });
  double floor_score = 1.0;
  for (long long i = recent0; i < g_size; i++)
    if (h_row[i] < floor_score && h_row[i] > 0.001) floor_score = h_row[i];
  floor_score = std::min(floor_score, 0.4);
  const double top = older[0].first;
  if (top < std::max(minScore, floor_score)) return 0;
  int support = 0;
  if (top >= floor_score)
    for (size_t i = 1; i < older.size(); i++)
      if (std::abs(older[i].second - older[0].second) <= lcKFMaxDist && older[i].first >= floor_score * 0.8) support++;
  if (support >= lcNKFClosest && top > minScore) {
    *kf_prev_idx = older[0].second;
    return 1;
  }
  return 0;
}

}  // extern "C"
