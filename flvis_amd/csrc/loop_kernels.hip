// flvis_amd: place recognition + pose-graph optimisation of the reference's loop closing for gfx950, next to
// the ORB extraction / matching of orb_kernels.hip (§8f-1):
//   * the DBoW3 bag-of-words of a keyframe's ORB descriptors   voc.transform(kf.lm_descriptor, kf_bv)   vo_loopclosing.cpp:249-253
//   * one row of the similarity matrix                          voc.score(kf_bv, kf_lc_tmp[i]->kf_bv)    vo_loopclosing.cpp:417-437
//   * the loop-candidate selection on that row (host logic)     isLoopCandidate                          vo_loopclosing.cpp:520-590
// DBoW3 semantics (3rdPartLib/DBow3/src): the tree descent takes the FIRST child of minimal Hamming distance
// (Vocabulary.cpp:836-874), a word's value is its idf weight added once per occurrence (BowVector.cpp:34-46), the vector is
// L1-normalised with the norm summed in ascending word order (BowVector.cpp:62-84), and the L1 score is summed over the common
// words in ascending order (ScoringObject.cpp:23-68).  The fp64 sums are kept SEQUENTIAL in exactly that order, so ids, values
// and scores are bit-identical to the CPU restatement (oracle/ref_bow.cpp): the candidate selection thresholds the scores.
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
    while (true) {
      const int c0 = v.child_ptr[node], c1 = v.child_ptr[node + 1];
      if (c0 == c1) break;  // leaf
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
    if (v.weight[node] > 0) out = v.word_id[node];
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

// one wave per database vector: L1Scoring::score(query, db[j]); db_nnz[j] < 0 marks an absent keyframe (score 0)
__global__ __launch_bounds__(256) void k_bow_score(const int* __restrict__ q_ids, const double* __restrict__ q_vals, const int* __restrict__ q_nnz,
                                                   const int* __restrict__ db_ids, const double* __restrict__ db_vals,
                                                   const int* __restrict__ db_nnz, int vcap, int n_db, double* __restrict__ scores) {
  __shared__ double s_term[4][64];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + wv;
  if (j >= n_db) return;
  const int nd = db_nnz[j], nq = q_nnz[0];
  if (nd < 0) {
    if (lane == 0) scores[j] = 0.0;
    return;
  }
  const int* di = db_ids + (size_t)j * vcap;
  const double* dv = db_vals + (size_t)j * vcap;
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
    s_term[wv][lane] = term;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      unsigned long long mm = m;
      while (mm) {  // common words in ascending order
        const int b = __ffsll((long long)mm) - 1;
        score += s_term[wv][b];
        mm &= mm - 1;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) scores[j] = -score / 2.0;
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
  for (int c = 0; c < n_edges; c++)
    if (h_child_idx[c] <= 0 || h_child_idx[c] >= n_nodes) return ctx->fail(FLVIS_ERR_INVALID_ARG, "bow_set_vocabulary: bad child index");
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
  return FLVIS_OK;
}

int flvis_hip_bow_transform(flvis_ctx* ctx, const uint8_t* d_desc, const int* d_count, int dcap, int n_img, int vcap, int* d_ids,
                            double* d_vals, int* d_nnz) {
  CHECK_CTX(ctx);
  if (!d_desc || !d_count || !d_ids || !d_vals || !d_nnz || dcap <= 0 || n_img <= 0 || vcap <= 0)
    return ctx->fail(FLVIS_ERR_INVALID_ARG, "bow_transform: bad args");
  if (dcap > BOW_MAXF) return ctx->fail(FLVIS_ERR_CAPACITY, "bow_transform: at most 2048 descriptors per keyframe");
  if (ctx->voc_nodes < 2) return ctx->fail(FLVIS_ERR_CONFIG, "bow_transform: no vocabulary (flvis_hip_bow_set_vocabulary)");
  VocDev v{(const int*)ctx->scratch("voc_child_ptr", 0), (const int*)ctx->scratch("voc_child_idx", 0),
           (const uint8_t*)ctx->scratch("voc_desc", 0), (const int*)ctx->scratch("voc_word_id", 0),
           (const double*)ctx->scratch("voc_weight", 0), (const double*)ctx->scratch("voc_word_weight", 0), ctx->voc_nodes, ctx->voc_words};
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
