// TEST INFRASTRUCTURE (CPU oracle, see oracle/README.md): restatement of the place-recognition half of the reference's loop closing
// (SURVEY 8f-4): DBoW3 bag-of-words of a keyframe's ORB descriptors, the L1 similarity against all earlier keyframes and the
// loop-candidate selection on one row of the similarity matrix.
//
//   src/backend/vo_loopclosing.cpp:249-253   voc.transform(kf.lm_descriptor, kf_bv)
//   src/backend/vo_loopclosing.cpp:417-437   sim_matrix row: voc.score(kf_bv, kf_lc_tmp[i]->kf_bv)
//   src/backend/vo_loopclosing.cpp:520-590   isLoopCandidate
//   3rdPartLib/DBow3/src/Vocabulary.cpp:628-688,836-874   transform (tree descent by Hamming distance, first minimum wins)
//   3rdPartLib/DBow3/src/BowVector.cpp:26-41,62-84        addWeight, normalize(L1)
//   3rdPartLib/DBow3/src/ScoringObject.cpp:23-68          L1Scoring::score
//   3rdPartLib/DBow3/src/DescManip.cpp:92-119             Hamming distance of 32-byte descriptors
//
// The vocabulary FILE the reference loads (vo_loopclosing.cpp:1097) is not shipped with it; the tree is handed over as flat
// arrays (children of node n: child_idx[child_ptr[n] .. child_ptr[n+1]), node descriptors, word id and weight of the leaves).
// Weighting TF_IDF and scoring L1_NORM (DBoW3's defaults and what the ORB vocabularies are built with) are the only ones restated.
// Pinning: the VALUE half of the transform (addWeight once per feature with a positive weight, normalize(L1)) is pinned on the
// reference's own BowVector.cpp, which builds from the standard library alone (oracle/bowvector_ref.cpp ->
// tests/golden/bowvector_ref.npz, bit-exact).  parity unpinned for the rest: the tree descent, the Hamming distance and the L1 score
// live in sources that need OpenCV (Vocabulary.cpp, DescManip.cpp, ScoringObject.cpp via Vocabulary.h) and DBoW3's tests
// (3rdPartLib/DBow3/tests) read vocabulary / image files that are not in the reference; those are checked against an independent
// plain-Python restatement of the same lines (tests/_voc.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

namespace ref {

struct Vocabulary {
  std::vector<int> child_ptr, child_idx, word_id;
  std::vector<uint8_t> desc;  // [n_nodes][32]
  std::vector<double> weight;
  int n_nodes = 0;
  bool is_leaf(int n) const { return child_ptr[n + 1] == child_ptr[n]; }
};

static inline int hamming256(const uint8_t* a, const uint8_t* b) {  // DescManip::distance, binary branch
  int d = 0;
  for (int i = 0; i < 4; i++) {
    uint64_t x, y;
    memcpy(&x, a + 8 * i, 8);
    memcpy(&y, b + 8 * i, 8);
    d += __builtin_popcountll(x ^ y);
  }
  return d;
}

// Vocabulary::transform(feature, word_id, weight): Vocabulary.cpp:836-874
static void word_of(const Vocabulary& v, const uint8_t* f, int& word, double& weight) {
  int node = 0;
  do {
    int best_d = INT32_MAX, best = node;
    for (int c = v.child_ptr[node]; c < v.child_ptr[node + 1]; c++) {
      const int id = v.child_idx[c];
      const int d = hamming256(f, &v.desc[(size_t)id * 32]);
      if (d < best_d) {
        best_d = d;
        best = id;
      }
    }
    node = best;
  } while (!v.is_leaf(node));
  word = v.word_id[node];
  weight = v.weight[node];
}

// Vocabulary::transform(features, BowVector) for TF_IDF + L1: Vocabulary.cpp:628-688
static void transform(const Vocabulary& v, int n, const uint8_t* desc, std::map<int, double>& bow) {
  bow.clear();
  if (v.n_nodes <= 1) return;
  for (int r = 0; r < n; r++) {
    int id;
    double w;
    word_of(v, desc + (size_t)r * 32, id, w);
    if (w > 0) {  // BowVector::addWeight
      auto it = bow.lower_bound(id);
      if (it != bow.end() && !(bow.key_comp()(id, it->first)))
        it->second += w;
      else
        bow.insert(it, {id, w});
    }
  }
  double norm = 0.0;  // BowVector::normalize(L1): ascending word id
  for (auto& kv : bow) norm += std::fabs(kv.second);
  if (norm > 0.0)
    for (auto& kv : bow) kv.second /= norm;
}

// L1Scoring::score on two sorted sparse vectors: ScoringObject.cpp:23-68
static double l1_score(int n1, const int* id1, const double* v1, int n2, const int* id2, const double* v2) {
  int a = 0, b = 0;
  double score = 0;
  while (a < n1 && b < n2) {
    if (id1[a] == id2[b]) {
      const double vi = v1[a], wi = v2[b];
      score += std::fabs(vi - wi) - std::fabs(vi) - std::fabs(wi);
      a++;
      b++;
    } else if (id1[a] < id2[b]) {
      a = (int)(std::lower_bound(id1 + a, id1 + n1, id2[b]) - id1);
    } else {
      b = (int)(std::lower_bound(id2 + b, id2 + n2, id1[a]) - id2);
    }
  }
  return -score / 2.0;
}

}  // namespace ref

extern "C" {

void* ref_voc_create(int n_nodes, const int* child_ptr, const int* child_idx, const uint8_t* desc, const double* weight,
                     const int* word_id) {
  ref::Vocabulary* v = new ref::Vocabulary();
  v->n_nodes = n_nodes;
  v->child_ptr.assign(child_ptr, child_ptr + n_nodes + 1);
  v->child_idx.assign(child_idx, child_idx + child_ptr[n_nodes]);
  v->desc.assign(desc, desc + (size_t)n_nodes * 32);
  v->weight.assign(weight, weight + n_nodes);
  v->word_id.assign(word_id, word_id + n_nodes);
  return v;
}
void ref_voc_destroy(void* h) { delete (ref::Vocabulary*)h; }

// word id per descriptor (the discrete half of the transform)
void ref_voc_words(void* h, int n, const uint8_t* desc, int* words) {
  const ref::Vocabulary& v = *(ref::Vocabulary*)h;
  for (int r = 0; r < n; r++) {
    double w;
    ref::word_of(v, desc + (size_t)r * 32, words[r], w);
  }
}
// BoW vector of n descriptors: ascending word ids + L1-normalised values; returns the number of entries
int ref_voc_transform(void* h, int n, const uint8_t* desc, int cap, int* ids, double* vals) {
  std::map<int, double> bow;
  ref::transform(*(ref::Vocabulary*)h, n, desc, bow);
  int k = 0;
  for (auto& kv : bow) {
    if (k < cap) {
      ids[k] = kv.first;
      vals[k] = kv.second;
    }
    k++;
  }
  return k;
}
double ref_bow_score(int n1, const int* id1, const double* v1, int n2, const int* id2, const double* v2) {
  return ref::l1_score(n1, id1, v1, n2, id2, v2);
}

// isLoopCandidate (vo_loopclosing.cpp:520-590) on the newest keyframe's row of the similarity matrix: row[i] =
// sim_matrix[i][g_size-1], present[i] = kf_lc_tmp[i] != nullptr.  Returns 1 and the index of the earlier keyframe when a
// candidate is found.  std::sort's order among EQUAL scores is unspecified in the reference; here ties keep the lower index first.
int ref_loop_candidate(int g_size, const double* row, const uint8_t* present, int lcKFDist, int lcKFMaxDist, int lcNKFClosest,
                       double minScore, int64_t* kf_prev_idx) {
  if (g_size < 40) return 0;
  if (g_size - lcKFDist <= 0) return 0;  // (the reference would index max_sim_mat[0] of an empty vector)
  const int64_t hi = (int64_t)g_size - lcKFDist;
  const int64_t lo = hi > 5000 ? hi - 5000 : 0;
  std::vector<std::pair<double, int>> cand;  // (score, index)
  for (int64_t i = lo; i < hi; i++)
    if (present[i]) cand.push_back({row[i], (int)i});
  if (cand.empty()) return 0;
  std::stable_sort(cand.begin(), cand.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first > b.first; });
  double lc_min_score = 1.0;  // the minimum score among the last lcKFDist keyframes (covisible with the newest one)
  for (int64_t i = hi; i < g_size; i++) {
    const double s = row[i];
    if (s < lc_min_score && s > 0.001) lc_min_score = s;
  }
  lc_min_score = std::min(lc_min_score, 0.4);
  if (cand[0].first < std::max(minScore, lc_min_score)) return 0;
  const int idx_max = cand[0].second;
  int nkf_closest = 0;
  if (cand[0].first >= lc_min_score)
    for (size_t i = 1; i < cand.size(); i++)
      if (std::abs(cand[i].second - idx_max) <= lcKFMaxDist && cand[i].first >= lc_min_score * 0.8) nkf_closest++;
  if (nkf_closest >= lcNKFClosest && cand[0].first > minScore) {
    *kf_prev_idx = idx_max;
    return 1;
  }
  return 0;
}
}
