// flvis_amd: the order std::sort leaves EQUAL keys in.
//
// FeatureDEM sorts the corner candidates of a region by their (quirky, integer-built) Harris score with
//   sort(region.begin(), region.end(), sortbysecdesc)            feature_dem.cpp:170,230
// std::sort is not stable: where two candidates tie, their order is whatever the implementation's algorithm leaves -- deterministic,
// and it decides which of them the greedy spacing walk sees first.  The reference is built with GCC, so "the implementation" is
// libstdc++'s introsort; its behaviour is restated here from the published algorithm (Musser's introsort as libstdc++ arranges it):
//   * 2 floor(log2 n) levels of quicksort on ranges longer than 16: median of (first + 1, middle, last - 1) moved to the front as
//     the pivot, unguarded Hoare partition of the rest; the right part is handled first, the loop continues on the left part;
//   * a range that exhausts the depth budget is heap-sorted (make-heap + pop-heap with the sift-down-to-a-leaf-then-push-up variant);
//   * one final insertion sort over the whole array: guarded over the first 16 elements, unguarded over the rest.
// The function sorts an array of candidate indices by score[index], descending, and leaves ties exactly where that algorithm leaves
// them.  Plain C++ (host and device): tests/cpp/dem_sort_check.cpp compares it with the real std::sort of this toolchain (the CPU
// checker of the parity tests calls std::sort itself, it does not use this header).
#pragma once

#if defined(__HIPCC__)
#define FLVIS_DS_HD __host__ __device__ inline
#else
#define FLVIS_DS_HD inline
#endif

namespace flvis {
namespace demsort {

constexpr int THRESHOLD = 16;

// sortbysecdesc on candidate indices
template <typename I>
FLVIS_DS_HD bool before(const float* score, I a, I b) {
  return score[a] > score[b];
}

template <typename I>
FLVIS_DS_HD void push_heap(I* v, int hole, int top, I value, const float* score) {
  int parent = (hole - 1) / 2;
  while (hole > top && before(score, v[parent], value)) {
    v[hole] = v[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  v[hole] = value;
}

template <typename I>
FLVIS_DS_HD void adjust_heap(I* v, int hole, int len, I value, const float* score) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (before(score, v[child], v[child - 1])) child--;
    v[hole] = v[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    v[hole] = v[child - 1];
    hole = child - 1;
  }
  push_heap(v, hole, top, value, score);
}

template <typename I>
FLVIS_DS_HD void heap_sort(I* v, int len, const float* score) {
  if (len >= 2) {
    int parent = (len - 2) / 2;
    while (true) {
      const I value = v[parent];
      adjust_heap(v, parent, len, value, score);
      if (parent == 0) break;
      parent--;
    }
  }
  int last = len;
  while (last > 1) {
    --last;
    const I value = v[last];
    v[last] = v[0];
    adjust_heap(v, 0, last, value, score);
  }
}

template <typename I>
FLVIS_DS_HD void unguarded_linear_insert(I* v, int last, const float* score) {
  const I val = v[last];
  int next = last - 1;
  while (before(score, val, v[next])) {
    v[last] = v[next];
    last = next;
    --next;
  }
  v[last] = val;
}

template <typename I>
FLVIS_DS_HD void insertion_sort(I* v, int first, int last, const float* score) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (before(score, v[i], v[first])) {
      const I val = v[i];
      for (int k = i; k > first; --k) v[k] = v[k - 1];
      v[first] = val;
    } else {
      unguarded_linear_insert(v, i, score);
    }
  }
}

// the quicksort levels; depth = levels this range may still use (std::sort starts with 2 floor(log2 n))
template <typename I>
FLVIS_DS_HD void introsort_loop(I* v, int first0, int last0, int depth0, const float* score) {
  // the recursion on the right part as an explicit stack: the parts are disjoint ranges, so the order they are handled in does not
  // change where anything ends up.  A range pushed at budget d is at most n / 2^(levels used): 64 entries hold any int-sized array.
  int st_first[64], st_last[64], st_depth[64];
  int sp = 0;
  st_first[0] = first0;
  st_last[0] = last0;
  st_depth[0] = depth0;
  sp = 1;
  while (sp > 0) {
    --sp;
    const int first = st_first[sp];
    int last = st_last[sp], depth = st_depth[sp];
    while (last - first > THRESHOLD) {
      if (depth == 0) {
        heap_sort(v + first, last - first, score);
        break;
      }
      --depth;
      // median of three to the front
      const int a = first + 1, b = first + (last - first) / 2, c = last - 1;
      int m;
      if (before(score, v[a], v[b])) {
        if (before(score, v[b], v[c])) m = b;
        else if (before(score, v[a], v[c])) m = c;
        else m = a;
      } else if (before(score, v[a], v[c])) m = a;
      else if (before(score, v[b], v[c])) m = c;
      else m = b;
      {
        const I t = v[first];
        v[first] = v[m];
        v[m] = t;
      }
      // unguarded partition of [first + 1, last) around the pivot at `first`
      const I pivot = v[first];
      int f = first + 1, l = last;
      while (true) {
        while (before(score, v[f], pivot)) ++f;
        --l;
        while (before(score, pivot, v[l])) --l;
        if (!(f < l)) break;
        const I t = v[f];
        v[f] = v[l];
        v[l] = t;
        ++f;
      }
      st_first[sp] = f;
      st_last[sp] = last;
      st_depth[sp] = depth;
      ++sp;
      last = f;
    }
  }
}

FLVIS_DS_HD int floor_log2(int n) {
  int k = 0;
  while (n > 1) {
    n >>= 1;
    ++k;
  }
  return k;
}

// v[0 .. n): candidate indices; afterwards sorted by score[index] descending, ties where libstdc++'s std::sort leaves them.
// depth < 0: std::sort's own budget.
template <typename I>
FLVIS_DS_HD void sort_desc(I* v, int n, const float* score, int depth = -1) {
  if (n <= 0) return;
  introsort_loop(v, 0, n, depth < 0 ? 2 * floor_log2(n) : depth, score);
  if (n > THRESHOLD) {
    insertion_sort(v, 0, THRESHOLD, score);
    for (int i = THRESHOLD; i != n; ++i) unguarded_linear_insert(v, i, score);
  } else {
    insertion_sort(v, 0, n, score);
  }
}

}  // namespace demsort
}  // namespace flvis
