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
// The algorithm is written over an ACCESSOR (get / set / before on element values), because the device runs it in two forms: on an
// index array in LDS (any size) and -- the fast one -- on an array that lives in one or two vector registers ACROSS the lanes of a
// wavefront, read and written with v_readlane / v_writelane: every index is wave-uniform, so the whole sequential algorithm runs on the
// scalar unit at a few cycles per element access instead of an LDS round trip (img_kernels.hip: k_feature_dem_prep).  Plain C++ for
// the array form (host and device): tests/cpp/dem_sort_check.cpp compares it with the real std::sort of this toolchain (the CPU checker
// of the parity tests calls std::sort itself, it does not use this header).
#pragma once

#if defined(__HIPCC__)
#define FLVIS_DS_HD __host__ __device__ inline
#else
#define FLVIS_DS_HD inline
#endif

namespace flvis {
namespace demsort {

constexpr int THRESHOLD = 16;
constexpr int STACK = 64;  // ranges waiting for their quicksort levels: a range pushed at level d is at most n / 2^d long

// accessor of an array of candidate indices sorted by score[index] descending (sortbysecdesc)
template <typename I>
struct IndexArray {
  I* v;
  const float* score;
  typedef I value_type;
  FLVIS_DS_HD I get(int i) const { return v[i]; }
  FLVIS_DS_HD void set(int i, I x) const { v[i] = x; }
  FLVIS_DS_HD bool before(I a, I b) const { return score[a] > score[b]; }
};

template <class A>
FLVIS_DS_HD void push_heap(const A& a, int first, int hole, int top, typename A::value_type value) {
  int parent = (hole - 1) / 2;
  while (hole > top && a.before(a.get(first + parent), value)) {
    a.set(first + hole, a.get(first + parent));
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a.set(first + hole, value);
}

template <class A>
FLVIS_DS_HD void adjust_heap(const A& a, int first, int hole, int len, typename A::value_type value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (a.before(a.get(first + child), a.get(first + child - 1))) child--;
    a.set(first + hole, a.get(first + child));
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    a.set(first + hole, a.get(first + child - 1));
    hole = child - 1;
  }
  push_heap(a, first, hole, top, value);
}

template <class A>
FLVIS_DS_HD void heap_sort(const A& a, int first, int len) {
  if (len >= 2) {
    int parent = (len - 2) / 2;
    while (true) {
      adjust_heap(a, first, parent, len, a.get(first + parent));
      if (parent == 0) break;
      parent--;
    }
  }
  int last = len;
  while (last > 1) {
    --last;
    const typename A::value_type value = a.get(first + last);
    a.set(first + last, a.get(first));
    adjust_heap(a, first, 0, last, value);
  }
}

template <class A>
FLVIS_DS_HD void unguarded_linear_insert(const A& a, int last) {
  const typename A::value_type val = a.get(last);
  int next = last - 1;
  while (true) {
    const typename A::value_type nv = a.get(next);
    if (!a.before(val, nv)) break;
    a.set(last, nv);
    last = next;
    --next;
  }
  a.set(last, val);
}

template <class A>
FLVIS_DS_HD void insertion_sort(const A& a, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    const typename A::value_type val = a.get(i);
    if (a.before(val, a.get(first))) {
      for (int k = i; k > first; --k) a.set(k, a.get(k - 1));
      a.set(first, val);
    } else {
      unguarded_linear_insert(a, i);
    }
  }
}

// the quicksort levels; depth = levels this range may still use (std::sort starts with 2 floor(log2 n)).  stack: 3 * STACK ints
// (the recursion on the right part as an explicit stack: the parts are disjoint ranges, so the order they are handled in does not
// change where anything ends up)
template <class A>
FLVIS_DS_HD void introsort_loop(const A& a, int first0, int last0, int depth0, int* stack) {
  int* const st_first = stack;
  int* const st_last = stack + STACK;
  int* const st_depth = stack + 2 * STACK;
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
        heap_sort(a, first, last - first);
        break;
      }
      --depth;
      // median of three to the front
      const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
      const typename A::value_type va = a.get(ia), vb = a.get(ib), vc = a.get(ic);
      int m;
      if (a.before(va, vb)) {
        if (a.before(vb, vc)) m = ib;
        else if (a.before(va, vc)) m = ic;
        else m = ia;
      } else if (a.before(va, vc)) m = ia;
      else if (a.before(vb, vc)) m = ic;
      else m = ib;
      const typename A::value_type pivot = m == ia ? va : (m == ib ? vb : vc);
      a.set(m, a.get(first));
      a.set(first, pivot);
      // unguarded partition of [first + 1, last) around the pivot at `first`
      int f = first + 1, l = last;
      while (true) {
        typename A::value_type vf = a.get(f);
        while (a.before(vf, pivot)) {
          ++f;
          vf = a.get(f);
        }
        --l;
        typename A::value_type vl = a.get(l);
        while (a.before(pivot, vl)) {
          --l;
          vl = a.get(l);
        }
        if (!(f < l)) break;
        a.set(f, vl);
        a.set(l, vf);
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

// std::sort's FIRST phase alone (the quicksort levels, heap sort where the budget runs out): what is left is one insertion sort over
// the whole array, and an insertion sort is STABLE -- so std::sort's result is the stable sort of the array this function leaves.  A
// caller that can rank in parallel (position = number of elements that sort before + number of equal elements earlier in THIS array)
// runs only this phase sequentially: n log2(n / 16) element visits instead of the insertion sort's n^2 / 64 on top.
template <class A>
FLVIS_DS_HD void quicksort_phase(const A& a, int n, int* stack, int depth = -1) {
  if (n <= 0) return;
  introsort_loop(a, 0, n, depth < 0 ? 2 * floor_log2(n) : depth, stack);
}

// a[0 .. n) sorted by a.before, ties where libstdc++'s std::sort leaves them.  depth < 0: std::sort's own budget.
template <class A>
FLVIS_DS_HD void sort_with(const A& a, int n, int* stack, int depth = -1) {
  if (n <= 0) return;
  introsort_loop(a, 0, n, depth < 0 ? 2 * floor_log2(n) : depth, stack);
  if (n > THRESHOLD) {
    insertion_sort(a, 0, THRESHOLD);
    for (int i = THRESHOLD; i != n; ++i) unguarded_linear_insert(a, i);
  } else {
    insertion_sort(a, 0, n);
  }
}

// v[0 .. n): candidate indices; afterwards sorted by score[index] descending
template <typename I>
FLVIS_DS_HD void sort_desc(I* v, int n, const float* score, int depth = -1) {
  int stack[3 * STACK];
  sort_with(IndexArray<I>{v, score}, n, stack, depth);
}

}  // namespace demsort
}  // namespace flvis
