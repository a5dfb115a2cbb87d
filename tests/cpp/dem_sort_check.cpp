// CPU check of flvis_amd/csrc/dem_sort.hpp (the order the device leaves tied FeatureDEM candidates in) against the REAL std::sort of
// this toolchain -- the call the reference makes (feature_dem.cpp:170,230: sort(..., sortbysecdesc) on vector<pair<Point2f, float>>).
// Tie-heavy inputs of every size, plus the heap-sort fallback forced through libstdc++'s own __introsort_loop with small budgets.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <utility>
#include <vector>

#include "dem_sort.hpp"

typedef std::pair<int, float> Cand;  // (candidate index, score): what sortbysecdesc sees of the reference's pairs
static bool sortbysecdesc(const Cand& a, const Cand& b) { return a.second > b.second; }

int main() {
  std::mt19937 rng(12345);
  long cases = 0, bad = 0, with_ties = 0, heap_cases = 0;
  for (int n = 0; n <= 2100; n += (n < 200 ? 1 : 37)) {
    for (int rep = 0; rep < (n < 200 ? 40 : 6); rep++) {
      const int distinct = 1 + (int)(rng() % (rep % 3 == 0 ? 4 : (rep % 3 == 1 ? 40 : 100000)));
      std::vector<float> score(n);
      for (int i = 0; i < n; i++) {
        const int q = (int)(rng() % distinct);
        score[i] = rep % 5 == 4 ? (float)(q % 7 == 0 ? 0 : q) * -0.05f : (float)q * 3.0f;  // (negative values and zeros as the quirky score produces them)
      }
      if (rep % 7 == 6) std::sort(score.begin(), score.end());            // presorted ascending: the worst case for a naive pivot
      if (rep % 7 == 5) std::sort(score.rbegin(), score.rend());
      for (int depth = -1; depth <= (n > 16 ? 3 : -1); depth++) {
        std::vector<Cand> ref(n);
        std::vector<short> idx(n);
        for (int i = 0; i < n; i++) {
          ref[i] = Cand(i, score[i]);
          idx[i] = (short)i;
        }
        if (depth < 0) {
          std::sort(ref.begin(), ref.end(), sortbysecdesc);
        } else if (n > 0) {  // std::sort's two phases with a smaller quicksort budget: ranges longer than 16 fall to the heap sort
          std::__introsort_loop(ref.begin(), ref.end(), (long)depth, __gnu_cxx::__ops::__iter_comp_iter(sortbysecdesc));
          std::__final_insertion_sort(ref.begin(), ref.end(), __gnu_cxx::__ops::__iter_comp_iter(sortbysecdesc));
          heap_cases++;
        }
        flvis::demsort::sort_desc(idx.data(), n, score.data(), depth);
        bool same = true, ties = false;
        for (int i = 0; i < n; i++) {
          same = same && ref[i].first == (int)idx[i];
          ties = ties || (i > 0 && ref[i].second == ref[i - 1].second);
        }
        // the device's form: the quicksort phase alone, then a STABLE sort of what it leaves (the final insertion sort is stable)
        {
          std::vector<short> part(n);
          for (int i = 0; i < n; i++) part[i] = (short)i;
          int stack[3 * flvis::demsort::STACK];
          flvis::demsort::quicksort_phase(flvis::demsort::IndexArray<short>{part.data(), score.data()}, n, stack, depth);
          std::stable_sort(part.begin(), part.end(), [&](short a, short b) { return score[a] > score[b]; });
          for (int i = 0; i < n; i++) same = same && ref[i].first == (int)part[i];
        }
        cases++;
        with_ties += ties;
        bad += !same;
      }
    }
  }
  printf("%ld cases (%ld with ties, %ld through the forced heap-sort budget), %ld differ\n", cases, with_ties, heap_cases, bad);
  return bad ? 1 : 0;
}
