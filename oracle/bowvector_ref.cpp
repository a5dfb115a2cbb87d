// TEST INFRASTRUCTURE: driver over the REFERENCE's own BowVector (3rdPartLib/DBow3/src/BowVector.cpp, compiled where it lies by
// oracle/Makefile into _ref/libdbow3_bowvector.so; it needs nothing but the standard library).  It pins the value half of the
// bag-of-words transform -- addWeight / addIfNotExist per feature, then normalize -- of oracle/ref_bow.cpp and of the product
// (tests/golden/make_bowvector_fixture.py writes the golden vectors from it).
#include "BowVector.h"

extern "C" int ref_dbow3_bowvector(int n, const unsigned* word, const double* weight, int norm, int add_if_not_exist, int cap, unsigned* ids,
                                   double* vals) {
  DBoW3::BowVector v;
  for (int i = 0; i < n; i++) {
    if (!(weight[i] > 0)) continue;  // "not stopped" (Vocabulary.cpp:657)
    if (add_if_not_exist)
      v.addIfNotExist(word[i], weight[i]);
    else
      v.addWeight(word[i], weight[i]);
  }
  if (norm == 1) v.normalize(DBoW3::L1);
  if (norm == 2) v.normalize(DBoW3::L2);
  int k = 0;
  for (DBoW3::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++k)
    if (k < cap) {
      ids[k] = it->first;
      vals[k] = it->second;
    }
  return k;
}
