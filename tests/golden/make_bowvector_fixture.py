"""Writes tests/golden/bowvector_ref.npz: word sequences with their leaf weights, and the bag-of-words vector the REFERENCE's own
BowVector class builds from them -- addWeight once per feature whose weight is positive, then normalize(L1) (what
Vocabulary::transform does for TF_IDF / L1_NORM, 3rdPartLib/DBow3/src/Vocabulary.cpp:648-685) -- through
oracle/_ref/libdbow3_bowvector.so (oracle/bowvector_ref.cpp + the reference's BowVector.cpp compiled where it lies).
Run in the build container (needs /root/reference); the file is data.

    python tests/golden/make_bowvector_fixture.py
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "..", "oracle", "_ref", "libdbow3_bowvector.so")


def reference_bowvector(words, weights, norm=1, add_if_not_exist=False):
    lib = C.CDLL(LIB)
    w = np.ascontiguousarray(words, np.uint32)
    v = np.ascontiguousarray(weights, np.float64)
    ids = np.zeros(len(w) + 1, np.uint32)
    vals = np.zeros(len(w) + 1, np.float64)
    k = lib.ref_dbow3_bowvector(len(w), w.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), norm, int(add_if_not_exist), len(ids),
                                ids.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p))
    return ids[:k].astype(np.int32), vals[:k].copy()


if __name__ == "__main__":
    assert os.path.exists(LIB), "make -C oracle (needs /root/reference)"
    rng = np.random.default_rng(77)
    n_words = 300
    word_weight = rng.uniform(0.01, 9.0, n_words)
    word_weight[rng.choice(n_words, 25, replace=False)] = 0.0          # stopped words (idf 0: seen in every training image)
    out = {"word_weight": word_weight}
    cases = [rng.integers(0, n_words, n) for n in (1, 2, 17, 300, 1000, 2048)]
    cases.append(np.full(500, 7))                                       # one word, many occurrences
    cases.append(rng.integers(0, 12, 900))                              # few words, long runs of repeats
    cases.append(np.array([w for w in range(n_words) if word_weight[w] == 0.0]))   # only stopped words: an empty vector
    cases.append(np.zeros(0, np.int64))
    for c, words in enumerate(cases):
        ids, vals = reference_bowvector(words, word_weight[words])
        out["words_%d" % c], out["ids_%d" % c], out["vals_%d" % c] = words.astype(np.int32), ids, vals
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "bowvector_ref.npz"), **out)
    print("cases:", [len(c) for c in cases], "->", [len(out["ids_%d" % c]) for c in range(len(cases))])
