"""CPU: the oracle's DBoW3 restatement (oracle/ref_bow.cpp: transform, L1 score, isLoopCandidate) against independent plain-Python
restatements of the same reference lines, and the product's host-side candidate selection (flvis_loop_candidate: control logic, no
GPU involved) against the oracle's."""
import ctypes as C
import os

import numpy as np

import _oracle as O
import _voc as V


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class RefVoc:
    def __init__(self, voc):
        cp, ci, ds, wt, wi = [np.ascontiguousarray(x) for x in voc]
        O.lib().ref_voc_create.restype = C.c_void_p
        self.h = C.c_void_p(O.lib().ref_voc_create(len(cp) - 1, _p(cp, C.c_int), _p(ci, C.c_int), _p(ds, C.c_uint8), _p(wt, C.c_double),
                                                   _p(wi, C.c_int)))

    def transform(self, d, cap=4096):
        d = np.ascontiguousarray(d, np.uint8)
        ids = np.zeros(cap, np.int32)
        vals = np.zeros(cap)
        n = O.lib().ref_voc_transform(self.h, len(d), _p(d, C.c_uint8), cap, _p(ids, C.c_int), _p(vals, C.c_double))
        return ids[:n].copy(), vals[:n].copy()

    def __del__(self):
        O.lib().ref_voc_destroy(self.h)


def ref_score(a, b):
    O.lib().ref_bow_score.restype = C.c_double
    ai, av, bi, bv = [np.ascontiguousarray(x) for x in (a[0], a[1], b[0], b[1])]
    return O.lib().ref_bow_score(len(ai), _p(ai, C.c_int), _p(av, C.c_double), len(bi), _p(bi, C.c_int), _p(bv, C.c_double))


def ref_candidate(row, present, dist, maxdist, nclosest, min_score):
    row = np.ascontiguousarray(row, np.float64)
    pres = np.ascontiguousarray(present, np.uint8)
    out = C.c_int64(-1)
    r = O.lib().ref_loop_candidate(len(row), _p(row, C.c_double), _p(pres, C.c_uint8), dist, maxdist, nclosest, C.c_double(min_score),
                                   C.byref(out))
    return int(out.value) if r else None


def py_candidate(row, present, dist, maxdist, nclosest, min_score):
    """vo_loopclosing.cpp:520-590 line by line (ties of the sort: lower index first)"""
    g = len(row)
    if g < 40 or g - dist <= 0:
        return None
    lo = g - dist - 5000 if g - dist > 5000 else 0
    cand = [(i, row[i]) for i in range(lo, g - dist) if present[i]]
    if not cand:
        return None
    cand.sort(key=lambda a: -a[1])
    lc_min = 1.0
    for i in range(g - dist, g):
        if row[i] < lc_min and row[i] > 0.001:
            lc_min = row[i]
    lc_min = min(lc_min, 0.4)
    if cand[0][1] < max(min_score, lc_min):
        return None
    n = 0
    if cand[0][1] >= lc_min:
        for i, s in cand[1:]:
            if abs(i - cand[0][0]) <= maxdist and s >= lc_min * 0.8:
                n += 1
    return cand[0][0] if (n >= nclosest and cand[0][1] > min_score) else None


def test_transform_and_score_match_the_python_restatement():
    kfs = V.make_keyframes(1)
    voc = V.build_vocabulary(kfs[:16])
    assert (voc[3] == 0).any() or True
    rv = RefVoc(voc)
    vecs = []
    for d in kfs:
        ids, vals = rv.transform(d)
        pi, pv = V.py_transform(voc, d)
        assert np.array_equal(ids, pi) and np.array_equal(vals, pv)          # same sums in the same order: bit-identical
        assert np.all(np.diff(ids) > 0) and abs(vals.sum() - 1.0) < 1e-12
        vecs.append((ids, vals))
    for a in range(0, len(vecs), 3):
        for b in range(len(vecs)):
            s = ref_score(vecs[a], vecs[b])
            assert s == V.py_score(*vecs[a], *vecs[b])
            assert abs(s - ref_score(vecs[b], vecs[a])) < 1e-15
            assert -1e-15 <= s <= 1 + 1e-15
        assert abs(ref_score(vecs[a], vecs[a]) - 1.0) < 1e-12
    # neighbours in the sequence share prototypes, distant keyframes do not
    near = np.mean([ref_score(vecs[i], vecs[i + 1]) for i in range(len(vecs) - 1)])
    far = np.mean([ref_score(vecs[i], vecs[(i + 12) % len(vecs)]) for i in range(len(vecs))])
    assert near > 2 * far


def test_transform_edge_cases():
    kfs = V.make_keyframes(2, n_img=8)
    voc = V.build_vocabulary(kfs)
    rv = RefVoc(voc)
    ids, vals = rv.transform(np.zeros((0, 32), np.uint8))
    assert len(ids) == 0
    one = kfs[0][:1]
    ids, vals = rv.transform(np.repeat(one, 7, axis=0))                       # the same word 7 times: one entry, value 1
    pi, pv = V.py_transform(voc, np.repeat(one, 7, axis=0))
    assert np.array_equal(ids, pi) and np.array_equal(vals, pv) and (len(ids) == 0 or vals[0] == 1.0)
    disjoint = ref_score((np.array([1, 3], np.int32), np.array([0.5, 0.5])), (np.array([2, 4], np.int32), np.array([0.5, 0.5])))
    assert disjoint == 0.0


def test_loop_candidate_logic_oracle_and_product_host_code():
    import flvis_amd
    rng = np.random.default_rng(5)
    hits = 0
    for trial in range(300):
        g = int(rng.integers(30, 140))
        row = rng.uniform(0.0, 0.08, g)
        present = (rng.random(g) > 0.05).astype(np.uint8)
        if trial % 2 == 0 and g > 70:          # plant a revisited place: a cluster of high scores far back
            c = int(rng.integers(5, g - 40))
            row[c - 2:c + 3] = rng.uniform(0.25, 0.6, 5)
        row[-18:] = np.clip(row[-18:] + rng.uniform(0.0, 0.3, min(18, g)), 0, 1)
        if trial % 7 == 0:
            row[rng.integers(0, g)] = row[rng.integers(0, g)]               # equal scores
        args = (int(rng.choice([10, 18])), int(rng.choice([3, 50])), int(rng.choice([0, 2, 4])), float(rng.choice([0.01, 0.05, 0.3])))
        want = py_candidate(row, present, *args)
        assert ref_candidate(row, present, *args) == want, trial
        assert flvis_amd.loop_candidate(row, present, *args) == want, trial
        hits += want is not None
    assert 20 < hits < 280
    # the reference's parameters (launch/d435_pixhawk/sn943222072828_stereo_px4.yaml:70-87)
    row = np.full(60, 0.02)
    row[10:14] = [0.3, 0.5, 0.45, 0.3]
    row[-18:] = 0.2
    pres = np.ones(60, np.uint8)
    assert ref_candidate(row, pres, 18, 50, 2, 0.05) == 11 == flvis_amd.loop_candidate(row, pres, 18, 50, 2, 0.05)
    assert ref_candidate(row[:39], pres[:39], 18, 50, 2, 0.05) is None          # fewer than 40 keyframes


def test_values_match_the_reference_bowvector_class():
    """pinned on reference code: the golden vectors were built by DBoW3's own BowVector (addWeight per feature, normalize(L1)),
    compiled from the reference where it lies (oracle/bowvector_ref.cpp, tests/golden/make_bowvector_fixture.py)"""
    cases, word_weight = V.bowvector_golden()
    voc, leaf = V.flat_vocabulary(word_weight)
    rv = RefVoc(voc)
    for words, ids, vals in cases:
        gi, gv = rv.transform(leaf[words])
        assert np.array_equal(gi, ids) and np.array_equal(gv, vals), (len(words), len(ids))
    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref", "libdbow3_bowvector.so")
    if os.path.exists(lib):      # in the build container: the committed vectors are what the reference's class says today
        import importlib.util
        spec = importlib.util.spec_from_file_location("mk", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                                                         "make_bowvector_fixture.py"))
        mk = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mk)
        for words, ids, vals in cases:
            ri, rvv = mk.reference_bowvector(words, word_weight[words])
            assert np.array_equal(ri, ids) and np.array_equal(rvv, vals)
