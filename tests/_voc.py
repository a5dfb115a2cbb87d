"""Test helper: a small DBoW3-shaped vocabulary tree (hierarchical k-majority clustering of binary descriptors, the shape
DBoW3's Vocabulary::create produces: 3rdPartLib/DBow3/src/Vocabulary.cpp HKmeansStep, bit-majority means in DescManip::meanValue),
built deterministically with numpy, and synthetic "keyframes" of 256-bit descriptors to feed it.  Not a restatement of DBoW3's
training (its k-means++ seeding draws random numbers); only the SHAPE of what it produces matters to transform / score."""
import numpy as np


def hamming(a, b):
    """a [n,32] uint8, b [m,32] uint8 -> [n,m] int"""
    x = np.bitwise_xor(a[:, None, :], b[None, :, :])
    return np.unpackbits(x, axis=2).sum(2).astype(np.int64)


def majority(d):
    bits = np.unpackbits(d, axis=1)
    return np.packbits((2 * bits.sum(0) >= len(d)).astype(np.uint8))


def make_keyframes(seed, n_img=24, n_proto=60, per_img=(300, 500), flip=12):
    """Descriptors around random prototypes; image i draws from a sliding subset of the prototypes, so that images close in index
    look alike.  Returns a list of [n_i, 32] uint8 arrays."""
    rng = np.random.default_rng(seed)
    protos = rng.integers(0, 256, (n_proto, 32), dtype=np.uint8)
    out = []
    for i in range(n_img):
        lo = (i * 2) % (n_proto - 20)
        n = int(rng.integers(per_img[0], per_img[1]))
        which = rng.integers(lo, lo + 20, n)
        d = protos[which].copy()
        bits = np.unpackbits(d, axis=1)
        for r in range(n):
            bits[r, rng.choice(256, int(rng.integers(0, flip)), replace=False)] ^= 1
        out.append(np.packbits(bits, axis=1))
    return out


def build_vocabulary(train, k=6, depth=3, iters=3):
    """train: list of [n,32] uint8 (one per training image).  Returns flat arrays (child_ptr, child_idx, desc, weight, word_id)."""
    allD = np.concatenate(train)
    img_of = np.concatenate([np.full(len(d), i) for i, d in enumerate(train)])
    nodes = [dict(desc=np.zeros(32, np.uint8), children=[], members=np.arange(len(allD)))]

    def split(nid, level):
        mem = nodes[nid]["members"]
        if level == depth or len(mem) <= k:
            return
        # seeds: the first k distinct descriptors of the node
        _, first = np.unique(allD[mem], axis=0, return_index=True)
        seeds = allD[mem][np.sort(first)[:k]]
        if len(seeds) < 2:
            return
        centers = seeds.copy()
        for _ in range(iters):
            lab = hamming(allD[mem], centers).argmin(1)
            for c in range(len(centers)):
                if np.any(lab == c):
                    centers[c] = majority(allD[mem][lab == c])
        lab = hamming(allD[mem], centers).argmin(1)
        for c in range(len(centers)):
            if not np.any(lab == c):
                continue
            nodes.append(dict(desc=centers[c].copy(), children=[], members=mem[lab == c]))
            cid = len(nodes) - 1
            nodes[nid]["children"].append(cid)
            split(cid, level + 1)

    split(0, 0)
    n = len(nodes)
    child_ptr = np.zeros(n + 1, np.int32)
    child_idx = []
    for i, nd in enumerate(nodes):
        child_idx += nd["children"]
        child_ptr[i + 1] = len(child_idx)
    desc = np.stack([nd["desc"] for nd in nodes])
    word_id = np.full(n, -1, np.int32)
    weight = np.zeros(n)
    w = 0
    for i, nd in enumerate(nodes):
        if not nd["children"]:
            word_id[i] = w
            w += 1
            ni = len(np.unique(img_of[nd["members"]]))
            weight[i] = np.log(len(train) / ni)          # idf (Vocabulary.cpp setNodeWeights): 0 for a word seen in every image
    return child_ptr, np.array(child_idx, np.int32), desc, weight, word_id


def py_transform(voc, d):
    """independent restatement of Vocabulary::transform (TF_IDF, L1) in plain Python: returns (ids, vals)"""
    child_ptr, child_idx, desc, weight, word_id = voc
    bow = {}
    for f in d:
        node = 0
        while child_ptr[node + 1] > child_ptr[node]:
            ch = child_idx[child_ptr[node]:child_ptr[node + 1]]
            dist = hamming(f[None], desc[ch])[0]
            node = int(ch[int(np.argmin(dist))])          # argmin: first minimum
        if weight[node] > 0:
            bow[int(word_id[node])] = bow.get(int(word_id[node]), 0.0) + float(weight[node])
    ids = sorted(bow)
    norm = 0.0
    for i in ids:
        norm += abs(bow[i])
    vals = [bow[i] / norm for i in ids] if norm > 0 else [bow[i] for i in ids]
    return np.array(ids, np.int32), np.array(vals)


def py_score(a_ids, a_vals, b_ids, b_vals):
    b = dict(zip(b_ids.tolist(), b_vals.tolist()))
    s = 0.0
    for i, v in zip(a_ids.tolist(), a_vals.tolist()):
        if i in b:
            s += abs(v - b[i]) - abs(v) - abs(b[i])
    return -s / 2.0


def flat_vocabulary(word_weight, seed=5):
    """a one-level tree: every word a child of the root with its own random descriptor, so that a SEQUENCE OF WORDS can be fed through
    transform() as the sequence of those descriptors (Hamming distance 0 to its own leaf).  Returns (voc arrays, leaf descriptors)."""
    n = len(word_weight)
    rng = np.random.default_rng(seed)
    leaf = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    assert len(np.unique(leaf, axis=0)) == n
    child_ptr = np.zeros(n + 2, np.int32)
    child_ptr[1:] = n
    child_idx = np.arange(1, n + 1, dtype=np.int32)
    desc = np.concatenate([np.zeros((1, 32), np.uint8), leaf])
    weight = np.concatenate([[0.0], np.asarray(word_weight, np.float64)])
    word_id = np.concatenate([[-1], np.arange(n)]).astype(np.int32)
    return (child_ptr, child_idx, desc, weight, word_id), leaf


def bowvector_golden():
    """tests/golden/bowvector_ref.npz (written by the reference's own BowVector class): [(words, ids, vals)], word weights"""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bowvector_ref.npz"))
    return [(z["words_%d" % c], z["ids_%d" % c], z["vals_%d" % c]) for c in range(int(z["n_cases"]))], z["word_weight"]
