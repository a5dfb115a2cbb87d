"""Test helper: writers of the three on-disk layouts DBoW3's Vocabulary::load accepts (3rdPartLib/DBow3/src/Vocabulary.cpp:
toStream :1180-1256, load_fromtxt :1259-1332, save(FileStorage) :1112-1178), fed with the flat tree arrays of tests/_voc.py, so
that the product's reader (flvis_voc_file_open) can be checked on files whose content is known.

Compression: the reference's own QuickLZ (oracle/_ref/libquicklz.so, built by oracle/Makefile from the reference's quicklz.c
where it lies) when it is there; `qlz1_compress` below is an independent encoder of the same block format for machines without
/root/reference (it mirrors the DECODER's table schedule, so what it emits is a valid level-1 stream, though not byte-identical
to the reference encoder's choice of matches)."""
import ctypes as C
import gzip
import os
import struct

import numpy as np

MAGIC = 88877711233
REFLIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref", "libquicklz.so")


def _children(voc, n):
    child_ptr, child_idx = voc[0], voc[1]
    return [int(c) for c in child_idx[child_ptr[n]:child_ptr[n + 1]]]


def save_order(voc):
    """(node, parent) in the order toStream / save write the nodes: a stack of parents, last in first out."""
    out, parents = [], [0]
    while parents:
        pid = parents.pop()
        for c in _children(voc, pid):
            out.append((c, pid))
            if _children(voc, c):
                parents.append(c)
    return out


def words_of(voc):
    """[(word id, node id)] ascending word id"""
    word_id = voc[4]
    w = [(int(word_id[n]), n) for n in range(len(word_id)) if not _children(voc, n) and n > 0]
    return sorted(w)


def payload(voc, k, L, scoring=0, weighting=0, desc_cols=32, desc_type=0):
    """the uncompressed stream behind the 13-byte file header"""
    _, _, desc, weight, _ = voc
    b = bytearray(struct.pack("<iiii", k, L, scoring, weighting))
    for n, pid in save_order(voc):
        b += struct.pack("<IId", n, pid, float(weight[n]))
        b += struct.pack("<iii", desc_cols, 1, desc_type)
        row = bytes(desc[n].tobytes())
        b += (row * ((desc_cols + 31) // 32))[:desc_cols]
    w = words_of(voc)
    b += struct.pack("<I", len(w))
    for wid, nid in w:
        b += struct.pack("<II", wid, nid)
    return bytes(b)


# ---- QuickLZ level 1 -------------------------------------------------------------------------------------------------------------
def _hash3(data, p):
    v = data[p] | (data[p + 1] << 8) | (data[p + 2] << 16)
    return ((v >> 12) ^ v) & 4095


def qlz1_compress(data):
    """one block; see the format notes in flvis_amd/csrc/voc_file.cpp"""
    n = len(data)
    long_header = n >= 216
    body = bytearray()
    table = {}
    state = {"hashed": -1, "cpos": None, "nbits": 31, "word": 0}

    def hash_upto(upto):
        while state["hashed"] < upto:
            state["hashed"] += 1
            table[_hash3(data, state["hashed"])] = state["hashed"]

    def flag(bit):
        if state["nbits"] == 31:
            if state["cpos"] is not None:
                struct.pack_into("<I", body, state["cpos"], state["word"] | (1 << 31))
            state["cpos"] = len(body)
            body.extend(b"\0\0\0\0")
            state["nbits"], state["word"] = 0, 0
        state["word"] |= bit << state["nbits"]
        state["nbits"] += 1

    d = 0
    last_matchstart = n - 1 - 6 - 4
    while d < n:
        took = False
        if d < last_matchstart:
            h = _hash3(data, d)
            src = table.get(h, -1)
            if 0 <= src <= d - 3 and data[src:src + 3] == data[d:d + 3]:
                ln = 3
                while ln < 255 and d + ln < n - 4 and data[src + ln] == data[d + ln]:
                    ln += 1
                flag(1)
                if ln < 18:
                    body.extend(struct.pack("<H", (h << 4) | (ln - 2)))
                else:
                    body.extend(struct.pack("<H", h << 4) + bytes([ln]))
                d += ln
                hash_upto(d - ln)
                state["hashed"] = d - 1
                took = True
        if not took:
            flag(0)
            body.append(data[d])
            d += 1
            hash_upto(d - 3)
    if state["cpos"] is not None:
        struct.pack_into("<I", body, state["cpos"], state["word"] | (1 << 31))
    hs = 9 if long_header else 3
    flags = 0x45 | (2 if long_header else 0)
    if long_header:
        return bytes([flags]) + struct.pack("<II", len(body) + hs, n) + bytes(body)
    return bytes([flags, len(body) + hs, n]) + bytes(body)


_ref = None


def ref_quicklz():
    """the reference's QuickLZ through ctypes, or None"""
    global _ref
    if _ref is None and os.path.exists(REFLIB):
        _ref = C.CDLL(REFLIB)
        _ref.qlz_compress.restype = C.c_size_t
        _ref.qlz_compress.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p]
        _ref.qlz_decompress.restype = C.c_size_t
        _ref.qlz_decompress.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
        _ref.qlz_size_decompressed.restype = C.c_size_t
        _ref.qlz_size_decompressed.argtypes = [C.c_char_p]
    return _ref


def ref_compress(data):
    q = ref_quicklz()
    state = C.create_string_buffer(q.qlz_get_setting(1))     # zeroed, as toStream's memset leaves it
    out = C.create_string_buffer(len(data) + 400)
    n = q.qlz_compress(bytes(data), out, len(data), state)
    return out.raw[:n]


def ref_decompress(block):
    q = ref_quicklz()
    state = C.create_string_buffer(q.qlz_get_setting(2))
    n = q.qlz_size_decompressed(block)
    out = C.create_string_buffer(max(n, 1) + 16)
    got = q.qlz_decompress(bytes(block), out, state)
    return out.raw[:got]


def write_binary(path, voc, k, L, compress=None, chunk=10000, **kw):
    """compress: None (plain) or a function bytes -> one QuickLZ block, applied per 10000-byte chunk as toStream does"""
    body = payload(voc, k, L, **kw)
    n_nodes = len(voc[0]) - 1
    with open(path, "wb") as f:
        f.write(struct.pack("<Q?I", MAGIC, compress is not None, n_nodes))
        if compress is None:
            f.write(body)
        else:
            chunks = [body[i:i + chunk] for i in range(0, len(body), chunk)]
            f.write(struct.pack("<I", len(chunks)))
            for c in chunks:
                f.write(compress(c))


def write_txt(path, voc, k, L, scoring=0, weighting=0):
    """ORB-SLAM2 layout: ids are implied by the line number, so the nodes go out in id order (tests/_voc.py numbers parents first)"""
    child_ptr, child_idx, desc, weight, _ = voc
    n = len(child_ptr) - 1
    parent = np.zeros(n, np.int64)
    for p in range(n):
        for c in _children(voc, p):
            parent[c] = p
    with open(path, "w") as f:
        f.write("%d %d %d %d\n" % (k, L, scoring, weighting))
        for i in range(1, n):
            leaf = 0 if _children(voc, i) else 1
            f.write("%d %d %s %r\n" % (parent[i], leaf, " ".join(str(int(b)) for b in desc[i]), float(np.float32(weight[i]))))


def write_yaml(path, voc, k, L, scoring=0, weighting=0, gz=False, tagged=True, flow=False):
    """what cv::FileStorage writes for Vocabulary::save(fs): a block sequence of flow mappings, wrapped before `descriptor`"""
    _, _, desc, weight, _ = voc
    lines = ["%YAML:1.0", "---", "vocabulary:", "   k: %d" % k, "   L: %d" % L, "   scoringType: %d" % scoring,
             "   weightingType: %d" % weighting, "   nodes:" + (" [" if flow else "")]
    items = []
    for n, pid in save_order(voc):
        ds = ("dbw3 0 32 " if tagged else "") + " ".join(str(int(b)) for b in desc[n]) + " "
        w = repr(float(weight[n]))
        if w.endswith(".0"):
            w = w[:-1]                                            # FileStorage prints 0. / 1.
        items.append('{ nodeId:%d, parentId:%d, weight:%s,\n          descriptor:"%s" }' % (n, pid, w, ds))
    if flow:
        lines.append(",\n".join("      " + it for it in items) + " ]")
    else:
        lines += ["      - " + it for it in items]
    lines.append("   words:" + (" [" if flow else ""))
    witems = ["{ wordId:%d, nodeId:%d }" % (wid, nid) for wid, nid in words_of(voc)]
    if flow:
        lines.append(",\n".join("      " + it for it in witems) + " ]")
    else:
        lines += ["      - " + it for it in witems]
    text = "\n".join(lines) + "\n"
    if gz:
        with gzip.open(path, "wb") as f:
            f.write(text.encode())
    else:
        with open(path, "w") as f:
            f.write(text)
