"""The DBoW3 vocabulary-file reader behind `Vocabulary voc(path)` (vo_loopclosing.cpp:1097; flvis_voc_file_open, host only):
every on-disk layout Vocabulary::load accepts must give back the tree that was written, child order included; the compressed
layout is pinned on a file whose QuickLZ blocks the reference's own compressor wrote (tests/golden/voc_k6_quicklz.dbow3)."""
import os
import struct

import numpy as np
import pytest

import _voc
import _vocfile as VF
import flvis_amd

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def voc():
    kfs = _voc.make_keyframes(seed=5, n_img=10, n_proto=40, per_img=(120, 200))
    return _voc.build_vocabulary(kfs, k=5, depth=3)


def same_tree(got, voc, weight32=False):
    child_ptr, child_idx, desc, weight, word_id = voc
    assert np.array_equal(got["child_ptr"], child_ptr)
    assert np.array_equal(got["child_idx"], child_idx)
    assert np.array_equal(got["desc"][1:], desc[1:])
    w = weight.astype(np.float32).astype(np.float64) if weight32 else weight
    assert np.array_equal(got["weight"][1:], w[1:])
    leaf = np.diff(child_ptr) == 0
    assert np.array_equal(got["word_id"][leaf], word_id[leaf])
    assert np.all(got["word_id"][~leaf] == -1)
    assert got["n_words"] == int(word_id[leaf].max()) + 1


def test_golden_file_compressed_by_the_reference(tmp_path):
    want = np.load(os.path.join(GOLD, "voc_k6.npz"))
    got = flvis_amd.read_vocabulary_file(os.path.join(GOLD, "voc_k6_quicklz.dbow3"))
    assert got["layout"] == "binary-quicklz" and (got["k"], got["L"], got["scoring"], got["weighting"]) == (6, 3, 0, 0)
    same_tree(got, (want["child_ptr"], want["child_idx"], want["desc"], want["weight"], want["word_id"]))


def test_binary_plain(voc, tmp_path):
    p = str(tmp_path / "v.dbow3")
    VF.write_binary(p, voc, 5, 3)
    got = flvis_amd.read_vocabulary_file(p)
    assert got["layout"] == "binary" and (got["k"], got["L"]) == (5, 3)
    same_tree(got, voc)


def test_binary_quicklz_blocks_of_an_independent_encoder(voc, tmp_path):
    p = str(tmp_path / "v.dbow3")
    VF.write_binary(p, voc, 5, 3, compress=VF.qlz1_compress, chunk=3000)   # several blocks, a short last one
    got = flvis_amd.read_vocabulary_file(p)
    assert got["layout"] == "binary-quicklz"
    same_tree(got, voc)
    assert os.path.getsize(p) < len(VF.payload(voc, 5, 3))                 # the matches were really taken


def test_stored_blocks_and_short_headers(voc, tmp_path):
    # QuickLZ stores a block it cannot shrink (flag bit 0 clear) and uses one-byte sizes below 216 bytes
    calls = [0]

    def mixed(chunk):
        calls[0] += 1
        if calls[0] % 2:
            hs = 9 if len(chunk) >= 216 else 3
            head = bytes([0x46]) + struct.pack("<II", len(chunk) + hs, len(chunk)) if hs == 9 else bytes([0x44, len(chunk) + 3, len(chunk)])
            return head + chunk
        return VF.qlz1_compress(chunk)

    for chunk in (150, 1000):
        p = str(tmp_path / "v.dbow3")
        VF.write_binary(p, voc, 5, 3, compress=mixed, chunk=chunk)
        same_tree(flvis_amd.read_vocabulary_file(p), voc)
    if VF.ref_quicklz() is not None:
        p = str(tmp_path / "r.dbow3")
        VF.write_binary(p, voc, 5, 3, compress=VF.ref_compress, chunk=150)     # the reference's own short blocks
        same_tree(flvis_amd.read_vocabulary_file(p), voc)


def test_tf_weighting_is_accepted(voc, tmp_path):
    p = str(tmp_path / "v.dbow3")
    VF.write_binary(p, voc, 5, 3, weighting=1)
    assert flvis_amd.read_vocabulary_file(p)["weighting"] == 1


@pytest.mark.skipif(VF.ref_quicklz() is None, reason="oracle/_ref/libquicklz.so needs /root/reference to build")
def test_quicklz_against_the_reference_both_ways(tmp_path):
    rng = np.random.default_rng(3)
    for trial in range(12):
        n = int(rng.integers(1, 12000))
        kind = trial % 4
        if kind == 0:
            data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()                        # incompressible -> stored block
        elif kind == 1:
            data = rng.integers(0, 4, n, dtype=np.uint8).tobytes()                          # many short matches
        elif kind == 2:
            data = (bytes(rng.integers(0, 256, 37, dtype=np.uint8)) * (n // 37 + 1))[:n]    # long overlapping matches
        else:
            data = bytes(n)                                                                   # one run
        # (a) the encoder of the tests writes what the reference's decoder reads back
        assert VF.ref_decompress(VF.qlz1_compress(data)) == data, (trial, n)
    # (b) through the file reader: vocabularies of several shapes compressed by the reference
    for seed, k, depth in ((1, 3, 2), (2, 8, 2), (3, 4, 4)):
        kfs = _voc.make_keyframes(seed=seed, n_img=8, n_proto=40, per_img=(100, 160))
        v = _voc.build_vocabulary(kfs, k=k, depth=depth)
        p = str(tmp_path / ("v%d.dbow3" % seed))
        VF.write_binary(p, v, k, depth, compress=VF.ref_compress)
        same_tree(flvis_amd.read_vocabulary_file(p), v)


def test_text_layout(voc, tmp_path):
    p = str(tmp_path / "ORBvoc.txt")
    VF.write_txt(p, voc, 5, 3)
    got = flvis_amd.read_vocabulary_file(p)
    assert got["layout"] == "text" and (got["k"], got["L"]) == (5, 3)
    same_tree(got, voc, weight32=True)      # load_fromtxt reads every number of a line as float


@pytest.mark.parametrize("gz,tagged,flow", [(False, True, False), (True, True, False), (False, False, False), (False, True, True)])
def test_yaml_layout(voc, tmp_path, gz, tagged, flow):
    p = str(tmp_path / ("v.yml.gz" if gz else "v.yml"))
    VF.write_yaml(p, voc, 5, 3, gz=gz, tagged=tagged, flow=flow)
    got = flvis_amd.read_vocabulary_file(p)
    assert got["layout"] == "yaml" and (got["k"], got["L"]) == (5, 3)
    same_tree(got, voc)


def test_child_order_is_the_file_order(tmp_path):
    # two children of the root at the same Hamming distance from a query: the one listed first must stay first
    child_ptr = np.array([0, 2, 2, 2], np.int32)
    desc = np.zeros((3, 32), np.uint8)
    desc[1, 0], desc[2, 0] = 1, 2
    for order in ([1, 2], [2, 1]):
        v = (child_ptr, np.array(order, np.int32), desc, np.array([0, 1.5, 2.5]), np.array([-1, 0, 1], np.int32))
        p = str(tmp_path / "o.dbow3")
        VF.write_binary(p, v, 2, 1)
        assert flvis_amd.read_vocabulary_file(p)["child_idx"].tolist() == order


def test_refusals(voc, tmp_path):
    def refused(path, needle):
        with pytest.raises(flvis_amd.FlvisError) as e:
            flvis_amd.read_vocabulary_file(path)
        assert needle in str(e.value), str(e.value)

    refused(str(tmp_path / "absent.dbow3"), "cannot open")
    p = str(tmp_path / "v.dbow3")
    VF.write_binary(p, voc, 5, 3, scoring=1)
    refused(p, "scoring type 1")
    VF.write_binary(p, voc, 5, 3, weighting=2)
    refused(p, "weighting type 2")
    VF.write_binary(p, voc, 5, 3, desc_cols=64)
    refused(p, "1x32 CV_8U")
    VF.write_binary(p, voc, 5, 3, desc_type=5)
    refused(p, "1x32 CV_8U")
    VF.write_binary(p, voc, 5, 3)
    raw = open(p, "rb").read()
    open(p, "wb").write(raw[:len(raw) // 2])
    refused(p, "node count exceeds")                               # half the node records are missing
    open(p, "wb").write(raw[:-5])
    refused(p, "unexpected end")                                   # the word table is cut short
    bad = bytearray(raw)
    struct.pack_into("<I", bad, 13 + 16, 10 ** 6)                  # first node id far out of range
    open(p, "wb").write(bytes(bad))
    refused(p, "out of range")
    bad = bytearray(raw)
    struct.pack_into("<I", bad, 13 + 16 + 4, struct.unpack_from("<I", raw, 13 + 16)[0])   # a node that is its own parent
    open(p, "wb").write(bytes(bad))
    refused(p, "out of range")
    bad = bytearray(raw)
    struct.pack_into("<I", bad, 9, 0xFFFFFFF0)                     # a node count no file of this size can hold: refused, not allocated
    open(p, "wb").write(bytes(bad))
    refused(p, "node count exceeds")
    open(p, "wb").write(b"neither binary nor yaml\n")
    refused(p, "neither a DBoW3 binary")
    t = str(tmp_path / "v.txt")
    open(t, "w").write("5 3 0 0\n0 1 1 2 3 0.5\n")
    refused(t, "a node line must hold")
    open(t, "w").write("99 3 0 0\n")
    refused(t, "not a vocabulary header")
    open(t, "w").write("1e300 3 0 0\n")                            # values whose casts to int would be undefined behaviour
    refused(t, "not a vocabulary header")
    open(t, "w").write("nan 3 0 0\n")
    refused(t, "not a vocabulary header")
    line = "0 1 " + " ".join(["7"] * 31)
    open(t, "w").write("5 1 0 0\n" + line + " 300 0.5\n")          # a descriptor byte that is no byte
    refused(t, "descriptor byte outside")
    open(t, "w").write("5 1 0 0\n" + line + " -1 0.5\n")
    refused(t, "descriptor byte outside")
    open(t, "w").write("5 1 0 0\n" + line + " 7 inf\n")
    refused(t, "weight that is not a number")
    y = str(tmp_path / "v.yml")
    VF.write_yaml(y, voc, 5, 3)
    txt = open(y).read()
    open(y, "w").write(txt.replace("nodeId:1,", "nodeId:99999999999,", 1))        # an id beyond 32 bits: refused, not wrapped
    refused(y, "bad nodeId")
    open(y, "w").write(txt.replace(" 32 ", " 32 999 ", 1).replace("dbw3 0 32 999 ", "dbw3 0 32 999 ", 1))
    with pytest.raises(flvis_amd.FlvisError):                      # 33 numbers or a byte of 999: either way refused
        flvis_amd.read_vocabulary_file(y)


def test_corrupt_compressed_streams_are_refused_not_crashed(voc, tmp_path):
    p = str(tmp_path / "v.dbow3")
    VF.write_binary(p, voc, 5, 3, compress=VF.qlz1_compress)
    raw = open(p, "rb").read()
    rng = np.random.default_rng(11)
    outcomes = set()
    for trial in range(200):
        bad = bytearray(raw)
        for _ in range(int(rng.integers(1, 4))):
            bad[int(rng.integers(17, len(bad)))] = int(rng.integers(0, 256))
        if trial % 10 == 0:
            bad = bad[:int(rng.integers(17, len(bad)))]
        open(p, "wb").write(bytes(bad))
        try:
            flvis_amd.read_vocabulary_file(p)          # a flipped literal can still be a well-formed file
            outcomes.add("ok")
        except flvis_amd.FlvisError:
            outcomes.add("refused")
    assert "refused" in outcomes


def test_implausible_block_size_is_refused_before_allocating(voc, tmp_path):
    p = str(tmp_path / "v.dbow3")
    VF.write_binary(p, voc, 5, 3, compress=VF.qlz1_compress)
    raw = bytearray(open(p, "rb").read())
    assert raw[17] & 2                                             # first block: long header (flags, u32 compressed, u32 decompressed)
    struct.pack_into("<I", raw, 17 + 5, 0xFFFFFF00)
    open(p, "wb").write(bytes(raw))
    with pytest.raises(flvis_amd.FlvisError) as e:
        flvis_amd.read_vocabulary_file(p)
    assert "implausible" in str(e.value)


def test_symbols_exported():
    lib = flvis_amd.load_library()
    for name in ("flvis_voc_file_open", "flvis_voc_file_info", "flvis_voc_file_arrays", "flvis_voc_file_close",
                 "flvis_hip_bow_load_vocabulary"):
        assert hasattr(lib, name), name
