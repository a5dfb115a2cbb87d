"""Writes tests/golden/voc_k6_quicklz.dbow3 + voc_k6.npz: a small ORB vocabulary in DBoW3's compressed binary layout
(Vocabulary::toStream, 3rdPartLib/DBow3/src/Vocabulary.cpp:1180-1256) whose QuickLZ blocks come from the REFERENCE's own
compressor -- oracle/_ref/libquicklz.so, built by `make -C oracle` from /root/reference/3rdPartLib/DBow3/src/quicklz.c where it
lies -- and the tree it holds as flat arrays.  Run in the build container (needs /root/reference); the two files are data.

    python tests/golden/make_voc_fixture.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import _voc  # noqa: E402
import _vocfile as VF  # noqa: E402

if __name__ == "__main__":
    assert VF.ref_quicklz() is not None, "oracle/_ref/libquicklz.so is missing: make -C oracle (needs /root/reference)"
    kfs = _voc.make_keyframes(seed=2024, n_img=12, n_proto=50, per_img=(150, 250))
    voc = _voc.build_vocabulary(kfs, k=6, depth=3)
    path = os.path.join(HERE, "voc_k6_quicklz.dbow3")
    VF.write_binary(path, voc, 6, 3, compress=VF.ref_compress)
    np.savez_compressed(os.path.join(HERE, "voc_k6.npz"), child_ptr=voc[0], child_idx=voc[1], desc=voc[2], weight=voc[3], word_id=voc[4])
    raw = len(VF.payload(voc, 6, 3))
    print("nodes %d, words %d, payload %d bytes -> file %d bytes (%d blocks)" % (len(voc[0]) - 1, int((voc[4] >= 0).sum()), raw,
                                                                                os.path.getsize(path), (raw + 9999) // 10000))
