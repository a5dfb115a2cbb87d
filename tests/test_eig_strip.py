"""CPU check of the strip-mined corner-response phases (flvis_amd/csrc/eig_strip.hpp, the arithmetic of the opt-in kernel
k_eig_cand_strip): the header is plain C++, so the very functions the kernel calls are compiled for the host, run tile by tile over
whole images as the kernel's workgroups would run them, and compared bit for bit with the oracle's cornerMinEigenVal map (which the
default kernel is bit-exact against on the GPU)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import _oracle as O
import _synth as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe():
    out = os.path.join(tempfile.gettempdir(), "flvis_eig_strip_check")
    cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-I", os.path.join(ROOT, "flvis_amd", "csrc"),
           os.path.join(ROOT, "tests", "cpp", "eig_strip_check.cpp"), "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()
    return out


@pytest.mark.parametrize("h,w,seed,kind", [(480, 640, 1, "corner"), (376, 1241, 2, "corner"), (97, 131, 3, "texture"), (16, 64, 4, "texture"),
                                           (33, 67, 5, "corner"), (200, 260, 6, "noise")])
def test_strip_phases_reproduce_the_response_map_bit_for_bit(exe, h, w, seed, kind):
    if kind == "corner":
        img = S.corner_img(h, w, seed)
    elif kind == "texture":
        img = S.texture_u8(h, w, seed)
    else:
        img = np.random.default_rng(seed).integers(0, 256, (h, w), dtype=np.uint8)
    ref = O.min_eigen_map(img)
    d = tempfile.mkdtemp(prefix="flvis_eig_")
    img.tofile(os.path.join(d, "img.u8"))
    ref.astype(np.float32).tofile(os.path.join(d, "ref.f32"))
    r = subprocess.run([exe, str(w), str(h), os.path.join(d, "img.u8"), os.path.join(d, "ref.f32")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, (r.stdout.decode(), r.stderr.decode())
    assert b" 0 differ" in r.stdout and b"candidate keys" in r.stdout
    n_keys = int(r.stdout.split(b"responses and ")[1].split()[0])
    assert n_keys > 0 or kind == "texture" and h <= 16


def test_strip_phases_stay_inside_their_arrays():
    """the same harness under AddressSanitizer / UBSan with host arrays of exactly the kernel's LDS array sizes: no strip reads or
    writes outside the tile, the (fx, fy) maps or the response map"""
    out = os.path.join(tempfile.gettempdir(), "flvis_eig_strip_check_asan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off",
           "-I", os.path.join(ROOT, "flvis_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "eig_strip_check.cpp"), "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        pytest.skip("no sanitizer runtime for this g++: " + r.stdout.decode()[-200:])
    for h, w, seed in ((97, 131, 3), (33, 67, 5), (17, 65, 9), (480, 640, 1)):
        img = S.corner_img(h, w, seed)
        d = tempfile.mkdtemp(prefix="flvis_eig_")
        img.tofile(os.path.join(d, "img.u8"))
        O.min_eigen_map(img).astype(np.float32).tofile(os.path.join(d, "ref.f32"))
        r = subprocess.run([out, str(w), str(h), os.path.join(d, "img.u8"), os.path.join(d, "ref.f32")], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert r.returncode == 0 and b" 0 differ" in r.stdout, (r.stdout.decode(), r.stderr.decode()[-1500:])
