"""The C ABI from a C++ caller (tests/cpp/caller.cpp): what the reference's nodelet would do -- host images and IMU samples in,
flvis_frame_out / flvis_keyframe out -- with no Python, torch or HIP headers on the caller's side.  Built here with g++ against
include/flvis_hip.h and the in-tree libflvis_hip.so.  Without a GPU the program must report FLVIS_ERR_NO_DEVICE (exit 3): the
library has no CPU fallback."""
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _build():
    import flvis_amd
    flvis_amd.load_library()
    exe = os.path.join(tempfile.gettempdir(), "flvis_cpp_caller")
    libdir = os.path.join(ROOT, "flvis_amd")
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "caller.cpp"),
           "-o", exe, "-L", libdir, "-lflvis_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()
    return exe


def _yaml():
    from flvis_amd import synth
    p = os.path.join(tempfile.gettempdir(), "flvis_cpp_caller.yaml")
    open(p, "w").write(synth.D435I_STEREO_YAML)
    return p


def _env():
    import torch
    env = dict(os.environ)
    # the caller links the system HIP runtime; make sure the loader finds one (torch bundles its own copy)
    env["LD_LIBRARY_PATH"] = ":".join(["/opt/rocm/lib", os.path.join(os.path.dirname(torch.__file__), "lib"), env.get("LD_LIBRARY_PATH", "")])
    return env


def test_cpp_caller_builds_and_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU (the GPU run is the test below)")
    exe = _build()
    r = subprocess.run([exe, _yaml()], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=_env(), timeout=120)
    assert r.returncode == 3 and b"no device" in r.stdout, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_cpp_caller_runs_the_path():
    exe = _build()
    r = subprocess.run([exe, _yaml(), "8"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=_env(), timeout=300)
    assert r.returncode == 0 and b"caller OK" in r.stdout, (r.returncode, r.stdout.decode(), r.stderr.decode())
