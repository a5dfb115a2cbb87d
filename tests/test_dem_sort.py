"""CPU check of csrc/dem_sort.hpp -- the order the device leaves tied FeatureDEM candidates in -- against the real std::sort of this
toolchain (libstdc++: what the reference's GCC build calls at feature_dem.cpp:170,230).  tests/cpp/dem_sort_check.cpp sorts tie-heavy
score arrays of every size 0 .. 2100 both ways and also forces the heap-sort fallback through libstdc++'s own __introsort_loop."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_restated_introsort_leaves_ties_where_std_sort_does():
    out = os.path.join(tempfile.gettempdir(), "flvis_dem_sort_check")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "flvis_amd", "csrc"),
           os.path.join(ROOT, "tests", "cpp", "dem_sort_check.cpp"), "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()
    r = subprocess.run([out], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and b" 0 differ" in r.stdout, (r.stdout.decode(), r.stderr.decode())
    n_ties = int(r.stdout.split(b"(")[1].split()[0])
    assert n_ties > 10000
