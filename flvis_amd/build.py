"""Builds libflvis_hip.so (gfx950) in-tree with hipcc.  Called by __graft_entry__.build() and lazily by the loader."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libflvis_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-result", "-Wno-unused-value"]
FLAGS += os.environ.get("FLVIS_EXTRA_HIPCC_FLAGS", "").split()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    for src in sources():
        obj = os.path.join(CSRC, "build", os.path.basename(src) + ".o")
        objs.append(obj)
        if (not force) and os.path.exists(obj) and os.path.getmtime(obj) > max(
                [os.path.getmtime(src)] + [os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.hpp"))] +
                [os.path.getmtime(h) for h in glob.glob(os.path.join(HERE, "..", "include", "*.h"))]):
            continue
        cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"), file=sys.stderr)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz"]  # zlib: gzipped FileStorage vocabularies (voc_file.cpp)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
