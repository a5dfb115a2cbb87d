#!/usr/bin/env python3
"""Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on this box with known-byte probes (scripts/pmc_probe.hip, prebuilt as
build_variants/pmc_probe): two separate --pmc passes (the two counters do not fit one pass on gfx950), per probe kernel the mean counter
value, the bytes the kernel is known to move, and factor = known bytes / (counter x 1024) -- the number a counter reading of a kernel
with that access pattern has to be multiplied by.  Writes <out>/<tag>_counter_calibration.json and .md.

usage (on the GPU box): python scripts/pmc_calibrate.py <tag> <outdir> [bytes]"""
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, outdir = sys.argv[1], sys.argv[2]
N = int(sys.argv[3], 0) if len(sys.argv) > 3 else 1 << 30
exe = os.path.join(ROOT, "build_variants", "pmc_probe")
rec = N // 6144 * 6144
# probe -> (bytes read, bytes written, what it stands for)
KNOWN = {
    "probe_copy<uint4>": (N, N, "16 B/lane copy (wide coalesced streaming)"),
    "probe_copy<unsigned int>": (N, N, "4 B/lane copy (dword loads / stores: the image kernels' staging)"),
    "probe_read<uint4>": (N, 0, "16 B/lane read"),
    "probe_read<uint2>": (N, 0, "8 B/lane read (fp64 SoA)"),
    "probe_read<unsigned int>": (N, 0, "4 B/lane read"),
    "probe_read1": (N // 4, 0, "1 B/lane read (byte gathers of the index-reflecting paths)"),
    "probe_write<uint4>": (0, N, "16 B/lane write"),
    "probe_write<double>": (0, N, "8 B/lane write (fp64 SoA: the local map's scratch)"),
    "probe_write<unsigned int>": (0, N, "4 B/lane write"),
    "probe_rows16_write": (0, rec, "non-temporal 16 B/lane rows of 1 KB (the LK template cache's stores)"),
    "probe_rows16_read": (rec, 0, "non-temporal 16 B/lane rows of 1 KB (the LK template cache's loads)"),
}


def norm(name):
    n = name.split("(")[0].replace("void ", "").strip()
    n = n.replace("HIP_vector_type<unsigned int, 4u>", "uint4").replace("HIP_vector_type<unsigned int, 2u>", "uint2")
    return n.replace(" >", ">")


env = dict(os.environ, TMPDIR="/tmp")
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = tempfile.mkdtemp(prefix="pmc_cal_%s_" % ctr, dir="/tmp")
    r = subprocess.run(["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", exe, str(N)],
                       cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if r.returncode != 0 or not files:
        raise SystemExit("rocprofv3 --pmc %s failed (rc %d):\n%s" % (ctr, r.returncode, r.stdout.decode(errors="replace")[-3000:]))
    per = {}
    for x in csv.DictReader(open(files[0])):
        if x["Counter_Name"] == ctr and "probe_" in x["Kernel_Name"]:
            per.setdefault(norm(x["Kernel_Name"]), []).append(float(x["Counter_Value"]))
    for k, v in per.items():
        res.setdefault(k, {})[ctr] = {"mean_kb": sum(v) / len(v), "min_kb": min(v), "max_kb": max(v), "n": len(v)}
out = {"bytes": N, "probes": {}, "note": "factor = known bytes / (counter x 1024): multiply a counter reading by it; None where the probe moves no bytes of that kind"}
lines = ["| probe | stands for | known read MB | FETCH_SIZE MB | fetch factor | known written MB | WRITE_SIZE MB | write factor |", "|---|---|---:|---:|---:|---:|---:|---:|"]
for k, v in sorted(res.items()):
    kr, kw, what = KNOWN.get(k, (None, None, "?"))
    f = v.get("FETCH_SIZE", {}).get("mean_kb")
    w = v.get("WRITE_SIZE", {}).get("mean_kb")
    ff = (kr / (f * 1024.0)) if (kr and f) else None
    wf = (kw / (w * 1024.0)) if (kw and w) else None
    out["probes"][k] = {"stands_for": what, "known_read_bytes": kr, "known_written_bytes": kw, "fetch_size_kb": v.get("FETCH_SIZE"),
                        "write_size_kb": v.get("WRITE_SIZE"), "fetch_factor": ff, "write_factor": wf}
    lines.append("| %s | %s | %s | %s | %s | %s | %s | %s |" % (
        k, what, "%.1f" % (kr / 1e6) if kr is not None else "?", "%.1f" % (f * 1024 / 1e6) if f is not None else "-",
        "%.3f" % ff if ff else "-", "%.1f" % (kw / 1e6) if kw is not None else "?", "%.1f" % (w * 1024 / 1e6) if w is not None else "-",
        "%.3f" % wf if wf else "-"))
os.makedirs(outdir, exist_ok=True)
json.dump(out, open(os.path.join(outdir, "%s_counter_calibration.json" % tag), "w"), indent=1)
open(os.path.join(outdir, "%s_counter_calibration.md" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
