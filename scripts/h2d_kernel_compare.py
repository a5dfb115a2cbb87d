#!/usr/bin/env python3
"""Per-kernel durations of bench.py's host-image leg against the resident frames before it, from one rocprofv3 --kernel-trace
--memory-copy-trace run of bench.py: which kernels last longer when the images arrive over PCIe, and how long a frame's chain is.
usage: h2d_kernel_compare.py <kernel_trace.csv> <memory_copy_trace.csv>"""
import collections
import csv
import sys

ker = []
for x in csv.DictReader(open(sys.argv[1])):
    n = x["Kernel_Name"]
    if "flvis::" in n:
        ker.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), n.split("flvis::")[1].split("(")[0]))
ker.sort()
rows = list(csv.DictReader(open(sys.argv[2])))
size_key = next((k for k in (rows[0].keys() if rows else []) if k.lower() in ("size", "bytes", "size_bytes")), None)
def is_big(x):   # an image upload: by size where the trace has one, else by duration (19.66 MB take ~350 us over PCIe)
    if size_key and x[size_key] not in ("", None) and float(x[size_key]) > 0:
        return float(x[size_key]) > 8e6
    return int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) > 150000
big = sorted(int(x["Start_Timestamp"]) for x in rows if is_big(x))
if not big:
    sys.exit("no image uploads in the trace; columns: %s" % (list(rows[0].keys()) if rows else "none"))
t_h2d = big[0]
heads = [s for s, e, n in ker if n in ("k_frame_head", "k_frame_head_prepare")]
res_heads = [h for h in heads if h < t_h2d][-41:]
h2d_heads = [h for h in heads if h > t_h2d][4:]
def table(lo, hi):
    d = collections.defaultdict(list)
    for s, e, n in ker:
        if lo <= s < hi:
            d[n].append((e - s) / 1e3)
    return d
a = table(res_heads[0], res_heads[-1])
b = table(h2d_heads[0], h2d_heads[-1])
fa = (res_heads[-1] - res_heads[0]) / 1e3 / (len(res_heads) - 1)
fb = (h2d_heads[-1] - h2d_heads[0]) / 1e3 / (len(h2d_heads) - 1)
print("frame period (k_frame_head to k_frame_head): resident %.1f us (%d frames), host images %.1f us (%d frames)" % (fa, len(res_heads) - 1, fb, len(h2d_heads) - 1))
print("%-28s %10s %10s %7s   per frame: resident / host (us)" % ("kernel", "resident", "host", "ratio"))
tot_a = tot_b = 0
for n in sorted(a, key=lambda k: -sum(b.get(k, [0])) ):
    if n not in b or n == "k_ba_worker":
        continue
    ma, mb = sum(a[n]) / len(a[n]), sum(b[n]) / len(b[n])
    pa, pb = sum(a[n]) / (len(res_heads) - 1), sum(b[n]) / (len(h2d_heads) - 1)
    tot_a += pa
    tot_b += pb
    print("%-28s %10.1f %10.1f %7.2f   %8.1f / %8.1f" % (n, ma, mb, mb / ma, pa, pb))
print("sum of kernel time per frame (all streams' queues): resident %.1f us, host images %.1f us" % (tot_a, tot_b))
cp = [(int(x["Start_Timestamp"]), int(x["End_Timestamp"])) for x in rows if is_big(x)]
cp = [c for c in cp if c[0] > h2d_heads[0]]
if cp:
    print("image uploads in the leg: %d, mean %.1f us each (%.1f GB/s)" % (len(cp), sum(e - s for s, e in cp) / len(cp) / 1e3,
          19.6608e6 * len(cp) / sum(e - s for s, e in cp)))
