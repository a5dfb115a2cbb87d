#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace csv of `bench.py` into a small markdown table.

usage: rocprof_summary.py <kernel_trace.csv> <steps> [out.md]
For every flvis kernel: launches, total time, average over ALL launches (what `--stats` prints; it includes the idle
start-up frames) and the average over the launches of the LAST <steps> frames (= bench.py's timed region), which is the
figure bench.py's `roofline.avg_launch_ms` has to agree with."""
import collections
import csv
import sys

path, steps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
by = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    if "flvis::" not in name:
        continue
    k = name.split("flvis::")[1].split("(")[0]
    by[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
out = []
out.append("| kernel | launches | total ms | avg us (all) | launches/frame | avg us (last %d frames) |" % steps)
out.append("|---|---:|---:|---:|---:|---:|")
tot_all = sum(e - s for v in by.values() for s, e in v)
nframes = max(len(v) for k, v in by.items() if k in ("k_frame_end", "k_frame_begin"))
nactive = len(by.get("k_ransac_f", [])) or nframes  # frames after the skipped start-up frames run the full sequence
for k, v in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    v.sort()
    d = [(e - s) / 1000.0 for s, e in v]
    per_frame = max(1, round(len(v) / nactive)) if len(v) > nframes else 1
    last = d[-steps * per_frame:] if len(d) >= steps * per_frame else d
    out.append("| %s | %d | %.2f | %.1f | %d | %.1f |" % (k, len(d), sum(d) / 1000.0, sum(d) / len(d), per_frame, sum(last) / len(last)))
out.append("")
out.append("flvis kernels total: %.2f ms over %d frames; non-flvis kernels in the trace (torch renderer, copies) are omitted."
           % (tot_all / 1e6, nframes))
text = "\n".join(out)
print(text)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(text + "\n")
