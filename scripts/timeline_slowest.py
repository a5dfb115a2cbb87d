#!/usr/bin/env python3
"""Prints the kernel timeline of the SLOWEST of the last N frames of a rocprofv3 kernel trace (frame = from one k_frame_head to the
next), all queues.   timeline_slowest.py trace.csv [last_n_frames]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "flvis::" in r["Kernel_Name"]]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["k"] = r["Kernel_Name"].split("flvis::")[1].split("(")[0]
rows.sort(key=lambda r: r["s"])
fb = [r for r in rows if r["k"] in ("k_frame_head", "k_frame_head_prepare", "k_frame_begin")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fb = fb[-(n + 1):]
spans = [(fb[i + 1]["s"] - fb[i]["s"], i) for i in range(len(fb) - 1)]
print("frame spans (us):", [round(d / 1e3) for d, _ in spans])
d, i = max(spans)
t0, t1 = fb[i]["s"], fb[i + 1]["s"]
qs = {q: k for k, q in enumerate(sorted(set(r["Queue_Id"] for r in rows), key=lambda x: int(x)))}
print("slowest frame: %d of the last %d, %.1f us" % (i, n, d / 1e3))
for r in rows:
    if r["e"] > t0 - 200000 and r["s"] < t1:
        print("q%-2d %-20s start %8.1f  dur %7.1f" % (qs[r["Queue_Id"]], r["k"], (r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3))
