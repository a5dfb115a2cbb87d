#!/usr/bin/env python3
"""Prints the kernel timeline of one frame (between two k_frame_begin launches) from a rocprofv3 kernel trace csv."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "flvis::" in r["Kernel_Name"]]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["k"] = r["Kernel_Name"].split("flvis::")[1].split("(")[0]
rows.sort(key=lambda r: r["s"])
fb = [i for i, r in enumerate(rows) if r["k"] in ("k_frame_head", "k_frame_head_prepare", "k_frame_begin")]  # first kernel of a frame
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(fb) - 5
a, b = fb[which], fb[which + 1]
t0 = rows[a]["s"]
qs = sorted(set(r["Queue_Id"] for r in rows))
print("frame %d: %.1f us between frame_begin launches; queues %s" % (which, (rows[b]["s"] - t0) / 1e3, qs))
# include kernels of other queues that overlap the window
win = [r for r in rows if r["e"] > t0 and r["s"] < rows[b]["s"]]
last_end = {}
for r in win:
    q = r["Queue_Id"]
    gap = (r["s"] - last_end[q]) / 1e3 if q in last_end else float("nan")
    print("q%-3s %-18s start %8.1f  dur %7.1f  gap-before %6.1f" % (q, r["k"], (r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3, gap))
    last_end[q] = r["e"]
