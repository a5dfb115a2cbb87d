#!/usr/bin/env python3
"""Prints every flvis kernel of a rocprofv3 kernel trace csv inside a time window (default: 5 ms starting at the 100th-from-last
k_frame_begin): queue, kernel, workgroups, start, duration.  For multi-lane runs, where one frame's kernels interleave with other
lanes'.   timeline_window.py trace.csv [window_ms] [frame_begins_from_end]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "flvis::" in r["Kernel_Name"]]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["k"] = r["Kernel_Name"].split("flvis::")[1].split("(")[0]
    gx, wx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
    gy = int(r.get("Grid_Size_Y", 1) or 1)
    wy = int(r.get("Workgroup_Size_Y", 1) or 1)
    r["wg"] = (gx // max(wx, 1)) * (gy // max(wy, 1))
rows.sort(key=lambda r: r["s"])
win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
back = int(sys.argv[3]) if len(sys.argv) > 3 else 100
fb = [r for r in rows if r["k"] in ("k_frame_head", "k_frame_head_prepare", "k_frame_begin")]  # first kernel of a frame
t0 = fb[max(0, len(fb) - back)]["s"]
t1 = t0 + int(win_ms * 1e6)
qs = {q: i for i, q in enumerate(sorted(set(r["Queue_Id"] for r in rows), key=lambda x: int(x)))}
print("window %.1f ms; queues %s" % (win_ms, list(qs)))
for r in rows:
    if r["e"] > t0 and r["s"] < t1:
        print("q%-2d %-20s wg %5d  start %8.1f  dur %7.1f" % (qs[r["Queue_Id"]], r["k"], r["wg"], (r["s"] - t0) / 1e3, (r["e"] - r["s"]) / 1e3))
