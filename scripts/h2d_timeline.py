#!/usr/bin/env python3
"""Memory copies and the first / last kernel of every frame from a rocprofv3 trace (--kernel-trace --memory-copy-trace) of the
host-feed leg: shows whether the upload of frame N+1 overlaps the kernels of frame N.  h2d_timeline.py <dir with the csv files>"""
import csv
import glob
import sys

d = sys.argv[1]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
mt = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(kt)):
    if "flvis::" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("flvis::")[1].split("(")[0]
        if k in ("k_frame_head", "k_frame_end"):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
for r in csv.DictReader(open(mt)):
    b = int(r.get("Bytes", r.get("Size", 0)) or 0)
    if b >= (1 << 20):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy %s %.1f MB" % (r.get("Direction", "?"), b / 1e6)))
ev.sort()
t0 = ev[-60][0] if len(ev) > 60 else ev[0][0]
for s, e, k in ev[-60:]:
    print("%-28s start %9.1f us  dur %8.1f us" % (k, (s - t0) / 1e3, (e - s) / 1e3))
