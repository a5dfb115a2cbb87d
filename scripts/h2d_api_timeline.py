#!/usr/bin/env python3
"""Host API calls (rocprofv3 --hip-trace) beside the uploads and the frame heads of the host-image leg: when was each image upload
ISSUED by the host, when did it run, and what was the host doing in between.
usage: h2d_api_timeline.py <hip_api_trace.csv> <kernel_trace.csv> <memory_copy_trace.csv>"""
import csv
import sys

api = list(csv.DictReader(open(sys.argv[1])))
ker = list(csv.DictReader(open(sys.argv[2])))
cpy = list(csv.DictReader(open(sys.argv[3])))
big = sorted((int(x["Start_Timestamp"]), int(x["End_Timestamp"])) for x in cpy if int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) > 150000)
if not big:
    sys.exit("no uploads")
# (bench.py's pose check re-runs the leg's frames afterwards with synchronous uploads: the leg's own uploads are the first half)
half = big[:len(big) // 2] if len(big) >= 16 else big
t_lo = half[-10][0] if len(half) >= 10 else half[0][0]
t_hi = half[-1][1] + 1500000
t0 = t_lo
ev = []
for s, e in big:
    if t_lo <= s <= t_hi:
        ev.append((s, "GPU  upload runs %.0f us" % ((e - s) / 1e3)))
for x in ker:
    n = x["Kernel_Name"]
    s = int(x["Start_Timestamp"])
    if t_lo <= s <= t_hi and ("k_frame_head" in n or "k_frame_end" in n or "k_store_progress" in n):
        ev.append((s, "GPU  %s (%.0f us)" % (n.split("flvis::")[-1].split("(")[0], (int(x["End_Timestamp"]) - s) / 1e3)))
fk = "Function" if "Function" in api[0] else next(k for k in api[0] if "unction" in k or "Name" in k)
slow = 0
for x in api:
    s, e = int(x["Start_Timestamp"]), int(x["End_Timestamp"])
    if s < t_lo - 2000000 or s > t_hi:
        continue
    n = x[fk]
    d = (e - s) / 1e3
    if "hipMemcpyAsync" in n or "hipMemcpy2DAsync" in n or "hipStreamWaitEvent" in n or d > 60:
        ev.append((s, "HOST %s takes %.0f us" % (n, d)))
ev.sort()
for t, what in ev:
    if t >= t_lo - 2000000:
        print("%10.1f  %s" % ((t - t0) / 1e3, what))
