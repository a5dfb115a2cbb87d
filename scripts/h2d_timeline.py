#!/usr/bin/env python3
"""Timeline of the host-image leg of bench.py from a rocprofv3 run with --kernel-trace --memory-copy-trace: every large H2D copy and the
first / last kernel of every frame (k_frame_head / k_frame_end), in start order -- shows whether the upload of frame n+1 runs under the
kernels of frame n.   usage: h2d_timeline.py <kernel_trace.csv> <memory_copy_trace.csv> [rows]"""
import csv
import sys

ev = []
for x in csv.DictReader(open(sys.argv[1])):
    n = x["Kernel_Name"]
    if "k_frame_head" in n or "k_frame_end" in n:
        ev.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), n.split("(")[0].split("::")[-1]))
rows = list(csv.DictReader(open(sys.argv[2])))
cols = rows[0].keys() if rows else []
size_key = next((k for k in cols if k.lower() in ("size", "bytes", "size_bytes")), None)
dir_key = next((k for k in cols if k.lower() in ("direction", "name", "kind")), None)
for x in rows:
    b = int(float(x[size_key])) if size_key and x[size_key] not in ("", None) else -1
    if 0 <= b < (1 << 20):
        continue
    ev.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), "COPY %s %.1f MB" % (x[dir_key] if dir_key else "?", b / 1e6)))
ev.sort()
keep = int(sys.argv[3]) if len(sys.argv) > 3 else 150
ev = ev[-keep:]
t0 = ev[0][0]
print("# columns of the copy trace:", ", ".join(cols))
for a, b, n in ev:
    print("%-44s start %10.1f us  dur %8.1f us" % (n, (a - t0) / 1e3, (b - a) / 1e3))
