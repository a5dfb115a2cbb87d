#!/usr/bin/env python3
"""Every kernel and every memory copy of the last frames of the host-image leg (rocprofv3 --kernel-trace --memory-copy-trace), in start
order with the gap to the previous event's end: where a frame waits when its images arrive over PCIe.
usage: h2d_full_timeline.py <kernel_trace.csv> <memory_copy_trace.csv> [frames]"""
import csv
import sys

ev = []
for x in csv.DictReader(open(sys.argv[1])):
    n = x["Kernel_Name"]
    if "flvis::" not in n:
        continue
    ev.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), n.split("flvis::")[1].split("(")[0], "q" + x.get("Queue_Id", "?")))
rows = list(csv.DictReader(open(sys.argv[2])))
cols = rows[0].keys() if rows else []
size_key = next((k for k in cols if k.lower() in ("size", "bytes", "size_bytes")), None)
for x in rows:
    b = int(float(x[size_key])) if size_key and x[size_key] not in ("", None) else -1
    ev.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), "COPY %.3f MB" % (b / 1e6), "dma"))
ev.sort()
heads = [i for i, e in enumerate(ev) if e[2] in ("k_frame_head", "k_frame_head_prepare")]
nf = int(sys.argv[3]) if len(sys.argv) > 3 else 3
# the frames of the host-image leg are those with a k_store_progress (the copy stream's progress word) beside them; what follows the leg
# in bench.py (the resident re-run that checks its poses) has none
marks = [i for i, e in enumerate(ev) if e[2] == "k_store_progress"]
if marks:
    last = marks[-1]
    leg_heads = [i for i in heads if i <= last]
    a = leg_heads[-nf - 1] if len(leg_heads) > nf else 0
    b = next((i for i in heads if i > last), len(ev))
    ev = ev[a:b]
else:
    a = heads[-nf - 1] if len(heads) > nf else 0
    ev = ev[a:]
t0 = ev[0][0]
last_end = ev[0][0]
for s, e, n, q in ev:
    if n.startswith("k_ba_worker"):
        continue
    print("%-4s %-26s start %9.1f  dur %7.1f  after-prev-end %7.1f" % (q, n, (s - t0) / 1e3, (e - s) / 1e3, (s - last_end) / 1e3))
    last_end = max(last_end, e)
