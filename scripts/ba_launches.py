#!/usr/bin/env python3
"""The local-map launches of a rocprofv3 --kernel-trace csv of bench.py: duration of every k_ba_worker launch of the last frames, how many
overlap, and the gap to the frame that queued them.  usage: ba_launches.py <kernel_trace.csv> [last_n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
last_n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
ba = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_ba_worker" in r["Kernel_Name"])
fe = sorted(int(r["End_Timestamp"]) for r in rows if "k_frame_end" in r["Kernel_Name"])
ba = ba[-last_n:]
if not ba:
    sys.exit("no k_ba_worker launches in the trace")
d = sorted((e - s) / 1e3 for s, e in ba)
q = lambda f: d[min(len(d) - 1, int(f * len(d)))]  # noqa: E731
print("k_ba_worker, last %d launches: min %.0f  p25 %.0f  p50 %.0f  p75 %.0f  p90 %.0f  max %.0f us; mean %.0f us" %
      (len(d), d[0], q(0.25), q(0.5), q(0.75), q(0.9), d[-1], sum(d) / len(d)))
# the frame whose k_frame_end precedes the launch's start: how long the launch waited behind it
import bisect
waits = []
for s, e in ba:
    i = bisect.bisect_right(fe, s) - 1
    if i >= 0:
        waits.append((s - fe[i]) / 1e3)
waits.sort()
print("start of a launch after the last k_frame_end before it: p50 %.0f  p90 %.0f  max %.0f us" % (waits[len(waits) // 2], waits[int(0.9 * len(waits))], waits[-1]))
span = (ba[-1][1] - ba[0][0]) / 1e3
print("span of these launches %.0f us = %.0f us per launch; busy (union) %.0f us" % (span, span / len(ba), sum(min(e, ba[i + 1][0] if i + 1 < len(ba) else e) - s for i, (s, e) in enumerate(ba)) / 1e3))
print("durations in launch order (us): " + " ".join("%.0f" % ((e - s) / 1e3) for s, e in ba[-60:]))
