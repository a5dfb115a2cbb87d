#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc counter_collection csv (printed as a small table)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
names = []
for r in rows:
    if "flvis::" not in r["Kernel_Name"]:
        continue
    k = r["Kernel_Name"].split("flvis::")[1].split("(")[0]
    c = r["Counter_Name"]
    if c not in names:
        names.append(c)
    acc[k][c] += float(r["Counter_Value"])
    if c == names[0]:
        cnt[k] += 1
print("%-20s %5s " % ("kernel", "n") + " ".join("%16s" % n[-16:] for n in names))
for k in sorted(acc, key=lambda k: -acc[k].get(names[0], 0)):
    n = max(cnt[k], 1)
    print("%-20s %5d " % (k, n) + " ".join("%16.0f" % (acc[k][c] / n) for c in names))
