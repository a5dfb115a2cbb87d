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
    k = r["Kernel_Name"].split("(")[0][-34:]
    c = r["Counter_Name"]
    if c not in names:
        names.append(c)
    acc[k][c] += float(r["Counter_Value"])
    if c == names[0]:
        cnt[k] += 1
print("%-36s %5s " % ("kernel", "n") + " ".join("%14s" % n[-14:] for n in names))
for k in sorted(acc, key=lambda k: -acc[k].get(names[0], 0)):
    n = max(cnt[k], 1)
    print("%-36s %5d " % (k, n) + " ".join("%14.0f" % (acc[k][c] / n) for c in names))
