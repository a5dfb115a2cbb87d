#!/usr/bin/env python3
"""Prints `name calls total_us avg_us` for the flvis kernels of a rocprofv3 `*kernel_stats.csv` (names contain commas)."""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1])):
    n = r.get("Name", "")
    if "flvis" in n:
        print("%-22s calls %5s total_us %10.1f avg_us %8.1f" % (n.split("(")[0].replace("flvis::", ""), r.get("Calls"),
              float(r.get("TotalDurationNs", 0)) / 1e3, float(r.get("AverageNs", 0)) / 1e3))
