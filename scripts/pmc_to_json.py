#!/usr/bin/env python3
"""Reduces two rocprofv3 --pmc passes (FETCH_SIZE pass, WRITE_SIZE pass) of `bench.py` to per-launch HBM traffic of one kernel.
usage: pmc_to_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel substring> <out.json>"""
import csv
import json
import sys


def avg(path, counter, kern):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if kern in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            n += 1
    return (tot / n if n else None), n


f, nf = avg(sys.argv[1], "FETCH_SIZE", sys.argv[3])
w, nw = avg(sys.argv[2], "WRITE_SIZE", sys.argv[3])
out = {"kernel": sys.argv[3], "fetch_size_kb_per_launch": f, "write_size_kb_per_launch": w, "launches": [nf, nw],
       "traffic_bytes_per_launch": (f + w) * 1024.0 if f is not None and w is not None else None,
       "note": "rocprofv3 FETCH_SIZE / WRITE_SIZE (KB) averaged over all launches of the kernel in `bench.py --steps 20 --warmup 60`, "
               "two separate --pmc passes; uncorrected: the gfx950 factor-2 under-count documented for 16 B/lane streaming "
               "reads is not calibrated for this kernel's dword accesses (MI355X_MICROARCH.md, HBM section)"}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out))
