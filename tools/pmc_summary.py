#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, the mean of every counter per dispatch.
usage: python tools/pmc_summary.py <dir> [name filter]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt in k:
            acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()):
        print(f"    {n:32s} {sum(v) / len(v):16.1f}  (n={len(v)})")
