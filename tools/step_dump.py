#!/usr/bin/env python
"""Dump one optimiser step of a rocprofv3 --kernel-trace CSV as a launch list: queue, start offset (us), duration (us), name.
usage: python tools/step_dump.py <kernel_trace.csv> [step index from the end, default 2]"""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
marks = [i for i, r in enumerate(rows) if "sample_posterior" in r[3]]
lo, hi = marks[-k - 1], marks[-k]
t0 = rows[lo][0]
qs = sorted({r[2] for r in rows[lo:hi]})
for s, e, q, n in rows[lo:hi]:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)_kernel", n)
    n = m.group(1) if m else n.split("(")[0][:60]
    print(f"q{qs.index(q)} {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f}  {n}")
