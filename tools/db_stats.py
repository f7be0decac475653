#!/usr/bin/env python
"""Per-kernel summary (calls, total, average, share) of a rocprofv3 --kernel-trace result database, as CSV — the same
table `--stats` prints.  usage: python tools/db_stats.py gpurun_out/prof/runc_results.db > profiles/x_kernel_stats.csv"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
                 f"max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for n, k, t, a, lo, hi in rows:
    print(f'"{n}",{k},{t},{a:.1f},{100.0 * t / tot:.3f},{lo},{hi}')
