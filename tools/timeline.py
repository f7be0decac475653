#!/usr/bin/env python
"""Timeline view of a rocprofv3 --kernel-trace CSV: over the last K optimiser steps (delimited by adamw_ema launches of the
largest grid... in practice by the `sample_posterior` kernel that opens every step), the time no kernel is running, the time at
least two overlap, per-queue busy time, and the idle gaps attributed to the kernel that ends them.
usage: python tools/timeline.py <kernel_trace.csv> [steps]"""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)_kernel", n)
    if m:
        return m.group(1)
    return n.split("(")[0][:48]


marks = [i for i, r in enumerate(rows) if "sample_posterior" in r[3]]
if len(marks) < K + 1:
    raise SystemExit(f"only {len(marks)} step markers")
lo, hi = marks[-K - 1], marks[-1]
seg = rows[lo:hi]
t0, t1 = seg[0][0], rows[hi][0]
print(f"{K} steps, {len(seg)} launches ({len(seg) / K:.0f} per step), wall {(t1 - t0) / K / 1e6:.3f} ms per step")
# union / overlap by sweep
ev = []
for s, e, q, n in seg:
    ev.append((s, 1))
    ev.append((min(e, t1), -1))
ev.sort()
depth, last, busy, multi = 0, t0, 0, 0
for t, d in ev:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        multi += t - last
    depth += d
    last = t
print(f"per step: some kernel running {busy / K / 1e6:.3f} ms, idle {(t1 - t0 - busy) / K / 1e6:.3f} ms, >= 2 kernels {multi / K / 1e6:.3f} ms")
perq = defaultdict(int)
for s, e, q, n in seg:
    perq[q] += e - s
print("busy per queue (ms per step):", {q: round(v / K / 1e6, 3) for q, v in sorted(perq.items())})
# idle gaps (no kernel at all) attributed to the kernel that starts after the gap
gaps = defaultdict(lambda: [0, 0])
cur_end = seg[0][1]
for s, e, q, n in seg[1:]:
    if s > cur_end:
        g = gaps[short(n)]
        g[0] += s - cur_end
        g[1] += 1
    cur_end = max(cur_end, e)
print("idle before (us per step, count per step):")
for n, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {n:48s} {g / K / 1e3:9.1f} {c / K:7.1f}")
# time per kernel name on the main queue, and the part of it overlapped by another queue
main_q = max(perq, key=perq.get)
tot = defaultdict(lambda: [0, 0])
for s, e, q, n in seg:
    t = tot[(q == main_q, short(n))]
    t[0] += e - s
    t[1] += 1
print(f"kernels (main queue = {main_q}): ms per step, launches per step, avg us")
for (m, n), (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:24]:
    print(f"  {'main' if m else 'side'} {n:48s} {d / K / 1e6:8.3f} {c / K:6.1f} {d / c / 1e3:8.1f}")
# gaps of the MAIN queue (another queue may be running meanwhile), attributed to the main-queue kernel that ends them, with the
# part of each gap during which a side-queue kernel was running (= the main stream waiting on the side stream's events)
mains = [(s, e, n) for s, e, q, n in seg if q == main_q]
sides = sorted((s, e) for s, e, q, n in seg if q != main_q)


def side_cover(a, b):
    c = 0
    for s, e in sides:
        if e <= a:
            continue
        if s >= b:
            break
        c += min(e, b) - max(s, a)
    return c


mg = defaultdict(lambda: [0, 0, 0])
cur_end = mains[0][1]
for s, e, n in mains[1:]:
    if s > cur_end:
        g = mg[short(n)]
        g[0] += s - cur_end
        g[1] += 1
        g[2] += side_cover(cur_end, s)
    cur_end = max(cur_end, e)
print("main-queue gaps: us per step, count per step, of which a side-queue kernel was running (us per step)")
for n, (g, c, sc) in sorted(mg.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {n:48s} {g / K / 1e3:9.1f} {c / K:7.1f} {sc / K / 1e3:9.1f}")
