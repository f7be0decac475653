#!/bin/bash
# Round 6, lease 5: the second-size launch time of tools/bench_wgrad_group.py — kernel time or launch overhead?  (kernel trace)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ITERS=10 timeout -k 10 300 rocprofv3 --kernel-trace -d $O/kt --output-format csv -- python3 $R/tools/bench_wgrad_group.py 64 256 > $O/kt.log 2>&1
tail -3 $O/kt.log
cd $R
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/kt/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(list(rows[0].keys()))
sel = [r for r in rows if "tn_group" in r["Kernel_Name"]]
prev = None
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0
    print(r["Kernel_Name"][22:60], "dur %.1f us" % ((e - s) / 1e3), "gap since the previous one's end %.1f us" % gap, "scratch", r.get("Private_Segment_Size", r.get("Scratch_Size", "?")), "grid", r.get("Grid_Size", "?"))
    prev = e
PY
rm -rf $O/kt
