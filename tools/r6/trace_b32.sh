#!/bin/bash
# Round 6: kernel trace of the b = 32 step with the main-queue gaps itemised (tools/timeline.py): where does the main stream wait
# for the optimiser's side stream?
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6t
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof32 --output-format csv -- python3 $R/bench.py --steps 8 --warmup 2 --global-batch ${B:-32} --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs --no-loss-vs-ref > $O/bench_b32.json 2> $O/rocprof32.err || { tail -5 $O/rocprof32.err; exit 1; }
cd $R
t=$(find $O/prof32 -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $t 4 > $O/b32_timeline.txt 2>&1
python - $t > $O/b32_step_start.txt <<'PY'
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "sample_posterior" in r[3]]
i0 = marks[-2]
# from 40 launches before the step marker (the end of the previous backward) to 120 after it
t0 = rows[i0][0]
for s, e, q, n in rows[max(0, i0 - 30): i0 + 130]:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} q{q} {n}")
PY
rm -rf $O/prof32
tail -20 $O/b32_timeline.txt
