#!/bin/bash
# round 4: rocprofv3 kernel stats of the step through bench.py's multi-GPU path at world 1 (REED_FORCE_REDUCER=1, plain plan only): which
# kernels the data-parallel backward launches beside its gradient buckets (gemm_tn_group_kernel, the one-shot four-wave GEMMs, the
# attention backward with four item lists per CU)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4V
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export REED_FORCE_REDUCER=1 REED_BENCH_TUNED=0
echo "[$(date +%T)] kernel trace, data-parallel path at world 1, b=256"
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs > $O/bench_dp1_under_rocprof.json 2> $O/rocprof.err
echo "rc=$?"
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/dp1_kernel_stats.csv
t=$(find $O/prof -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t 3 > $O/dp1_timeline.txt 2>&1
python - <<PY > $O/dp1_grids.txt 2>&1
import csv, collections
acc = collections.defaultdict(lambda: collections.Counter())
for r in csv.DictReader(open("$t")):
    n = r["Kernel_Name"]
    if any(k in n for k in ("attn_bwd", "tn_group", "gemm256w")):
        acc[n[:80]][(r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))] += 1
for n, c in acc.items():
    print(n, dict(c))
PY
rm -rf $O/prof
cut -c1-200 $O/bench_dp1_under_rocprof.json; head -30 $O/dp1_timeline.txt; cat $O/dp1_grids.txt
echo done
