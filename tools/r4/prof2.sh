#!/bin/bash
# round 4, final tree: rocprofv3 kernel stats of the bench at b = 256 and b = 32 with the bench records of the SAME runs
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4C
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "[$(date +%T)] kernel trace b=256"
rocprofv3 --kernel-trace --stats -d $O/prof256 --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs > $O/bench_n1_under_rocprof.json 2> $O/rocprof256.err
echo "[$(date +%T)] kernel trace b=32"
rocprofv3 --kernel-trace --stats -d $O/prof32 --output-format csv -- python3 $R/bench.py --steps 8 --warmup 2 --global-batch 32 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs > $O/bench_n1_b32_under_rocprof.json 2> $O/rocprof32.err
cd $R
for d in prof256 prof32; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $f $O/${d}_kernel_stats.csv; t=$(find $O/$d -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t 4 > $O/${d}_timeline.txt 2>&1; done
rm -rf $O/prof256 $O/prof32
cut -c1-200 $O/bench_n1_under_rocprof.json; cut -c1-200 $O/bench_n1_b32_under_rocprof.json
echo done
