#!/bin/bash
# Round-2 final refresh at HEAD (after the persistent GEMM, the host-side fixes and the weight-gradient load hoist): bench at
# b = 256 / 128 / 64 / 32 and rocprofv3 kernel stats at b = 256 and 32.  Outputs under gpurun_out/r2d/ (copied into profiles/).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2d
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
for b in 128 64 32; do python bench.py --steps 10 --warmup 3 --global-batch $b --no-cpu-baseline > $O/bench_n1_b$b.json 2>> $O/bench_n1.err; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof256 --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-table > $O/bench_n1_under_rocprof.json 2> $O/rocprof256.err
rocprofv3 --kernel-trace --stats -d $O/prof32 --output-format csv -- python3 $R/bench.py --steps 8 --warmup 2 --global-batch 32 --no-cpu-baseline --no-kernel-table > $O/bench_n1_b32_under_rocprof.json 2> $O/rocprof32.err
cd $R
for f in $O/bench_n1.json $O/bench_n1_b128.json $O/bench_n1_b64.json $O/bench_n1_b32.json $O/bench_n1_under_rocprof.json $O/bench_n1_b32_under_rocprof.json; do python3 -c "
import json,sys
d=json.load(open(sys.argv[1])); r=d.get('roofline') or {}
print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['step_mfma_frac'], r.get('achieved'), r.get('frac'), r.get('avg_ms_per_launch'))" $f; done
find $O -name "*kernel_stats.csv" | head
echo done
