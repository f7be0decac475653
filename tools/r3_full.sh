#!/bin/bash
# Round 3: the whole GPU suite, then the driver-equivalent bench (N = 1 with the C3 per-GPU leg), the self-launcher's error path,
# and a rocprofv3 --stats pass of the same command.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3_full
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -q -m gpu 2>&1 | tail -25 > $O/gputest_summary.txt; echo "pytest rc=$?" | tee -a $O/gputest_summary.txt; tail -6 $O/gputest_summary.txt
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python3 -c "
import json
d=json.load(open('$O/bench_n1.json')); r=d.get('roofline') or {}
print('N=1', d['value'], d['ms_per_step'], d['step_mfma_frac'], r.get('achieved'), r.get('frac'), r.get('avg_ms_per_launch'))
print('c3 leg', d.get('c3_per_gpu_leg'))
print('cpu', d.get('cpu_baseline',{}).get('value'))"
echo done
