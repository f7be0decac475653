#!/bin/bash
# Round 3, first GPU call: the new fp32-operand build's tests, the changed data-parallel / CLI tests, the self-launching bench
# (N = 1 with the C3 per-GPU leg; --gpus 2 on a one-GPU box must exit non-zero with one line, no hang).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_fp32_gpu.py -q -x -m gpu 2>&1 | tail -40 > $O/pytest_fp32.txt; echo "fp32 tests rc=$?"; tail -15 $O/pytest_fp32.txt
timeout -k 10 600 python -m pytest tests/test_cli_gpu.py -q -x -m gpu -k "cu_reserve or default_mixed or roundtrip" 2>&1 | tail -30 > $O/pytest_cli.txt; echo "cli tests rc=$?"; tail -8 $O/pytest_cli.txt
timeout -k 10 120 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_gpus2.out 2> $O/bench_gpus2.err; echo "bench --gpus 2 rc=$? (expected 2)"; tail -2 $O/bench_gpus2.err
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"
python3 -c "
import json
d=json.load(open('$O/bench_n1.json')); r=d.get('roofline') or {}
print('N=1', d['value'], d['ms_per_step'], d['step_mfma_frac'], r.get('achieved'), r.get('frac'), r.get('avg_ms_per_launch'))
print('c3 leg', d.get('c3_per_gpu_leg'))
print('cpu', d.get('cpu_baseline',{}).get('value'))
for row in d['gemm_family_isolated']['table']: print(row)
"
echo done
