#!/bin/bash
# Round 6: the column split of the two 4608-wide GEMMs at b = 32 per GPU: tests, launch times, step A/B (REED_GEMM_COLSPLIT=0 = without)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6s
mkdir -p $O
cd $R
# (tests: tests/test_gemm_gpu.py -k column_split)

timeout -k 10 200 python tools/r6/nsplit.py 32 2>&1 | grep -v amdgpu.ids > $O/launch.txt; cat $O/launch.txt
for rep in 1 2 3; do
  for cs in 0 1; do
    echo "== REED_GEMM_COLSPLIT=$cs"
    REED_GEMM_COLSPLIT=$cs timeout -k 10 300 python bench.py --global-batch 32 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs --no-loss-vs-ref 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
  done
done > $O/step.txt 2>&1
cat $O/step.txt
