#!/bin/bash
# Round 6: the 256x288 kernel in the b = 32 step: REED_GEMM288=1 lets the heuristic take it (fc1 forward; with --dgrad-nt 1 also the fc2
# input gradient as an NT GEMM on W2^T)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6q
mkdir -p $O
cd $R
timeout -k 10 200 python tools/r6/t288.py 32 16 2>&1 | grep -v amdgpu.ids > $O/launch.txt; cat $O/launch.txt
for rep in 1 2 3; do
  for cfg in "0 auto" "1 auto" "0 1" "1 1"; do
    set -- $cfg
    echo "== REED_GEMM288=$1 --dgrad-nt $2"
    REED_GEMM288=$1 timeout -k 10 300 python bench.py --global-batch 32 --steps 30 --warmup 5 --dgrad-nt $2 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs --no-loss-vs-ref 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
  done
done > $O/step.txt 2>&1
cat $O/step.txt
