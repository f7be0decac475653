#!/bin/bash
# Round 6: the NT input gradients on transposed weight copies at the per-GPU batches of the 2- / 4-GPU strong-scaling runs (b = 128, 64)
# and below the threshold (b = 48, 32): bench.py --dgrad-nt 0 / 1 alternating
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6d
mkdir -p $O
cd $R
for b in 128 64 48 32; do
  for rep in 1 2; do
    for nt in 0 1; do
      echo "== b=$b --dgrad-nt $nt"
      timeout -k 10 400 python bench.py --global-batch $b --steps $((b >= 128 ? 12 : 24)) --warmup 4 --dgrad-nt $nt --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs --no-loss-vs-ref 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
    done
  done
done > $O/step.txt 2>&1
cat $O/step.txt
