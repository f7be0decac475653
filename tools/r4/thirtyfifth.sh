#!/bin/bash
# round 4, thirty-fifth lease: whole step with / without the tall / wide tile pairs in the four-wave weight gradients; b = 32 too
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4P
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for pr in 1 0; do
    echo "b=256 REED_WGRAD_W4_PAIRS=$pr" | tee -a $O/pairs.txt; REED_WGRAD_W4_PAIRS=$pr timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['avg_ms_per_launch'], r['roofline']['frac'])" | tee -a $O/pairs.txt
  done
done
for rep in 1 2; do
  for pr in 1 0; do
    echo "b=32 REED_WGRAD_W4_PAIRS=$pr" | tee -a $O/pairs.txt; REED_WGRAD_W4_PAIRS=$pr timeout -k 10 300 python bench.py --global-batch 32 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['avg_ms_per_launch'], r['roofline']['frac'])" | tee -a $O/pairs.txt
  done
  echo "b=32 REED_WGRAD_W4=0" | tee -a $O/pairs.txt; REED_WGRAD_W4=0 timeout -k 10 300 python bench.py --global-batch 32 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['avg_ms_per_launch'], r['roofline']['frac'])" | tee -a $O/pairs.txt
done
echo done
