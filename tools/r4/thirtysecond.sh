#!/bin/bash
# round 4, thirty-second lease: whole step with the four-wave TN form (ring + static split) against the shipped grouped kernel
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4L
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for w4 in 0 1; do
    echo "bench b=256 REED_WGRAD_W4=$w4" | tee -a $O/w4step.txt; REED_WGRAD_W4=$w4 timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['avg_ms_per_launch'], r['roofline']['frac'], r['roofline'].get('kernel','')[:40])" | tee -a $O/w4step.txt
  done
done
echo done
