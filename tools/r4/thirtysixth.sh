#!/bin/bash
# round 4, thirty-sixth lease: non-temporal loads of the weight gradients' X operand: whole step A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4Q
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for lib in "" tools/_ab/libreed_qnt.so; do
    echo "b=256 lib=${lib:-product}" | tee -a $O/qnt.txt; REED_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['avg_ms_per_launch'], r['roofline']['frac'])" | tee -a $O/qnt.txt
  done
done
echo done
