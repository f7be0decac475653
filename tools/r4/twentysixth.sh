#!/bin/bash
# round 4, twenty-sixth lease: ln_mod_fwd with a non-temporal read of the residual stream: whole step A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4D
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for lib in "" tools/_ab/libreed_lnfnt.so; do
    echo "bench b=256 lib=${lib:-product}" | tee -a $O/lnfnt.txt; REED_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | cut -c1-140 | tee -a $O/lnfnt.txt
    echo "bench b=32 lib=${lib:-product}" | tee -a $O/lnfnt.txt; REED_HIP_LIB=$lib timeout -k 10 300 python bench.py --global-batch 32 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | cut -c1-140 | tee -a $O/lnfnt.txt
  done
done
echo done
