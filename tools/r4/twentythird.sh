#!/bin/bash
# round 4, twenty-third lease: ln_mod_bwd2 with non-temporal saved-activation / residual-gradient accesses: whole step A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4z
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for lib in "" tools/_ab/libreed_lnbnt.so; do
    echo "bench b=256 lib=${lib:-product}" | tee -a $O/lnbnt.txt; REED_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | cut -c1-140 | tee -a $O/lnbnt.txt
    echo "bench b=32 lib=${lib:-product}" | tee -a $O/lnbnt.txt; REED_HIP_LIB=$lib timeout -k 10 300 python bench.py --global-batch 32 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | cut -c1-140 | tee -a $O/lnbnt.txt
  done
done
echo done
