#!/bin/bash
# round 4, twenty-fifth lease: non-temporal loads of the residual stream in the gate + residual epilogue (alone, and with the saved-array policy): whole step
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4B
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for lib in "" tools/_ab/libreed_ldnt.so tools/_ab/libreed_ldntsaved.so; do
    echo "bench b=256 lib=${lib:-product}" | tee -a $O/ldnt.txt; REED_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | cut -c1-140 | tee -a $O/ldnt.txt
  done
done
echo done
