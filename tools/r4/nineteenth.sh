#!/bin/bash
# round 4, nineteenth lease: non-temporal LDS-DMA loads in the attention kernels: kernels alone, whole step
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4v
mkdir -p $O
cd $R
for rep in 1 2; do
  for lib in "" tools/_ab/libreed_attnnt.so; do
    echo "lib=${lib:-product}" | tee -a $O/attn_nt.txt; REED_HIP_LIB=$lib timeout -k 10 200 python tools/time_attn.py 2>&1 | tail -8 | tee -a $O/attn_nt.txt || exit 1
  done
done
for rep in 1 2; do
  for lib in "" tools/_ab/libreed_attnnt.so; do
    echo "bench b=256 lib=${lib:-product}" | tee -a $O/attn_nt.txt; REED_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | cut -c1-140 | tee -a $O/attn_nt.txt
    echo "bench b=32 lib=${lib:-product}" | tee -a $O/attn_nt.txt; REED_HIP_LIB=$lib timeout -k 10 300 python bench.py --global-batch 32 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | cut -c1-140 | tee -a $O/attn_nt.txt
  done
done
echo done
