#!/bin/bash
# round 4, seventeenth lease: optimiser pass with non-temporal state accesses: the pass alone, whole step at b = 256 and b = 32
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4t
mkdir -p $O
cd $R
for lib in "" tools/_ab/libreed_optnt.so; do
  echo "lib=${lib:-product}" | tee -a $O/optnt.txt
  REED_HIP_LIB=$lib timeout -k 10 200 python tools/_ab/time_adam.py 2>&1 | tail -4 | tee -a $O/optnt.txt
done
for rep in 1 2 3; do
  for lib in "" tools/_ab/libreed_optnt.so; do
    echo "bench b=256 lib=${lib:-product}" | tee -a $O/optnt.txt; REED_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>/dev/null | cut -c1-140 | tee -a $O/optnt.txt
    echo "bench b=32 lib=${lib:-product}" | tee -a $O/optnt.txt; REED_HIP_LIB=$lib timeout -k 10 300 python bench.py --local-batch 32 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>/dev/null | cut -c1-140 | tee -a $O/optnt.txt
  done
done
echo done
