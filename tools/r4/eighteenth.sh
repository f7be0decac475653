#!/bin/bash
# round 4, eighteenth lease: non-temporal stores of the weight gradients: whole step at b = 32 and b = 256, optimiser tests
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4u
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "adam or optim or clip" 2>&1 | tail -3 | tee $O/pytest.txt || exit 1
for rep in 1 2 3; do
  for lib in "" tools/_ab/libreed_wgnt.so; do
    echo "bench b=32 lib=${lib:-product}" | tee -a $O/wgnt.txt; REED_HIP_LIB=$lib timeout -k 10 300 python bench.py --global-batch 32 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | cut -c1-140 | tee -a $O/wgnt.txt
  done
done
for rep in 1 2; do
  for lib in "" tools/_ab/libreed_wgnt.so; do
    echo "bench b=256 lib=${lib:-product}" | tee -a $O/wgnt.txt; REED_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | cut -c1-140 | tee -a $O/wgnt.txt
  done
done
echo done
