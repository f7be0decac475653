#!/bin/bash
# A/B of the weight-gradient side stream and the overlapped optimiser over the local batch sizes the 1/2/4/8-GPU
# strong-scaling runs see.   usage (GPU box): bash tools/sweep_batch.sh "32 64 128 256" > gpurun_out/sweep.log
for b in ${1:-32 64 128 256}; do
  for cfg in "0 0" "1 0" "0 1" "1 1"; do
    set -- $cfg
    echo "== b=$b REED_WGRAD_STREAM=$1 REED_OPT_OVERLAP=$2"
    REED_WGRAD_STREAM=$1 REED_OPT_OVERLAP=$2 python bench.py --global-batch $b --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table 2>&1 | tail -1 | cut -c1-200
  done
done
