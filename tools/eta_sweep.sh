#!/bin/bash
# In-step A/B of the 256x144 tile's selection threshold (REED_GEMM144_ETA: 0 = never, larger = more shapes take it).
for b in ${1:-32 64 128 256}; do
  for eta in ${2:-0.01 0.8 1.0 1.3}; do
    echo "== b=$b eta=$eta"
    REED_GEMM144_ETA=$eta python bench.py --global-batch $b --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table 2>&1 | tail -1 | cut -c1-200
  done
done
