#!/bin/bash
# Same-box A/B of two builds of libreed_hip.so through bench.py (alternating, so that clock / box drift cancels).
# usage: bash tools/ab_lib.sh tools/_ab/libreed_hip_old.so "32 256"
OLD=$1
for b in ${2:-32 256}; do
  for rep in 1 2; do
    for lib in "$OLD" ""; do
      echo "== b=$b lib=${lib:-current}"
      REED_HIP_LIB=$lib python bench.py --global-batch $b --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table 2>&1 | tail -1 | cut -c1-130
    done
  done
done
