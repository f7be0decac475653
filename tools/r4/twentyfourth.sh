#!/bin/bash
# round 4, twenty-fourth lease: non-temporal policy for the saved-for-backward arrays of the GEMM epilogues: whole step A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4A
mkdir -p $O
cd $R
for rep in 1 2 3 4; do
  for lib in "" tools/_ab/libreed_savednt.so; do
    echo "bench b=256 lib=${lib:-product}" | tee -a $O/savednt.txt; REED_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | cut -c1-140 | tee -a $O/savednt.txt
  done
done
for lib in "" tools/_ab/libreed_savednt.so; do
  echo "table lib=${lib:-product}" | tee -a $O/savednt.txt; REED_HIP_LIB=$lib timeout -k 10 200 python tools/gemm_table.py 256 20 | tee -a $O/savednt.txt || exit 1
done
echo done
