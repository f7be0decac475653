#!/bin/bash
# round 4, thirteenth lease: VALU issue costs for one wave per SIMD; the GEMM table with and without the compiler's packed-f32 (SLP) ops
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4o
mkdir -p $O
cd $R
timeout -k 10 120 tools/micro/valu_rate | tee $O/valu_rate.txt || exit 1
for rep in 1 2 3; do
  for lib in tools/_ab/libreed_noslp.so ""; do
    echo "lib=${lib:-product}" | tee -a $O/noslp_ab.txt; REED_HIP_LIB=$lib timeout -k 10 200 python tools/gemm_table.py 256 20 | tee -a $O/noslp_ab.txt || exit 1
  done
done
echo done
