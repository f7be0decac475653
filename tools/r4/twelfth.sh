#!/bin/bash
# round 4, twelfth lease: MFMA-layout epilogues (no LDS patch) in the four-wave kernels: tests, GEMM table A/B, stamps
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4n
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu 2>&1 | tail -6 | tee $O/pytest.txt || exit 1
for rep in 1 2 3; do
  for lib in tools/_ab/libreed_ldsepi.so ""; do
    echo "lib=${lib:-product (MFMA-layout epilogues)}" | tee -a $O/direct_ab.txt; REED_HIP_LIB=$lib timeout -k 10 200 python tools/gemm_table.py 256 20 | tee -a $O/direct_ab.txt || exit 1
  done
done
for lib in tools/_ab/libreed_ldsepi.so "" tools/_ab/libreed_ldsepi.so ""; do
  echo "bench lib=${lib:-product}" | tee -a $O/direct_bench.txt; REED_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>/dev/null | cut -c1-140 | tee -a $O/direct_bench.txt
done
echo done
