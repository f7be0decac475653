#!/bin/bash
# round 4, sixteenth lease: non-temporal epilogue stores (selective = product, everywhere, nowhere): GEMM tests, whole-step A/B at b = 256 and b = 32
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4s
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu 2>&1 | tail -3 | tee $O/pytest.txt || exit 1
for rep in 1 2 3; do
  for lib in tools/_ab/libreed_st0.so tools/_ab/libreed_stnt.so ""; do
    echo "bench lib=${lib:-product (nt except gate+res)}" | tee -a $O/policy_bench.txt; REED_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>/dev/null | cut -c1-140 | tee -a $O/policy_bench.txt
  done
done
for lib in tools/_ab/libreed_st0.so tools/_ab/libreed_stnt.so ""; do
  echo "table lib=${lib:-product}" | tee -a $O/policy_table.txt; REED_HIP_LIB=$lib timeout -k 10 200 python tools/gemm_table.py 256 20 | tee -a $O/policy_table.txt || exit 1
  echo "table b=32 lib=${lib:-product}" | tee -a $O/policy_table.txt; REED_HIP_LIB=$lib timeout -k 10 200 python tools/gemm_table.py 32 50 | tee -a $O/policy_table.txt || exit 1
done
echo done
