#!/bin/bash
# round 4, twenty-seventh lease: GELU / dGELU epilogues on packed fp32 pairs: tests, GEMM table and whole step against the scalar forms
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4E
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu -k "epilogue or gelu" 2>&1 | tail -3 | tee $O/pytest.txt || exit 1
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -q -x -m gpu 2>&1 | tail -3 | tee -a $O/pytest.txt || exit 1
for rep in 1 2 3; do
  for lib in tools/_ab/libreed_unpacked.so ""; do
    echo "table lib=${lib:-product (packed)}" | tee -a $O/packed.txt; REED_HIP_LIB=$lib timeout -k 10 200 python tools/gemm_table.py 256 20 | tee -a $O/packed.txt || exit 1
  done
done
for rep in 1 2 3; do
  for lib in tools/_ab/libreed_unpacked.so ""; do
    echo "bench b=256 lib=${lib:-product (packed)}" | tee -a $O/packed.txt; REED_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | cut -c1-140 | tee -a $O/packed.txt
  done
done
echo done
