#!/bin/bash
# round 4, twenty-ninth lease: final tree after the TN ring: GEMM tests (default environment), block table, whole step
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4I
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu 2>&1 | tail -3 | tee $O/pytest.txt || exit 1
timeout -k 10 200 python tools/gemm_table.py 256 20 | tee $O/table.txt
for rep in 1 2; do
  timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | cut -c1-140 | tee -a $O/bench.txt
done
echo done
