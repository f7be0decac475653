#!/bin/bash
# round 4, thirty-fourth lease: tall / wide tile pairs in the four-wave weight gradients: tests, time per launch with / without the pairs, stamps
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4O
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu -k "wgrad" 2>&1 | tail -3 | tee $O/pytest.txt || exit 1
for rep in 1 2; do
  echo "pairs" | tee -a $O/w4.txt; timeout -k 10 200 python tools/bench_wgrad_group.py 256 32 256 32 2>&1 | tail -2 | tee -a $O/w4.txt
  echo "no pairs" | tee -a $O/w4.txt; REED_WGRAD_W4_PAIRS=0 timeout -k 10 200 python tools/bench_wgrad_group.py 256 32 256 32 2>&1 | tail -2 | tee -a $O/w4.txt
done
REED_HIP_LIB=tools/_ab/libreed_clk.so timeout -k 10 200 python tools/_ab/clk_tn_w4.py 2>&1 | tail -7 | tee $O/clk.txt
echo done
