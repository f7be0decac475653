#!/bin/bash
# round 4, thirty-first lease: four-wave TN form with the excess ragged tiles cut in two along K: tests, time per launch with / without the split
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4K
mkdir -p $O
cd $R
REED_WGRAD_W4=1 timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu -k "wgrad" 2>&1 | tail -3 | tee $O/pytest.txt || exit 1
for rep in 1 2; do
  echo "split" | tee -a $O/w4.txt; REED_WGRAD_W4=1 timeout -k 10 200 python tools/bench_wgrad_group.py 256 2>&1 | tail -1 | tee -a $O/w4.txt
  echo "no split" | tee -a $O/w4.txt; REED_WGRAD_W4=1 REED_WGRAD_W4_SPLIT=0 timeout -k 10 200 python tools/bench_wgrad_group.py 256 2>&1 | tail -1 | tee -a $O/w4.txt
done
REED_WGRAD_W4=1 REED_HIP_LIB=tools/_ab/libreed_clk.so timeout -k 10 200 python tools/_ab/clk_tn_w4.py 2>&1 | tail -5 | tee $O/clk.txt
echo done
