#!/bin/bash
# round 4, thirtieth lease: four-wave TN ring with XOR fragment addresses and the bias-gradient copy of the loop: tests, stamps, time per launch
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4J
mkdir -p $O
cd $R
REED_WGRAD_W4=1 timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu -k "wgrad" 2>&1 | tail -3 | tee $O/pytest.txt || exit 1
REED_WGRAD_W4=1 REED_HIP_LIB=tools/_ab/libreed_clk.so timeout -k 10 200 python tools/_ab/clk_tn_w4.py 2>&1 | tail -5 | tee $O/clk.txt
REED_WGRAD_W4=1 timeout -k 10 200 python tools/bench_wgrad_group.py 256 2>&1 | tail -2 | tee $O/w4.txt
echo done
