#!/bin/bash
# Round 6, lease 4: attention under a CU reserve (bits); where the items of the grouped weight gradients land when a second size runs
# in one process (stamps: XCC per blockIdx class, K-loop start / end)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6d
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_attention_gpu.py -x -q -k "cu_reserve" > $O/tests.txt 2>&1; tail -15 $O/tests.txt
echo "[$(date +%T)] stamps, two sizes in one process"
REED_HIP_LIB=tools/_ab/libreed_clk.so timeout -k 10 300 python tools/_ab/clk_tn_w4.py 64 256 > $O/stamps_64_256.txt 2>&1; cat $O/stamps_64_256.txt
KEEP=1 REED_HIP_LIB=tools/_ab/libreed_clk.so timeout -k 10 300 python tools/_ab/clk_tn_w4.py 64 256 > $O/stamps_64_256_keep.txt 2>&1; cat $O/stamps_64_256_keep.txt
echo "[$(date +%T)] done"
