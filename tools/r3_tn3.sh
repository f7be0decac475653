#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3n
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu -k "wgrad_group" 2>&1 | tail -8 | tee $O/pytest.txt
for rep in 1 2; do
  echo "tn3 (128x192, 3 per CU):"; timeout -k 10 200 python tools/bench_wgrad_group.py 256 128 64 32 2>&1 | cut -c1-60 | tee -a $O/tn3.txt
  echo "old (256x128, 2 per CU):"; REED_WGRAD_TN3=0 timeout -k 10 200 python tools/bench_wgrad_group.py 256 128 64 32 2>&1 | cut -c1-60 | tee -a $O/old.txt
done
echo done
