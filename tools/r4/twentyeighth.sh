#!/bin/bash
# round 4, twenty-eighth lease: the four-wave TN form (weight gradients) with its K-tile buffers as a ring of four 32-row slices: tests, time per launch
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4H
mkdir -p $O
cd $R
REED_WGRAD_W4=1 timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu -k "wgrad" 2>&1 | tail -3 | tee $O/pytest.txt || exit 1
for rep in 1 2; do
  echo "REED_WGRAD_W4=1 ring (product build)" | tee -a $O/w4ring.txt; REED_WGRAD_W4=1 timeout -k 10 200 python tools/bench_wgrad_group.py 256 32 2>&1 | tail -4 | tee -a $O/w4ring.txt
  echo "REED_WGRAD_W4=1 two-phase loop (libreed_noring)" | tee -a $O/w4ring.txt; REED_WGRAD_W4=1 REED_HIP_LIB=tools/_ab/libreed_noring.so timeout -k 10 200 python tools/bench_wgrad_group.py 256 32 2>&1 | tail -4 | tee -a $O/w4ring.txt
done
echo done
