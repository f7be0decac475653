#!/bin/bash
# round 4, eighth lease: the ragged-M split of the GEMM dispatcher: tests, the towers at batch 64 with and without it
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4i
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py tests/test_encoder_gpu.py -q -x -m gpu -k "ragged_m or encoder or clip or tower or dinov2" 2>&1 | tail -5 | tee $O/pytest.txt
for sp in 1 0; do
  echo "split $sp"
  REED_GEMM_SPLIT_M=$sp timeout -k 10 200 python tools/bench_tower.py dinov2-vit-l 64 2>&1 | tail -1 | tee -a $O/enc.txt
  REED_GEMM_SPLIT_M=$sp timeout -k 10 200 python tools/bench_encoder.py 64 2>&1 | tail -1 | tee -a $O/enc.txt
  REED_GEMM_SPLIT_M=$sp timeout -k 10 200 python tools/bench_tower.py dinov2-vit-l 32 2>&1 | tail -1 | tee -a $O/enc.txt
done
echo done
