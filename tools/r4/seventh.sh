#!/bin/bash
# round 4, seventh lease: gemm128c with the operand stream / the MFMAs switched off; the towers with the split GELU epilogues
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4h
mkdir -p $O
cd $R
for d in 0 1 2; do
  echo "gemm128c dbg=$d:"; REED_GEMM128C_DBG=$d REED_FORCE_TILE=129 timeout -k 10 200 python tools/gemm_table.py 256 20 | tee -a $O/gemm128c_dbg.txt || exit 1
done
timeout -k 10 600 python -m pytest tests/test_encoder_gpu.py -q -x -m gpu 2>&1 | tail -5 | tee $O/pytest_enc.txt
for w in 0 1; do
  echo "towers, QGELU on the four-wave kernel = $w"
  REED_QGELU_W4=$w timeout -k 10 200 python tools/bench_encoder.py 2>&1 | tail -3 | tee -a $O/enc.txt
  REED_QGELU_W4=$w timeout -k 10 200 python tools/bench_tower.py dinov2-vit-l 64 2>&1 | tail -2 | tee -a $O/enc.txt
done
echo done
