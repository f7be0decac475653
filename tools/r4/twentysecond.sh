#!/bin/bash
# round 4, twenty-second lease: GELU(erf) from the A&S form: encoder tests, DINOv2 ViT-L with the fc1 GEMM on the 8-wave / 4-wave kernel, CLIP ViT-L
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4y
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_encoder_gpu.py -q -x -m gpu 2>&1 | tail -3 | tee $O/pytest.txt || exit 1
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu -k "erf or gelu or epilogue" 2>&1 | tail -3 | tee -a $O/pytest.txt || exit 1
for rep in 1 2; do
  for w4 in 0 1; do
    echo "REED_QGELU_W4=$w4 dinov2-vit-l 64" | tee -a $O/towers_erf.txt
    REED_QGELU_W4=$w4 timeout -k 10 300 python tools/bench_tower.py dinov2-vit-l 64 2>&1 | tail -n 1 | cut -c1-330 | tee -a $O/towers_erf.txt
    echo "REED_QGELU_W4=$w4 dinov2-vit-l 256" | tee -a $O/towers_erf.txt
    REED_QGELU_W4=$w4 timeout -k 10 300 python tools/bench_tower.py dinov2-vit-l 256 2>&1 | tail -n 1 | cut -c1-330 | tee -a $O/towers_erf.txt
  done
  echo "clip 64" | tee -a $O/towers_erf.txt; timeout -k 10 300 python tools/bench_encoder.py 64 --no-cpu 2>&1 | tail -n 1 | cut -c1-330 | tee -a $O/towers_erf.txt
  echo "clip 256" | tee -a $O/towers_erf.txt; timeout -k 10 300 python tools/bench_encoder.py 256 --no-cpu 2>&1 | tail -n 1 | cut -c1-330 | tee -a $O/towers_erf.txt
done
echo done
