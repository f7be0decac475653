#!/bin/bash
# round 4, twenty-first lease: the skinny kernel for the tails of the ragged-M split: GEMM + encoder tests, towers with / without it
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4x
mkdir -p $O
cd $R
(timeout -k 10 1000 python -m pytest tests/test_gemm_gpu.py tests/test_encoder_gpu.py -q -x -m gpu > $O/pytest_full.txt 2>&1; echo "rc=$?" >> $O/pytest_full.txt) &
PID=$!
while kill -0 $PID 2>/dev/null; do sleep 30; echo "[$(date +%T)] tests running: $(tail -c 80 $O/pytest_full.txt | tr '\n' ' ')"; done
tail -n 4 $O/pytest_full.txt
grep -q "rc=0" $O/pytest_full.txt || exit 1
for rep in 1 2; do
  for sk in 0 1; do
    for enc in "dinov2-vit-l 64" "dinov2-vit-l 256"; do
      echo "REED_GEMM_SKINNY=$sk $enc" | tee -a $O/towers_skinny.txt
      REED_GEMM_SKINNY=$sk timeout -k 10 300 python tools/bench_tower.py $enc 2>&1 | tail -n 1 | cut -c1-330 | tee -a $O/towers_skinny.txt
    done
  done
done
echo done
