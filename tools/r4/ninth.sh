#!/bin/bash
# round 4, ninth lease: the row kernel for T = 257 / 261: tests, the towers with and without it
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4j
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_attention_gpu.py tests/test_encoder_gpu.py -q -x -m gpu -k "attention_fwd or encoder or clip or tower or dinov2" 2>&1 | tail -5 | tee $O/pytest.txt
for tl in 1 0; do
  echo "tail kernel $tl"
  REED_ATTN_TAIL=$tl timeout -k 10 200 python tools/bench_tower.py dinov2-vit-l 64 2>&1 | tail -1 | tee -a $O/enc.txt
  REED_ATTN_TAIL=$tl timeout -k 10 200 python tools/bench_encoder.py 64 2>&1 | tail -1 | tee -a $O/enc.txt
  REED_ATTN_TAIL=$tl timeout -k 10 200 python tools/bench_encoder.py 256 2>&1 | tail -1 | tee -a $O/enc.txt
done
echo done
