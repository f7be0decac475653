#!/bin/bash
# Round 3: the key-stationary attention backward, persistent form — tests, then same-box A/B against the one-shot key-stationary
# kernel (REED_ATTN_BWD=ks1) and the two-phase kernel (REED_ATTN_BWD=2p).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_attention_gpu.py -q -x -m gpu 2>&1 | tail -15 > $O/pytest_attn.txt; rc=$?; echo "attention tests rc=$rc"; tail -6 $O/pytest_attn.txt
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  echo "two-phase:"; REED_ATTN_BWD=2p timeout -k 10 120 python tools/time_attn.py 32 64 256 | tee -a $O/time_attn_2p.txt
  echo "key-stationary one-shot:"; REED_ATTN_BWD=ks1 timeout -k 10 120 python tools/time_attn.py 32 64 256 | tee -a $O/time_attn_ks1.txt
  echo "key-stationary persistent:"; timeout -k 10 120 python tools/time_attn.py 32 64 256 | tee -a $O/time_attn_ksp.txt
done
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -q -x -m gpu -k "tiny or c2_xl2 or full_size or side_stream" 2>&1 | tail -25 > $O/pytest_model.txt; echo "model tests rc=$?"; tail -5 $O/pytest_model.txt
echo done
