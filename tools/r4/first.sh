#!/bin/bash
# round 4, first lease: the new forward (tests, A/B against round 2's, the diagnosis switches), the GEMM phase stagger
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4a
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_attention_gpu.py -q -x -m gpu 2>&1 | tail -15 > $O/pytest_attn.txt; rc=$?
echo "attention tests rc=$rc"; tail -5 $O/pytest_attn.txt
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  echo "fwd r4:"; timeout -k 10 120 python tools/time_attn.py 32 256 | tee -a $O/time_attn_r4.txt || exit 1
  echo "fwd r2:"; REED_ATTN_FWD=r2 timeout -k 10 120 python tools/time_attn.py 32 256 | tee -a $O/time_attn_r2.txt || exit 1
done
for d in 1 2 4 7 8 16 24 31; do
  echo "fwd dbg=$d"; REED_ATTN_FWD_DBG=$d timeout -k 10 120 python tools/time_attn.py 256 | tee -a $O/time_fwd_dbg.txt || exit 1
done
for us in 0 6 12 18 0 12; do
  echo "stagger $us us:"; REED_GEMM_STAGGER_US=$us timeout -k 10 200 python tools/gemm_table.py 256 20 | tee -a $O/stagger.txt || exit 1
done
echo done
