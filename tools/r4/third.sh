#!/bin/bash
# round 4, third lease: staggered forward + pipelined phase B / 16-byte dQ stores of the backward: tests, A/B timing, stamps
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4c
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_attention_gpu.py -q -x -m gpu 2>&1 | tail -15 > $O/pytest_attn.txt; rc=$?
echo "attention tests rc=$rc"; tail -5 $O/pytest_attn.txt
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  echo "fwd stag:"; timeout -k 10 120 python tools/time_attn.py 32 256 | tee -a $O/time_attn_stag.txt || exit 1
  echo "fwd nostag:"; REED_ATTN_FWD=nostag timeout -k 10 120 python tools/time_attn.py 32 256 | tee -a $O/time_attn_nostag.txt || exit 1
done
echo "fwd stamps (stag)"; REED_ATTN_FWD_DBG=32 timeout -k 10 120 python tools/r4/fwd_stamps.py 256 | tee -a $O/fwd_stamps.txt || exit 1
echo "fwd stamps (nostag)"; REED_ATTN_FWD=nostag REED_ATTN_FWD_DBG=32 timeout -k 10 120 python tools/r4/fwd_stamps.py 256 | tee -a $O/fwd_stamps.txt || exit 1
echo "bwd stamps"; REED_ATTN_KSP_DBG=4 timeout -k 10 120 python tools/r4/bwd_stamps.py 256 | tee -a $O/bwd_stamps.txt || exit 1
echo "bwd stamps b=32"; REED_ATTN_KSP_DBG=4 timeout -k 10 120 python tools/r4/bwd_stamps.py 32 | tee -a $O/bwd_stamps.txt || exit 1
echo done
