#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3j
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_attention_gpu.py -q -x -m gpu 2>&1 | tail -15 > $O/pytest_attn.txt; rc=$?; echo "attention tests rc=$rc"; tail -4 $O/pytest_attn.txt
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  echo "two-phase:"; REED_ATTN_BWD=2p timeout -k 10 120 python tools/time_attn.py 32 256 | tee -a $O/time_attn_2p.txt
  echo "persistent, tiles:"; REED_ATTN_RING=0 timeout -k 10 120 python tools/time_attn.py 32 256 | tee -a $O/time_attn_ksp.txt
  echo "persistent, rings:"; timeout -k 10 120 python tools/time_attn.py 32 256 | tee -a $O/time_attn_ring.txt
done
for d in 1 2 3; do echo "ring dbg=$d"; REED_ATTN_KSP_DBG=$d timeout -k 10 120 python tools/time_attn.py 256 | tee -a $O/time_dbg.txt; done
echo done
