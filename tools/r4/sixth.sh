#!/bin/bash
# round 4, sixth lease: pipelined phase A of the attention backward (tests, timing, stamps), the b = 256 plan tie, row kernels
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4f
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_attention_gpu.py -q -x -m gpu 2>&1 | tail -15 > $O/pytest_attn.txt; rc=$?
echo "attention tests rc=$rc"; tail -5 $O/pytest_attn.txt
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do timeout -k 10 120 python tools/time_attn.py 32 256 | tee -a $O/time_attn.txt || exit 1; done
echo "bwd stamps"; REED_ATTN_KSP_DBG=4 timeout -k 10 120 python tools/r4/bwd_stamps.py 256 | tee -a $O/bwd_stamps.txt || exit 1
timeout -k 10 120 python tools/time_rows.py 256 | tee $O/time_rows.txt || exit 1
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -q -x -m gpu -s -k "plan_matches or tile_choice" 2>&1 | tail -15 > $O/pytest_plan.txt; rc=$?
echo "plan tests rc=$rc"; tail -8 $O/pytest_plan.txt
exit $rc
