#!/bin/bash
# round 4, second lease: forward phase stamps, new tests (attention item loops, embed tails, bench plain-then-tuned)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4b
mkdir -p $O
cd $R
for d in 32 56; do
  echo "stamps dbg=$d"; REED_ATTN_FWD_DBG=$d timeout -k 10 120 python tools/r4/fwd_stamps.py 256 | tee -a $O/fwd_stamps.txt || exit 1
done
timeout -k 10 900 python -m pytest tests/test_attention_gpu.py tests/test_kernels_gpu.py -q -x -m gpu 2>&1 | tail -15 > $O/pytest_a.txt; rc=$?
echo "attention + kernel tests rc=$rc"; tail -5 $O/pytest_a.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 1000 python -m pytest tests/test_cli_gpu.py -q -x -m gpu -k "bench" 2>&1 | tail -30 > $O/pytest_b.txt; rc=$?
echo "bench tests rc=$rc"; tail -12 $O/pytest_b.txt
exit $rc
