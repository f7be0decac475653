#!/bin/bash
# round 4, fourteenth lease: counted waits at a persistent tile's start (the previous epilogue's stores drain under K-tile 0): tests, A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4p
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu 2>&1 | tail -3 | tee $O/pytest.txt || exit 1
for rep in 1 2 3; do
  for lib in tools/_ab/libreed_drain.so ""; do
    echo "lib=${lib:-product (counted waits)}" | tee -a $O/drain_ab.txt; REED_HIP_LIB=$lib timeout -k 10 200 python tools/gemm_table.py 256 20 | tee -a $O/drain_ab.txt || exit 1
  done
done
echo done
