#!/bin/bash
# round 4, fifth lease: the two-workgroups-per-CU GEMM (gemm128c): tests with tile 129, the block's table against the default
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4e
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu -k "129" 2>&1 | tail -15 > $O/pytest_gemm129.txt; rc=$?
echo "gemm tests (tile 129) rc=$rc"; tail -6 $O/pytest_gemm129.txt
[ $rc -ne 0 ] && exit $rc
for t in 0 129 0 129; do
  echo "force tile $t:"; REED_FORCE_TILE=$t timeout -k 10 200 python tools/gemm_table.py 256 20 | tee -a $O/force_tile.txt || exit 1
done
for gm in 4 16; do
  echo "tile 129, group rows $gm:"; REED_GEMM128C_GM=$gm REED_FORCE_TILE=129 timeout -k 10 200 python tools/gemm_table.py 256 20 | tee -a $O/force_tile.txt || exit 1
done
echo "b=32:"; for t in 0 129; do REED_FORCE_TILE=$t timeout -k 10 200 python tools/gemm_table.py 32 50 | tee -a $O/force_tile_b32.txt || exit 1; done
echo done
