#!/bin/bash
# round 4, tenth lease: where the gate + residual epilogue's time is: the block's GEMM table on builds whose epilogue drops its
# operand loads (libreed_noload) / its stores (libreed_nostore), against the product build
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4k
mkdir -p $O
cd $R
for rep in 1 2; do
  for lib in "" tools/_ab/libreed_noload.so tools/_ab/libreed_nostore.so; do
    echo "lib=${lib:-product}"; REED_HIP_LIB=$lib timeout -k 10 200 python tools/gemm_table.py 256 20 | tee -a $O/epi_diag.txt || exit 1
  done
done
echo done
