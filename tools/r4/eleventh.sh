#!/bin/bash
# round 4, eleventh lease: the gate vector hoisted out of the gate + residual epilogue (operand loads 8 groups deep): tests, A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4l
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu -k "epilogues or many_tiles" 2>&1 | tail -4 | tee $O/pytest.txt
for rep in 1 2 3; do
  for lib in tools/_ab/libreed_nohoist.so tools/_ab/libreed_pfh6.so ""; do
    echo "lib=${lib:-product (hoisted, 8 deep)}"; REED_HIP_LIB=$lib timeout -k 10 200 python tools/gemm_table.py 256 20 | tee -a $O/hoist_ab.txt || exit 1
  done
done
for lib in tools/_ab/libreed_nohoist.so "" tools/_ab/libreed_nohoist.so ""; do
  echo "bench lib=${lib:-product}"; REED_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>/dev/null | cut -c1-140 | tee -a $O/hoist_bench.txt
done
echo done
