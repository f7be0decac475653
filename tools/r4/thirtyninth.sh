#!/bin/bash
# round 4, thirty-ninth lease: bias-gradient tiles on the even XCDs: tests, per-XCC stamps, whole step against the compact problem-order runs
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4V
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu -k "wgrad" 2>&1 | tail -3 | tee $O/pytest.txt || exit 1
REED_HIP_LIB=tools/_ab/libreed_clk.so timeout -k 10 200 python tools/_ab/clk_tn_w4.py 2>&1 | tail -9 | cut -c1-200 | tee $O/clk.txt
for rep in 1 2 3; do
  for lib in tools/_ab/libreed_natural.so ""; do
    echo "lib=${lib:-product}" | tee -a $O/bench.txt; REED_HIP_LIB=$lib timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['avg_ms_per_launch'], r['roofline']['frac'])" | tee -a $O/bench.txt
  done
done
echo done
