#!/bin/bash
# round 4, thirty-third lease: compact XCD blocks of full tiles in the four-wave weight gradients: tests, time, traffic
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4N
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py -q -x -m gpu -k "wgrad" 2>&1 | tail -3 | tee $O/pytest.txt || exit 1
timeout -k 10 300 python tools/bench_wgrad_group.py 256 32 256 32 2>&1 | tail -4 | tee $O/w4.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/gemm_f --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/gemm_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/gemm_w --output-format csv -- python3 $R/tools/pmc_gemm.py > $O/gemm_w.log 2>&1
cd $R
for d in gemm_f gemm_w; do python tools/pmc_summary.py $O/$d tn_group; done | tee $O/pmc.txt
rm -rf $O/gemm_f $O/gemm_w
for rep in 1 2; do
  timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['avg_ms_per_launch'], r['roofline']['frac'])" | tee -a $O/bench.txt
done
echo done
