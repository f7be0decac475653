#!/bin/bash
# Round 6, lease 6: the scratch-free weight-gradient kernel (direct AGPR stores, role-local bias sums): tests, launch times incl. the
# second-size case, stamps, step A/B against round 5's build
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6f
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gemm_gpu.py -x -q -k "wgrad_group" > $O/tests.txt 2>&1 || { tail -40 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
for lib in tools/_ab/libreed_r5.so ""; do echo "== lib=${lib:-current}"; REED_HIP_LIB=$lib timeout -k 10 300 python tools/bench_wgrad_group.py 256 128 64 32 2>&1 | grep "b="; done > $O/ab_launch.txt 2>&1
cat $O/ab_launch.txt
REED_HIP_LIB=tools/_ab/libreed_clk.so timeout -k 10 300 python tools/_ab/clk_tn_w4.py 256 > $O/stamps.txt 2>&1; grep -v amdgpu $O/stamps.txt | head -14
for rep in 1 2; do
  for lib in tools/_ab/libreed_r5.so ""; do
    echo "== lib=${lib:-current}"
    REED_HIP_LIB=$lib timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs --no-loss-vs-ref 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); r = d['roofline']; print(d['value'], d['ms_per_step'], r['avg_ms_per_launch'], r['frac'])" || exit 1
  done
done > $O/ab_step.txt 2>&1
cat $O/ab_step.txt
