#!/bin/bash
# round 4, fortieth lease: the host's deal of the full tiles at b = 32 (whole step, per launch)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4W
mkdir -p $O
cd $R
for rep in 1 2 3; do
  for lib in tools/_ab/libreed_natural.so ""; do
    echo "b=32 lib=${lib:-product}" | tee -a $O/b32.txt; REED_HIP_LIB=$lib timeout -k 10 300 python bench.py --global-batch 32 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['roofline']['avg_ms_per_launch'], r['roofline']['frac'])" | tee -a $O/b32.txt
  done
done
for lib in tools/_ab/libreed_natural.so ""; do echo "per launch lib=${lib:-product}" | tee -a $O/b32.txt; REED_HIP_LIB=$lib timeout -k 10 200 python tools/bench_wgrad_group.py 256 32 64 32 2>&1 | tail -3 | tee -a $O/b32.txt; done
echo done
