#!/bin/bash
# Round 6: the staggered deal: tests, launch times per setting (alternating), step A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6g
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gemm_gpu.py -x -q -k "staggered or column_split" > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -1 $O/tests.txt
for rep in 1 2; do for st in 0 1 2; do REED_W_STAGGER=$st timeout -k 10 200 python tools/r6/stagger.py 256 2>&1 | grep -v amdgpu.ids; done; done > $O/launch.txt 2>&1
cat $O/launch.txt
for rep in 1 2 3; do
  for st in 0 1; do
    echo "== REED_W_STAGGER=$st"
    REED_W_STAGGER=$st timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs --no-loss-vs-ref 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
  done
done > $O/step.txt 2>&1
cat $O/step.txt
