#!/bin/bash
# Round 6: the ring backward's vector-memory issue spread over phase B's key steps (REED_ATTN_BWD_SPREAD=0: round 5's burst in front
# of the product): attention tests, launch times alternating, step A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6h
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_attention_gpu.py -x -q > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -1 $O/tests.txt
for rep in 1 2 3; do for sp in 0 1; do echo "REED_ATTN_BWD_SPREAD=$sp"; REED_ATTN_BWD_SPREAD=$sp timeout -k 10 200 python tools/time_attn.py 32 256 2>&1 | grep "b="; done; done > $O/launch.txt 2>&1
cat $O/launch.txt
for rep in 1 2 3; do
  for sp in 0 1; do
    echo "== REED_ATTN_BWD_SPREAD=$sp"
    REED_ATTN_BWD_SPREAD=$sp timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs --no-loss-vs-ref 2>/dev/null | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" || exit 1
  done
done > $O/step.txt 2>&1
cat $O/step.txt
