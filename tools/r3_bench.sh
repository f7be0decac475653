#!/bin/bash
# Round 3: step-level bench at b = 256 (with the C3 per-GPU leg) — same box, attention backward two-phase vs persistent.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3e
mkdir -p $O
cd $R
for tag in ksp 2p ksp 2p; do
  if [ $tag = 2p ]; then export REED_ATTN_BWD=2p; else unset REED_ATTN_BWD; fi
  timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table > $O/bench_$tag.json 2>> $O/bench.err
  python3 -c "
import json
d=json.load(open('$O/bench_$tag.json'))
print('$tag', d['value'], d['ms_per_step'], d['step_mfma_frac'], 'c3', d['c3_per_gpu_leg']['images_per_sec_per_gpu'], d['c3_per_gpu_leg']['ms_per_step'])"
done
echo done
