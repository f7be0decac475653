#!/bin/bash
# A/B on one box: delta from the dgrad epilogue (REED_ATTN_DP=1, default) vs the row kernel (0), b = 256 and b = 128
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for gb in 256 128; do
for v in 1 0 1 0; do
  REED_ATTN_DP=$v timeout -k 10 300 python bench.py --steps 15 --warmup 4 --global-batch $gb --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b=$gb DP=$v', d['value'], d['ms_per_step'])"
done; done
