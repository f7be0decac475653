#!/bin/bash
# Round 6, lease 3: the half-batch pipeline gate; the weight gradients without bias gradients (what the fourth waves cost) and with
# operands at staggered addresses (the b = 128 launch time seen after other allocations)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6c
mkdir -p $O
cd $R
echo "[$(date +%T)] half-batch pipeline"
timeout -k 10 900 python tools/r6/half_batch.py > $O/half_batch.txt 2>&1 || { tail -30 $O/half_batch.txt; exit 1; }
cat $O/half_batch.txt
echo "[$(date +%T)] wgrad: padding / no bias"
for b in 256 128; do
  for pad in 0 4352 69632; do
    NOBIAS=0 PAD=$pad timeout -k 10 300 python tools/bench_wgrad_group.py 64 $b 2>&1 | grep "b=$b" | sed "s/^/pad $pad: /"
  done
done > $O/wgrad_pad.txt 2>&1
NOBIAS=1 timeout -k 10 300 python tools/bench_wgrad_group.py 256 2>&1 | grep "b=" | sed "s/^/no bias gradients: /" >> $O/wgrad_pad.txt
cat $O/wgrad_pad.txt
echo "[$(date +%T)] done"
