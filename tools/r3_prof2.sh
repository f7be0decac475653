#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3g
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof32 --output-format csv -- python3 $R/bench.py --steps 6 --warmup 2 --global-batch 32 --no-cpu-baseline --no-kernel-table > $O/b32.json 2> $O/rocprof32.err
cd $R
t=$(find $O/prof32 -name "*kernel_trace.csv" | head -1)
python tools/step_dump.py $t 2 > $O/step_b32.txt
rm -rf $O/prof32
grep -n "copyBuffer" $O/step_b32.txt | head
wc -l $O/step_b32.txt
