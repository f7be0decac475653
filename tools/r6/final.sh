#!/bin/bash
# Round 6: the final records of the tree: the default bench (the driver's command), then tools/profile.sh (PMC passes, kernel stats, timelines)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout -k 10 700 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { tail -20 $O/bench_n1.err; exit 1; }
cut -c1-400 $O/bench_n1.json
TAG=r6 timeout -k 10 450 bash tools/profile.sh
