#!/bin/bash
# C5: per-kernel time of the sampler's evaluations (n = 32 -> batch 64, fp16 operands)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/c5prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5prof -o c5 -- python3 $R/tools/bench_generate.py 32 8 fp16 > $O/c5_bench.json 2> $O/c5_bench.err
cd $R
find $O/c5prof -name "*kernel_stats.csv" -exec cp {} $O/c5_kernel_stats.csv \;
rm -rf $O/c5prof
tail -c 300 $O/c5_bench.json
