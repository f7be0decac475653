#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/encprof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/encprof -o enc -- python3 $R/tools/bench_encoder.py 256 --no-cpu > $O/enc_bench.json 2> $O/enc_bench.err
cd $R
find $O/encprof -name "*kernel_stats.csv" -exec cp {} $O/enc_kernel_stats.csv \;
rm -rf $O/encprof
tail -c 200 $O/enc_bench.json
