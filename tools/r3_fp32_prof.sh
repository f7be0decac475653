#!/bin/bash
# fp32-operand build: where an XL/2 train step spends its time (b = 32)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/f32prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/f32prof -o f32 -- python3 $R/bench.py --mixed-precision fp32 --global-batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs > $O/f32_bench.json 2> $O/f32_bench.err
cd $R
find $O/f32prof -name "*kernel_stats.csv" -exec cp {} $O/f32_kernel_stats.csv \;
rm -rf $O/f32prof
tail -c 400 $O/f32_bench.json
