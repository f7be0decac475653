#!/bin/bash
# round 4, twentieth lease: the vectorised reduce_mod_parts (test, time), rocprofv3 kernel stats of the DINOv2 ViT-L tower at batch 64
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4w
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu 2>&1 | tail -3 | tee $O/pytest.txt || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/tower --output-format csv -- python3 $R/tools/bench_tower.py dinov2-vit-l 64 > $O/tower64.json 2> $O/tower.err
cd $R
f=$(find $O/tower -name "*kernel_stats.csv" | head -1); cp $f $O/tower64_kernel_stats.csv; rm -rf $O/tower
cut -c1-200 $O/tower64.json
for rep in 1 2; do
  echo "bench b=256" | tee -a $O/bench.txt; timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-table --no-c3-leg --no-vae-leg --no-config-legs 2>&1 | tail -n 1 | cut -c1-140 | tee -a $O/bench.txt
done
echo done
