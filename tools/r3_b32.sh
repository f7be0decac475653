#!/bin/bash
# Round 3: the step boundary at b = 32 (optimiser chunks beside the next forward): same-box variants.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3h
mkdir -p $O
cd $R
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 4 --global-batch 32 --no-cpu-baseline --no-kernel-table > $O/b32_$tag.json 2>> $O/err.txt; python3 -c "
import json
d=json.load(open('$O/b32_$tag.json')); print('$tag', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run base X=1
run prio REED_MAIN_PRIO=1
run adam2048 REED_ADAM_BLOCKS=2048
run adam1024 REED_ADAM_BLOCKS=1024
run noovl REED_OPT_OVERLAP=0
run prio_adam1024 REED_MAIN_PRIO=1 REED_ADAM_BLOCKS=1024
done
echo done
