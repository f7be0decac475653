#!/bin/bash
# Rehearsal of the N > 1 bench command on the one-GPU box (ranks share the device, gloo collectives): XL/2, the default workload
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
export REED_BENCH_REHEARSE=gloo
( time timeout -k 10 500 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-table > $O/reh_n2.json 2> $O/reh_n2.err ) 2> $O/reh_n2.time
echo "n2 rc=$?" ; tail -c 600 $O/reh_n2.json; tail -3 $O/reh_n2.time
( time REED_COMM_CUS=0 timeout -k 10 500 python bench.py --gpus 4 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-table > $O/reh_n4.json 2> $O/reh_n4.err ) 2> $O/reh_n4.time
echo "n4 rc=$?" ; tail -c 600 $O/reh_n4.json; tail -3 $O/reh_n4.time
