#!/bin/bash
# round 4: the whole GPU suite, smoke, the default bench (the driver's round-end sequence)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4g
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -q -x -m gpu 2>&1 | tail -25 > $O/pytest_gpu.txt; rc=$?
echo "gpu tests rc=$rc"; tail -8 $O/pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; rc=$?
echo "bench rc=$rc"; cut -c1-400 $O/bench_n1.json
exit $rc
