#!/bin/bash
# usage: TAG=r5_full bash tools/full.sh — the whole GPU suite, smoke, the default bench (the driver's round-end sequence).  Progress goes to files under
# gpurun_out/ and a line per minute to stdout (a run that writes nothing for 7 minutes is taken to be hung).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG:-full}
mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
( timeout -k 10 1050 python -m pytest tests -v -x -m gpu -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; echo $? > $O/pytest_rc ) &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 45; echo "[$(date +%T)] $(grep -c -E 'PASSED|SKIPPED|FAILED' $O/pytest_gpu.txt) tests done"; done
rc=$(cat $O/pytest_rc)
echo "gpu tests rc=$rc"; tail -4 $O/pytest_gpu.txt
[ "$rc" != "0" ] && { grep -E "FAILED|Error" $O/pytest_gpu.txt | head -20; exit 1; }
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "bench:"; timeout -k 10 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; rc=$?
echo "bench rc=$rc"; cut -c1-400 $O/bench_n1.json
exit $rc
