#!/bin/bash
# round 4: last check of the final tree: GEMM + model + CLI tests, smoke
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r4U
mkdir -p $O
cd $R
( timeout -k 10 1000 python -m pytest tests/test_gemm_gpu.py tests/test_model_gpu.py tests/test_cli_gpu.py -q -x -m gpu -p no:cacheprovider > $O/pytest.txt 2>&1; echo $? > $O/rc ) &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 45; echo "[$(date +%T)] $(tail -c 60 $O/pytest.txt | tr '\n' ' ')"; done
tail -n 3 $O/pytest.txt
[ "$(cat $O/rc)" != "0" ] && exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo done
