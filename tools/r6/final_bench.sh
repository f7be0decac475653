#!/bin/bash
# Round 6: the default bench of the final tree (the driver's command) + the new record's test
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r6
mkdir -p $O
cd $R
timeout -k 10 200 python -m pytest tests/test_cli_gpu.py -x -q -k "attention_record" 2>&1 | tail -1
timeout -k 10 700 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || { tail -20 $O/bench_n1.err; exit 1; }
python -c "
import json; d = json.loads(open('$O/bench_n1.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['step_mfma_frac'], d['roofline']['frac'], d['roofline']['avg_ms_per_launch'], d['c3_per_gpu_leg']['images_per_sec_per_gpu'], json.dumps(d['attention']))"
