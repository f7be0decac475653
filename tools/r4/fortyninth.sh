#!/bin/bash
# round 4: the data-parallel path at world 1, b = 32 per GPU: torch's RCCL process group against the library's own communicator
# (REED_COMM=native) — per-bucket host cost at the shape where the host is closest to being the limit; plain plan only
set -e
mkdir -p gpurun_out/r4S
B="python bench.py --global-batch 32 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-table --no-config-legs --no-vae-leg --no-c3-leg"
for rep in 1 2; do
for c in torch native; do
  env REED_FORCE_REDUCER=1 REED_BENCH_TUNED=0 REED_COMM=$c timeout -k 10 110 $B > gpurun_out/r4S/$c$rep.json 2> gpurun_out/r4S/$c$rep.err || echo "bench $c failed"
  python - $c$rep <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r4S/{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], "images/s", d["ms_per_step"], "ms |", d["plans"]["plain"]["plan"], flush=True)
PY
done; done
env timeout -k 10 110 $B > gpurun_out/r4S/single.json 2> gpurun_out/r4S/single.err
python -c "
import json; d = json.loads(open('gpurun_out/r4S/single.json').read().strip().splitlines()[-1]); print('single GPU path', d['value'], 'images/s', d['ms_per_step'], 'ms')"
