#!/bin/bash
# round 4: the one-GPU rehearsal of fortysixth.sh at the per-GPU batch of the 8-GPU run (b = 32): bench.py through its multi-GPU path at
# world 1 (GB=<per-GPU batch>, default 32; ST=<steps>), alone and while another process holds 16 CUs, with and without the kernel forms selected beside collectives
set -e
export GB=${GB:-32}
mkdir -p gpurun_out/r4T${GB:-32}
B="python bench.py --global-batch ${GB:-32} --steps ${ST:-20} --warmup 5 --no-cpu-baseline --no-kernel-table --no-config-legs --no-vae-leg --no-c3-leg"
run() {  # name, hog CUs per stream (0 = none), env...
  name=$1; n=$2; shift 2
  if [ "$n" != 0 ]; then timeout -k 5 120 tools/_ab/hog_main $n 100 2> gpurun_out/r4T${GB:-32}/hog_$name.txt & hp=$!; sleep 2; fi
  env REED_FORCE_REDUCER=1 "$@" timeout -k 10 110 $B > gpurun_out/r4T${GB:-32}/$name.json 2> gpurun_out/r4T${GB:-32}/$name.err || echo "bench $name failed"
  if [ "$n" != 0 ]; then kill $hp 2>/dev/null || true; wait $hp 2>/dev/null || true; fi
  python - "$name" <<'PY'
import json, os, sys
name = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r4T{os.environ.get('GB', '32')}/{name}.json").read().strip().splitlines()[-1])
    pl = d["plans"]
    print(f"{name:14s} plain plan {pl['plain']['images_per_sec']:8.1f} images/s | tuned plan {pl['tuned'].get('images_per_sec')} "
          f"({pl['tuned'].get('plan', '').split('buckets, ')[-1].replace(', replicated optimiser pass', '')}) | tuner ms per step "
          f"{d['data_parallel'].get('cu_reserve_tuning_ms')}", flush=True)
except Exception as e:
    print(name, "no record:", repr(e)[:200], flush=True)
PY
}
run alone_now 0
run alone_static 0 REED_COMM_FORMS=0
run held_now 8
run held_static 8 REED_COMM_FORMS=0
