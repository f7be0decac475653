#!/bin/bash
# round 4: a one-GPU rehearsal of the data-parallel step's kernel forms — bench.py through its multi-GPU path at world 1
# (REED_FORCE_REDUCER=1: RCCL reducer, backward with ops.set_concurrent_comm) while ANOTHER PROCESS holds 16 CUs the whole time
# (tools/_ab/hog_main.hip) — with the forms the library now selects beside a collective, and with the ones it selected before
# (REED_COMM_FORMS=0: the static weight-gradient launch, one attention-backward workgroup per CU, persistent four-wave GEMMs; the
# first pass of this script had REED_WGRAD_W4=1 REED_ATTN_BWD_GRID=1 instead: 1038 against 1103-1105 images/s).
# The stand-in also sits beside the forward (RCCL would not): absolute numbers mean little, the A/B is what counts.
set -e
mkdir -p gpurun_out/r4W
B="python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-table --no-config-legs --no-vae-leg --no-c3-leg"
run() {  # name, hog CUs per stream (0 = none), env...
  name=$1; n=$2; shift 2
  if [ "$n" != 0 ]; then timeout -k 5 170 tools/_ab/hog_main $n 140 2> gpurun_out/r4W/hog_$name.txt & hp=$!; sleep 2; fi
  env REED_FORCE_REDUCER=1 "$@" timeout -k 10 160 $B > gpurun_out/r4W/$name.json 2> gpurun_out/r4W/$name.err || echo "bench $name failed"
  if [ "$n" != 0 ]; then kill $hp 2>/dev/null || true; wait $hp 2>/dev/null || true; fi
  python - "$name" <<'PY'
import json, sys
name = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/r4W/{name}.json").read().strip().splitlines()[-1])
    p = d.get("plans", {})
    print(name, "value", d["value"], "ms", d["ms_per_step"], "| plain", (p.get("plain") or {}).get("value"), "| tuned", (p.get("tuned") or {}).get("value"),
          (p.get("tuned") or {}).get("cu_reserve"), flush=True)
except Exception as e:
    print(name, "no record:", repr(e)[:200], flush=True)
PY
}
run alone_now 0
run alone_static 0 REED_COMM_FORMS=0
run held_now 8
run held_static 8 REED_COMM_FORMS=0
run held_now2 8
run held_static2 8 REED_COMM_FORMS=0
