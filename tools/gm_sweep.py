#!/usr/bin/env python
"""gemm256 tile-walk sweep: the block's forward / dgrad GEMMs (bench.time_gemms) under REED_GEMM256_GM = 1, 2, 4, 8, 16
(tile rows per XCD-local group; one process per value: the override is read once).  usage: python tools/gm_sweep.py [b]"""
import json, os, subprocess, sys
b = sys.argv[1] if len(sys.argv) > 1 else "256"
code = ("import sys, json; sys.path.insert(0, '.'); import bench; from reed_amd import ops; ops.gemm_force_tile(256); "
        f"print(json.dumps([r for r in bench.time_gemms({b}) if 'wgrad' not in r['kernel']]))")
res = {}
for gm in (4, 1, 2, 8, 16, 4):
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, REED_GEMM256_GM=str(gm)), capture_output=True, text=True)
    rows = json.loads(out.stdout.strip().splitlines()[-1])
    for r in rows:
        res.setdefault(r["kernel"], {}).setdefault(gm, []).append(r["ms"])
print(f"b={b}  (ms per launch; GM = tile rows per group)")
for k, v in res.items():
    print(f"  {k:24s} " + "  ".join(f"GM{gm}: {min(t):.4f}" for gm, t in sorted(v.items())))
