#!/usr/bin/env python
"""Print bench.py's per-shape GEMM table (the block's 12 launches through the engine's entry points) for A/B runs under
environment switches. usage: python tools/gemm_table.py [b] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
it = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if os.environ.get("REED_FORCE_TILE"):   # 128 / 144 / 256 / 257 / 258: ops.gemm_force_tile (A/B of the tile kernels)
    from reed_amd import ops
    ops.gemm_force_tile(int(os.environ["REED_FORCE_TILE"]))
rows = bench.time_gemms(b, iters=it)
print(" | ".join(f"{r['kernel'].split()[0][0]}{r['kernel'].split()[1]}:{r['ms']:.3f}" for r in rows), f"| sum {sum(r['ms'] for r in rows):.3f}")
