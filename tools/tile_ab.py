#!/usr/bin/env python
"""A/B of the forward / dgrad block GEMMs on the heuristic's kernel against the 128^2 and 256^2 kernels forced, through
bench.time_gemms (product entry points, fused epilogues).  usage: python tools/tile_ab.py [b ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from reed_amd import ops
for b in [int(x) for x in sys.argv[1:]] or [256]:
    res = {}
    for tile in (0, 128, 256, 144, 0, 128, 256, 144):
        ops.gemm_force_tile(tile)
        for r in bench.time_gemms(b):
            if "wgrad" not in r["kernel"]:
                res.setdefault(r["kernel"], {}).setdefault(tile, []).append(r["ms"])
    ops.gemm_force_tile(0)
    print(f"b={b}")
    for k, v in res.items():
        print(f"   {k:24s} heuristic {min(v[0]):.4f}   128^2 {min(v[128]):.4f}   256^2 {min(v[256]):.4f}   256x144 {min(v[144]):.4f} ms")
