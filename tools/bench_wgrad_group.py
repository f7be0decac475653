#!/usr/bin/env python
"""The block's four weight gradients as one grouped launch: gemm_tn.hip's 256x128 / 128x256 tiles with two workgroups per CU
(force_tile 128) against gemm256w.hip's 256^2 tiles with four 128x128 waves (default), ms per launch.
usage: python tools/bench_wgrad_group.py [b ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reed_amd import ops  # noqa: E402

dev = torch.device("cuda")
D, Hm, T = 1152, 4608, 256
shapes = [(D, Hm), (Hm, D), (D, D), (3 * D, D)]
for b in [int(v) for v in sys.argv[1:]] or [256, 32]:
    M = b * T
    probs = []
    pad = int(os.environ.get("PAD", "0")) // 2     # PAD=<bytes>: operand j starts j * PAD bytes into its allocation (a multiple of 256)
    nobias = os.environ.get("NOBIAS", "0") == "1"  # NOBIAS=1: no bias gradients (the items' fourth waves only stage and wait)

    def operand(cols, j):
        t = (torch.randn(M * cols + 8 * pad, device=dev) * 0.05).to(torch.bfloat16)
        return t[j * pad:j * pad + M * cols].view(M, cols)

    for j, (n_out, k_in) in enumerate(shapes):
        dy, x = operand(n_out, 2 * j), operand(k_in, 2 * j + 1)
        out = torch.zeros(n_out * k_in + n_out, device=dev)
        probs.append((dy, x, out[:n_out * k_in].view(n_out, k_in), None if nobias else out[n_out * k_in:], n_out, k_in))
    flop = sum(2.0 * M * n * k for n, k in shapes)
    res = {}
    for tile in (128, 0):
        ops.gemm_force_tile(tile)
        for _ in range(3):
            ops.wgrad_group(probs, M)
        torch.cuda.synchronize()
        iters = int(os.environ.get("ITERS", "20"))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.wgrad_group(probs, M)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        res[tile] = (ms, [q[2].clone() if os.environ.get('SKIP_CLONE') != '1' else q[2] for q in probs])
    ops.gemm_force_tile(0)
    if os.environ.get("STAMPS") == "1":   # diagnostic build (REED_HIP_LIB=tools/_ab/libreed_clk.so): the last launch's K loops
        import ctypes, statistics
        from reed_amd import _lib
        rd = _lib.load("bf16").reed_clk_probe_read
        rd.restype = ctypes.c_int; rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
        buf = (ctypes.c_ulonglong * (8 * 4096))()
        assert rd(buf, 8 * 4096) == 0
        W = [[buf[8 * i + j] for j in range(8)] for i in range(2000, 2256) if buf[8 * i + 1] > 0]
        t00 = min(w[4] for w in W)
        for k in sorted(set(int(w[6]) for w in W)):
            ww = [w for w in W if int(w[6]) == k]
            print(f"   XCC {k}: {len(ww)} items, blockIdx % 8 = {sorted(set(int(w[7]) % 8 for w in ww))}, clock {statistics.median([w[0] / w[1] * 0.1 for w in ww]):.3f} GHz, "
                  f"{statistics.median([w[0] / w[2] for w in ww]):.0f} cycles per K-tile, K loops start {min((w[4] - t00) / 100 for w in ww):.0f}..{max((w[4] - t00) / 100 for w in ww):.0f} us, "
                  f"end {min((w[5] - t00) / 100 for w in ww):.0f}..{max((w[5] - t00) / 100 for w in ww):.0f} us")
    d = max((a_ - b_).abs().max().item() for a_, b_ in zip(res[128][1], res[0][1]))
    print(f"b={b}: 2 x 4-wave 256x128 tiles {res[128][0]:.4f} ms {flop / res[128][0] / 1e9:7.1f} TF | 4-wave 256^2 tiles {res[0][0]:.4f} ms "
          f"{flop / res[0][0] / 1e9:7.1f} TF | max |diff| {d:.2e}", flush=True)
