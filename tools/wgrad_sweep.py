#!/usr/bin/env python
"""Sweep tile x split-K x prefetch for the four SiT-XL/2 block weight gradients (TN GEMM + slab reduce) at batch b.
usage: python tools/wgrad_sweep.py [b]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reed_amd import _lib, ops
L = _lib.load(); dev = torch.device("cuda"); b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = b * 256
shapes = {"qkv": (3456, 1152), "proj": (1152, 1152), "fc1": (4608, 1152), "fc2": (1152, 4608)}
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
def timeit(fn, it=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for name, (N, K) in shapes.items():
    dy, x = bf(M, N), bf(M, K)
    out = torch.zeros(N, K, device=dev); ref = None
    ws = torch.empty(8 * (N * K + N) + 64, device=dev)
    fl = 2.0 * M * N * K
    res = []
    for tile in (128, 256):
        for split in (1, 2, 3, 4, 6, 8):
            for pf in (0,):
                L.reed_gemm_force_tile(tile)
                f = lambda: ops.linear_wgrad(dy, x, out, split_k=split, ws=ws, Mtok=M, N=N, K=K)
                ms = timeit(f)
                if ref is None: ref = out.clone()
                err = (out - ref).abs().max().item()
                res.append((ms, tile, split, pf, err))
    res.sort()
    print(f"{name} N={N} K={K} M={M}: best " + " | ".join(f"t{t} s{s} pf{p} {ms:.3f}ms {fl/ms/1e9:.0f}TF e={e:.1e}" for ms, t, s, p, e in res[:6]), flush=True)
    print("    worst " + " | ".join(f"t{t} s{s} pf{p} {ms:.3f}ms" for ms, t, s, p, e in res[-3:]), flush=True)
L.reed_gemm_force_tile(0)
