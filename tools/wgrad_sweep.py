#!/usr/bin/env python
"""Sweep tile x split-K for the four SiT-XL/2 block weight gradients (TN GEMM + slab reduce, bias gradient fused for
qkv / fc1) at local batch b: the 128^2 kernel (gemm.hip) against gemm_tn.hip's 256x128 and 128x256 tiles; every
variant is checked against the 128^2 split-1 result.   usage: python tools/wgrad_sweep.py [b]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reed_amd import ops
dev = torch.device("cuda"); b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = b * 256
shapes = {"qkv": (3456, 1152, True), "proj": (1152, 1152, False), "fc1": (4608, 1152, True), "fc2": (1152, 4608, False)}
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
def timeit(fn, it=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
names = {ops.TN: "128x128", ops.TN_TALL: "256x128", ops.TN_WIDE: "128x256"}
for name, (N, K, with_bias) in shapes.items():
    dy, x = bf(M, N), bf(M, K)
    out = torch.zeros(N * K + N, device=dev)
    dw, db = out[:N * K], out[N * K:]
    ws = torch.empty(12 * (N * K + N) + 64, device=dev)
    fl = 2.0 * M * N * K
    ops.linear_wgrad(dy, x, dw, dbias=db if with_bias else None, split_k=1, Mtok=M, N=N, K=K, ws=ws)
    ref = out.clone()
    res = []
    for lay in (ops.TN, ops.TN_TALL, ops.TN_WIDE):
        if lay == ops.TN_WIDE and K % 256:
            continue
        for split in (1, 2, 3, 4, 5, 6, 8):
            if (M // 64) // split < 16:
                continue
            f = lambda: ops.linear_wgrad(dy, x, dw, dbias=db if with_bias else None, split_k=split, ws=ws, Mtok=M, N=N, K=K, lay=lay)
            out.zero_()
            ms = timeit(f)
            err = (out - ref).abs().max().item() / ref.abs().max().item()
            res.append((ms, names[lay], split, err))
    res.sort()
    print(f"{name} dW[{N}x{K}] tokens={M}: " + " | ".join(f"{t} s{s} {ms:.3f}ms {fl/ms/1e9:.0f}TF e={e:.0e}" for ms, t, s, e in res[:7]), flush=True)
    worst = max(res, key=lambda r: r[3])
    print(f"    max rel err over all variants {worst[3]:.1e} ({worst[1]} s{worst[2]})", flush=True)
