#!/usr/bin/env python
"""A/B the 128^2 and 256^2 GEMM kernels on the SiT-XL/2 block shapes (events on the launch stream), and check
that both kernels produce the same numbers. usage: python tools/bench_gemm.py [b] [layouts e.g. NT,NN,TN]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reed_amd import _lib, ops  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lays = (sys.argv[2] if len(sys.argv) > 2 else "NT,NN,TN").split(",")
dev = torch.device("cuda")
L = _lib.load()
D, Hm, T = 1152, 4608, 256
M = b * T
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)  # noqa: E731


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


shapes = {"qkv": (3 * D, D), "proj": (D, D), "fc1": (Hm, D), "fc2": (D, Hm)}
for lay in lays:
    for name, (N, K) in shapes.items():
        if lay == "NT":
            x, w = bf(M, K), bf(N, K)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            fn = lambda: ops.linear_fwd(x, w, None, out)  # noqa: E731
            flop = 2.0 * M * N * K
        elif lay == "NN":   # dx[M,K] = dy[M,N] @ w[N,K]
            x, w = bf(M, N), bf(N, K)
            out = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
            fn = lambda: ops.linear_dgrad(x, w, out)  # noqa: E731
            flop = 2.0 * M * N * K
        else:               # dw[N,K] = dy[M,N]^T x[M,K]
            x, w = bf(M, N), bf(M, K)
            out = torch.empty(N, K, device=dev)
            ws = torch.empty(16 * N * K + ops.colsum_ws_floats(M, N), device=dev)
            gb = torch.empty(N, device=dev)

            def fn():
                tile = ops._FORCED
                if tile == 256:
                    _lay, split = ops.plan_wgrad(M, N, K)
                    ops.colsum_bf16(x, N, ws, gb, M, N)
                    ops.linear_wgrad(x, w, out, split_k=split, ws=ws.data_ptr() + 4 * ops.colsum_ws_floats(M, N))
                else:
                    ops.linear_wgrad(x, w, out, dbias=gb)
            flop = 2.0 * M * N * K
        res = {}
        for tile in (128, 256):
            L.reed_gemm_force_tile(tile)
            ops._FORCED = tile
            out.zero_()
            try:
                ms = timeit(fn)
            except RuntimeError as e:
                print(lay, name, tile, "ERR", str(e)[:80]); continue
            res[tile] = (ms, out.float().clone())
            print(f"{lay} {name:5s} M={M} N={N} K={K} tile={tile}: {ms:.4f} ms  {flop / ms / 1e9:7.1f} TF/s", flush=True)
        if len(res) == 2:
            d = (res[128][1] - res[256][1]).abs().max().item()
            print(f"    max |128 - 256| = {d:.3e} (max |ref| {res[128][1].abs().max().item():.3e})")
L.reed_gemm_force_tile(0)
