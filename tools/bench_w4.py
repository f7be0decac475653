#!/usr/bin/env python
"""A/B of the 4-wave 128x128-per-wave kernel (csrc/gemm256w.hip, force_tile 257) against the 8-wave 256^2 kernel (256) on the
SiT-XL/2 block shapes: bit-identity of the outputs, then ms per launch (events on the launch stream).
usage: python tools/bench_w4.py [b]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reed_amd import ops  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
D, Hm, T = 1152, 4608, 256
M = b * T
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)  # noqa: E731


def timeit(fn, iters=int(os.environ.get("ITERS", "20"))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


shapes = {"qkv": (3 * D, D), "proj": (D, D), "fc1": (Hm, D), "fc2": (D, Hm)}
tot = {256: 0.0, 257: 0.0}
for lay in ("NT", "NN"):
    for name, (N, K) in shapes.items():
        flop = 2.0 * M * N * K
        if lay == "NT":
            x, w = bf(M, K), bf(N, K)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            fn = lambda: ops.linear_fwd(x, w, None, out)  # noqa: E731
        else:
            x, w = bf(M, N), bf(N, K)
            out = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
            fn = lambda: ops.linear_dgrad(x, w, out)  # noqa: E731
        res = {}
        for tile in (256, 257):
            ops.gemm_force_tile(tile)
            out.fill_(float("nan"))
            ms = timeit(fn)
            res[tile] = (ms, out.clone())
            tot[tile] += ms
        ops.gemm_force_tile(0)
        same = torch.equal(res[256][1], res[257][1])
        d = (res[256][1].float() - res[257][1].float()).abs().max().item()
        print(f"{lay} {name:5s}: 8-wave {res[256][0]:.4f} ms {flop / res[256][0] / 1e9:7.1f} TF | 4-wave {res[257][0]:.4f} ms "
              f"{flop / res[257][0] / 1e9:7.1f} TF | bit-identical {same} (max diff {d:.3e})", flush=True)
print(f"sum: 8-wave {tot[256]:.3f} ms, 4-wave {tot[257]:.3f} ms")
