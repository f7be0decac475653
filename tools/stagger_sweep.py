#!/usr/bin/env python
"""First-round start stagger of the 256^2 GEMM kernel vs epilogue-heavy shapes. usage: python tools/stagger_sweep.py [b]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reed_amd import _lib, ops
L = _lib.load(); dev = torch.device("cuda"); b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T, D, Hm = 256, 1152, 4608; M = b * T
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
x, w_qkv, w_proj, w1, w2 = bf(M, D), bf(3 * D, D), bf(D, D), bf(Hm, D), bf(D, Hm)
big, big2, o3, ybuf = bf(M, Hm), bf(M, Hm), bf(M, 3 * D), bf(M, D)
xo = torch.empty(M, D, device=dev); xi = torch.randn(M, D, device=dev); gate = bf(b, 6 * D); bias = bf(Hm)
def dgrad(epi, dy, w, N, K, out, **kw): ops.gemm(ops.NN, epi, dy, w, M, K, N, out, N, K, K, **kw)
cases = [
    ("fwd qkv", lambda: ops.linear_fwd(x, w_qkv, bias[:3 * D], o3)),
    ("fwd proj gate+res", lambda: ops.linear_fwd(x, w_proj, bias[:D], xo, epi=ops.EPI_GATE_RES, R=xi, gate=gate, ldgate=6 * D, rows_per_gate=T, y_out=ybuf)),
    ("fwd fc1 gelu", lambda: ops.linear_fwd(x, w1, bias, big, epi=ops.EPI_GELU, act_out=big2)),
    ("fwd fc2 gate+res", lambda: ops.linear_fwd(big, w2, bias[:D], xo, epi=ops.EPI_GATE_RES, R=xi, gate=gate, ldgate=6 * D, rows_per_gate=T, y_out=ybuf)),
    ("dgrad fc2 dgelu", lambda: dgrad(ops.EPI_DGELU, x, w2, D, Hm, big, R=big2, ldr=Hm)),
    ("dgrad fc1", lambda: dgrad(ops.EPI_BF16, big, w1, Hm, D, ybuf)),
    ("dgrad qkv", lambda: dgrad(ops.EPI_BF16, o3, w_qkv, 3 * D, D, ybuf)),
]
def timeit(fn, it=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
STS = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else (0, 100, 200, 400, 700, 1000)
for tile in (0, 256):
    L.reed_gemm_force_tile(tile)
    for name, fn in cases:
        row = []
        for st in STS:
            L.reed_gemm_set_stagger(st)
            row.append(f"{st}:{timeit(fn):.3f}")
        print(f"tile={tile} {name:20s} " + "  ".join(row), flush=True)
L.reed_gemm_set_stagger(-1); L.reed_gemm_force_tile(0)
