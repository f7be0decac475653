#!/usr/bin/env python
"""Where does the epilogue time go? Same GEMM, output rows collapsed onto one row (ldc=0: same store instructions, no
HBM write stream / cache pollution) vs the real output. usage: python tools/epi_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reed_amd import _lib, ops
L = _lib.load(); dev = torch.device("cuda")
b, T, D, Hm = 256, 256, 1152, 4608; M = b * T
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
x, big = bf(M, D), bf(M, Hm)
w_qkv, w1, w2 = bf(3 * D, D), bf(Hm, D), bf(D, Hm)
o3, obig, obig2 = bf(M, 3 * D), bf(M, Hm), bf(M, Hm)
def timeit(fn, it=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for name, fn0, fn1 in (
    ("fwd qkv bf16", lambda: ops.gemm(ops.NT, ops.EPI_BF16, x, w_qkv, M, 3 * D, D, o3, D, D, 3 * D),
                     lambda: ops.gemm(ops.NT, ops.EPI_BF16, x, w_qkv, M, 3 * D, D, o3, D, D, 0)),
    ("fwd fc1 gelu", lambda: ops.gemm(ops.NT, ops.EPI_GELU, x, w1, M, Hm, D, obig, D, D, Hm, C2=obig2, ldc2=Hm),
                     lambda: ops.gemm(ops.NT, ops.EPI_GELU, x, w1, M, Hm, D, obig, D, D, 0, C2=obig2, ldc2=0)),
    ("dgrad fc1 bf16", lambda: ops.gemm(ops.NN, ops.EPI_BF16, big, w1, M, D, Hm, o3, Hm, D, D),
                       lambda: ops.gemm(ops.NN, ops.EPI_BF16, big, w1, M, D, Hm, o3, Hm, D, 0)),
):
    for pers in (1, 0):
        L.reed_gemm_set_persistent(pers)
        print(f"{name:16s} persistent={pers}: real output {timeit(fn0):.3f} ms | rows collapsed (ldc=0) {timeit(fn1):.3f} ms", flush=True)
L.reed_gemm_set_persistent(1)
