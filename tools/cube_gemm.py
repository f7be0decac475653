#!/usr/bin/env python
"""NT vs NN vs TN on the same cubic problem (operands resident in the Infinity Cache): isolates the LDS/issue path of
the k-strided (transposing-read) layouts from the HBM side. usage: python tools/cube_gemm.py [n]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reed_amd import _lib, ops
L = _lib.load(); dev = torch.device("cuda"); n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
a, b = bf(n, n), bf(n, n)
ob, of = torch.empty(n, n, dtype=torch.bfloat16, device=dev), torch.empty(n, n, device=dev)
def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for tile in (128, 256):
    L.reed_gemm_force_tile(tile)
    for name, fn in (("NT", lambda: ops.gemm(ops.NT, ops.EPI_BF16, a, b, n, n, n, ob, n, n, n)),
                     ("NN", lambda: ops.gemm(ops.NN, ops.EPI_BF16, a, b, n, n, n, ob, n, n, n)),
                     ("TN", lambda: ops.gemm(ops.TN, ops.EPI_F32, a, b, n, n, n, of, n, n, n))):
        ms = timeit(fn)
        print(f"tile {tile} {name}: {ms:.4f} ms {2.0*n**3/ms/1e9:7.1f} TF/s", flush=True)
L.reed_gemm_force_tile(0)
