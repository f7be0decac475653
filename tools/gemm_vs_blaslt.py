#!/usr/bin/env python
"""Calibration, not product: the SiT-XL/2 block GEMMs through reed_amd's kernels and through hipBLASLt / rocBLAS (torch.matmul
on bf16 tensors) on the same operands, same stream, HIP events. Says how much of the distance to the 2.5 PFLOP/s dense peak the
vendor library closes on these shapes. usage: python tools/gemm_vs_blaslt.py [b]"""
import json
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


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


rows = []
shapes = {"qkv": (3 * D, D), "proj": (D, D), "fc1": (Hm, D), "fc2": (D, Hm)}
for lay in ("NT", "NN", "TN"):
    for name, (N, K) in shapes.items():
        flop = 2.0 * M * N * K
        if lay == "NT":      # y[M,N] = x[M,K] w[N,K]^T
            x, w = bf(M, K), bf(N, K)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            ours = lambda: ops.linear_fwd(x, w, None, out)  # noqa: E731
            lib = lambda: torch.matmul(x, w.t(), out=out)  # noqa: E731
        elif lay == "NN":    # dx[M,K] = dy[M,N] w[N,K]
            x, w = bf(M, N), bf(N, K)
            out = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
            ours = lambda: ops.linear_dgrad(x, w, out)  # noqa: E731
            lib = lambda: torch.matmul(x, w, out=out)  # noqa: E731
        else:                # dw[N,K] = dy[M,N]^T x[M,K]   (ours: fp32 output, split-K + slab reduce as the engine runs it)
            x, w = bf(M, N), bf(M, K)
            out = torch.empty(N, K, device=dev)
            out16 = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
            wl, split = ops.plan_wgrad(M, N, K)
            ws = torch.empty(16 * N * K, device=dev)
            ours = lambda: ops.linear_wgrad(x, w, out, split_k=split, ws=ws.data_ptr(), lay=wl)  # noqa: E731
            lib = lambda: torch.matmul(x.t(), w, out=out16)  # noqa: E731
        t_o, t_l = timeit(ours), timeit(lib)
        r = dict(layout=lay, name=name, M=M, N=N, K=K, ours_ms=round(t_o, 4), ours_tflops=round(flop / t_o / 1e9, 1),
                 lib_ms=round(t_l, 4), lib_tflops=round(flop / t_l / 1e9, 1), ours_over_lib=round(t_l / t_o, 3))
        rows.append(r)
        print(json.dumps(r), flush=True)
tot_o = sum(r["ours_ms"] for r in rows)
tot_l = sum(r["lib_ms"] for r in rows)
print(json.dumps({"b": b, "sum_ours_ms": round(tot_o, 3), "sum_lib_ms": round(tot_l, 3), "torch": torch.__version__,
                  "blas_backend": str(torch.backends.cuda.preferred_blas_library())}))
