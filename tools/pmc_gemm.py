#!/usr/bin/env python
"""Tiny workload for rocprofv3 --pmc passes: 3 launches each of the dominant GEMM kernels at the bench shapes
(b=256), through the product entry points: wgrad fc1 (TN 128^2, wave-quantised split-K + slab reduce), fwd fc1
(NT 256^2, GELU epilogue), dgrad fc1 (NN 256^2)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reed_amd import ops
dev = torch.device("cuda"); M, D, Hm = 65536, 1152, 4608
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
x, w1, b1 = bf(M, D), bf(Hm, D), bf(Hm)
pre, act = torch.empty(M, Hm, dtype=torch.bfloat16, device=dev), torch.empty(M, Hm, dtype=torch.bfloat16, device=dev)
dw = torch.empty(Hm, D, device=dev); db = torch.empty(Hm, device=dev); ws = torch.empty(8 * (Hm * D + Hm), device=dev)
dx = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
lay, split = ops.plan_wgrad(M, Hm, D)
for _ in range(3):
    ops.linear_fwd(x, w1, b1, pre, epi=ops.EPI_GELU, act_out=act)
    ops.linear_wgrad(act, x, dw, dbias=db, split_k=split, Mtok=M, N=Hm, K=D, ws=ws, lay=lay)
    ops.gemm(ops.NN, ops.EPI_BF16, act, w1, M, D, Hm, dx, Hm, D, D)
torch.cuda.synchronize()
print("done")
