#!/usr/bin/env python
"""Workload for rocprofv3 --pmc passes over the step's main kernels at the bench shapes (b=256, SiT-XL/2), 3 launches
each through the product entry points: the three fc1 GEMMs (forward NT+GELU 256^2, wgrad TN 128^2 split-K, dgrad NN
256^2 with the ragged last column tile), attention forward/backward (hd 72) and the LayerNorm+modulate backward."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reed_amd import ops
dev = torch.device("cuda"); b, T, H, hd = 256, 256, 16, 72
M, D, Hm = b * T, 1152, 4608
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
x, w1, b1 = bf(M, D), bf(Hm, D), bf(Hm)
pre, act = torch.empty(M, Hm, dtype=torch.bfloat16, device=dev), torch.empty(M, Hm, dtype=torch.bfloat16, device=dev)
dw = torch.empty(Hm * D + Hm, device=dev); ws = torch.empty(8 * (Hm * D + Hm), device=dev)
dx = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
qkv, o, do = bf(M, 3 * D) * 10, torch.empty(M, D, dtype=torch.bfloat16, device=dev), bf(M, D)
dqkv, lse = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev), torch.empty(b, H, T, device=dev)
xf, dxf = torch.randn(M, D, device=dev), torch.randn(M, D, device=dev)
mod = bf(b, 6 * D); mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
part, pg = torch.empty(M // 16, 2, D, device=dev), torch.empty(M // 16, D, device=dev)
dy = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
lay, split = ops.plan_wgrad(M, Hm, D)
mp = mod.data_ptr()
for _ in range(3):
    ops.linear_fwd(x, w1, b1, pre, epi=ops.EPI_GELU, act_out=act)
    ops.linear_wgrad(act, x, dw.data_ptr(), dbias=dw.data_ptr() + 4 * Hm * D, split_k=split, Mtok=M, N=Hm, K=D, ws=ws, lay=lay)
    ops.gemm(ops.NN, ops.EPI_BF16, act, w1, M, D, Hm, dx, Hm, D, D)
    ops.attention_fwd(qkv, o, lse, b, T, H, hd)
    ops.attention_bwd(qkv, o, do, lse, dqkv, b, T, H, hd)
    ops.ln_modulate_bwd_gate(do, xf, mean, rstd, mp + 2 * D, 6 * D, dxf, part, x, mp + 4 * D, 6 * D, dy, pg, None, M, D, T)
torch.cuda.synchronize()
print("done")
