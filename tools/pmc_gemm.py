#!/usr/bin/env python
"""Tiny workload for rocprofv3 --pmc passes: 3 launches each of the dominant GEMM kernels at the bench shapes (b=256), through
the product entry points: the block's grouped weight gradients (gemm256w_tn_group_kernel + wgrad_split_reduce_kernel since round 4; gemm_tn_group_kernel with REED_WGRAD_W4=0), fwd fc1 (NT 256^2 four-wave kernel, the
derivative-saving GELU epilogue 14 of round 5), dgrad fc2 (NN, the one-multiply epilogue 16), dgrad fc1 (NN, plain), fwd proj (NT, gate + residual)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reed_amd import ops
dev = torch.device("cuda"); M, D, Hm = 65536, 1152, 4608
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
x, w1, b1 = bf(M, D), bf(Hm, D), bf(Hm)
pre, act = torch.empty(M, Hm, dtype=torch.bfloat16, device=dev), torch.empty(M, Hm, dtype=torch.bfloat16, device=dev)
dx = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
probs = []
for n_out, k_in in ((D, Hm), (Hm, D), (D, D), (3 * D, D)):
    out = torch.zeros(n_out * k_in + n_out, device=dev)
    probs.append((bf(M, n_out), bf(M, k_in), out[:n_out * k_in].view(n_out, k_in), out[n_out * k_in:], n_out, k_in))
w2, wp = bf(D, Hm), bf(D, D)
da1 = torch.empty(M, Hm, dtype=torch.bfloat16, device=dev)
xin, xout, y = torch.randn(M, D, device=dev), torch.empty(M, D, device=dev), torch.empty(M, D, dtype=torch.bfloat16, device=dev)
gate = bf(M // 256, 6 * D)
for _ in range(3):
    ops.linear_fwd(x, w1, b1, pre, epi=ops.EPI_GELU_G, act_out=act)
    ops.wgrad_group(probs, M)
    ops.gemm(ops.NN, ops.EPI_MUL, x, w2, M, Hm, D, da1, D, Hm, Hm, R=pre, ldr=Hm)
    ops.gemm(ops.NN, ops.EPI_BF16, act, w1, M, D, Hm, dx, Hm, D, D)
    ops.linear_fwd(x, wp, b1[:D], xout, epi=ops.EPI_GATE_RES, R=xin, gate=gate, ldgate=6 * D, rows_per_gate=256, y_out=y)
torch.cuda.synchronize()
print("done")
