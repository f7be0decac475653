#!/usr/bin/env python
"""Workload for rocprofv3 --pmc passes over the third session's kernels, 3 launches each through the product entry points:
gemm144_kernel<NN, bf16> (dgrad fc1 at b = 32: N 1152, K 4608) and gemm144_kernel<NT, gate+res> (fc2 forward at b = 32),
the 256^2 kernel forced on the same dgrad for comparison, and attention forward / backward (hd 72, b = 256) with the
LDS-staged row stores."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reed_amd import ops
dev = torch.device("cuda"); T, H, hd = 256, 16, 72
D, Hm = 1152, 4608
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
b = 32; M = b * T
a1, w1, w2, b2 = bf(M, Hm), bf(Hm, D), bf(D, Hm), bf(D)
dx = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
gate, xin, xout = bf(b, 6 * D), torch.randn(M, D, device=dev), torch.empty(M, D, device=dev)
y = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    ops.gemm_force_tile(0)
    ops.gemm(ops.NN, ops.EPI_BF16, a1, w1, M, D, Hm, dx, Hm, D, D)                       # gemm144<NN, bf16>
    ops.linear_fwd(a1, w2, b2, xout, epi=ops.EPI_GATE_RES, R=xin, gate=gate[:, 5 * D:], ldgate=6 * D, rows_per_gate=T,
                   y_out=y)                                                             # gemm144<NT, gate+res>
    ops.gemm_force_tile(256)
    ops.gemm(ops.NN, ops.EPI_BF16, a1, w1, M, D, Hm, dx, Hm, D, D)                       # gemm256<NN, bf16> (160 workgroups)
ops.gemm_force_tile(0)
b = 256; M = b * T
qkv, o, do = bf(M, 3 * D) * 10, torch.empty(M, D, dtype=torch.bfloat16, device=dev), bf(M, D)
dqkv, lse = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev), torch.empty(b, H, T, device=dev)
for _ in range(3):
    ops.attention_fwd(qkv, o, lse, b, T, H, hd)
    ops.attention_bwd(qkv, o, do, lse, dqkv, b, T, H, hd)
torch.cuda.synchronize()
print("done")
