#!/usr/bin/env python
"""Workload for rocprofv3 --pmc passes over the attention kernels (hd 72, T 256, b = 256 unless given): 3 launches each of
forward and backward through the product entry points."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reed_amd import ops
dev = torch.device("cuda"); T, H, hd = 256, 16, 72
D = H * hd
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = b * T
bf = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
qkv, o, do = bf(M, 3 * D), torch.empty(M, D, dtype=torch.bfloat16, device=dev), bf(M, D)
dqkv, lse = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev), torch.empty(b, H, T, device=dev)
ws = torch.empty(ops.attention_bwd_ws_floats(b, T, H), device=dev)
for _ in range(3):
    ops.attention_fwd(qkv, o, lse, b, T, H, hd)
    ops.attention_bwd(qkv, o, do, lse, dqkv, b, T, H, hd, ws=ws)
torch.cuda.synchronize()
print("done")
