#!/usr/bin/env python
"""Tiny workload for rocprofv3 --pmc passes of the grouped weight-gradient launch alone: 4 launches at the bench's shape
(b = 256: 65536 tokens; `python3 tools/pmc_wgrad.py 32` for the 8-GPU shape), through the product entry point."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reed_amd import ops
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda"); M, D, Hm = b * 256, 1152, 4608
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
probs = []
for n_out, k_in in ((D, Hm), (Hm, D), (D, D), (3 * D, D)):
    out = torch.zeros(n_out * k_in + n_out, device=dev)
    probs.append((bf(M, n_out), bf(M, k_in), out[:n_out * k_in].view(n_out, k_in), out[n_out * k_in:], n_out, k_in))
for _ in range(4):
    ops.wgrad_group(probs, M)
torch.cuda.synchronize()
print("done")
