#!/usr/bin/env python
"""Per-phase shader-clock time of attn_fwd256p_kernel (diagnosis instantiation, REED_ATTN_FWD_DBG bit 5): the kernel sums, per
wave, the cycles between ten stamps over its items and leaves them at the start of `lse`.
usage (GPU box, diagnosis build): python tools/_ab/build_variant.py diag -DREED_ATTN_DIAG; REED_HIP_LIB=tools/_ab/libreed_diag.so
REED_ATTN_FWD_DBG=32 python tools/attn_fwd_stamps.py [b]   (other dbg bits may be OR-ed in)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reed_amd import ops

assert int(os.environ.get("REED_ATTN_FWD_DBG", "0")) & 32, "set REED_ATTN_FWD_DBG=32 (+ other bits)"
dev = torch.device("cuda")
T, H, hd = 256, 16, 72
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
D = H * hd
M = b * T
qkv = (torch.randn(M, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
o = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
lse = torch.zeros(b, H, T, device=dev)
for _ in range(3):
    ops.attention_fwd(qkv, o, lse, b, T, H, hd)
torch.cuda.synchronize()
nwg = min(b * H, 256)
t = lse.flatten().view(torch.int64)[: nwg * 8 * 10].view(nwg, 8, 10).double().cpu()
items = (b * H) / nwg
names = ["wait Q(n)", "Q frags + issue Q(n+1)", "S = K Q^T", "softmax", "wait V(n)", "barrier 2", "PV", "wait K(n+1)", "barrier 3",
         "epilogue + V(n+1) + stores"]
print(f"b={b}: {items:.1f} items per workgroup; cycles per item and wave (mean over workgroups), waves 0-3 | waves 4-7")
tot = [0.0, 0.0]
for k, n in enumerate(names):
    a, c = t[:, :4, k].mean().item() / items, t[:, 4:, k].mean().item() / items
    tot[0] += a
    tot[1] += c
    print(f"  {n:30s} {a:9.0f} | {c:9.0f}")
print(f"  {'sum':30s} {tot[0]:9.0f} | {tot[1]:9.0f}")
