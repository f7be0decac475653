#!/usr/bin/env python
"""Per-phase shader-clock time of attn_bwd_ring_kernel (REED_ATTN_KSP_DBG bit 2): cycles between eight stamps, summed per wave over
its items, left at the start of dqkv.  usage (GPU box, diagnosis build: python tools/_ab/build_variant.py diag -DREED_ATTN_DIAG): REED_HIP_LIB=tools/_ab/libreed_diag.so
REED_ATTN_KSP_DBG=4 python tools/attn_bwd_stamps.py [b]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reed_amd import ops

assert int(os.environ.get("REED_ATTN_KSP_DBG", "0")) & 4, "set REED_ATTN_KSP_DBG=4 (+ other bits)"
dev = torch.device("cuda")
T, H, hd = 256, 16, 72
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
D = H * hd
M = b * T
qkv = (torch.randn(M, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
o = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
do = torch.randn(M, D, device=dev).to(torch.bfloat16)
dqkv = torch.zeros(M, 3 * D, dtype=torch.bfloat16, device=dev)
lse = torch.empty(b, H, T, device=dev)
ws = torch.empty(ops.attention_bwd_ws_floats(b, T, H), device=dev)
ops.attention_fwd(qkv, o, lse, b, T, H, hd)
for _ in range(3):
    ops.attention_bwd(qkv, o, do, lse, dqkv, b, T, H, hd, ws=ws)
torch.cuda.synchronize()
nwg = min(b * H, 256)
t = dqkv.flatten().view(torch.int64)[: nwg * 8 * 8].view(nwg, 8, 8).double().cpu()
items = (b * H) / nwg
names = ["item-start barrier", "phase A (4 chunks)", "barrier after A", "issue + phase B + dQ stores", "ring wait (vmcnt)", "barrier after B",
         "dK, dV through LDS + stores", "item-end wait + own rows"]
print(f"b={b}: {items:.1f} items per workgroup; cycles per item and wave (mean over workgroups), waves 0-3 | waves 4-7")
tot = [0.0, 0.0]
for k, n in enumerate(names):
    a, c = t[:, :4, k].mean().item() / items, t[:, 4:, k].mean().item() / items
    tot[0] += a
    tot[1] += c
    print(f"  {n:32s} {a:9.0f} | {c:9.0f}")
print(f"  {'sum':32s} {tot[0]:9.0f} | {tot[1]:9.0f}")
