#!/usr/bin/env python
"""Throughput of a frozen ViT target encoder on the HIP path: images/s through preprocess_raw_image + tower
and the tower alone as a fraction of the bf16 MFMA roofline.  usage (GPU box): python tools/bench_tower.py [enc-type] [batch]
enc-type: dinov2-vit-l (default; the C2 configuration's encoder), dinov2reg-vit-l, dinov2-vit-b, jepa-vit-h, mae-vit-l, mocov3-vit-l"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reed_amd.encoders import VIT_TOWERS, VitEncoder  # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "dinov2-vit-l"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda")
kw = VIT_TOWERS[key]
enc = VitEncoder(**kw)
enc.enc_type = key.split("-")[0]
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    for n, p in enc.named_parameters():
        if p.ndim >= 2 and "token" not in n and "pos_embed" not in n:
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * (3.0 / p[0].numel()) ** 0.5)
        elif n.endswith("gamma"):
            p.fill_(0.5)
        elif "norm" in n and n.endswith("weight"):
            p.fill_(1.0)
        else:
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * 0.05)
enc = enc.to(dev).eval()
E, L, T = enc.embed, enc.depth, enc.tokens
mac = L * (T * 12 * E * E + 2 * T * T * E) + enc.npatch * 3 * enc.patch ** 2 * E
raw = torch.randint(0, 256, (B, 3, 256, 256), dtype=torch.uint8, device=dev)
for _ in range(2):
    out = enc.encode_raw(raw)
torch.cuda.synchronize()
iters = 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    out = enc.encode_raw(raw)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
from reed_amd.encoders import preprocess_raw_image  # noqa: E402
x = preprocess_raw_image(raw, enc.enc_type)
torch.cuda.synchronize()
e0.record()
for _ in range(iters):
    out = enc(x)
e1.record()
torch.cuda.synchronize()
ms_tower = e0.elapsed_time(e1) / iters
assert bool(torch.isfinite(out).all()) and out.shape == (B, enc.npatch, E)
print(json.dumps({"metric": f"{key} frozen encoder forward images/sec (1 x MI355X, bf16)", "batch": B, "tokens": T,
                  "value": round(B / ms * 1e3, 1), "ms_per_batch": round(ms, 2), "ms_tower_only": round(ms_tower, 2),
                  "gflop_per_image": round(2 * mac / 1e9, 2),
                  "roofline": {"bound": "mfma", "achieved": round(2 * mac * B / ms_tower / 1e9, 1), "peak": 2500.0,
                               "unit": "TFLOP/s", "frac": round(2 * mac * B / ms_tower / 1e9 / 2500.0, 4)}}))
