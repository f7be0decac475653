#!/usr/bin/env python
"""Throughput of the frozen CLIP ViT-L/14 image encoder forward on the HIP path: images/s and
fraction of the bf16 MFMA roofline (2 x 80.9 GMAC per image: 24 blocks x 257 tokens x 12 W^2 + attention + patch
embedding), with the oracle (CPU restatement, fp32) timed on the host cores beside it.
usage (GPU box): python tools/bench_encoder.py [batch] [--no-cpu]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import clip_vit as oclip          # noqa: E402  (checker / CPU baseline only)
from reed_amd.encoders import ClipVisionEncoder  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
dev = torch.device("cuda")
cfg = oclip.make_config()   # ViT-L/14 @ 224
W, L, T = cfg["width"], cfg["layers"], (cfg["image"] // cfg["patch"]) ** 2 + 1
mac = L * (T * 12 * W * W + 2 * T * T * W) + (T - 1) * 3 * cfg["patch"] ** 2 * W
enc = ClipVisionEncoder(**cfg)
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    for n, p in enc.named_parameters():
        if p.ndim >= 2:
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * (3.0 / p[0].numel()) ** 0.5)
        elif "ln_" in n and n.endswith("weight"):
            p.fill_(1.0)
        else:
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * 0.05)
enc = enc.to(dev).eval()
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
x = enc.preprocess(raw)
torch.cuda.synchronize()
e0.record()
for _ in range(iters):
    out = enc(x)
e1.record()
torch.cuda.synchronize()
ms_tower = e0.elapsed_time(e1) / iters
res = {"metric": "CLIP ViT-L/14 frozen encoder forward images/sec (1 x MI355X, bf16)", "batch": B,
       "value": round(B / ms * 1e3, 1), "ms_per_batch": round(ms, 2), "ms_tower_only": round(ms_tower, 2),
       "gflop_per_image": round(2 * mac / 1e9, 2),
       "roofline": {"bound": "mfma", "achieved": round(2 * mac * B / ms_tower / 1e9, 1), "peak": 2500.0, "unit": "TFLOP/s",
                    "frac": round(2 * mac * B / ms_tower / 1e9 / 2500.0, 4)}}
if "--no-cpu" not in sys.argv:
    P = {k: v.detach().float().cpu() for k, v in enc.state_dict().items()}
    xb = oclip.preprocess(raw[:4].cpu())
    with torch.no_grad():
        oclip.forward(P, cfg, xb[:1])
        t0 = time.time()
        ref = oclip.forward(P, cfg, xb)
        dt = time.time() - t0
    res["cpu_baseline"] = {"value": round(4 / dt, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
                           "sample": f"oracle/clip_vit.py fp32, 4 images, {dt:.1f}s"}
    d = (out[:4].float().cpu() - ref).abs().max().item() / ref.abs().max().item()
    res["max_rel_dev_vs_fp32_oracle"] = round(d, 4)
print(json.dumps(res))
