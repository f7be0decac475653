#!/usr/bin/env python
"""Host-side view of one training step: every library call (ops._call) timed on the host; prints the calls that took longer than
a threshold (where the host blocks on the GPU) and the per-entry-point totals.  usage: python tools/host_slow_calls.py [b] [us]"""
import copy
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from reed_amd import ops  # noqa: E402
from reed_amd.loss import SILoss  # noqa: E402
from reed_amd.models.sit import SiT_models  # noqa: E402
from reed_amd.optim import FusedAdamWEMA  # noqa: E402
from reed_amd.trainer import TrainStep  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = SiT_models["SiT-XL/2"](z_dims=[1024], z_types=["i"], encoder_depth=8).to(dev).train()
bench.random_fill(model, 1234)
ema = copy.deepcopy(model).requires_grad_(False).eval()
opt = FusedAdamWEMA(model, ema, lr=1e-4, max_grad_norm=1.0)
loss_fn = SILoss(enc_names=["dinov2-vit-l"], loss_weights={"dinov2-vit-l": 1.0})
step = TrainStep(model, loss_fn, opt, None, proj_coeff=0.5, diffusion_warm_up_steps=0)
g = torch.Generator(device=dev).manual_seed(100)
mean = torch.randn(b, 4, 32, 32, device=dev, generator=g) * 5.49
moments = torch.cat([mean, torch.full_like(mean, 0.5)], dim=1)
labels = torch.randint(0, 1000, (b,), device=dev, generator=g)
zs = [torch.randn(b, 256, 1024, device=dev, generator=g)]

log = []
orig = ops._call


def timed(name, *a):
    t = time.perf_counter()
    r = orig(name, *a)
    log.append((name, t, time.perf_counter() - t))
    return r


for _ in range(4):
    step(None, labels, zs, moments=moments)
torch.cuda.synchronize()
ops._call = timed
K = 6
t0 = time.perf_counter()
starts = []
for _ in range(K):
    starts.append(time.perf_counter())
    step(None, labels, zs, moments=moments)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ops._call = orig
print(f"b = {b}: {dt / K * 1e3:.2f} ms per step; host {t_enq / K * 1e3:.2f} ms per step; {len(log) / K:.0f} library calls per step, "
      f"{sum(d for _, _, d in log) / K * 1e3:.2f} ms inside them")
tot = defaultdict(lambda: [0.0, 0])
for n, t, d in log:
    tot[n][0] += d
    tot[n][1] += 1
for n, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:10]:
    print(f"  {n:34s} {d / K * 1e3:8.3f} ms per step, {c / K:6.1f} calls, {d / c * 1e6:8.1f} us each")
s = starts[-2]
print(f"calls over {thr:.0f} us in the last-but-one step (ms since step start, us):")
idx = 0
for n, t, d in log:
    if starts[-2] <= t < starts[-1]:
        idx += 1
        if d * 1e6 > thr:
            print(f"  #{idx:4d} {(t - s) * 1e3:8.3f} {d * 1e6:9.1f} {n}")
