#!/usr/bin/env python
"""Where the host is while the GPU runs a training step: host time per phase (forward enqueue, backward, optimiser) and, at the
end of each phase, whether the GPU had already drained everything enqueued before it (= the host is behind, the GPU idles).
usage: python tools/host_phases.py [local_batch] [steps]"""
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from reed_amd.loss import SILoss  # noqa: E402
from reed_amd.models.sit import SiT_models  # noqa: E402
from reed_amd.optim import FusedAdamWEMA  # noqa: E402
from reed_amd.trainer import TrainStep  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
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

marks = []


def mark(name):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.append((name, time.perf_counter(), ev))


orig_loss = step.loss_fn


def loss_wrapped(*a, **k):
    mark("step start -> loss_fn")
    out = orig_loss(*a, **k)
    mark("forward enqueued")
    return out


step.loss_fn = loss_wrapped
orig_opt = opt.step


def opt_wrapped():
    mark("backward returned")
    orig_opt()
    mark("optimiser enqueued")


opt.step = opt_wrapped
for _ in range(4):
    step(None, labels, zs, moments=moments)
torch.cuda.synchronize()
marks.clear()
e0 = torch.cuda.Event(enable_timing=True)
e0.record()
torch.cuda.synchronize()
t0 = time.perf_counter()
t0_gpu = 0.0   # e0 completed (about) now: GPU time of a later event = e0.elapsed_time(ev), host time = perf_counter - t0
rows = []
for _ in range(K):
    step(None, labels, zs, moments=moments)
    # at each mark of this step: had the GPU already finished what was enqueued before the PREVIOUS mark?
t_enq = time.perf_counter() - t0
# replay: for each mark i, query whether mark i-1's event was complete at the time mark i was taken is not recoverable after the
# fact; instead record completion lag now: time from mark to event completion measured by polling once per mark below
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"b = {b}: {dt / K * 1e3:.2f} ms per step, host enqueue {t_enq / K * 1e3:.2f} ms per step")
names = ["step start -> loss_fn", "forward enqueued", "backward returned", "optimiser enqueued"]
per = {n: [] for n in names}
for i in range(1, len(marks)):
    per[marks[i][0]].append((marks[i][1] - marks[i - 1][1]) * 1e3)
for n in names:
    v = per[n][1:]
    if v:
        print(f"  host ms up to '{n}': mean {sum(v) / len(v):7.3f}  min {min(v):7.3f}  max {max(v):7.3f}")
# how far ahead of the GPU the host is at each mark: (GPU time at which the stream reaches the mark) - (host time of the mark);
# near zero or negative = the GPU had nothing queued there (the host is the limit at that point of the step)
lead = {n: [] for n in names}
for n, th, ev in marks:
    lead[n].append(e0.elapsed_time(ev) - (th - t0) * 1e3)
for n in names:
    v = lead[n][2:]
    if v:
        print(f"  host lead at '{n}' (ms): mean {sum(v) / len(v):7.3f}  min {min(v):7.3f}  max {max(v):7.3f}")
