#!/usr/bin/env python
"""Is the host ahead of the GPU?  K training steps of SiT-XL/2 at per-GPU batch b without any host synchronisation: the host time the
Python side needs to ENQUEUE a step (forward+backward / optimiser / the rest) against the wall time the GPU needs to run it.  The b = 32 kernel
trace shows the main queue empty for 1.1 ms behind clip_finalize while the optimiser's chunks run on the side queue.
usage: python tools/r6/host_ahead.py [b] [steps]"""
import copy, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reed_amd.loss import SILoss
from reed_amd.models.sit import SiT_models
from reed_amd.optim import FusedAdamWEMA
from reed_amd.trainer import TrainStep
import bench
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda")
torch.manual_seed(0)
model = SiT_models["SiT-XL/2"](z_dims=[1024], z_types=["i"], encoder_depth=8).to(dev).train()
model.precision = "bf16"
bench.random_fill(model, 1234)
ema = copy.deepcopy(model).requires_grad_(False).eval()
opt = FusedAdamWEMA(model, ema, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.0, eps=1e-8, max_grad_norm=1.0)
step = TrainStep(model, SILoss(enc_names=["dinov2-vit-l"], loss_weights={"dinov2-vit-l": 1.0}), opt, None, proj_coeff=0.5,
                 diffusion_warm_up_steps=0)
g = torch.Generator(device=dev).manual_seed(100)
mean = torch.randn(b, 4, 32, 32, device=dev, generator=g) * 5.49
moments = torch.cat([mean, torch.full_like(mean, 0.5)], dim=1)
labels = torch.randint(0, 1000, (b,), device=dev, generator=g)
zs = [torch.randn(b, 256, 1024, device=dev, generator=g)]
for _ in range(3):
    step(None, labels, zs, moments=moments)
torch.cuda.synchronize()
# host time inside opt.step (its launches and events) by wrapping it
host = {"opt": 0.0}
_os = opt.step


ev = []


def timed_step():
    t = time.perf_counter()
    _os()
    host["opt"] += time.perf_counter() - t
    e = torch.cuda.Event(enable_timing=True)
    e.record()            # main stream: right behind clip_finalize
    ev.append([e])


import reed_amd.trainer as _tr
_sp = _tr.sample_posterior


def marked_sp(*a, **k):
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()           # main stream: the first thing of the next step
    r = _sp(*a, **k)
    e = torch.cuda.Event(enable_timing=True)
    e.record()            # main stream: behind the next step's first two kernels (RNG, sample_posterior)
    if ev and len(ev[-1]) == 1:
        ev[-1] += [e0, e]
    return r


_tr.sample_posterior = marked_sp


opt.step = timed_step
t0 = time.perf_counter()
marks = []
for _ in range(K):
    step(None, labels, zs, moments=moments)
    marks.append(time.perf_counter())
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
per = [(marks[i] - (marks[i - 1] if i else t0)) * 1e3 for i in range(K)]
gaps = [(x[0].elapsed_time(x[1]) * 1e3, x[1].elapsed_time(x[2]) * 1e3) for x in ev if len(x) == 3]
print("main stream, us: clip_finalize done -> next step's first launch reached | RNG + sample_posterior:",
      " ".join(f"{a:.0f}|{c:.0f}" for a, c in gaps))
print(f"b = {b}: host enqueue {t_enq / K * 1e3:.2f} ms per step (of which optimiser.step {host['opt'] / K * 1e3:.2f}), "
      f"GPU {t_all / K * 1e3:.2f} ms per step; per-step host ms: {' '.join(f'{v:.1f}' for v in per)}")
