#!/usr/bin/env python
"""cProfile of the host side of the training step at a tiny batch (the GPU is never the limit there).
usage: python tools/host_profile.py [b] [steps]"""
import copy
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from reed_amd.loss import SILoss  # noqa: E402
from reed_amd.models.sit import SiT_models  # noqa: E402
from reed_amd.optim import FusedAdamWEMA  # noqa: E402
from reed_amd.trainer import TrainStep  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
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
for _ in range(4):
    step(None, labels, zs, moments=moments)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    step(None, labels, zs, moments=moments)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(30)
