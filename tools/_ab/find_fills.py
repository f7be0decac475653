"""Which torch fills / zeros / copies run inside one train step, with their sizes (torch profiler, record_shapes)."""
import copy, sys, torch
sys.path.insert(0, ".")
from bench import random_fill
from reed_amd.loss import SILoss
from reed_amd.models.sit import SiT_models
from reed_amd.optim import FusedAdamWEMA
from reed_amd.trainer import TrainStep
dev = torch.device("cuda:0")
b = 256
model = SiT_models["SiT-XL/2"](z_dims=[1024], z_types=["i"], encoder_depth=8).to(dev).train()
random_fill(model, 1234)
ema = copy.deepcopy(model).requires_grad_(False).eval()
opt = FusedAdamWEMA(model, ema, lr=1e-4, max_grad_norm=1.0)
lf = SILoss(enc_names=["dinov2-vit-l"], loss_weights={"dinov2-vit-l": 1.0})
step = TrainStep(model, lf, opt, None, proj_coeff=0.5, diffusion_warm_up_steps=0)
g = torch.Generator(device=dev).manual_seed(1)
mean = torch.randn(b, 4, 32, 32, device=dev, generator=g) * 5.49
moments = torch.cat([mean, torch.full_like(mean, 0.5)], dim=1)
labels = torch.randint(0, 1000, (b,), device=dev, generator=g)
zs = [torch.randn(b, 256, 1024, device=dev, generator=g)]
for _ in range(3): step(None, labels, zs, moments=moments)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    step(None, labels, zs, moments=moments)
    torch.cuda.synchronize()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::mul", "aten::add") and ev.input_shapes:
        n = 1
        for d in (ev.input_shapes[0] or []): n *= d
        if n >= 1 << 20:
            st = [f"{f.split('/')[-1]}" for f in (ev.stack or [])[:6] if "reed_amd" in f or "bench" in f]
            print(ev.name, ev.input_shapes[0], f"{n * 4 / 1e6:.0f} MB(fp32)", st[:3])
