"""The fused clip + AdamW + EMA + shadow pass alone (no forward beside it): bytes / time."""
import copy, os, sys, torch
sys.path.insert(0, ".")
os.environ["REED_OPT_OVERLAP"] = "0"
from reed_amd.models.sit import SiT_models
from reed_amd.optim import FusedAdamWEMA
dev = torch.device("cuda:0")
m = SiT_models["SiT-XL/2"](z_dims=[1024], z_types=["i"], encoder_depth=8).to(dev).train()
ema = copy.deepcopy(m).requires_grad_(False).eval()
opt = FusedAdamWEMA(m, ema, lr=1e-4)
A, L = m._arena, m._layout
A.ensure_grad() if hasattr(A, "ensure_grad") else None
if A.grad is None:
    A.grad = torch.zeros_like(A.master)
A.grad.normal_(0, 1e-3)
for _ in range(3): opt.step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): opt.step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
n = L.n_total
print(f"n = {n/1e6:.1f} M parameters: {ms:.3f} ms per step (norm + clip + update) = {(4 + 38) * n / ms / 1e9:.2f} TB/s of 42 B/param")
