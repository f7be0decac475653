import sys, torch
sys.path.insert(0, ".")
from reed_amd import ops
from tests.test_attention_gpu import _ref
dev = torch.device("cuda")
B, T, H, hd = 2, 256, 16, 72
g = torch.Generator().manual_seed(5 + T)
qkv = (torch.randn(B, T, 3, H, hd, generator=g)).to(torch.bfloat16).to(dev)
do = (torch.randn(B, T, H * hd, generator=g)).to(torch.bfloat16).to(dev)
o = torch.zeros(B, T, H * hd, dtype=torch.bfloat16, device=dev)
lse = torch.zeros(B, H, T, device=dev)
ops.attention_fwd(qkv, o, lse, B, T, H, hd)
dqkv = torch.full_like(qkv, float("nan"))
import os
ws = torch.full((ops.attention_bwd_ws_floats(B, T, H),), float("nan"), device=dev) if os.environ.get("FORM", "ws") == "ws" else None
for _ in range(int(os.environ.get("PRE", "0"))):
    ops.attention_bwd(qkv, o, do, lse, torch.empty_like(dqkv), B, T, H, hd, ws=torch.empty(ops.attention_bwd_ws_floats(B, T, H), device=dev))
ops.attention_bwd(qkv, o, do, lse, dqkv, B, T, H, hd, ws=ws)
q32 = qkv.float().requires_grad_(True)
ro, _ = _ref(q32, B, T, H, hd)
ro.backward(do.float())
ref = q32.grad
err = (dqkv.float() - ref).abs()
bad = ~(err < 0.05 * ref.abs().max())
print("nonfinite", int((~torch.isfinite(dqkv.float())).sum()))
print("bad total", int(bad.sum()), "of", bad.numel())
for w, n in enumerate("qkv"):
    bw = bad[:, :, w]
    print(n, "bad", int(bw.sum()))
    if bw.any():
        idx = bw.nonzero()
        print("  b", idx[:, 0].unique().tolist()[:8], "h", idx[:, 2].unique().tolist()[:20])
        print("  t", idx[:, 1].unique().tolist()[:40])
        print("  d", idx[:, 3].unique().tolist()[:80])
