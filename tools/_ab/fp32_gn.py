"""Which gradient tensors make up the XL/2 fp32 step-1 gradient norm (diagnostic)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from tests.test_model_gpu import _hip_trainer, _run_traj
from tests.test_oracle_golden import load
dev = torch.device("cuda:0")
g = load("xl2_c2")
m, ema, opt, lf = _hip_trainer("SiT-XL/2", dict(z_dims=[1024], z_types=["i"], encoder_depth=8), dev, ["dinov2"], [1.0])
m.precision = ema.precision = "fp32"
out = {}
def grab(step):
    if step == 0:
        torch.cuda.synchronize()
        tot = 0.0
        groups = {}
        for k, p in m.named_parameters():
            if p.grad is None:
                print("no grad:", k); continue
            n2 = float(p.grad.double().pow(2).sum())
            tot += n2
            key = ".".join(x for x in k.split(".") if not x.isdigit())
            groups[key] = groups.get(key, 0.0) + n2
        ga = load("xl2_c2_gnorms")
        bad = []
        for k, p in m.named_parameters():
            if p.grad is None: continue
            r = float(p.grad.double().norm()) / float(ga["gnorm." + k])
            if abs(r - 1) > 1e-5: bad.append((k, r, float(ga["gnorm." + k])))
        print("parameters off by > 1e-5:", len(bad))
        for b in bad[:60]: print("   ", b)
        print("sum over named parameters:", tot ** 0.5)
        print("arena grad norm:", float(m._arena.grad.double().norm()))
        for k, v in sorted(groups.items(), key=lambda kv: -kv[1]):
            print(f"  {k:45s} {v ** 0.5:.6e}")
rec = _run_traj(m, opt, lf, dev, 8, 1, [(1024, "i")], True, after_backward=grab)
print("opt.grad_norm", rec["grad_norm"], "ref", g["fp32.grad_norm"][0])
