#!/usr/bin/env python
"""C5 (BASELINE.json configs[4]): generate.py's sampling hot loop on one MI355X — SiT-XL/2, Heun ODE sampler with
classifier-free guidance over the whole interval (every model evaluation at batch 2n; reference samplers.py:46-104).
Times a short run (default 8 Heun steps = 15 evaluations) and scales to the 250-step recipe (499 evaluations): the
per-evaluation cost does not depend on the step index. Reports generated images/s, model evaluations/s and the
fraction of the bf16 MFMA roofline (237.23 GFLOP per image per evaluation).
usage (GPU box): python tools/bench_generate.py [n_per_gpu] [steps] [fp16|bf16]   (fp16 = generate.py's default: the
IEEE-half build of the kernels, the mantissa of the reference's TF32 evaluations)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import random_fill                    # noqa: E402
from reed_amd.models.sit import SiT_models       # noqa: E402
from reed_amd.samplers import euler_sampler      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
precision = sys.argv[3] if len(sys.argv) > 3 else "fp16"
dev = torch.device("cuda")
torch.manual_seed(0)
model = SiT_models["SiT-XL/2"](z_dims=[1024], z_types=["i"], encoder_depth=8, use_cfg=True).to(dev).eval()
random_fill(model, 1234)
model.precision = precision
z = torch.randn(n, 4, 32, 32, device=dev)
y = torch.randint(0, 1000, (n,), device=dev)
euler_sampler(model, z, y, num_steps=2, heun=True, cfg_scale=1.5)   # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
out = euler_sampler(model, z, y, num_steps=steps, heun=True, cfg_scale=1.5, guidance_low=0.0, guidance_high=1.0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
evals = 2 * steps - 1
per_eval = dt / evals
flop_eval = 237.23e9 * 2 * n          # CFG: batch 2n per evaluation
res = {"metric": "SiT-XL/2 250-step Heun CFG sampling, generated images/sec (1 x MI355X)", "n_per_gpu": n, "precision": precision,
       "timed": f"{steps} Heun steps = {evals} model evaluations at batch {2 * n} in {dt:.2f}s",
       "ms_per_evaluation": round(per_eval * 1e3, 2), "evaluations_per_sec": round(1 / per_eval, 2),
       "value": round(n / (499 * per_eval), 3), "seconds_per_image_250_heun": round(499 * per_eval / n, 3),
       "roofline": {"bound": "mfma", "achieved": round(flop_eval / per_eval / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                    "frac": round(flop_eval / per_eval / 2.5e15, 4)},
       "finite": bool(torch.isfinite(out).all()), "out_dtype": str(out.dtype)}
print(json.dumps(res))
