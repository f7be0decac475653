#!/usr/bin/env python
"""The staggered deal of the persistent 256^2 kernel (csrc/gemm256w.hip: w_stag_tile) against the plain static deal, same process
cannot switch (REED_W_STAGGER is read once): run once per setting.  b = 256: the two gate + residual GEMMs (proj, fc2 forward) and, with
REED_W_STAGGER=2, the three plain 1152-wide input gradients; event-timed us per launch, 3 passes of 20.
usage: REED_W_STAGGER=0|1|2 python tools/r6/stagger.py [b]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reed_amd import ops
dev = torch.device("cuda"); b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M, D, Hm = b * 256, 1152, 4608
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
xin, xo, y = torch.randn(M, D, device=dev), torch.empty(M, D, device=dev), torch.empty(M, D, dtype=torch.bfloat16, device=dev)
gate, bias = bf(b, D), bf(D)
cases = []
for name, K in (("proj forward (gate + residual)", D), ("fc2 forward (gate + residual)", Hm)):
    x, w = bf(M, K), bf(D, K)
    cases.append((name, K, lambda x=x, w=w, K=K: ops.gemm(ops.NT, ops.EPI_GATE_RES, x, w, M, D, K, xo, K, K, D, C2=y, ldc2=D, R=xin, ldr=D, bias=bias,
                                                        gate=gate, ldgate=D, rows_per_gate=256)))
for name, K in (("dgrad proj (plain)", D), ("dgrad qkv (plain)", 3 * D), ("dgrad fc1 (plain)", Hm)):
    dy, wt = bf(M, K), bf(D, K)
    cases.append((name, K, lambda dy=dy, wt=wt, K=K: ops.gemm(ops.NT, ops.EPI_BF16, dy, wt, M, D, K, y, K, K, D)))
for name, K, fn in cases:
    res = []
    for rep in range(3):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"REED_W_STAGGER={os.environ.get('REED_W_STAGGER', '1')} b = {b} {name}: {' '.join(f'{v:.1f}' for v in res)} us "
          f"({2.0 * M * D * K / min(res) / 1e6:.0f} TF)", flush=True)
