#!/usr/bin/env python
"""The 256x288 kernel (csrc/gemm288.hip, NT) against the other tiles on the two 4608-wide GEMMs of a block at b = 32 per GPU (and b = 16):
fc1 forward (GELU: two outputs) and the fc2 input gradient as an NT GEMM on W2^T (dGELU).  Event-timed us per call, 3 alternating
passes of 20; every form must give the bits of the 256x144 kernel.
usage: python tools/r6/t288.py [b ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reed_amd import ops
dev = torch.device("cuda"); D, Hm = 1152, 4608
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
for b in [int(a) for a in sys.argv[1:]] or [32, 16]:
    M = b * 256
    x, w, bias, r = bf(M, D), bf(Hm, D), bf(Hm), bf(M, Hm)
    outs = {}
    def fc1(tile):
        def f():
            ops.gemm_force_tile(tile)
            c, c2 = outs.setdefault(("fc1", tile), (torch.empty(M, Hm, dtype=torch.bfloat16, device=dev), torch.empty(M, Hm, dtype=torch.bfloat16, device=dev)))
            ops.gemm(ops.NT, ops.EPI_GELU, x, w, M, Hm, D, c, D, D, Hm, C2=c2, ldc2=Hm, bias=bias)
        return f
    def dfc2(tile):
        def f():
            ops.gemm_force_tile(tile)
            c, = outs.setdefault(("dfc2", tile), (torch.empty(M, Hm, dtype=torch.bfloat16, device=dev),))
            ops.gemm(ops.NT, ops.EPI_DGELU, x, w, M, Hm, D, c, D, D, Hm, R=r, ldr=Hm)
        return f
    for name, mk in (("fc1 forward + GELU", fc1), ("fc2 input gradient (NT on W^T) x dGELU", dfc2)):
        forms = [(144, "256x144"), (288, "256x288"), (257, "256^2 four-wave"), (0, "heuristic")]
        res = {t: [] for t, _ in forms}
        for rep in range(3):
            for t, _ in forms:
                fn = mk(t)
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    fn()
                e1.record(); torch.cuda.synchronize()
                res[t].append(e0.elapsed_time(e1) / 20 * 1e3)
        ops.gemm_force_tile(0)
        key = name.split()[0] if name.startswith("fc1") else "dfc2"
        key = "fc1" if name.startswith("fc1") else "dfc2"
        same = all(all(torch.equal(a, bb) for a, bb in zip(outs[(key, 144)], outs[(key, t)])) for t, _ in forms)
        flop = 2.0 * M * Hm * D
        print(f"b = {b} {name}: " + " | ".join(f"{lab} {' '.join(f'{v:.1f}' for v in res[t])} us ({flop / min(res[t]) / 1e6:.0f} TF)" for t, lab in forms)
              + f" | bit-identical: {same}", flush=True)
