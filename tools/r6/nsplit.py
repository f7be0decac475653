#!/usr/bin/env python
"""b = 32 per GPU: the two 4608-wide outputs of a block (fc1 forward, fc2 input gradient; M = 8192 tokens) are 32 x 18 = 576 tiles of
256^2 = 2.25 rounds of 256 CUs, so they ran on 256x144 tiles (1024 tiles = 4 rounds).  Would a COLUMN SPLIT do better: the first
16 tile columns (512 tiles = exactly 2 rounds) on the four-wave 256^2 kernel, the last 512 columns as a second launch on another
kernel?  Same products in the same order whichever kernel forms an element (bit-identical).  Event-timed us per call, alternating.
usage: python tools/r6/nsplit.py [b]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reed_amd import ops
dev = torch.device("cuda"); M, D, Hm = (int(sys.argv[1]) if len(sys.argv) > 1 else 32) * 256, 1152, 4608
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
x, w, bias = bf(M, D), bf(Hm, D), bf(Hm)
o = [torch.empty(M, Hm, dtype=torch.bfloat16, device=dev) for _ in range(2)]   # pre-activations (C)
u = [torch.empty(M, Hm, dtype=torch.bfloat16, device=dev) for _ in range(2)]   # activations (C2), as the engine's fc1 forward stores both


def whole(tile, i):
    def f():
        ops.gemm_force_tile(tile)
        ops.gemm(ops.NT, ops.EPI_GELU, x, w, M, Hm, D, o[i], D, D, Hm, C2=u[i], ldc2=Hm, bias=bias)
    return f


def split(n1, t1, t2, i):
    def f():
        ops.gemm_force_tile(t1)
        ops.gemm(ops.NT, ops.EPI_GELU, x, w, M, n1, D, o[i], D, D, Hm, C2=u[i], ldc2=Hm, bias=bias)
        ops.gemm_force_tile(t2)
        ops.gemm(ops.NT, ops.EPI_GELU, x, w[n1:], M, Hm - n1, D, o[i][:, n1:], D, D, Hm, C2=u[i][:, n1:], ldc2=Hm, bias=bias[n1:])
    return f


forms = [("256x144 tiles, 4 rounds (round 5's choice)", whole(144, 0)), ("dispatcher's column split (force_tile 259)", whole(259, 1)),
         ("256^2 four-wave, 3 rounds", whole(257, 1)),
         ("4096 cols 256^2 + 512 cols 128^2", split(4096, 257, 128, 1)),
         ("4096 cols 256^2 + 512 cols heuristic", split(4096, 257, 0, 1)),
         ("4096 cols 256^2 + 512 cols 256^2 (64 tiles)", split(4096, 257, 257, 1))]
res = {k: [] for k, _ in forms}
same = {}
for rep in range(3):
    for tag, fn in forms:
        o[1].zero_(); u[1].zero_()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        same[tag] = (torch.equal(o[0], o[1]) and torch.equal(u[0], u[1])) if tag != forms[0][0] else True
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        res[tag].append(e0.elapsed_time(e1) / 20 * 1e3)
ops.gemm_force_tile(0)
# the fc2 input gradient (NN on W2 [D, Hm], epilogue 16 = times the saved GELU'): default dispatch against the 256x144 tiles
dy, w2, gp = bf(M, D), bf(D, Hm), bf(M, Hm)
d = [torch.empty(M, Hm, dtype=torch.bfloat16, device=dev) for _ in range(2)]
dres = {144: [], 259: []}
for rep in range(3):
    for tile in (144, 259):
        ops.gemm_force_tile(tile)
        fn = lambda: ops.gemm(ops.NN, ops.EPI_MUL, dy, w2, M, Hm, D, d[tile == 259], D, Hm, Hm, R=gp, ldr=Hm)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        dres[tile].append(e0.elapsed_time(e1) / 20 * 1e3)
ops.gemm_force_tile(0)
print(f"fc2 input gradient M = {M}: 256x144 tiles {' '.join(f'{v:.1f}' for v in dres[144])} us | column split "
      f"{' '.join(f'{v:.1f}' for v in dres[259])} us | bit-identical: {torch.equal(d[0], d[1])}", flush=True)
flop = 2.0 * M * Hm * D
for tag, _ in forms:
    print(f"fc1 forward M = {M}: {tag}: {' '.join(f'{v:.1f}' for v in res[tag])} us ({flop / min(res[tag]) / 1e6:.0f} TF) bit-identical: {same[tag]}",
          flush=True)
