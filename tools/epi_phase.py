#!/usr/bin/env python
"""How long a tile's epilogue takes against how many other CUs are in theirs at the same moment (persistent four-wave GEMM,
stamped build: python tools/_ab/build_variant.py clk -DREED_CLK_PROBE).  Every tile records its
K loop's end and the moment its last store is issued on the 100 MHz clock; the epilogue of a tile is that interval, its
concurrency the number of tiles chip-wide whose interval contains its midpoint.
usage: REED_HIP_LIB=tools/_ab/libreed_clk.so python tools/epi_phase.py [b]"""
import ctypes
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reed_amd import _lib, ops  # noqa: E402
from reed_amd.ops import NT, NN, EPI_BF16, EPI_GELU, EPI_GATE_RES, EPI_DGELU  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
D, Hm, T = 1152, 4608, 256
M = b * T
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)  # noqa: E731
f32 = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
L = _lib.load("bf16")
rd = L.reed_clk_probe_read
rd.restype = ctypes.c_int
rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
med = statistics.median


def case(name):
    if name == "fqkv":
        x, w, out = bf(M, D), bf(3 * D, D), bf(M, 3 * D)
        return (lambda: ops.gemm(NT, EPI_BF16, x, w, M, 3 * D, D, out, D, D, 3 * D)), 3 * D, D
    if name == "fproj" or name == "ffc2":
        K = D if name == "fproj" else Hm
        x, w, xin, xout, y, gate, bias = bf(M, K), bf(D, K), f32(M, D), f32(M, D), bf(M, D), bf(b, 6 * D), bf(D)
        return (lambda: ops.gemm(NT, EPI_GATE_RES, x, w, M, D, K, xout, K, K, D, C2=y, ldc2=D, R=xin, ldr=D, bias=bias,
                                 gate=gate, ldgate=6 * D, rows_per_gate=T)), D, K
    if name == "ffc1":
        x, w, a1, u, bias = bf(M, D), bf(Hm, D), bf(M, Hm), bf(M, Hm), bf(Hm)
        return (lambda: ops.gemm(NT, EPI_GELU, x, w, M, Hm, D, a1, D, D, Hm, C2=u, ldc2=Hm, bias=bias)), Hm, D
    if name == "dfc2":
        dy, w, da, a1 = bf(M, D), bf(D, Hm), bf(M, Hm), bf(M, Hm)
        return (lambda: ops.gemm(NN, EPI_DGELU, dy, w, M, Hm, D, da, D, Hm, Hm, R=a1, ldr=Hm)), Hm, D
    if name == "dfc1":
        dy, w, dx = bf(M, Hm), bf(Hm, D), bf(M, D)
        return (lambda: ops.gemm(NN, EPI_BF16, dy, w, M, D, Hm, dx, Hm, D, D)), D, Hm
    raise KeyError(name)


for name in ("fqkv", "fproj", "ffc2", "ffc1", "dfc2", "dfc1"):
    fn, n_out, kk = case(name)
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    ntn = (n_out + 255) // 256
    nwg = ((M + 255) // 256) * ntn
    buf = (ctypes.c_ulonglong * (8 * nwg))()
    assert rd(buf, 8 * nwg) == 0
    tiles = []
    for i in range(nwg):
        w = [buf[8 * i + j] for j in range(8)]
        if w[1] == 0:
            continue
        tiles.append(dict(mode=w[3], start=w[6], l0=w[4], l1=w[5], end=w[7] >> 16, cu=w[7] & 0xFFFF))
    t_first = min(t["start"] for t in tiles)
    t_last = max(t["end"] for t in tiles)
    full = [t for t in tiles if t["mode"] == 0]
    iv = sorted((t["l1"], t["end"]) for t in tiles)
    for t in full:
        mid = (t["l1"] + t["end"]) // 2
        t["conc"] = sum(1 for a, z in iv if a <= mid <= z)
        t["epi"] = (t["end"] - t["l1"]) / 100.0
        t["loop"] = (t["l1"] - t["l0"]) / 100.0
        t["pro"] = (t["l0"] - t["start"]) / 100.0
    percu = {}
    for t in tiles:
        percu.setdefault(t["cu"], []).append(t)
    for v in percu.values():
        v.sort(key=lambda t: t["start"])
        for k, t in enumerate(v):
            t["pos"] = k
    print(f"{name:6s} {ms:.4f} ms | {len(tiles)} tiles on {len(percu)} CUs | kernel span {(t_last - t_first) / 100.0:.1f} us | full tiles: "
          f"prologue {med(t['pro'] for t in full):.2f} + loop {med(t['loop'] for t in full):.2f} + epilogue {med(t['epi'] for t in full):.2f} us (median)", flush=True)
    bins = [(0, 32), (32, 64), (64, 96), (96, 128), (128, 160), (160, 192), (192, 224), (224, 257)]
    line = "   epilogue us by concurrent epilogues:"
    for lo, hi in bins:
        e = [t["epi"] for t in full if lo <= t["conc"] < hi]
        if e:
            line += f"  [{lo},{hi}): {med(e):.2f} (n={len(e)})"
    print(line, flush=True)
    line = "   by position on its CU:"
    for k in range(max(t["pos"] for t in full) + 1):
        e = [t for t in full if t["pos"] == k]
        if e:
            line += f"  #{k}: {med(t['epi'] for t in e):.2f} us at {med(t['conc'] for t in e):.0f} conc (n={len(e)})"
    print(line, flush=True)
