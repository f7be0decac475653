#!/usr/bin/env python
"""In-kernel clock and cycles per K-tile of the four-wave 256^2 GEMM under a full chip (diagnostic build with stamps around the
K loop: python tools/_ab/build_variant.py clk -DREED_CLK_PROBE; run with REED_HIP_LIB=tools/_ab/libreed_clk.so).
Each shape is launched back to back for ~2 s on random data, then the last launch's per-workgroup stamps are read.
usage: REED_HIP_LIB=tools/_ab/libreed_clk.so python tools/clk_probe.py [b]"""
import ctypes
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reed_amd import _lib, ops  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
D, Hm, T = 1152, 4608, 256
M = b * T
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)  # noqa: E731
L = _lib.load("bf16")
rd = L.reed_clk_probe_read
rd.restype = ctypes.c_int
rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
shapes = {"qkv": (3 * D, D), "proj": (D, D), "fc1": (Hm, D), "fc2": (D, Hm)}
ops.gemm_force_tile(257)
for lay in ("NT", "NN"):
    for name, (N, K) in shapes.items():
        if lay == "NT":
            x, w = bf(M, K), bf(N, K)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            fn = lambda: ops.linear_fwd(x, w, None, out)  # noqa: E731
            n_out, kk = N, K
        else:
            x, w = bf(M, N), bf(N, K)
            out = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
            fn = lambda: ops.linear_dgrad(x, w, out)  # noqa: E731
            n_out, kk = K, N
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 2.0:
            for _ in range(50):
                fn()
            torch.cuda.synchronize(); n += 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        nwg = ((M + 255) // 256) * ((n_out + 255) // 256)
        buf = (ctypes.c_ulonglong * (8 * nwg))()
        assert rd(buf, 8 * nwg) == 0
        W = [[buf[8 * i + j] for j in range(8)] for i in range(nwg)]
        full = [w for w in W if w[3] == 0 and w[1] > 0]
        clk = [w[0] / w[1] * 0.1 for w in full]           # GHz: shader cycles per 100 MHz tick
        cyc = [w[0] / w[2] for w in full]                 # shader cycles per K-tile
        us = [w[1] / 100.0 / w[2] for w in full]          # microseconds per K-tile
        pro = [(w[4] - w[6]) / 100.0 for w in full]       # kernel entry -> K loop (address set-up, first two K-tiles' DMA)
        epi = [((w[7] >> 16) - w[5]) / 100.0 for w in full]   # K loop end -> stores acknowledged
        loop = [w[1] / 100.0 for w in full]
        # gap between a workgroup's exit and the next workgroup's entry on the same CU
        percu = {}
        for w in W:
            if w[1] > 0:
                percu.setdefault(w[7] & 0xFFFF, []).append((w[6], w[7] >> 16))
        gaps = []
        for v in percu.values():
            v.sort()
            gaps += [(v[i + 1][0] - v[i][1]) / 100.0 for i in range(len(v) - 1)]
        med = statistics.median
        flop = 2.0 * M * n_out * kk
        print(f"{lay} {name:5s} K={kk:5d}: {ms:.4f} ms {flop / ms / 1e9:7.1f} TF | clock {med(clk):.3f} GHz | K-tile {med(cyc):7.1f} cyc "
              f"{med(us):.3f} us | full tile: prologue {med(pro):5.2f} + loop {med(loop):6.2f} + epilogue {med(epi):5.2f} us, then "
              f"{med(gaps) if gaps else float('nan'):5.2f} us until the CU's next workgroup enters ({len(percu)} CUs seen)", flush=True)
