#!/usr/bin/env python
"""In-kernel clock and cycles per K-tile of the grouped weight-gradient kernel (gemm_tn_group_kernel) under a full chip; needs
the diagnostic build (python tools/_ab/build_variant.py clk -DREED_CLK_PROBE; REED_HIP_LIB=tools/_ab/libreed_clk.so).
usage: REED_HIP_LIB=tools/_ab/libreed_clk.so python tools/clk_probe_tn.py [b]"""
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
shapes = [(D, Hm), (Hm, D), (D, D), (3 * D, D)]
M = b * T
probs = []
for n_out, k_in in shapes:
    dy = (torch.randn(M, n_out, device=dev) * 0.05).to(torch.bfloat16)
    x = (torch.randn(M, k_in, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.zeros(n_out * k_in + n_out, device=dev)
    probs.append((dy, x, out[:n_out * k_in].view(n_out, k_in), out[n_out * k_in:], n_out, k_in))
flop = sum(2.0 * M * n * k for n, k in shapes)
L = _lib.load("bf16")
rd = L.reed_clk_probe_read_tn
rd.restype = ctypes.c_int
rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
ops.wgrad_group(probs, M); torch.cuda.synchronize()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 2.0:
    for _ in range(20):
        ops.wgrad_group(probs, M)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.wgrad_group(probs, M)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
n = 512
buf = (ctypes.c_ulonglong * (4 * n))()
assert rd(buf, 4 * n) == 0
W = [[buf[4 * i + j] for j in range(4)] for i in range(n) if buf[4 * i + 1] > 0]
med = statistics.median
for wide in (0, 1):
    w = [x for x in W if x[3] == wide]
    if not w:
        continue
    clk = [x[0] / x[1] * 0.1 for x in w]
    cyc = [x[0] / x[2] for x in w]
    print(f"{'128x256' if wide else '256x128'} tiles: {len(w):4d} workgroups, clock {med(clk):.3f} GHz, {med(cyc):7.1f} cycles per K-tile of 32 "
          f"(MFMA floor: 32 MFMAs x 16 clk x 2 waves per SIMD = 1024 with two workgroups per CU), loop {med([x[1] for x in w]) / 100.0:8.1f} us")
print(f"b = {b}: {ms:.4f} ms per launch, {flop / ms / 1e9:.1f} TFLOP/s")
