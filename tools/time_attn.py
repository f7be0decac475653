"""Attention forward / backward at the SiT-XL/2 shape (T 256, 16 heads, head_dim 72): time per launch, algorithmic
TFLOP/s (4 T^2 hd forward, 10 T^2 hd backward per head) and GB/s of the qkv/o/dqkv streams.
usage (GPU box): python tools/time_attn.py [b ...]"""
import sys
import torch
sys.path.insert(0, ".")
from reed_amd import ops

dev = torch.device("cuda")
T, H, hd = 256, 16, 72
D = H * hd


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


for b in [int(a) for a in sys.argv[1:]] or [32, 256]:
    M = b * T
    qkv = (torch.randn(M, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
    o = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
    do = torch.randn(M, D, device=dev).to(torch.bfloat16)
    dqkv = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(b, H, T, device=dev)
    import os
    ws = torch.empty(ops.attention_bwd_ws_floats(b, T, H), device=dev)
    tf = timeit(lambda: ops.attention_fwd(qkv, o, lse, b, T, H, hd))
    tb = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, dqkv, b, T, H, hd, ws=ws))
    # the form the engine runs where the dO GEMM is on the four-wave 256^2 kernel: delta from that GEMM's epilogue 13 (the GEMM
    # with and without it is timed beside)
    w = (torch.randn(D, D, device=dev) / D ** 0.5).to(torch.bfloat16)
    dy = torch.randn(M, D, device=dev).to(torch.bfloat16)
    dpart = torch.empty(H, 2, M, device=dev)
    line2 = ""
    if ws is not None and ops.dgrad_with_head_dots(dy, w, do, o, dpart, M, D, D, hd):
        tdp = timeit(lambda: ops.attention_bwd_dp(qkv, do, lse, dpart, dqkv, ws, b, T, H, hd))
        tg1 = timeit(lambda: ops.dgrad_with_head_dots(dy, w, do, o, dpart, M, D, D, hd))
        tg0 = timeit(lambda: ops.gemm(ops.NN, ops.EPI_BF16, dy, w, M, D, D, do, D, D, D))
        line2 = (f" | bwd with delta from the dO GEMM {tdp*1e6:8.1f} us + {max(0.0, tg1 - tg0)*1e6:5.1f} us in that GEMM "
                 f"({tg0*1e6:.1f} -> {tg1*1e6:.1f} us)")
    ff, fb = 4.0 * T * T * hd * b * H, 10.0 * T * T * hd * b * H
    bytes_f, bytes_b = M * D * 2 * 4, M * D * 2 * 8
    print(f"b={b:4d} fwd {tf*1e6:8.1f} us {ff/tf/1e12:7.1f} TFLOP/s {bytes_f/tf/1e12:5.2f} TB/s | "
          f"bwd {tb*1e6:8.1f} us {fb/tb/1e12:7.1f} TFLOP/s {bytes_b/tb/1e12:5.2f} TB/s" + line2)
