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
    ws = None if os.environ.get("REED_ATTN_BWD") else torch.empty(ops.attention_bwd_ws_floats(b, T, H), device=dev)
    tf = timeit(lambda: ops.attention_fwd(qkv, o, lse, b, T, H, hd))
    tb = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, dqkv, b, T, H, hd, ws=ws))
    ff, fb = 4.0 * T * T * hd * b * H, 10.0 * T * T * hd * b * H
    bytes_f, bytes_b = M * D * 2 * 4, M * D * 2 * 8
    print(f"b={b:4d} fwd {tf*1e6:8.1f} us {ff/tf/1e12:7.1f} TFLOP/s {bytes_f/tf/1e12:5.2f} TB/s | "
          f"bwd {tb*1e6:8.1f} us {fb/tb/1e12:7.1f} TFLOP/s {bytes_b/tb/1e12:5.2f} TB/s")
