"""HBM-bound row kernels of the SiT block at the XL/2 shape: achieved TB/s per launch (events on the launch stream).
usage (GPU box): python tools/time_rows.py [b ...]"""
import sys
import torch
sys.path.insert(0, ".")
from reed_amd import ops

dev = torch.device("cuda")
T, D = 256, 1152


def timeit(fn, iters=20):
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


for b in [int(a) for a in sys.argv[1:]] or [32, 64, 256]:
    M = b * T
    x = torch.randn(M, D, device=dev)
    mod = (torch.randn(b, 6 * D, device=dev) * 0.3).to(torch.bfloat16)
    h = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    dh = torch.randn(M, D, device=dev).to(torch.bfloat16)
    y = torch.randn(M, D, device=dev).to(torch.bfloat16)
    dx = torch.randn(M, D, device=dev)
    dy = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
    part = torch.empty(M // 16, 2, D, device=dev)
    pg, pd = torch.empty(M // 16, D, device=dev), torch.empty(M // 16, D, device=dev)
    mp = mod.data_ptr()
    n = M * D
    t = timeit(lambda: ops.ln_modulate_fwd(x, mp, mp + 2 * D, 6 * D, h, mean, rstd, M, D, T))
    print(f"b={b:4d} ln_modulate_fwd       {t*1e6:8.1f} us  {6 * n / t / 1e12:5.2f} TB/s")
    t = timeit(lambda: ops.ln_modulate_bwd(dh, x, mean, rstd, mp + 2 * D, 6 * D, dx, part, M, D, T))
    print(f"b={b:4d} ln_modulate_bwd       {t*1e6:8.1f} us  {14 * n / t / 1e12:5.2f} TB/s")
    t = timeit(lambda: ops.ln_modulate_bwd_gate(dh, x, mean, rstd, mp + 2 * D, 6 * D, dx, part, y, mp + 4 * D, 6 * D, dy,
                                                pg, pd, M, D, T))
    print(f"b={b:4d} ln_modulate_bwd_gate  {t*1e6:8.1f} us  {18 * n / t / 1e12:5.2f} TB/s")
    t = timeit(lambda: ops.gate_bwd(dx, y, mp + 4 * D, 6 * D, dy, pg, M, D, T, part_dy=pd))
    print(f"b={b:4d} gate_bwd              {t*1e6:8.1f} us  {8 * n / t / 1e12:5.2f} TB/s")
