#!/usr/bin/env python
"""Why does tools/bench_wgrad_group.py report 2.5-3 ms per grouped weight-gradient launch for the SECOND size it times in a process
(0.8 / 1.6 ms when the size comes first, and under rocprofv3 either way)?  Per size: host time per call, GPU time per launch (events),
with and without gemm_tn.hip's launches in between (VARIANT=a: only the default kernel; b: force_tile 128 first, as the bench does;
c: as b with a device synchronize + empty_cache between the sizes)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reed_amd import ops
dev = torch.device("cuda")
D, Hm, T = 1152, 4608, 256
shapes = [(D, Hm), (Hm, D), (D, D), (3 * D, D)]
variant = os.environ.get("VARIANT", "a")
for b in [int(v) for v in sys.argv[1:]] or [64, 256]:
    M = b * T
    probs = []
    for n_out, k_in in shapes:
        dy = (torch.randn(M, n_out, device=dev) * 0.05).to(torch.bfloat16)
        x = (torch.randn(M, k_in, device=dev) * 0.05).to(torch.bfloat16)
        out = torch.zeros(n_out * k_in + n_out, device=dev)
        probs.append((dy, x, out[:n_out * k_in].view(n_out, k_in), out[n_out * k_in:], n_out, k_in))
    for tile in ((0,) if variant == "a" else (128, 0)):
        ops.gemm_force_tile(tile)
        for _ in range(3):
            ops.wgrad_group(probs, M)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        host = []
        t_all = time.perf_counter()
        e0.record()
        for _ in range(20):
            t0 = time.perf_counter()
            ops.wgrad_group(probs, M)
            host.append((time.perf_counter() - t0) * 1e6)
        e1.record()
        t_issue = (time.perf_counter() - t_all) * 1e3
        torch.cuda.synchronize()
        t_wall = (time.perf_counter() - t_all) * 1e3
        print(f"variant {variant} b={b} tile {tile}: GPU {e0.elapsed_time(e1) / 20:.4f} ms per launch; host per call median {sorted(host)[10]:.0f} us, max {max(host):.0f} us; "
              f"20 calls issued in {t_issue:.2f} ms, done after {t_wall:.2f} ms; allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB, reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB", flush=True)
    ops.gemm_force_tile(0)
    if variant == "c":
        del probs, dy, x, out
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
