#!/usr/bin/env python
"""Would the input-gradient GEMMs run faster as NT GEMMs on a transposed copy of the weights?  dx[M, k_in] = dy[M, n_out] W[n_out, k_in]:
NN today (W read k-strided: two transposing LDS reads per fragment), NT with W^T [k_in, n_out] (both operands k-contiguous: one
ds_read_b128 per fragment).  Same products in the same order (bit-identical).  b = 256, plain bf16 store, the four shapes of a block,
alternating, event-timed us per launch.
usage: python tools/r6/nn_vs_nt.py [b]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reed_amd import ops
dev = torch.device("cuda"); M, D, Hm = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) * 256, 1152, 4608
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
for name, n_out, k_in in (("dgrad fc1", Hm, D), ("dgrad fc2", D, Hm), ("dgrad qkv", 3 * D, D), ("dgrad proj", D, D)):
    dy, w = bf(M, n_out), bf(n_out, k_in)
    wt = w.t().contiguous()
    o1 = torch.empty(M, k_in, dtype=torch.bfloat16, device=dev)
    o2 = torch.empty(M, k_in, dtype=torch.bfloat16, device=dev)
    nn = lambda: ops.gemm(ops.NN, ops.EPI_BF16, dy, w, M, k_in, n_out, o1, n_out, k_in, k_in)
    nt = lambda: ops.gemm(ops.NT, ops.EPI_BF16, dy, wt, M, k_in, n_out, o2, n_out, n_out, k_in)
    res = {"NN": [], "NT": []}
    for rep in range(3):
        for tag, fn in (("NN", nn), ("NT", nt)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            res[tag].append(e0.elapsed_time(e1) / 20 * 1e3)
    flop = 2.0 * M * n_out * k_in
    print(f"{name} (N = {k_in}, K = {n_out}): NN {' '.join(f'{v:.1f}' for v in res['NN'])} us ({flop / min(res['NN']) / 1e6:.0f} TF) | "
          f"NT on W^T {' '.join(f'{v:.1f}' for v in res['NT'])} us ({flop / min(res['NT']) / 1e6:.0f} TF) | bit-identical: {torch.equal(o1, o2)}", flush=True)
