"""Forced-tile timing of single GEMM shapes (us): which kernel the heuristic should take at small local batches."""
import sys, torch
sys.path.insert(0, ".")
from reed_amd import ops
dev = torch.device("cuda:0")
def t(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
for b in (32, 64, 128):
    M = b * 256
    for name, lay, N, K, epi in (("fwd qkv", ops.NT, 3456, 1152, ops.EPI_BF16), ("fwd fc1 gelu", ops.NT, 4608, 1152, ops.EPI_GELU),
                                 ("dgrad qkv", ops.NN, 1152, 3456, ops.EPI_BF16), ("dgrad fc1", ops.NN, 1152, 4608, ops.EPI_BF16),
                                 ("dgrad fc2 dgelu", ops.NN, 4608, 1152, ops.EPI_DGELU), ("dgrad proj", ops.NN, 1152, 1152, ops.EPI_BF16)):
        P = (torch.randn(M, K, device=dev) * 0.1).to(torch.bfloat16)
        Q = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16) if lay == ops.NT else (torch.randn(K, N, device=dev) * 0.05).to(torch.bfloat16)
        C = torch.empty(M, N, dtype=torch.bfloat16, device=dev); C2 = torch.empty_like(C); R = torch.randn(M, N, device=dev).to(torch.bfloat16)
        bias = torch.randn(N, device=dev).to(torch.bfloat16)
        kw = dict(bias=bias) if lay == ops.NT else {}
        if epi == ops.EPI_GELU: kw.update(C2=C2, ldc2=N)
        if epi == ops.EPI_DGELU: kw.update(R=R, ldr=N)
        out = []
        for force in (0, 144, 257, 258, 128):
            ops.gemm_force_tile(force)
            try:
                us = t(lambda: ops.gemm(lay, epi, P, Q, M, N, K, C, K, K if lay == ops.NT else N, N, **kw))
                out.append(f"{force}:{us:6.1f}")
            except Exception as e:
                out.append(f"{force}:  n/a")
        ops.gemm_force_tile(0)
        print(f"b={b:3d} {name:16s} " + "  ".join(out))
