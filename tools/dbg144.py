"""gemm144.hip bring-up aid: identity / random operands on the smallest shapes (1, 2, 3 and 8 K-tiles, one and two column
tiles, both layouts) with the positions of the first mismatches — the tool that located the element-wise re-pack of
bf16 fragment vectors.  usage (GPU box): python tools/dbg144.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reed_amd import ops
dev = torch.device("cuda")
def run(lay, M, N, K, ident=False):
    g = torch.Generator().manual_seed(1)
    if ident:
        x = torch.eye(M, K).to(torch.bfloat16).to(dev)
        w = (torch.arange(N * K).reshape(N, K) % 251).float().to(torch.bfloat16).to(dev)
    else:
        x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.gemm_force_tile(144)
    if lay == "NT":
        ops.gemm(ops.NT, ops.EPI_BF16, x, w, M, N, K, out, K, K, N)
    else:
        ops.gemm(ops.NN, ops.EPI_BF16, x, w.t().contiguous(), M, N, K, out, K, N, N)
    torch.cuda.synchronize()
    ref = (x.float() @ w.float().t()).to(torch.bfloat16)
    bad = ~((out.float() - ref.float()).abs() <= 0.02 + 0.02 * ref.float().abs())
    print(lay, M, N, K, "ident" if ident else "rand", "bad", int(bad.sum()), "of", M * N)
    if bad.any():
        rows = bad.any(1).nonzero().flatten().tolist()
        cols = bad.any(0).nonzero().flatten().tolist()
        print("  bad rows", rows[:40], "... n=", len(rows))
        print("  bad cols", cols[:160], "... n=", len(cols))
        i, j = bad.nonzero()[0].tolist()
        print("  first", i, j, float(out[i, j]), float(ref[i, j]))
for lay in ("NT", "NN"):
    run(lay, 256, 144, 64, True)
    run(lay, 256, 144, 64)
    run(lay, 256, 144, 128)
    run(lay, 256, 144, 192)
    run(lay, 256, 144, 512)
    run(lay, 256, 288, 256, True)
