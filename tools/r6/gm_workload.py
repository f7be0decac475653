#!/usr/bin/env python
"""Round 6, verdict item 5: the forward / dgrad GEMMs of the block at b = 256 under another XCD deal of the persistent 256^2 kernel
(diagnostic build -DREED_TILE_GM_ENV: REED_TILE_GM = tile rows per XCD-local group; a group is GM rows x all tile columns walked
column by column, an XCD's 32 workgroups run 32 consecutive positions of its contiguous eighth: GM x 32/GM tiles in K-lockstep).
Prints event-timed us per launch; under rocprofv3 --pmc the same launches give FETCH_SIZE / GRBM_GUI_ACTIVE per kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reed_amd import ops
dev = torch.device("cuda"); M, D, Hm = 65536, 1152, 4608
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
x, w1, b1 = bf(M, D), bf(Hm, D), bf(Hm)
pre, act = torch.empty(M, Hm, dtype=torch.bfloat16, device=dev), torch.empty(M, Hm, dtype=torch.bfloat16, device=dev)
dx = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
w2, wq = bf(D, Hm), bf(3 * D, D)
da1 = torch.empty(M, Hm, dtype=torch.bfloat16, device=dev)
o3 = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev)
cases = {
    "fwd fc1 <0,14>": lambda: ops.linear_fwd(x, w1, b1, pre, epi=ops.EPI_GELU_G, act_out=act),
    "dgrad fc1 <1,0>": lambda: ops.gemm(ops.NN, ops.EPI_BF16, act, w1, M, D, Hm, dx, Hm, D, D),
    "dgrad fc2 <1,16>": lambda: ops.gemm(ops.NN, ops.EPI_MUL, x, w2, M, Hm, D, da1, D, Hm, Hm, R=pre, ldr=Hm),
    "fwd qkv <0,0>": lambda: ops.linear_fwd(x, wq, b1[:3 * D], o3),
}
n = int(os.environ.get("N", "10"))
for name, fn in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f"GM={os.environ.get('REED_TILE_GM', 'default')} {name}: {e0.elapsed_time(e1) / n * 1e3:.1f} us")
