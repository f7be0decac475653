import os, sys, torch
sys.path.insert(0, os.getcwd())
from reed_amd import ops
dev = torch.device("cuda")
M = 256 * 257
E = 1024
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
def timeit(fn, iters=100):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
h, qkv, o, u = bf(M, E), bf(M, 3 * E), bf(M, E), bf(M, 4 * E)
xa, xb = torch.randn(M, E, device=dev), torch.empty(M, E, device=dev)
wq, wp, w1, w2 = bf(3 * E, E), bf(E, E), bf(4 * E, E), bf(E, 4 * E)
bq, bp, b1, b2 = bf(3 * E), bf(E), bf(4 * E), bf(E)
gamma = torch.rand(E, device=dev)
ones = torch.ones(E, dtype=torch.bfloat16, device=dev)
cases = [
 ("qkv  bf16", 2.0*M*3*E*E, lambda: ops.gemm(ops.NT, ops.EPI_BF16, h, wq, M, 3*E, E, qkv, E, E, 3*E, bias=bq)),
 ("proj ls_res", 2.0*M*E*E, lambda: ops.gemm(ops.NT, ops.EPI_LS_RES, o, wp, M, E, E, xb, E, E, E, R=xa, ldr=E, bias=bp, gate=gamma)),
 ("proj gate_res(ones)", 2.0*M*E*E, lambda: ops.gemm(ops.NT, ops.EPI_GATE_RES, o, wp, M, E, E, xb, E, E, E, R=xa, ldr=E, bias=bp, gate=ones, ldgate=0, rows_per_gate=257)),
 ("fc1  gelu_erf", 2.0*M*4*E*E, lambda: ops.gemm(ops.NT, ops.EPI_GELU_ERF, h, w1, M, 4*E, E, None, E, E, 4*E, C2=u, ldc2=4*E, bias=b1)),
 ("fc1  qgelu", 2.0*M*4*E*E, lambda: ops.gemm(ops.NT, ops.EPI_QGELU, h, w1, M, 4*E, E, None, E, E, 4*E, C2=u, ldc2=4*E, bias=b1)),
 ("fc2  ls_res", 2.0*M*4*E*E, lambda: ops.gemm(ops.NT, ops.EPI_LS_RES, u, w2, M, E, 4*E, xb, 4*E, 4*E, E, R=xa, ldr=E, bias=b2, gate=gamma)),
]
for name, flop, fn in cases:
    r = []
    for tile in (256, 257):
        ops.gemm_force_tile(tile); r.append(timeit(fn))
    ops.gemm_force_tile(0)
    print(f"{name:22s}: 8-wave {r[0]:.4f} ms {flop/r[0]/1e9:7.1f} TF | 4-wave {r[1]:.4f} ms {flop/r[1]/1e9:7.1f} TF")
