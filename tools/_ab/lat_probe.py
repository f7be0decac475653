import os, sys, torch
sys.path.insert(0, os.getcwd())
from reed_amd import ops
dev = torch.device("cuda")
D, Hm = 1152, 4608
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for N, K in ((Hm, D), (D, Hm)):
  for r in (1, 3, 7, 14, 28, 56):
    M = 256 * r
    x, w = bf(M, K), bf(N, K)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    res = []
    for tile in (256, 257):
        ops.gemm_force_tile(tile)
        ms = timeit(lambda: ops.linear_fwd(x, w, None, out))
        res.append(ms)
    ops.gemm_force_tile(0)
    tiles = r * ((N + 255) // 256)
    print(f"N={N} K={K} M={M} tiles={tiles}: 8-wave {res[0]*1e3:.1f} us, 4-wave {res[1]*1e3:.1f} us; per K-tile {res[0]*1e3/(K/64)/max(1,-(-tiles//256)):.2f} / {res[1]*1e3/(K/64)/max(1,-(-tiles//256)):.2f} us")
