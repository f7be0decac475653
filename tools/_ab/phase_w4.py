import ctypes, statistics, sys
import torch
sys.path.insert(0, "/root/repo")
from reed_amd import _lib, ops
b = 256
dev = torch.device("cuda")
D, Hm, T = 1152, 4608, 256
M = b * T
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
L = _lib.load("bf16")
rd = L.reed_clk_probe_read; rd.restype = ctypes.c_int; rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
ops.gemm_force_tile(257)   # one tile per workgroup
for lay, name, (N, K) in (("NT", "fc1", (Hm, D)), ("NT", "fc2", (D, Hm)), ("NN", "fc1", (Hm, D)), ("NN", "qkv", (3 * D, D))):
    if lay == "NT":
        x, w = bf(M, K), bf(N, K); out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        fn = lambda: ops.linear_fwd(x, w, None, out); n_out, kk = N, K
    else:
        x, w = bf(M, N), bf(N, K); out = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
        fn = lambda: ops.linear_dgrad(x, w, out); n_out, kk = K, N
    for _ in range(300): fn()
    torch.cuda.synchronize()
    nwg = ((M + 255) // 256) * ((n_out + 255) // 256)
    buf = (ctypes.c_ulonglong * (8 * nwg))()
    assert rd(buf, 8 * nwg) == 0
    W = [[buf[8 * i + j] for j in range(4)] for i in range(nwg) if buf[8 * i + 3] > 0]
    med = statistics.median
    print(f"{lay} {name} K={kk}: per K-tile (wave 0, full tiles, {len(W)} records): phase A {med([w[0]/w[3] for w in W]):7.1f} | waits + barrier {med([w[1]/w[3] for w in W]):6.1f} | phase B {med([w[2]/w[3] for w in W]):7.1f} cycles (64 MFMAs = 1024 each; three stamps ~120)")
