import ctypes, os, statistics, sys, time
import torch
sys.path.insert(0, "/root/repo")
from reed_amd import _lib, ops
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
D, Hm, T = 1152, 4608, 256
shapes = [(D, Hm), (Hm, D), (D, D), (3 * D, D)]
M = b * T
probs = []
for n_out, k_in in shapes:
    dy = (torch.randn(M, n_out, device=dev) * 0.05).to(torch.bfloat16)
    x = (torch.randn(M, k_in, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.zeros(n_out * k_in + n_out, device=dev)
    probs.append((dy, x, out[:n_out * k_in].view(n_out, k_in), out[n_out * k_in:], n_out, k_in))
L = _lib.load("bf16")
rd = L.reed_clk_probe_read_tn; rd.restype = ctypes.c_int; rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(30): ops.wgrad_group(probs, M)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.wgrad_group(probs, M)
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1)/10:.4f} ms per launch (with stamps)")
n = 128 * 4
buf = (ctypes.c_ulonglong * (8 * n))()
assert rd(buf, 8 * n) == 0
W = [[buf[8 * i + j] for j in range(8)] for i in range(n) if buf[8 * i + 5] > 0]
names = ["DMA issue (6)", "24 tr reads issued + returned", "32 MFMAs issued", "wait: next K-tile's DMA landed", "barrier + loop"]
tot = 0
for k in range(5):
    v = statistics.median([w[k] / w[5] for w in W]); tot += v
    print(f"  {names[k]:34s} {v:8.1f} cycles per K-tile (median over {len(W)} waves)")
print(f"  sum {tot:.1f}")
