import os, sys, torch
sys.path.insert(0, os.getcwd())
from reed_amd import ops
dev = torch.device("cuda")
D, Hm, T = 1152, 4608, 256
b = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = b * T
allshapes = {"fc2": (D, Hm), "fc1": (Hm, D), "proj": (D, D), "qkv": (3 * D, D), "sq2048": (2048, 2048), "sq1024": (1024, 1024)}
def run(names):
    probs = []
    for nm in names:
        n_out, k_in = allshapes[nm]
        dy = (torch.randn(M, n_out, device=dev) * 0.05).to(torch.bfloat16)
        x = (torch.randn(M, k_in, device=dev) * 0.05).to(torch.bfloat16)
        out = torch.zeros(n_out * k_in + n_out, device=dev)
        probs.append((dy, x, out[:n_out * k_in].view(n_out, k_in), out[n_out * k_in:], n_out, k_in))
    flop = sum(2.0 * M * allshapes[nm][0] * allshapes[nm][1] for nm in names)
    for tile in (128, 0):
        ops.gemm_force_tile(tile)
        for _ in range(3): ops.wgrad_group(probs, M)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.wgrad_group(probs, M)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"  {'+'.join(names):24s} tile={tile:3d}: {ms:.4f} ms {flop/ms/1e9:7.1f} TF; per K-tile(64) {ms*1e3/(M/64):.3f} us", flush=True)
    ops.gemm_force_tile(0)
for names in (["sq1024"], ["sq2048"], ["fc1"], ["fc2"], ["qkv"], ["proj"], ["fc1", "fc2"], ["fc2", "fc1", "proj", "qkv"]):
    run(names)
