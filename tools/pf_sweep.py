import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from reed_amd import _lib, ops
L = _lib.load(); dev = torch.device("cuda")
M, D, Hm = 65536, 1152, 4608
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)
def timeit(fn, it=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for name, (N, K) in {"fc1": (Hm, D), "fc2": (D, Hm), "qkv": (3 * D, D), "proj": (D, D)}.items():
    dy, x = bf(M, N), bf(M, K)
    out = torch.empty(N, K, device=dev); ws = torch.empty(8 * N * K, device=dev)
    L.reed_gemm_force_tile(256)
    ref = None
    for pf in (0, 3, 4, 5, 6, 0, 3, 4, 5, 6):
        L.reed_gemm_set_prefetch(pf)
        ms = timeit(lambda: ops.linear_wgrad(dy, x, out, split_k=8, ws=ws))
        if ref is None: ref = out.clone()
        ok = torch.equal(ref, out)
        print(f"TN {name} pf={pf:2d}: {ms:.4f} ms {2.0*M*N*K/ms/1e9:7.1f} TF/s same={ok}", flush=True)
L.reed_gemm_set_prefetch(-1); L.reed_gemm_force_tile(0)
