import sys, torch
sys.path.insert(0, ".")
from reed_amd import ops
dev = torch.device("cuda:0")
ops.use("fp32")
for lay, name in ((ops.NT, "NT"), (ops.NN, "NN"), (ops.TN, "TN")):
    for M, N, K in ((8192, 4608, 1152), (8192, 1152, 4608), (65536, 4608, 1152), (65536, 1152, 4608), (65536, 128, 1152)):
        P = torch.randn(M if lay != ops.TN else K, K if lay != ops.TN else M, device=dev)
        Q = torch.randn(N if lay == ops.NT else K, K if lay == ops.NT else N, device=dev)
        C = torch.empty(M, N, device=dev)
        ldp = P.shape[1]; ldq = Q.shape[1]
        f = lambda: ops.gemm(lay, ops.EPI_F32, P, Q, M, N, K, C, ldp, ldq, N)
        out = []
        for force in (0,):
            ops.gemm_force_tile(force)
            f(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            out.append(f"{'auto' if not force else force}: {ms:.3f} ms {2.0*M*N*K/ms/1e9:6.1f} TF")
        ops.gemm_force_tile(0)
        print(f"{name} M={M} N={N} K={K}: " + " | ".join(out))
