cd /root/repo
python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ["REED_GEMM_RING"]="1"
from reed_amd import ops
dev=torch.device("cuda")
# correctness of the ring variant vs the 8-wave kernel (bit-identical expected)
for lay in ("NT","NN"):
  for (M,N,K) in ((512,256,128),(1024,1152,1152),(2048,4608,1152),(768,1152,4608),(300,384,256)):
    g=torch.Generator().manual_seed(1)
    if lay=="NT":
        x=(torch.randn(M,K,generator=g)*0.1).to(torch.bfloat16).to(dev); w=(torch.randn(N,K,generator=g)*0.1).to(torch.bfloat16).to(dev)
        fn=lambda o: ops.linear_fwd(x,w,None,o)
    else:
        x=(torch.randn(M,K,generator=g)*0.1).to(torch.bfloat16).to(dev); w=(torch.randn(K,N,generator=g)*0.1).to(torch.bfloat16).to(dev)
        fn=lambda o: ops.gemm(ops.NN, ops.EPI_BF16, x, w, M, N, K, o, K, N, N)
    outs=[]
    for tile in (256,257):
        ops.gemm_force_tile(tile); o=torch.full((M,N),float("nan"),dtype=torch.bfloat16,device=dev); fn(o); torch.cuda.synchronize(); outs.append(o)
    ops.gemm_force_tile(0)
    ref=(x.float()@(w.float().t() if lay=="NT" else w.float()))
    print(lay,M,N,K,"ring==8wave:",torch.equal(outs[0],outs[1]),"max err vs fp32",(outs[1].float()-ref).abs().max().item())
PY
for r in 0 1 0 1; do echo "== ring=$r"; REED_GEMM_RING=$r ITERS=300 timeout -k 10 200 python tools/bench_w4.py 256 2>&1 | grep -E "4-wave|sum" | sed 's/8-wave.*| 4-wave/4-wave/'; done
