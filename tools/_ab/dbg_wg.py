import os, sys, torch
sys.path.insert(0, os.getcwd())
from reed_amd import ops
dev = torch.device("cuda")
tokens = 512
g = torch.Generator().manual_seed(17)
shapes = [(1152, 4608), (4608, 1152), (1152, 1152), (3456, 1152)]
probs, refs = [], []
for n_out, k_in in shapes:
    dy = torch.randn(tokens, n_out, generator=g).to(torch.bfloat16).to(dev)
    x = torch.randn(tokens, k_in, generator=g).to(torch.bfloat16).to(dev)
    out = torch.full((n_out * k_in + n_out,), float("nan"), device=dev)
    probs.append((dy, x, out[:n_out * k_in].view(n_out, k_in), out[n_out * k_in:], n_out, k_in))
    refs.append((dy.float().t() @ x.float(), dy.float().sum(0)))
assert ops.wgrad_group(probs, tokens)
torch.cuda.synchronize()
for p, (q, (rw, rb)) in enumerate(zip(probs, refs)):
    bad = ((q[3] - rb).abs() > 1e-2).nonzero().flatten().tolist()
    print("problem", p, shapes[p], "dw max err", (q[2] - rw).abs().max().item(), "db bad rows", len(bad), bad[:40])
    if bad:
        i = bad[0]
        print("   got", q[3][i].item(), "ref", rb[i].item(), "diff", (q[3][i]-rb[i]).item())
        # is got == partial sum over some k range?
        dyf = probs[p][0].float()
        for k0, k1 in ((0, 64), (0, 256), (64, 512), (0, 448), (32, 512), (0, 480)):
            print("   sum", k0, k1, dyf[k0:k1, i].sum().item())
