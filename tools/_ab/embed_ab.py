"""Patch-embed and final-layer forward: the new forms against REED_EMBED_OLD=1, bit for bit, and their times (b = 256, XL/2)."""
import os, subprocess, sys, torch
sys.path.insert(0, ".")
if len(sys.argv) > 1:
    from reed_amd import ops
    dev = torch.device("cuda:0")
    B, C, HW, P, D, T = 256, 4, 32, 2, 1152, 256
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, C, HW, HW, generator=g).to(dev)
    w = (torch.randn(D, 16, generator=g) * 0.2).to(torch.bfloat16).to(dev)
    bias = torch.randn(D, generator=g).to(torch.bfloat16).to(dev)
    pos = torch.randn(T, D, generator=g).to(dev)
    tok = torch.empty(B * T, D, device=dev)
    xt = torch.randn(B * T, D, generator=g).to(dev)
    mod = torch.randn(B, 2 * D, generator=g).to(torch.bfloat16).to(dev)
    wf = (torch.randn(16, D, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    bf = torch.randn(16, generator=g).to(torch.bfloat16).to(dev)
    out = torch.empty(B, C, HW, HW, device=dev)
    mean, rstd = torch.empty(B * T, device=dev), torch.empty(B * T, device=dev)
    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10 * 1e3
    t1 = t(lambda: ops.patch_embed_fwd(x, w, bias, pos, tok, B, C, HW, P, D))
    t2 = t(lambda: ops.final_layer_fwd(xt, mod.data_ptr(), mod.data_ptr() + 2 * D, 2 * D, wf, bf, out, mean, rstd, B, T, D, C, P))
    dout = torch.randn(B, C, HW, HW, generator=g).to(dev)
    hbuf = torch.empty(B * T, D, dtype=torch.bfloat16, device=dev); dh = torch.empty_like(hbuf)
    dlin = torch.empty(B * T, 16, dtype=torch.bfloat16, device=dev)
    t3 = t(lambda: ops.final_layer_bwd_rows(dout, xt, mean, rstd, mod.data_ptr(), mod.data_ptr() + 2 * D, 2 * D, wf, hbuf, dlin, dh, B, T, D, C, P))
    torch.save({"tok": tok.cpu(), "out": out.cpu(), "mean": mean.cpu(), "rstd": rstd.cpu(), "hbuf": hbuf.cpu(), "dh": dh.cpu(), "dlin": dlin.cpu()}, sys.argv[1])
    print(f"patch_embed_fwd {t1:.1f} us, final_layer_fwd {t2:.1f} us, final_layer_bwd_rows {t3:.1f} us")
else:
    for tag, env in (("old", "1"), ("new", "0")):
        r = subprocess.run([sys.executable, __file__, f"/tmp/embed_{tag}.pt"], env=dict(os.environ, REED_EMBED_OLD=env), capture_output=True, text=True)
        print(tag, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:])
    a, b = torch.load("/tmp/embed_old.pt"), torch.load("/tmp/embed_new.pt")
    print({k: bool(torch.equal(a[k], b[k])) for k in a})
