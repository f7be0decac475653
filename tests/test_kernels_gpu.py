"""Kernel-level parity on the GPU: each HIP kernel against a plain PyTorch fp32 (fp64 for the samplers) restatement
of the same op on the same inputs, plus bit-exact checks of the index bookkeeping against the reference goldens."""
import numpy as np
import pytest
import torch

from oracle import detfill
from tests.test_oracle_golden import load

pytestmark = pytest.mark.gpu


def bfr(x):
    return x.to(torch.bfloat16).float()


@pytest.mark.parametrize("B,T,D", [(2, 256, 1152), (3, 16, 128), (1, 64, 384), (2, 32, 1024), (1, 64, 768), (1, 16, 1280)])
def test_ln_modulate_fwd_bwd(dev, B, T, D):
    from reed_amd import ops
    g = torch.Generator().manual_seed(D + T)
    M = B * T
    x = (torch.randn(M, D, generator=g) * 2 + 0.3).to(dev)
    mod = (torch.randn(B, 3 * D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    shift, scale = mod[:, :D], mod[:, D:2 * D]
    h = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.ln_modulate_fwd(x, shift, scale, 3 * D, h, mean, rstd, M, D, T)
    xr = x.clone().requires_grad_(True)
    s1 = bfr(1 + scale.float()).repeat_interleave(T, 0).requires_grad_(True)
    sh = shift.float().repeat_interleave(T, 0).requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), eps=1e-6) * s1 + sh
    torch.testing.assert_close(h.float(), bfr(ref.detach()), atol=2e-2, rtol=1e-2)
    torch.testing.assert_close(mean, x.mean(-1), atol=1e-5, rtol=1e-5)
    # backward
    dh = torch.randn(M, D, generator=g).to(torch.bfloat16).to(dev)
    ref.backward(dh.float())
    dx = torch.full((M, D), 0.25, device=dev)
    part = torch.empty(M // 16, 2, D, device=dev)
    ops.ln_modulate_bwd(dh, x, mean, rstd, scale, 3 * D, dx, part, M, D, T)
    torch.testing.assert_close(dx - 0.25, xr.grad, atol=2e-4, rtol=2e-3)
    dmod = torch.zeros(B, 2 * D, dtype=torch.bfloat16, device=dev)
    ops.reduce_mod_parts([(part.data_ptr(), 2 * D, 0), (part.data_ptr() + 4 * D, 2 * D, D)], dmod, 2 * D, B, D, T // 16)
    torch.testing.assert_close(dmod[:, :D].float(), bfr(sh.grad.view(B, T, D).sum(1)), atol=3e-2, rtol=2e-2)
    torch.testing.assert_close(dmod[:, D:].float(), bfr(s1.grad.view(B, T, D).sum(1)), atol=3e-2, rtol=2e-2)
    # plain cast mode
    ops.ln_modulate_fwd(x, None, None, 0, h, None, None, M, D, T)
    assert torch.equal(h.float(), bfr(x))


@pytest.mark.parametrize("chunks", [16, 5, 24])
def test_reduce_mod_parts_vector_form_equals_the_scalar_form(dev, chunks):
    """The four-columns-per-thread form (aligned parts) and the one-column form (a part whose offset is not a multiple of 4
    columns) sum the chunks in the same order: bit-equal, and equal to the fp64 sum within bf16 rounding."""
    from reed_amd import ops
    B, D = 6, 1152
    g = torch.Generator().manual_seed(11)
    part = torch.randn(B * chunks, 2, D, generator=g).to(dev)
    a = torch.zeros(B, 2 * D + 8, dtype=torch.bfloat16, device=dev)
    b = torch.zeros(B, 2 * D + 8, dtype=torch.bfloat16, device=dev)
    ops.reduce_mod_parts([(part.data_ptr(), 2 * D, 0), (part.data_ptr() + 4 * D, 2 * D, D)], a, 2 * D + 8, B, D, chunks)
    ops.reduce_mod_parts([(part.data_ptr(), 2 * D, 2), (part.data_ptr() + 4 * D, 2 * D, D + 6)], b, 2 * D + 8, B, D, chunks)
    assert torch.equal(a[:, :D], b[:, 2:D + 2]) and torch.equal(a[:, D:2 * D], b[:, D + 6:2 * D + 6])
    ref = part.double().view(B, chunks, 2, D).sum(1)
    torch.testing.assert_close(a[:, :D].double(), ref[:, 0], atol=4e-2, rtol=1e-2)
    torch.testing.assert_close(a[:, D:2 * D].double(), ref[:, 1], atol=4e-2, rtol=1e-2)


def test_gate_bwd(dev):
    from reed_amd import ops
    B, T, D = 2, 64, 384
    M = B * T
    g = torch.Generator().manual_seed(3)
    dx = torch.randn(M, D, generator=g).to(dev)
    y = torch.randn(M, D, generator=g).to(torch.bfloat16).to(dev)
    gate = torch.randn(B, D, generator=g).to(torch.bfloat16).to(dev)
    dy = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
    part = torch.empty(M // 16, D, device=dev)
    pdy = torch.empty(M // 16, D, device=dev)
    ops.gate_bwd(dx, y, gate, D, dy, part, M, D, T, part_dy=pdy)
    db = torch.zeros(D, device=dev)
    ops.rowsum_f32(pdy, M // 16, db, D)
    torch.testing.assert_close(db, dy.float().sum(0), atol=1e-3, rtol=1e-4)
    tall = torch.randn(1000, 384, generator=g).to(dev)
    o2, wsr = torch.zeros(384, device=dev), torch.empty(16 * 384, device=dev)
    ops.rowsum_f32(tall, 1000, o2, 384, ws=wsr)
    torch.testing.assert_close(o2, tall.sum(0), atol=1e-3, rtol=1e-4)
    dg = bfr(dx)
    assert torch.equal(dy.float(), bfr(dg * gate.float().repeat_interleave(T, 0)))
    ref = bfr(dg * y.float()).view(B, T, D).sum(1)
    torch.testing.assert_close(part.view(B, T // 16, D).sum(1), ref, atol=1e-3, rtol=1e-4)


def test_ln_bwd_with_gate_matches_two_passes(dev):
    """reed_ln_modulate_bwd_gate == reed_ln_modulate_bwd followed by reed_gate_bwd on the finished dx."""
    from reed_amd import ops
    for B, T, D in ((2, 64, 384), (3, 16, 1152)):
        M = B * T
        g = torch.Generator().manual_seed(7 + D)
        x = (torch.randn(M, D, generator=g) * 2).to(dev)
        scale = (torch.randn(B, D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        h = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
        mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
        ops.ln_modulate_fwd(x, scale, scale, D, h, mean, rstd, M, D, T)
        dh = torch.randn(M, D, generator=g).to(torch.bfloat16).to(dev)
        dx0 = torch.randn(M, D, generator=g).to(dev)
        y = torch.randn(M, D, generator=g).to(torch.bfloat16).to(dev)
        gate = torch.randn(B, 2 * D, generator=g).to(torch.bfloat16).to(dev)[:, D:]
        # two passes
        dx_a, part_a = dx0.clone(), torch.empty(M // 16, 2, D, device=dev)
        dy_a = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
        pg_a, pd_a = torch.empty(M // 16, D, device=dev), torch.empty(M // 16, D, device=dev)
        ops.ln_modulate_bwd(dh, x, mean, rstd, scale, D, dx_a, part_a, M, D, T)
        ops.gate_bwd(dx_a, y, gate, 2 * D, dy_a, pg_a, M, D, T, part_dy=pd_a)
        # one pass
        dx_b, part_b = dx0.clone(), torch.empty(M // 16, 2, D, device=dev)
        dy_b = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
        pg_b, pd_b = torch.empty(M // 16, D, device=dev), torch.empty(M // 16, D, device=dev)
        ops.ln_modulate_bwd_gate(dh, x, mean, rstd, scale, D, dx_b, part_b, y, gate, 2 * D, dy_b, pg_b, pd_b, M, D, T)
        # same arithmetic; the two instantiations may contract multiply-adds differently (last-bit differences in dx,
        # hence the odd bf16 rounding flip downstream)
        torch.testing.assert_close(dx_a, dx_b, atol=1e-5, rtol=1e-5)
        torch.testing.assert_close(part_a, part_b, atol=1e-4, rtol=1e-5)
        assert ((dy_a.float() - dy_b.float()).abs() <= 2 ** -7 * dy_a.float().abs() + 1e-6).all()
        assert (dy_a != dy_b).float().mean().item() < 1e-3
        torch.testing.assert_close(pg_a, pg_b, atol=5e-2, rtol=1e-3)
        torch.testing.assert_close(pd_a, pd_b, atol=5e-2, rtol=1e-3)


def test_patch_embed_and_final_layer_index_maps_bit_exact(dev):
    """patchify order (c,pi,pj) on the way in, (pi,pj,c) on the way out, row-major tokens: exact vs the reference."""
    from reed_amd import ops
    gs = load("static")
    B, C, HW, P, D = 1, 4, 32, 2, 128
    T = 256
    # one-hot conv weights: token feature k = input element k of the patch (values < 256 are exact in bf16)
    w = torch.zeros(D, C * P * P)
    for k in range(16):
        w[k, k] = 1.0
    pos = torch.zeros(T, D, device=dev)
    flat = torch.arange(C * HW * HW)
    for lo in range(0, 4096, 256):   # sweep the input in exact-in-bf16 slabs
        x = torch.where((flat >= lo) & (flat < lo + 256), flat - lo + 1, torch.zeros_like(flat)).float().reshape(1, C, HW, HW)
        tok = torch.empty(T, D, device=dev)
        ops.patch_embed_fwd(x.to(dev), w.to(torch.bfloat16).to(dev), None, pos, tok, B, C, HW, P, D)
        got = tok[:, :16].cpu()
        src = torch.from_numpy(gs["patchify_idx"])          # [T,16] flat input index feeding (token,k)
        exp = torch.where((src >= lo) & (src < lo + 256), src - lo + 1, torch.zeros_like(src)).float()
        assert torch.equal(got, exp)
    # patchify kernel (order 0) and unpatchify order through the final layer
    xb = torch.empty(T, 16, dtype=torch.bfloat16, device=dev)
    xs = (flat % 251).float().reshape(1, C, HW, HW)
    ops.patchify_bf16(xs.to(dev), xb, B, C, HW, P, 0)
    assert torch.equal(xb.float().cpu(), xs.flatten()[torch.from_numpy(gs["patchify_idx"])])
    # final layer: zero modulation, W = one-hot rows -> out = unpatchify(bf16(LN(x))[:, :16])
    xt = torch.randn(T, D, generator=torch.Generator().manual_seed(1)).to(dev)
    mod = torch.zeros(1, 2 * D, dtype=torch.bfloat16, device=dev)
    wf = torch.zeros(16, D)
    for j in range(16):
        wf[j, j] = 1.0
    out = torch.empty(1, C, HW, HW, device=dev)
    ops.final_layer_fwd(xt, mod, mod[:, D:], 2 * D, wf.to(torch.bfloat16).to(dev), None, out, None, None, 1, T, D, C, P)
    hln = bfr(torch.nn.functional.layer_norm(xt, (D,), eps=1e-6))[:, :16].cpu()
    un = torch.from_numpy(gs["unpatchify_idx"]).flatten()    # out.flat[i] = lin.flat[un[i]]
    torch.testing.assert_close(out.cpu().flatten(), hln.flatten()[un], atol=2e-2, rtol=0)   # values (LayerNorm in between)
    # The index map itself, bit for bit and independent of any rounding: 12 passes, pass r carries bit r of the linear
    # position (token * 16 + j) as the SIGN of element (token, j). Columns 16..31 hold the negated pattern, so every row
    # sums to exactly 0, LayerNorm keeps each sign, and the one-hot W copies it through: decoding the signs of the output
    # must give the reference's unpatchify map exactly (sit.py:256-269, golden `unpatchify_idx`).
    pos_id = torch.arange(T * 16).reshape(T, 16)
    decoded = torch.zeros(C * HW * HW, dtype=torch.long)
    for r in range(12):
        sgn = ((pos_id >> r) & 1).float() * 2 - 1
        xs_ = torch.zeros(T, D)
        xs_[:, :16], xs_[:, 16:32] = sgn, -sgn
        ops.final_layer_fwd(xs_.to(dev), mod, mod[:, D:], 2 * D, wf.to(torch.bfloat16).to(dev), None, out, None, None, 1, T,
                            D, C, P)
        o = out.cpu().flatten()
        assert (o.abs() > 1.0).all()
        decoded |= (o > 0).long() << r
    assert torch.equal(decoded, un)
    # and the backward's patchify of the output gradient in the (pi, pj, c) order (reed_patchify_bf16 order 1) is its inverse
    gi = torch.empty(T, 16, dtype=torch.bfloat16, device=dev)
    for lo in range(0, 4096, 256):
        img = torch.where((flat >= lo) & (flat < lo + 256), flat - lo + 1, torch.zeros_like(flat)).float()
        ops.patchify_bf16(img.reshape(1, C, HW, HW).to(dev), gi, B, C, HW, P, 1)
        exp = torch.zeros(T * 16)
        exp[un] = img                                         # lin.flat[un[i]] = out.flat[i]
        assert torch.equal(gi.float().cpu().flatten(), exp)


def test_patch_embed_values_and_smallk_wgrad(dev):
    from reed_amd import ops
    B, C, HW, P, D = 3, 4, 16, 2, 384
    T, K = 64, 16
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, HW, HW, generator=g)
    w = (torch.randn(D, C, P, P, generator=g) * 0.2).to(torch.bfloat16)
    b = (torch.randn(D, generator=g) * 0.1).to(torch.bfloat16)
    pos = torch.randn(T, D, generator=g)
    tok = torch.empty(B * T, D, device=dev)
    ops.patch_embed_fwd(x.to(dev), w.to(dev), b.to(dev), pos.to(dev), tok, B, C, HW, P, D)
    ref = torch.nn.functional.conv2d(bfr(x), w.float(), b.float(), stride=P).flatten(2).transpose(1, 2)
    ref = bfr(ref) + pos
    torch.testing.assert_close(tok.cpu().view(B, T, D), ref, atol=2e-2, rtol=1e-2)
    # wgrad: dW[d,k] = sum_m bf16(dtok)[m,d] * patch[m,k]
    dtok = torch.randn(B * T, D, generator=g).to(dev)
    xb = torch.empty(B * T, K, dtype=torch.bfloat16, device=dev)
    ops.patchify_bf16(x.to(dev), xb, B, C, HW, P, 0)
    ws = torch.empty(ops.smallk_ws_floats(D, K), device=dev)
    dw, db = torch.zeros(D, K, device=dev), torch.zeros(D, device=dev)
    ops.smallk_wgrad(dtok, True, xb, ws, dw, db, None, B * T, D, K, 0, False)
    torch.testing.assert_close(dw, bfr(dtok).t() @ xb.float(), atol=2e-3, rtol=1e-3)
    torch.testing.assert_close(db, bfr(dtok).sum(0), atol=2e-3, rtol=1e-3)
    ops.smallk_wgrad(dtok, True, xb, ws, dw, db, None, B * T, D, K, 0, True)
    torch.testing.assert_close(dw, 2 * (bfr(dtok).t() @ xb.float()), atol=4e-3, rtol=1e-3)


def test_timestep_sinusoid_and_label_cond(dev):
    from reed_amd import ops
    gs = load("static")
    t = torch.from_numpy(gs["sinus_t"]).to(dev)
    out = torch.empty(len(t), 256, dtype=torch.bfloat16, device=dev)
    ops.timestep_sinusoid(t, out, len(t))
    ref = torch.from_numpy(gs["sinus"])
    torch.testing.assert_close(out.float().cpu(), bfr(ref), atol=8e-3, rtol=0)   # one bf16 ulp at |x|<=1
    # label embedding + conditioning (integer bookkeeping bit-exact)
    B, D, NC = 6, 128, 10
    g = torch.Generator().manual_seed(2)
    table = torch.randn(NC + 1, D, generator=g).to(dev)
    temb = torch.randn(B, D, generator=g).to(torch.bfloat16).to(dev)
    labels = torch.tensor([0, 9, 3, 3, 7, 1], device=dev)
    drop = torch.tensor([0, 1, 0, 0, 1, 0], dtype=torch.uint8, device=dev)
    eff = torch.empty(B, dtype=torch.int64, device=dev)
    c, sc = torch.empty(B, D, device=dev), torch.empty(B, D, dtype=torch.bfloat16, device=dev)
    ops.label_cond(labels, drop, NC, table, temb, eff, c, sc, B, D)
    assert eff.tolist() == [0, 10, 3, 3, 10, 1]
    cref = temb.float() + table[eff]
    assert torch.equal(c, cref)
    torch.testing.assert_close(sc.float(), bfr(torch.nn.functional.silu(cref)), atol=1e-2, rtol=1e-2)
    # backward: deterministic scatter-add with duplicate labels
    ds = torch.randn(B, D, generator=g).to(dev)
    dt = torch.empty(B, D, dtype=torch.bfloat16, device=dev)
    dtab = torch.zeros(NC + 1, D, device=dev)
    ops.label_cond_bwd(ds, c, eff, dt, dtab, B, D)
    cr = cref.clone().requires_grad_(True)
    torch.nn.functional.silu(cr).backward(bfr(ds))
    torch.testing.assert_close(dt.float(), bfr(cr.grad), atol=1e-2, rtol=1e-2)
    exp = torch.zeros_like(dtab).index_add_(0, eff, cr.grad)
    torch.testing.assert_close(dtab, exp, atol=1e-5, rtol=1e-5)


def test_loss_kernels(dev):
    from reed_amd import ops
    from reed_amd.loss import _Cosine, _MSE
    g = torch.Generator().manual_seed(9)
    B, per = 5, 4096
    x, n = torch.randn(B, per, generator=g).to(dev), torch.randn(B, per, generator=g).to(dev)
    t = torch.rand(B, generator=g).to(dev)
    for pt in (0, 1):
        xt, tg = torch.empty_like(x), torch.empty_like(x)
        ops.interpolant(x, n, t, xt, tg, B, per, pt)
        tt = t[:, None]
        if pt == 0:
            a, s, da, ds = 1 - tt, tt, -1.0, 1.0
        else:
            h = np.pi / 2
            a, s, da, ds = torch.cos(tt * h), torch.sin(tt * h), -h * torch.sin(tt * h), h * torch.cos(tt * h)
        torch.testing.assert_close(xt, a * x + s * n, atol=2e-6, rtol=1e-6)
        torch.testing.assert_close(tg, da * x + ds * n, atol=2e-6, rtol=1e-6)
    out = x.clone().requires_grad_(True)
    l = _MSE.apply(out, n)
    torch.testing.assert_close(l, ((x - n) ** 2).mean(1), atol=1e-5, rtol=1e-5)
    w = torch.rand(B, generator=g).to(dev)
    (l * w).sum().backward()
    torch.testing.assert_close(out.grad, w[:, None] * 2 * (x - n) / per, atol=1e-7, rtol=1e-5)
    # cosine alignment: 3-D (image tokens) and 2-D (pooled text)
    for shape in ((B, 16, 256), (B, 512)):
        zt = torch.randn(*shape, generator=g).to(torch.bfloat16).to(dev).requires_grad_(True)
        z = torch.randn(*shape, generator=g).to(dev)
        cur = _Cosine.apply(zt, z)
        zr = zt.detach().float().requires_grad_(True)
        a_, b_ = torch.nn.functional.normalize(zr, dim=-1), torch.nn.functional.normalize(z, dim=-1)
        if len(shape) == 2:
            a_, b_ = a_.unsqueeze(1), b_.unsqueeze(1)
        ref = -(a_ * b_).sum(-1).mean(-1)
        torch.testing.assert_close(cur, ref, atol=1e-5, rtol=1e-4)
        (cur * w).sum().backward()
        (ref * w).sum().backward()
        torch.testing.assert_close(zt.grad.float(), bfr(zr.grad), atol=1e-4, rtol=2e-2)
    # sample_posterior
    mom = torch.randn(B, 8, 8, 8, generator=g).to(dev)
    eps = torch.randn(B, 4, 8, 8, generator=g).to(dev)
    o = torch.empty_like(eps)
    ops.sample_posterior(mom, eps, o, B, 4 * 64, 0.18215, 0.0)
    torch.testing.assert_close(o, (mom[:, :4] + mom[:, 4:] * eps) * 0.18215, atol=1e-6, rtol=1e-6)


def test_siloss_hip_vs_reference_units(dev):
    """reed_amd.loss.SILoss (HIP interpolant / MSE / cosine kernels + the reference's time-weight broadcasting) on the
    reference's own unit vectors (tools/gen_golden.py:g_loss_units): 6 time schedules x 2 path types with two encoders
    (image tokens + pooled text), the zero-weight branch, single-encoder keying, and the lognormal t transform
    (image/loss.py:118-151,160-168,204-237). fp32 arithmetic on both sides: 2e-6 relative."""
    from reed_amd.loss import SILoss
    g = load("loss_units")
    from tests.test_oracle_golden import inputs
    B = 6
    x, noise, t, y, _, zs = inputs(B, 4, 8, 21, [(32, "i"), (16, "t")], 16, 10)
    zt = [detfill.normal((B, 16, 32), 901).bfloat16().to(dev), detfill.normal((B, 16), 902).bfloat16().to(dev)]
    vel = detfill.normal((B, 4, 8, 8), 903).to(dev)
    zs_d = [z.to(dev) for z in zs]
    worst = 0.0
    for sched in ["constant", "linear", "cosine", "sigmoid", "loglinear", "cutoff"]:
        for path in ["linear", "cosine"]:
            lf = SILoss(path_type=path, enc_names=["clip", "text_embeds_qwenvl"],
                        loss_weights={"clip": 1.0, "text_embeds_qwenvl": 0.5}, time_schedule=sched, cutoffs=[0.2, 0.8])
            o = lf(lambda xx, tt, **k: (vel + 0.1 * xx, zt), x.to(dev), dict(y=y.to(dev)), zs=zs_d, time_input=t, noises=noise)
            for k in ("denoising_loss", "proj_loss", "img_proj_loss", "text_proj_loss"):
                got, ref = torch.as_tensor(o[k]).detach().cpu().numpy(), g[f"{sched}.{path}.{k}"]
                worst = max(worst, float(np.max(np.abs(got - ref) / (np.abs(ref) + 1e-6))))
                np.testing.assert_allclose(got, ref, rtol=2e-6, atol=2e-7, err_msg=f"{sched}.{path}.{k}")
    print("SILoss HIP vs reference units: worst relative deviation", worst)
    lf = SILoss(enc_names=["text_embeds_qwenvl"], loss_weights={"text_embeds_qwenvl": 0.0}, time_schedule="linear")
    o = lf(lambda xx, tt, **k: (vel, [zt[0]]), x.to(dev), dict(y=y.to(dev)), zs=[zs_d[0]], time_input=t, noises=noise)
    np.testing.assert_allclose(float(o["proj_loss"]), g["zero_weight.proj_loss"], rtol=2e-6)
    np.testing.assert_allclose(float(o["img_proj_loss"]), g["zero_weight.img_proj_loss"], rtol=2e-6)
    assert o["text_proj_loss"] == 0.0
    # lognormal weighting: t = sigma / (1 + sigma) or (2 / pi) atan(sigma), sigma = exp(N(0,1)) drawn on the CPU generator
    rn = detfill.normal((B, 1, 1, 1), 77)
    o_randn = torch.randn
    for path in ["linear", "cosine"]:
        cap = {}
        torch.randn = lambda *s, **k: rn.clone()
        try:
            lf = SILoss(path_type=path, weighting="lognormal", enc_names=["dinov2"], loss_weights={"dinov2": 1.0})
            lf(lambda xx, tt, **k: (cap.setdefault("t", tt), (vel, [zt[0]]))[1], x.to(dev), dict(y=y.to(dev)), zs=[zs_d[0]],
               noises=noise)
        finally:
            torch.randn = o_randn
        np.testing.assert_allclose(cap["t"].cpu().numpy(), g[f"lognormal.{path}.t"], rtol=1e-6)


def test_label_out_of_range_is_reported(dev):
    """A label outside the embedding table (nn.Embedding raises, sit.py:98) must neither read nor scatter outside the
    table (it lives inside the flat parameter arena): the kernel uses row 0, sets the device flag, and
    Engine.check_errors() / the samplers raise."""
    from reed_amd import ops, samplers
    from tests.test_model_gpu import build_hip_model
    from tests.test_oracle_golden import tiny_cfg
    B, D, NC = 4, 128, 10
    table = torch.randn(NC + 1, D, generator=torch.Generator().manual_seed(1)).to(dev)
    temb = torch.zeros(B, D, dtype=torch.bfloat16, device=dev)
    labels = torch.tensor([0, 11, -3, 10], device=dev)
    eff = torch.empty(B, dtype=torch.int64, device=dev)
    c, sc = torch.empty(B, D, device=dev), torch.empty(B, D, dtype=torch.bfloat16, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.label_cond(labels, None, NC, table, temb, eff, c, sc, B, D, table_rows=NC + 1, err=err)
    assert eff.tolist() == [0, 0, 0, 10] and int(err.item()) == 1
    assert torch.equal(c[1], table[0]) and torch.equal(c[3], table[10])
    err.zero_()
    ops.label_cond(labels.clamp(0, 10), None, NC, table, temb, eff, c, sc, B, D, table_rows=NC + 1, err=err)
    assert int(err.item()) == 0
    m = build_hip_model(tiny_cfg(), dev, 3).eval()     # num_classes = 10 (+ null row)
    x = torch.randn(2, 4, 8, 8, device=dev)
    with torch.no_grad():
        m(x, torch.full((2,), 0.5, device=dev), torch.tensor([3, 10], device=dev))
    m.engine().check_errors()
    with torch.no_grad():
        m(x, torch.full((2,), 0.5, device=dev), torch.tensor([3, 11], device=dev))
    with pytest.raises(IndexError):
        m.engine().check_errors()
    m.engine().check_errors()                           # the flag is cleared once reported
    with pytest.raises(IndexError):                     # CFG's hard-coded null id 1000 on an 11-row table
        samplers.euler_sampler(m, x, torch.tensor([1, 2], device=dev), num_steps=2, cfg_scale=1.5)
    with pytest.raises(IndexError):
        samplers.euler_sampler(m, x, torch.tensor([1, 12], device=dev), num_steps=2)


def test_fused_optimizer_vs_reference_toy(dev):
    """clip_grad_norm_(1.0) + AdamW(wd=0.01) + EMA: golden from torch's own implementations (tools/gen_golden.py)."""
    from reed_amd import ops
    g = load("optim_toy")
    shapes = [(5, 7), (11,), (3, 4, 2)]
    sizes = [int(np.prod(s)) for s in shapes]
    offs = [0, 36, 48]          # 4-aligned segment offsets (35 -> 36, 11 -> 12)
    n = 72
    p = torch.zeros(n)
    for i, s in enumerate(shapes):
        p[offs[i]:offs[i] + sizes[i]] = detfill.normal(s, 70 + i).flatten()
    p = p.to(dev)
    ema, m, v = p.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    shadow = torch.empty(n, dtype=torch.bfloat16, device=dev)
    partial, nc = torch.empty(64, device=dev), torch.empty(2, device=dev)
    for s in range(3):
        gr = torch.zeros(n)
        for i, sh in enumerate(shapes):
            gr[offs[i]:offs[i] + sizes[i]] = (detfill.normal(sh, 80 + 10 * s + i) * (3.0 if s == 0 else 0.1)).flatten()
        gr = gr.to(dev)
        ops.grad_sqnorm(gr, n, partial, 64)
        ops.clip_finalize(partial, 64, 1.0, nc)
        np.testing.assert_allclose(float(nc[0]), g[f"gn{s}"], rtol=1e-6)
        ops.adamw_ema(p, gr, m, v, ema, shadow, n, n, nc, 1e-2, 0.9, 0.999, 1e-8, 0.01, 1 - 0.9 ** (s + 1),
                      1 - 0.999 ** (s + 1), 0.99)
        for i in range(3):
            np.testing.assert_allclose(p[offs[i]:offs[i] + sizes[i]].cpu().numpy(), g[f"p{s}_{i}"].flatten(), rtol=2e-6, atol=5e-7)  # fp32 fma-contraction noise
            np.testing.assert_allclose(ema[offs[i]:offs[i] + sizes[i]].cpu().numpy(), g[f"e{s}_{i}"].flatten(), rtol=2e-6, atol=5e-7)
    assert torch.equal(shadow.float(), bfr(p))


def test_sampler_kernels_bit_exact_fp64(dev):
    from reed_amd import ops
    g = torch.Generator().manual_seed(4)
    n = 3 * 4 * 8 * 8
    x = torch.randn(n, generator=g, dtype=torch.float64).to(dev)
    mo = torch.randn(2 * n, generator=g).to(dev)
    dprev = torch.randn(n, generator=g, dtype=torch.float64).to(dev)
    xin = torch.empty(2 * n, device=dev)
    ops.sampler_input(x, xin, n, True)
    assert torch.equal(xin[:n], x.float()) and torch.equal(xin[n:], x.float())
    dt, s = -1.0 / 250, 1.7
    dc, du = mo[:n].double(), mo[n:].double()
    d = du + s * (dc - du)
    out, dst = torch.empty_like(x), torch.empty_like(x)
    ops.sampler_update(x, mo, None, dst, out, n, True, s, dt, 1.0, 0.0)
    assert torch.equal(dst, d) and torch.equal(out, x + dt * d)
    ops.sampler_update(x, mo, dprev, None, out, n, True, s, dt, 0.5, 0.5)
    assert torch.equal(out, x + dt * (0.5 * dprev + 0.5 * d))
    ops.sampler_update(x, mo, None, None, out, n, False, 1.0, dt, 1.0, 0.0)
    assert torch.equal(out, x + dt * mo[:n].double())
    # SDE step, linear path
    eps = torch.randn(n, generator=g, dtype=torch.float64).to(dev)
    t = torch.tensor(0.6160, dtype=torch.float64)
    dtt = torch.tensor(-0.0039, dtype=torch.float64)
    ops.sde_update(x, mo, eps, out, n, True, s, float(t), float(dtt), 0, False)

    def drift(v):
        a, da, sg, dsg = 1 - t, -1.0, t, 1.0
        r = a / da
        var = sg ** 2 - r * dsg * sg
        return v - 0.5 * (2 * t) * ((r * v - x.cpu()) / var)
    dcc, duu = drift(mo[:n].double().cpu()), drift(mo[n:].double().cpu())
    dd = duu + s * (dcc - duu)
    ref = x.cpu() + dd * dtt + torch.sqrt(2 * t) * (eps.cpu() * torch.sqrt(torch.abs(dtt)))
    torch.testing.assert_close(out.cpu(), ref, atol=1e-13, rtol=1e-13)


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_samplers_end_to_end_vs_reference(dev, precision):
    """Euler / Heun / Euler-Maruyama with (interval) CFG on a tiny SiT: HIP (bf16 model, fp64 state) vs the reference
    (fp32 model, fp64 state). 6 steps. The deviation is the bf16 evaluation error integrated over the trajectory
    (tests/test_model_gpu.py::test_long_horizon_heun_cfg_drift_s2 measures it over 50 steps at S/2 size); the bars are
    relative to the latents' scale and printed beside the measured values."""
    from reed_amd import samplers
    from tests.test_model_gpu import build_hip_model
    from tests.test_oracle_golden import tiny_cfg
    g = load("samplers")
    cfg = tiny_cfg(num_classes=1000)
    m = build_hip_model(cfg, dev, 5).eval()
    m.precision = precision   # fp16 = the sampling build (generate.py's default; the reference evaluates in fp32 / TF32)
    # measured (round 2), max abs on latents of scale 3.6: bf16 4.5e-3 .. 7.1e-3, fp16 6.7e-4 .. 1.06e-3 (x 6-7 smaller)
    tol = {"bf16": (1.5e-2, 1.5e-2), "fp16": (2.5e-3, 2.5e-3)}[precision]
    z = detfill.normal((3, 4, 8, 8), 41).to(dev)
    y = torch.tensor([3, 500, 999], device=dev)
    cfgs = {"euler": dict(heun=False, cfg_scale=1.0), "heun": dict(heun=True, cfg_scale=1.0),
            "euler_cfg": dict(heun=False, cfg_scale=2.5), "heun_cfg": dict(heun=True, cfg_scale=1.5),
            "heun_cfg_interval": dict(heun=True, cfg_scale=3.0, guidance_low=0.3, guidance_high=0.75)}
    for name, c in cfgs.items():
        out = samplers.euler_sampler(m, z, y, num_steps=6, prediction="v", **c)
        assert out.dtype == torch.float64 and out.shape == z.shape
        ref = torch.from_numpy(g[name])
        err = (out.cpu() - ref).abs().max().item()
        print(f"sampler {name} [{precision}]: max abs deviation {err:.3e}, latent scale {ref.abs().max().item():.2f}")
        assert err < tol[0], (name, err)
    eps = [detfill.normal((3, 4, 8, 8), 600 + i).double() for i in range(8)]
    for name, c in {"sde": dict(cfg_scale=1.0), "sde_cfg": dict(cfg_scale=2.0, guidance_high=0.9),
                    "sde_cosine": dict(cfg_scale=1.0, path_type="cosine")}.items():
        out = samplers.euler_maruyama_sampler(m, z, y, num_steps=6, noises=eps, **c)
        err = (out.cpu() - torch.from_numpy(g[name])).abs().max().item()
        print(f"sampler {name} [{precision}]: max abs deviation {err:.3e}")
        assert err < tol[1], (name, err)


def test_fp16_library_gemm_and_attention(dev):
    """libreed_hip_f16.so (the same sources with IEEE-half operands): NT GEMM with bias + GELU and attention against fp32
    references on fp16-rounded inputs — the error must be at fp16's resolution (2^-11), i.e. ~8x below the bf16 build's."""
    from reed_amd import ops
    from tests.test_attention_gpu import _ref
    g = torch.Generator().manual_seed(3)
    M, N, K = 512, 1152, 1152
    errs = {}
    for prec in ("bf16", "fp16"):
        dt = ops.half_dtype(prec)
        x = torch.randn(M, K, generator=g).to(dt).to(dev)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dt).to(dev)
        b = (torch.randn(N, generator=g) * 0.1).to(dt).to(dev)
        pre, act = torch.empty(M, N, dtype=dt, device=dev), torch.empty(M, N, dtype=dt, device=dev)
        prev = ops.use(prec)
        try:
            ops.linear_fwd(x, w, b, pre, epi=ops.EPI_GELU, act_out=act)
            B, T, H, hd = 2, 256, 4, 72
            qkv = torch.randn(B, T, 3, H, hd, generator=g).to(dt).to(dev)
            o = torch.empty(B, T, H * hd, dtype=dt, device=dev)
            lse = torch.empty(B, H, T, device=dev)
            ops.attention_fwd(qkv, o, lse, B, T, H, hd)
        finally:
            ops.use(prev)
        ref = x.float() @ w.float().t() + b.float()
        ro, rl = _ref(qkv, B, T, H, hd)
        errs[prec] = ((pre.float() - ref).abs().max().item(), (o.float() - ro).abs().max().item())
        torch.testing.assert_close(act.float(), torch.nn.functional.gelu(pre.float(), approximate="tanh"), atol=2e-2, rtol=2e-2)
        torch.testing.assert_close(lse, rl, atol=2e-3, rtol=1e-4)
    print("GEMM / attention max abs error vs fp32: bf16 build", errs["bf16"], " fp16 build", errs["fp16"])
    assert errs["fp16"][0] < 0.25 * errs["bf16"][0] and errs["fp16"][1] < 0.25 * errs["bf16"][1]
    # the IEEE-half build of every 256-wide tile kernel (eight waves, four 128x128 waves with asm MFMAs, 256x144), forward
    # and input-gradient layouts, ragged last column tile (N = 1152), against fp32 and against each other (bit-identical)
    M, N, K = 1024, 1152, 1152
    x = torch.randn(M, K, generator=g).to(torch.float16).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.float16).to(dev)
    wn = (torch.randn(K, N, generator=g) / K ** 0.5).to(torch.float16).to(dev)
    outs = {}
    prev = ops.use("fp16")
    try:
        for tile in (256, 257, 258, 144):
            ops.gemm_force_tile(tile)
            o1 = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
            o2 = torch.full((M, N), float("nan"), dtype=torch.float16, device=dev)
            ops.linear_fwd(x, w, None, o1)
            ops.gemm(ops.NN, ops.EPI_BF16, x, wn, M, N, K, o2, K, N, N)
            outs[tile] = (o1, o2)
    finally:
        ops.gemm_force_tile(0)
        ops.use(prev)
    torch.testing.assert_close(outs[257][0].float(), x.float() @ w.float().t(), atol=4e-3, rtol=2e-3)
    torch.testing.assert_close(outs[257][1].float(), x.float() @ wn.float(), atol=4e-3, rtol=2e-3)
    assert torch.equal(outs[256][0], outs[257][0]) and torch.equal(outs[256][1], outs[257][1])
    assert torch.equal(outs[258][0], outs[257][0]) and torch.equal(outs[258][1], outs[257][1])   # persistent form
    torch.testing.assert_close(outs[144][0].float(), outs[257][0].float(), atol=4e-3, rtol=2e-3)


@pytest.mark.parametrize("B,D,HW", [(40, 1152, 32), (3, 384, 32), (7, 1024, 32), (3, 384, 8), (5, 1152, 16), (1, 384, 8)])
def test_embed_and_final_layer_second_forms_bit_identical(dev, B, D, HW):
    """The patch-embed forward with the weight rows in registers (16-byte stores) and the final-layer forward / backward rows with
    the weight staged in LDS per 64 rows (csrc/embed.hip) against the first forms (the library's fallback for weights that are
    not 16-byte aligned: the second run passes the same weights at an 8-byte offset): the same per-lane accumulation
    order and the same wave reductions, so every output is bit-identical.  HW = 32: B * 256 rows, always whole 64-row groups;
    HW = 8 / 16 (T = 16 / 64 tokens per image) with odd B: 48, 320 and 16 rows — the kernels' tail guards (a last group with fewer
    than 64 rows, the second half of a row pair missing)."""
    from reed_amd import ops
    C, P = 4, 2
    T = (HW // P) ** 2
    g = torch.Generator().manual_seed(B + D)
    x = torch.randn(B, C, HW, HW, generator=g).to(dev)
    w = (torch.randn(D, 16, generator=g) * 0.2).to(torch.bfloat16).to(dev)
    bias = torch.randn(D, generator=g).to(torch.bfloat16).to(dev)
    pos = torch.randn(T, D, generator=g).to(dev)
    xt = torch.randn(B * T, D, generator=g).to(dev)
    mod = torch.randn(B, 2 * D, generator=g).to(torch.bfloat16).to(dev)
    wf = (torch.randn(16, D, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    bf = torch.randn(16, generator=g).to(torch.bfloat16).to(dev)
    dout = torch.randn(B, C, HW, HW, generator=g).to(dev)

    def off8(t):   # the same values at an address that is 8 (mod 16)
        buf = torch.empty(t.numel() + 12, dtype=t.dtype, device=dev)
        k = 4 if buf.data_ptr() % 16 == 0 else 0
        v = buf[k:k + t.numel()].view(t.shape)
        v.copy_(t)
        assert v.data_ptr() % 16 == 8
        return v

    def run(w, wf):
        tok = torch.full((B * T, D), float("nan"), device=dev)
        out = torch.full((B, C, HW, HW), float("nan"), device=dev)
        mean, rstd = torch.empty(B * T, device=dev), torch.empty(B * T, device=dev)
        hbuf = torch.full((B * T, D), float("nan"), dtype=torch.bfloat16, device=dev)
        dh, dlin = torch.full_like(hbuf, float("nan")), torch.full((B * T, 16), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.patch_embed_fwd(x, w, bias, pos, tok, B, C, HW, P, D)
        ops.final_layer_fwd(xt, mod.data_ptr(), mod.data_ptr() + 2 * D, 2 * D, wf, bf, out, mean, rstd, B, T, D, C, P)
        ops.final_layer_bwd_rows(dout, xt, mean, rstd, mod.data_ptr(), mod.data_ptr() + 2 * D, 2 * D, wf, hbuf, dlin, dh, B, T, D, C, P)
        torch.cuda.synchronize()
        return tok, out, mean, rstd, hbuf, dh, dlin

    new = run(w, wf)
    old = run(off8(w), off8(wf))
    for a, b in zip(new, old):
        assert torch.isfinite(a.float()).all() and torch.equal(a, b)


@pytest.mark.parametrize("R,C", [(1152, 4608), (3456, 1152), (384, 1152), (100, 72), (64, 200)])
def test_transpose_bf16(dev, R, C):
    """reed_transpose_bf16 (the transposed weight copies of round 6's NT input gradients; the 8-byte-lane form for multiples of 64,
    the 2-byte-lane form otherwise): exact."""
    from reed_amd import ops
    g = torch.Generator().manual_seed(R + C)
    src = torch.randn(R, C, generator=g).to(torch.bfloat16).to(dev)
    dst = torch.full((C, R), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.transpose_bf16(src, dst, R, C)
    assert torch.equal(dst, src.t().contiguous())
