"""End-to-end parity of the HIP path (reed_amd SiT + SILoss through the C ABI) on the GPU.

Checked against (a) golden vectors produced by the reference itself (tests/golden, fp32 or bf16-autocast) and
(b) the oracle (CPU restatement) run under bf16 autocast on the same seeded inputs — the same-precision
comparison. Tolerances are stated per test; index bookkeeping is checked bit-exactly in test_kernels_gpu.py.
"""
import os

import numpy as np
import pytest
import torch

from oracle import detfill
from oracle import loss as oloss
from oracle import sit as osit
from tests.test_oracle_golden import TINY_CASES, inputs, load

pytestmark = pytest.mark.gpu


def build_hip_model(cfg, dev, seed):
    from reed_amd.models.sit import SiT
    m = SiT(input_size=cfg["input_size"], patch_size=cfg["patch_size"], in_channels=cfg["in_channels"],
            hidden_size=cfg["hidden_size"], decoder_hidden_size=cfg["hidden_size"], depth=cfg["depth"],
            num_heads=cfg["num_heads"], num_classes=cfg["num_classes"], z_dims=cfg["z_dims"], z_types=cfg["z_types"],
            encoder_depth=cfg["encoder_depth"], encoder_depth_text=cfg["encoder_depth_text"],
            projector_dim=cfg["projector_dim"], class_dropout_prob=cfg["class_dropout_prob"],
            fused_attn=cfg["fused_attn"], qk_norm=cfg["qk_norm"])
    detfill.fill_state_dict(m.state_dict(), base_seed=seed)
    return m.to(dev)


# Per-tensor gradient bars against the SAME-PRECISION oracle (bf16 autocast on the CPU): (min cosine, max |norm ratio - 1|).
# Set from the values the kernels deliver on MI355X (printed by the test); anything looser is listed with its reason.
# Measured (round 2, every TINY case): worst cosine 0.99998, worst |norm ratio - 1| 0.0016 vs the bf16 oracle and 0.0059 vs
# the fp32 reference (q_norm / k_norm affine and qkv biases: small tensors summed over every token in bf16).
GRAD_BAR = (0.9999, 0.004)        # also bounds the fp32-reference norm ratio at 3 x 0.004
GRAD_BAR_REF_COS = 0.9999         # element-wise cosine against the fp32 reference's gradient tensors (measured >= 0.99996)
GRAD_BARS = {}                    # no exceptions needed


def cos(a, b):
    return torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0).item()


@pytest.mark.parametrize("name", [k for k, v in TINY_CASES.items() if v["hip"]])
def test_tiny_vs_reference_and_oracle(dev, name):
    from reed_amd.loss import SILoss
    g = load("tiny")
    c = TINY_CASES[name]
    cfg = c["cfg"]
    T = (cfg["input_size"] // cfg["patch_size"]) ** 2
    x, noise, t, y, drop_u, zs = inputs(4, 4, cfg["input_size"], 11, c["zspec"], T, cfg["num_classes"])
    drop = drop_u < cfg["class_dropout_prob"]
    # ---- HIP path
    m = build_hip_model(cfg, dev, 11)
    m.train()
    m.force_drop_mask = drop
    lf = SILoss(enc_names=c["enc"], loss_weights=dict(zip(c["enc"], c["co"])))
    out = lf(m, x.to(dev), dict(y=y.to(dev)), zs=[z.to(dev) for z in zs], time_input=t, noises=noise)
    total = out["denoising_loss"].mean() + 0.5 * out["proj_loss"]
    total.backward()
    torch.cuda.synchronize()
    # ---- oracle, bf16 autocast (same precision as the HIP path)
    P = detfill.fill_state_dict(osit.init_params(cfg), base_seed=11)
    P = {k: v.requires_grad_(k != "pos_embed") for k, v in P.items()}
    om = osit.OracleModel(P, cfg, autocast_bf16=True, training=True)
    om.drop_mask = drop
    oo = oloss.si_loss(om, x, dict(y=y), zs, enc_names=c["enc"], loss_weights=dict(zip(c["enc"], c["co"])), t=t,
                       noise=noise)
    ototal = oo["denoising_loss"].mean() + 0.5 * oo["proj_loss"]
    ototal.backward()
    # losses: vs same-precision oracle 5e-3 rel; vs the fp32 reference golden 2e-2 rel
    np.testing.assert_allclose(out["denoising_loss"].detach().cpu().numpy(), oo["denoising_loss"].detach().numpy(),
                               rtol=5e-3)
    np.testing.assert_allclose(float(out["proj_loss"]), float(oo["proj_loss"]), rtol=2e-2, atol=2e-3)
    np.testing.assert_allclose(out["denoising_loss"].detach().cpu().numpy(), g[f"{name}.denoising_loss"], rtol=2e-2)
    np.testing.assert_allclose(float(total), float(g[f"{name}.total"]), rtol=2e-2)
    # gradients: direction and size of every parameter's gradient vs the same-precision oracle, and the norm of every
    # parameter's gradient vs the fp32 reference golden
    bad, worst = [], [1.0, "", 0.0, "", 0.0, ""]
    for k, p in m.named_parameters():
        if not p.requires_grad:
            continue
        gh, go = p.grad.detach().cpu().float(), P[k].grad
        nh, no = gh.norm().item(), go.norm().item()
        if no < 5e-5:   # analytically zero gradients (e.g. k_norm.bias: softmax is invariant to a common key shift)
            assert nh < 5e-4, (k, nh, no)
            continue
        cs, dn = cos(gh, go), abs(nh / no - 1)
        dref = abs(nh / float(g[f"{name}.gnorm.{k}"]) - 1)
        if cs < worst[0]:
            worst[0:2] = [cs, k]
        if dn > worst[2]:
            worst[2:4] = [dn, k]
        if dref > worst[4]:
            worst[4:6] = [dref, k]
        cmin, nmax = GRAD_BARS.get((name, k), GRAD_BARS.get(k.split(".")[-2] + "." + k.split(".")[-1], GRAD_BAR))
        if cs < cmin or dn > nmax or dref > 3 * nmax:
            bad.append((k, cs, dn, dref))
    print(f"[{name}] worst cosine {worst[0]:.6f} ({worst[1]}), worst |norm ratio - 1| vs bf16 oracle {worst[2]:.5f} "
          f"({worst[3]}), vs fp32 reference {worst[4]:.5f} ({worst[5]})")
    assert not bad, bad[:8]
    # element values vs the fp32 reference golden
    worst_c = 1.0
    for k in ("final_layer.linear.bias", "x_embedder.proj.bias", "projectors.0.4.bias", "final_layer.linear.weight",
              "blocks.0.attn.qkv.bias", "x_embedder.proj.weight", "blocks.1.adaLN_modulation.1.bias",
              "blocks.2.mlp.fc1.bias"):
        gh = dict(m.named_parameters())[k].grad.detach().cpu().numpy()
        ref = g[f"{name}.grad.{k}"]
        cs = cos(torch.from_numpy(gh), torch.from_numpy(ref))
        worst_c = min(worst_c, cs)
        assert cs > GRAD_BAR_REF_COS, (k, cs)
    print(f"[{name}] worst cosine vs fp32 reference elements {worst_c:.6f}")
    # eval-mode inference forward
    m.eval()
    m.force_drop_mask = None
    xi, _, ti, yi, _, _ = inputs(4, 4, cfg["input_size"], 11, [], 0, 10)
    with torch.no_grad():
        o, z = m(xi.to(dev), ti.to(dev), yi.to(dev))
    assert z is None
    ref = torch.from_numpy(g[f"{name}.infer"])
    err = (o.cpu() - ref).abs().max().item()
    assert err <= 3e-2 * ref.abs().max().item() + 1e-3, err


def _hip_trainer(model_name, cfgkw, dev, enc, co, seed=0):
    from reed_amd.loss import SILoss
    from reed_amd.models.sit import SiT_models
    from reed_amd.optim import FusedAdamWEMA
    import copy
    m = SiT_models[model_name](**cfgkw)
    detfill.fill_state_dict(m.state_dict(), base_seed=seed)
    m = m.to(dev).train()
    ema = copy.deepcopy(m).requires_grad_(False).eval()
    opt = FusedAdamWEMA(m, ema, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.0, eps=1e-8, max_grad_norm=1.0)
    lf = SILoss(enc_names=list(enc), loss_weights=dict(zip(enc, co)))
    return m, ema, opt, lf


def _run_traj(m, opt, lf, dev, B, steps, zspec, align, proj_coeff=0.5, after_backward=None):
    rec = {"loss": [], "denoising_loss": [], "proj_loss": [], "grad_norm": []}
    for s in range(steps):
        x, noise, t, y, drop_u, zs = inputs(B, 4, 32, s, zspec, 256, 1000)
        m.force_drop_mask = drop_u < 0.1
        out = lf(m, x.to(dev), dict(y=y.to(dev)), zs=[z.to(dev) for z in zs], time_input=t, noises=noise)
        den = out["denoising_loss"].mean()
        proj = out["proj_loss"].mean() if torch.is_tensor(out["proj_loss"]) else torch.zeros((), device=dev)
        total = den + (proj * proj_coeff if align else 0.0)
        opt.zero_grad()
        total.backward()
        if after_backward is not None:
            after_backward(s)
        opt.step()
        rec["loss"].append(float(total)); rec["denoising_loss"].append(float(den))
        rec["proj_loss"].append(float(proj)); rec["grad_norm"].append(float(opt.grad_norm))
    return rec


def test_c1_s2_trajectory_vs_reference(dev):
    """C1: SiT-S/2, 64 random 32x32x4 latents, 10 steps, alignment off. Golden = the reference in fp32; the HIP path
    computes in bf16, so the per-step loss tolerance is 5e-3 absolute (losses ~2.0-2.3)."""
    g = load("s2_c1")
    m, ema, opt, lf = _hip_trainer("SiT-S/2", dict(z_dims=[], z_types=[]), dev, [], [])
    rec = _run_traj(m, opt, lf, dev, 64, 10, [], False)
    print("HIP :", [f"{v:.5f}" for v in rec["denoising_loss"]])
    print("REF :", [f"{v:.5f}" for v in g["denoising_loss"]])
    np.testing.assert_allclose(rec["denoising_loss"], g["denoising_loss"], atol=5e-3)
    np.testing.assert_allclose(rec["grad_norm"], g["grad_norm"], rtol=5e-2)
    sd = m.state_dict()
    for k in ("blocks.0.attn.qkv.weight", "final_layer.linear.weight", "t_embedder.mlp.2.bias"):
        w = sd[k].flatten()[:64].cpu().numpy()
        np.testing.assert_allclose(w, g["w." + k], atol=3e-4)   # 10 Adam steps of lr 1e-4: sign-level agreement
        e = ema.state_dict()[k].flatten()[:64].cpu().numpy()
        np.testing.assert_allclose(e, g["ema." + k], atol=1e-5)


@pytest.mark.parametrize("act_grad", [False, True])
def test_c2_xl2_trajectory_vs_reference_bf16(dev, act_grad):
    """C2: SiT-XL/2 + 1024-d (DINOv2-L-shaped) alignment, B=8, 5 optimiser steps on injected (x,t,eps,labels,zs).
    Golden = the reference under bf16 autocast (and fp32). Bar (BASELINE.json): per-step total loss within 1e-3 of
    the same-precision (bf16-autocast) reference; against the fp32 reference the bound is the reference's own
    bf16-vs-fp32 gap (2.3e-3 at step 1).  act_grad: the recomputing activation backward (what the engine picks at this token
    count) and the derivative-saving epilogues of round 5 (what it picks above 12288 tokens: the bench's b = 256) — same bars."""
    g = load("xl2_c2")
    kw = dict(z_dims=[1024], z_types=["i"], encoder_depth=8)
    m, ema, opt, lf = _hip_trainer("SiT-XL/2", kw, dev, ["dinov2"], [1.0])
    m.engine().save_act_grad = act_grad
    probes = {}

    def grab(step):   # step-1 gradients (unclipped: the clip coefficient is applied inside the fused update)
        if step == 0:
            torch.cuda.synchronize()
            named = dict(m.named_parameters())
            for k in [k[len("bf16.gnorm."):] for k in g.files if k.startswith("bf16.gnorm.")]:
                f = named[k].grad.detach().flatten()
                probes[k] = (f.double().norm().item(), f[:: max(1, f.numel() // 64)][:64].float().cpu().numpy())

    rec = _run_traj(m, opt, lf, dev, 8, 5, [(1024, "i")], True, after_backward=grab)
    d_bf16 = np.abs(np.array(rec["loss"]) - g["bf16.loss"])
    d_fp32 = np.abs(np.array(rec["loss"]) - g["fp32.loss"])
    ref_gap = np.abs(g["bf16.loss"] - g["fp32.loss"])
    print("HIP  loss:", [f"{v:.6f}" for v in rec["loss"]])
    print("REF bf16 :", [f"{v:.6f}" for v in g["bf16.loss"]])
    print("REF fp32 :", [f"{v:.6f}" for v in g["fp32.loss"]])
    print("|HIP-bf16|:", d_bf16, " |HIP-fp32|:", d_fp32, " ref bf16-fp32 gap:", ref_gap)
    assert (d_bf16 <= 1e-3).all(), d_bf16                       # BASELINE.json bar: within 1e-3 of the reference
    assert (d_fp32 <= np.maximum(2e-3, 1.5 * ref_gap)).all(), (d_fp32, ref_gap)  # vs fp32: the reference's own bf16 gap
    np.testing.assert_allclose(rec["grad_norm"], g["bf16.grad_norm"], rtol=3e-2)
    np.testing.assert_allclose(rec["proj_loss"], g["bf16.proj_loss"], atol=2e-3)
    # per-tensor gradients at XL/2 size, step 1, against the reference under bf16 autocast (and fp32): the norm of the
    # whole tensor and 64 elements spread over it (tools/gen_golden.py:GRAD_PROBES_XL: qkv / proj / fc1 / fc2 of blocks
    # 0, 8, 27, adaLN of blocks 8 and 27, projector layers, final layer, embedders)
    assert len(probes) >= 20
    rows = []
    for k, (nh, sl) in probes.items():
        nb, nf = float(g["bf16.gnorm." + k]), float(g["fp32.gnorm." + k])
        rb, rf = g["bf16.gslice." + k], g["fp32.gslice." + k]
        if not np.any(rb):     # label table: the 64 sampled elements lie in rows no label of the batch selects (exact zeros)
            assert not np.any(sl), k
            cb = cf = cref = 1.0
        else:
            cb = cos(torch.from_numpy(sl), torch.from_numpy(rb))
            cf = cos(torch.from_numpy(sl), torch.from_numpy(rf))
            cref = cos(torch.from_numpy(rb), torch.from_numpy(rf))     # the reference's own bf16-vs-fp32 agreement
        rows.append((k, nh / nb - 1, nh / nf - 1, nb / nf - 1, cb, cf, cref))
    for r in rows:
        print("  %-42s |g| vs bf16 %+.4f vs fp32 %+.4f (ref bf16 vs fp32 %+.4f)  slice cos vs bf16 %.5f vs fp32 %.5f (ref %.5f)" % r)
    for k, dnb, dnf, dref, cb, cf, cref in rows:
        assert abs(dnb) <= XL_GRAD_NORM_BAR + abs(dref), (k, dnb, dref)
        assert cb >= min(XL_GRAD_COS_BAR, cref - (1 - XL_GRAD_COS_BAR)), (k, cb, cref)
    # weights after the 5 AdamW steps (64 leading elements of three tensors) vs the reference's
    sd = m.state_dict()
    for k in ("blocks.0.attn.qkv.weight", "final_layer.linear.weight", "t_embedder.mlp.2.bias"):
        w = sd[k].flatten()[:64].cpu().numpy()
        # every step moves a weight by <= lr = 1e-4; a sign flip of a near-zero gradient element costs 2e-4 per step
        np.testing.assert_allclose(w, g["bf16.w." + k], atol=3e-4)
        frac_close = float(np.mean(np.abs(w - g["bf16.w." + k]) < 2e-5))
        print(f"  weights after 5 steps {k}: {frac_close:.2f} of 64 elements within 2e-5 of the reference")
        assert frac_close >= 0.7


# measured (round 2): every probe's norm within 1e-4 of the reference's bf16-autocast gradient, slice cosines >= 0.99998
FP16_EVAL_BAR = 1.2e-3
XL_GRAD_NORM_BAR = 0.002
XL_GRAD_COS_BAR = 0.9999


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_xl2_cfg_inference_vs_reference(dev, precision):
    """C5 at its real size: the CFG-doubled no-tap evaluation of SiT-XL/2 exactly as samplers.py:66-78 issues it
    ([x; x], labels [y; 1000]) at two times, against the reference in fp32 and under bf16 autocast
    (tools/gen_golden.py:g_xl_infer), plus 3 Heun steps with CFG 1.5 on top (5 evaluations, fp64 state).
    precision="fp16": the sampling build (IEEE-half operands, generate.py's default: the mantissa of the reference's TF32
    evaluations) — its deviation from the fp32 reference must be several times below the bf16 path's."""
    from reed_amd.models.sit import SiT_models
    from reed_amd.samplers import euler_sampler
    g = load("xl2_infer")
    m = SiT_models["SiT-XL/2"](z_dims=[1024], z_types=["i"], encoder_depth=8, use_cfg=True)
    detfill.fill_state_dict(m.state_dict(), base_seed=0)
    m = m.to(dev).eval()
    m.precision = precision
    x, _, _, y, _, _ = inputs(2, 4, 32, 77, [], 256, 1000)
    xx, yy = torch.cat([x, x]).to(dev), torch.cat([y, torch.tensor([1000, 1000])]).to(dev)
    for tv in (0.9, 0.35):
        with torch.no_grad():
            o, z = m(xx, torch.full((4,), tv, device=dev), yy)
        assert z is None
        o = o.cpu()
        rf, rb = torch.from_numpy(g[f"fp32.t{tv}"]), torch.from_numpy(g[f"bf16.t{tv}"])
        sc = rf.abs().max().item()
        e_b, e_f, e_ref = (o - rb).abs().max().item() / sc, (o - rf).abs().max().item() / sc, (rb - rf).abs().max().item() / sc
        print(f"XL/2 CFG eval [{precision}] t={tv}: max|HIP-ref_bf16| {e_b:.2e}  max|HIP-ref_fp32| {e_f:.2e}  "
              f"(reference's own bf16-vs-fp32 {e_ref:.2e}), all relative to max|v| = {sc:.3f}; "
              f"cos vs fp32 {cos(o, rf):.6f}")
        # max-abs over 16 K elements is the tail of the bf16 noise of BOTH sides (measured e_b 3.4e-3 .. 6.8e-3 across two
        # forward kernels with the same cosine); the cosine is the stable statistic
        if precision == "bf16":
            assert e_f <= 1.3 * e_ref + 5e-4 and e_b <= 2.0 * e_ref and cos(o, rf) > 0.9999 and cos(o, rb) > 0.9999
        else:   # 3 more mantissa bits: measured 6.6e-4 of the scale (bf16: 4.8e-3), cosine 0.9999998
            assert e_f <= FP16_EVAL_BAR and cos(o, rf) > 0.999999
    with torch.no_grad():
        s = euler_sampler(m, x.to(dev), y.to(dev), num_steps=3, heun=True, cfg_scale=1.5).cpu()
    ref = torch.from_numpy(g["heun3_cfg"])
    err = (s - ref).abs().max().item()
    print(f"XL/2 3-step Heun + CFG 1.5 [{precision}]: max abs deviation from the fp32 reference {err:.3e} (latent scale {ref.abs().max().item():.2f})")
    assert s.dtype == torch.float64
    assert err <= (4e-3 if precision == "bf16" else 6e-4) * ref.abs().max().item()     # measured 2.0e-3 / see print


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_long_horizon_heun_cfg_drift_s2(dev, precision):
    """What the 16-bit model evaluations cost over a long sampling horizon: SiT-S/2, n = 2, 50-step Heun with CFG 1.5 over the whole
    interval (99 evaluations at batch 4), HIP (bf16 operands, fp64 state) against the reference with an fp32 model
    (tools/gen_golden.py:g_samplers_long). The state fed to every 9th evaluation is compared: the printed curve is the
    drift; the bound asserted is on its end point, relative to the latents' scale."""
    from reed_amd.models.sit import SiT_models
    from reed_amd.samplers import euler_sampler
    g = load("samplers_long")
    m = SiT_models["SiT-S/2"](z_dims=[768], z_types=["i"], encoder_depth=8, use_cfg=True)
    detfill.fill_state_dict(m.state_dict(), base_seed=3)
    m = m.to(dev).eval()
    m.precision = precision
    z = detfill.normal((2, 4, 32, 32), 91).to(dev)
    y = torch.tensor([17, 833], device=dev)
    states = []

    class Rec:
        num_classes, class_dropout_prob = m.num_classes, m.class_dropout_prob

        def engine(self):
            return m.engine()

        def __call__(self, xx, tt, **kw):
            states.append(xx[:2].detach().cpu().clone())
            return m(xx, tt, **kw)

    with torch.no_grad():
        out = euler_sampler(Rec(), z, y, num_steps=50, heun=True, cfg_scale=1.5).cpu()
    assert len(states) == int(g["n_evals"]) == 99
    ref_states = torch.from_numpy(g["states"])
    curve = [(s - r).abs().max().item() for s, r in zip(states[::9], ref_states)]
    ref = torch.from_numpy(g["final"])
    scale = ref.abs().max().item()
    end = (out - ref).abs().max().item()
    print(f"drift of the {precision}-operand sampler vs the fp32 reference, max abs, at evaluations 0, 9, ..., 90:",
          " ".join(f"{c:.2e}" for c in curve), f"| final {end:.3e} (latent scale {scale:.2f}, rms "
          f"{(out - ref).pow(2).mean().sqrt().item():.3e})")
    assert curve[0] == 0.0
    assert end <= (LONG_DRIFT_BAR if precision == "bf16" else LONG_DRIFT_BAR_FP16) * scale


# measured (round 2): the drift grows linearly in t to 5.8e-3 abs = 1.4e-3 of the latents' scale (rms 3.7e-4 of it) at the
# end of the trajectory; it is the integral of the bf16 evaluation error over the unit time interval, so a 250-step run
# ends in the same place (sampling precision)
LONG_DRIFT_BAR = 3e-3
LONG_DRIFT_BAR_FP16 = 5e-4


# XL/2 bars: scaled from the S/2 curve by the printed measurements (the drift is the integral of the evaluation error over the
# unit time interval, linear in t at either size; 28 blocks of width 1152 instead of 12 of 384 carry a larger evaluation error)
LONG_DRIFT_BAR_XL = {"bf16": 6e-3, "fp16": 1e-3, "fp32": 2e-5}


@pytest.mark.parametrize("precision", ["bf16", "fp16", "fp32"])
def test_long_horizon_heun_cfg_drift_xl2(dev, precision):
    """The same pin at C5's real model size: SiT-XL/2, n = 2, 25-step Heun with CFG 1.5 over the whole
    interval (49 evaluations at batch 4) against the reference with an fp32 model (tools/gen_golden.py:g_samplers_long_xl).
    The state fed to every 6th evaluation is compared (the printed drift curve), the end point asserted relative to the
    latents' scale; precision "fp32" = the reference's own arithmetic under --no-tf32."""
    from reed_amd.models.sit import SiT_models
    from reed_amd.samplers import euler_sampler
    g = load("samplers_long_xl")
    m = SiT_models["SiT-XL/2"](z_dims=[1024], z_types=["i"], encoder_depth=8, use_cfg=True)
    detfill.fill_state_dict(m.state_dict(), base_seed=0)
    m = m.to(dev).eval()
    m.precision = precision
    z = detfill.normal((2, 4, 32, 32), 191).to(dev)
    y = torch.tensor([207, 980], device=dev)
    states = []

    class Rec:
        num_classes, class_dropout_prob = m.num_classes, m.class_dropout_prob

        def engine(self):
            return m.engine()

        def __call__(self, xx, tt, **kw):
            states.append(xx[:2].detach().cpu().clone())
            return m(xx, tt, **kw)

    with torch.no_grad():
        out = euler_sampler(Rec(), z, y, num_steps=25, heun=True, cfg_scale=1.5).cpu()
    assert len(states) == int(g["n_evals"]) == 49
    ref_states = torch.from_numpy(g["states"])
    curve = [(s - r).abs().max().item() for s, r in zip(states[::6], ref_states)]
    ref = torch.from_numpy(g["final"])
    scale = ref.abs().max().item()
    end = (out - ref).abs().max().item()
    print(f"XL/2 drift of the {precision}-operand sampler vs the fp32 reference, max abs, at evaluations 0, 6, ..., 48:",
          " ".join(f"{c:.2e}" for c in curve), f"| final {end:.3e} (latent scale {scale:.2f}, rms "
          f"{(out - ref).pow(2).mean().sqrt().item():.3e})")
    assert curve[0] == 0.0
    assert end <= LONG_DRIFT_BAR_XL[precision] * scale


def test_wgrad_side_stream_bit_identical(dev):
    """The blocks' weight-gradient GEMMs on the second HIP stream (engine.wgrad_stream) are the same launches in a
    different interleaving: every gradient must be bit-identical to the single-stream backward, also when the step is
    repeated (buffers recycled by the caching allocator while the side stream is still reading would show up here)."""
    from reed_amd.loss import SILoss
    c = TINY_CASES["xl3"]   # hd 72, 16 heads, D 1152: the XL block shape
    cfg = c["cfg"]
    T = (cfg["input_size"] // cfg["patch_size"]) ** 2
    x, noise, t, y, drop_u, zs = inputs(4, 4, cfg["input_size"], 5, c["zspec"], T, cfg["num_classes"])
    drop = drop_u < cfg["class_dropout_prob"]
    grads = {}
    for mode in (False, True):
        m = build_hip_model(cfg, dev, 5)
        m.train()
        m.force_drop_mask = drop
        m.engine().wgrad_stream = mode
        lf = SILoss(enc_names=c["enc"], loss_weights=dict(zip(c["enc"], c["co"])))
        for _ in range(3):
            for p in m.parameters():
                p.grad = None
            m.engine().zero_grad()
            out = lf(m, x.to(dev), dict(y=y.to(dev)), zs=[z.to(dev) for z in zs], time_input=t, noises=noise)
            (out["denoising_loss"].mean() + 0.5 * out["proj_loss"]).backward()
        torch.cuda.synchronize()
        grads[mode] = m._arena.grad.clone()
    assert torch.equal(grads[False], grads[True])


def test_saved_activation_derivative_vs_recomputing_backward(dev):
    """Round 5: the fc1 / t-MLP / projector forward epilogues save act'(pre) (bf16) where the pre-activation used to be saved
    and the backward multiplies by it (gemm.h EPI_GELU_G / EPI_SILU_G / EPI_MUL) instead of recomputing the derivative from the
    pre-activation (EPI_DGELU / EPI_DSILU).  Same forward bit for bit; the backward differs by ONE extra bf16 rounding of one
    factor (the derivative), i.e. 2^-9 relative per element of d(pre), uncorrelated: every gradient tensor stays within
    cosine 0.99999 / 1e-3 in norm of the recomputing form (both forms are held to the oracle bars by the tests above)."""
    from reed_amd.loss import SILoss
    c = TINY_CASES["xl3"]
    cfg = c["cfg"]
    T = (cfg["input_size"] // cfg["patch_size"]) ** 2
    x, noise, t, y, drop_u, zs = inputs(4, 4, cfg["input_size"], 5, c["zspec"], T, cfg["num_classes"])
    drop = drop_u < cfg["class_dropout_prob"]
    grads, losses = {}, {}
    for mode in (False, True):
        m = build_hip_model(cfg, dev, 5)
        m.train()
        m.force_drop_mask = drop
        m.engine().save_act_grad = mode
        lf = SILoss(enc_names=c["enc"], loss_weights=dict(zip(c["enc"], c["co"])))
        out = lf(m, x.to(dev), dict(y=y.to(dev)), zs=[z.to(dev) for z in zs], time_input=t, noises=noise)
        tot = out["denoising_loss"].mean() + 0.5 * out["proj_loss"]
        tot.backward()
        torch.cuda.synchronize()
        losses[mode] = tot.detach().clone()
        grads[mode] = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.requires_grad}
    assert torch.equal(losses[False], losses[True])
    worst = [1.0, "", 0.0, ""]
    for k, g0 in grads[False].items():
        g1 = grads[True][k]
        n0 = g0.norm().item()
        if n0 < 5e-5:
            continue
        cs, dn = cos(g0, g1), abs(g1.norm().item() / n0 - 1)
        if cs < worst[0]:
            worst[0:2] = [cs, k]
        if dn > worst[2]:
            worst[2:4] = [dn, k]
    print(f"[act-grad] worst cosine {worst[0]:.7f} ({worst[1]}), worst |norm ratio - 1| {worst[2]:.6f} ({worst[3]})")
    assert worst[0] > 0.99999 and worst[2] < 1e-3, worst


def test_overlapped_optimizer_bit_identical(dev):
    """FusedAdamWEMA(overlap=True): per-bucket update launches on the optimiser's own stream with the next forward
    waiting per bucket. Same arithmetic, different schedule -> master weights, EMA and the bf16 shadow after 4 steps
    must equal the single-launch optimiser bit for bit (a forward that read a bucket before its update landed, or an
    update overtaking a backward still writing gradients, would show up here)."""
    res = {}
    for overlap in (False, True):
        m, ema, opt, lf = _hip_trainer("SiT-S/2", dict(z_dims=[128], z_types=["i"], encoder_depth=4, projector_dim=256),
                                       dev, ["dinov2"], [1.0], seed=2)
        opt.overlap = overlap
        _run_traj(m, opt, lf, dev, 8, 4, [(128, "i")], True)
        sd = m.state_dict()   # orders the current stream after the pending update
        ema.state_dict()
        torch.cuda.synchronize()
        res[overlap] = (m._arena.master.clone(), ema._arena.master.clone(), m._arena.shadow.clone(), sd)
    for a, b in zip(res[False][:3], res[True][:3]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("overlap", [True, False])
def test_transposed_weight_copies_stay_fresh_and_change_no_bit(dev, overlap):
    """Round 6: engine.dgrad_nt — the blocks' input gradients as NT GEMMs on transposed 16-bit copies of the weights
    (arena.shadow_t), built at the first backward and refreshed by the fused optimiser on its side stream behind every update
    (overlap=True), or rebuilt at the next backward (the single-launch optimiser).  Four steps of SiT-S/2 with alignment: the
    trajectory, the master weights and the EMA equal the NN path's bit for bit, and after the last step every W^T is the
    transpose of the shadow."""
    res = {}
    for nt in (False, True):
        m, ema, opt, lf = _hip_trainer("SiT-S/2", dict(z_dims=[128], z_types=["i"], encoder_depth=4, projector_dim=256),
                                       dev, ["dinov2"], [1.0], seed=3)
        opt.overlap = overlap
        m.engine().dgrad_nt = nt
        rec = _run_traj(m, opt, lf, dev, 8, 4, [(128, "i")], True)
        m.state_dict()
        ema.state_dict()
        A = m._arena
        if nt:
            assert A.shadow_t is not None and len(A.t_seg) == 4 * 12
            if overlap:
                assert A.shadow_t_gen == A.shadow_gen and A.pending_t is not None    # the optimiser kept them fresh
            A.ensure_shadow_t("bf16")
            torch.cuda.synchronize()
            for name, (toff, n_out, k_in) in A.t_seg.items():
                w = A.view(A.shadow, name)
                assert torch.equal(A.shadow_t[toff:toff + n_out * k_in].view(k_in, n_out), w.t().contiguous()), name
        else:
            assert A.shadow_t is None
        torch.cuda.synchronize()
        res[nt] = (rec, A.master.clone(), ema._arena.master.clone())
    assert res[False][0] == res[True][0]
    assert torch.equal(res[False][1], res[True][1]) and torch.equal(res[False][2], res[True][2])


def test_full_size_properties(dev):
    """C2 at the bench's own size (SiT-XL/2 + 1024-d projector, local batch 256: no oracle finishes there), through
    properties that hold at any size:
      (1) reference init: adaLN / final layers are zero -> every block is the identity and the velocity is exactly 0
          (sit.py:246-254), the projector output is not;
      (2) determinism: the same step twice from the same state gives bit-identical losses and gradients;
      (3) linearity of backward: 2 x loss -> every gradient exactly doubled (bf16 / fp32 roundings commute with a
          power-of-two scale), which a dropped or double-counted term, or an accumulation into stale memory, breaks."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import random_fill
    from reed_amd.loss import SILoss
    from reed_amd.models.sit import SiT_models
    torch.manual_seed(0)
    m = SiT_models["SiT-XL/2"](z_dims=[1024], z_types=["i"], encoder_depth=8).to(dev).train()
    B = 256
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.randn(B, 4, 32, 32, device=dev, generator=g)
    noise = torch.randn(B, 4, 32, 32, device=dev, generator=g)
    t = torch.rand(B, device=dev, generator=g) * 0.9 + 0.05
    y = torch.randint(0, 1000, (B,), device=dev, generator=g)
    zs = [torch.randn(B, 256, 1024, device=dev, generator=g)]
    m.force_drop_mask = torch.rand(B, device=dev, generator=g) < 0.1
    with torch.no_grad():
        v, z = m(x, t, y, inference=False)
    assert float(v.abs().max()) == 0.0 and float(z[0].float().abs().max()) > 0.0
    random_fill(m, 1234)
    lf = SILoss(enc_names=["dinov2-vit-l"], loss_weights={"dinov2-vit-l": 1.0})

    def step(scale):
        for p in m.parameters():
            p.grad = None
        m.engine().zero_grad()
        out = lf(m, x, dict(y=y), zs=zs, time_input=t.cpu(), noises=noise)
        loss = out["denoising_loss"].mean() + 0.5 * out["proj_loss"]
        (loss * scale).backward()
        torch.cuda.synchronize()
        return float(loss.detach()), m._arena.grad.clone()

    l1, g1 = step(1.0)
    l2, g2 = step(1.0)
    assert l1 == l2 and torch.equal(g1, g2)
    assert np.isfinite(l1) and float(g1.abs().max()) > 0 and bool(torch.isfinite(g1).all())
    _, g3 = step(2.0)
    assert torch.equal(g3, 2.0 * g1)


def test_tile_choice_is_bit_invisible_at_small_local_batch(dev):
    """The 8-GPU point (SiT-XL/2 + 1024-d projector, local batch 32): the step whose 1152-wide GEMMs run on the 256x144
    tile (the heuristic's choice there, csrc/gemm144.hip) against the same step with every GEMM forced onto the 128^2
    kernel — the kernels are bit-identical per GEMM, so loss and every gradient must be bit-identical too."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import random_fill
    from reed_amd import ops
    from reed_amd.loss import SILoss
    from reed_amd.models.sit import SiT_models
    torch.manual_seed(0)
    m = SiT_models["SiT-XL/2"](z_dims=[1024], z_types=["i"], encoder_depth=8).to(dev).train()
    random_fill(m, 4321)
    B = 32
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn(B, 4, 32, 32, device=dev, generator=g)
    noise = torch.randn(B, 4, 32, 32, device=dev, generator=g)
    t = torch.rand(B, device=dev, generator=g) * 0.9 + 0.05
    y = torch.randint(0, 1000, (B,), device=dev, generator=g)
    zs = [torch.randn(B, 256, 1024, device=dev, generator=g)]
    m.force_drop_mask = torch.rand(B, device=dev, generator=g) < 0.1
    lf = SILoss(enc_names=["dinov2-vit-l"], loss_weights={"dinov2-vit-l": 1.0})

    def step(tile):
        ops.gemm_force_tile(tile)
        try:
            for p in m.parameters():
                p.grad = None
            m.engine().zero_grad()
            out = lf(m, x, dict(y=y), zs=zs, time_input=t.cpu(), noises=noise)
            loss = out["denoising_loss"].mean() + 0.5 * out["proj_loss"]
            loss.backward()
            torch.cuda.synchronize()
            return float(loss.detach()), m._arena.grad.clone()
        finally:
            ops.gemm_force_tile(0)

    l0, g0 = step(0)
    l1, g1 = step(128)
    assert np.isfinite(l0) and float(g0.abs().max()) > 0
    assert l0 == l1
    # every gradient is the same bits: since round 6 the grouped weight gradients' one-item-per-CU form (csrc/gemm256w.hip) walks
    # every tile's tokens as one sequence, like force_tile 128's kernel (csrc/gemm_tn.hip) — round 4's K-cut ragged tiles had
    # differed in fp32 summation order
    assert torch.equal(g0, g1)


def test_bench_plan_matches_the_golden_pinned_plan_at_b256(dev, monkeypatch):
    """The tie between the kernel plan the XL/2 goldens pin (B = 8: 128^2 / 256x144 tiles, per-GEMM split-K weight gradients, the
    row-kernel delta of the attention backward) and the plan the timed b = 256 step runs (persistent 256^2 four-wave kernels, the
    grouped weight gradients without split-K, the ring backward with delta from the dO GEMM's epilogue), at the bench's own size:
    ONE step of SiT-XL/2 + 1024-d projector at local batch 256, four ways —
      bench   the heuristic plan (what bench.py times)
      A       the same with the row-kernel delta (engine.fused_delta = False)
      B       A with every GEMM forced onto the 128^2 kernel            -> loss and every gradient bit-identical
      C       B with the per-GEMM split-K weight gradients (REED_WGRAD_GROUP=0) = the plan the goldens pin
                                                                         -> differs from B by fp32 summation order only (<= 1e-5)
    and bench vs A differs only where the last fp32 bit of delta (another summation order) flips a bf16 rounding of dS: the loss
    is bit-identical, every gradient tensor within 3e-3 of its norm (measured 1.5e-3, the label table), the arena within 2e-4
    (measured 7.9e-5) — the level of the bf16 activations' own rounding noise."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import random_fill
    from reed_amd import ops
    from reed_amd.loss import SILoss
    from reed_amd.models.sit import SiT_models
    torch.manual_seed(0)
    m = SiT_models["SiT-XL/2"](z_dims=[1024], z_types=["i"], encoder_depth=8).to(dev).train()
    random_fill(m, 4321)
    B = 256
    g = torch.Generator(device=dev).manual_seed(12)
    x = torch.randn(B, 4, 32, 32, device=dev, generator=g)
    noise = torch.randn(B, 4, 32, 32, device=dev, generator=g)
    t = torch.rand(B, device=dev, generator=g) * 0.9 + 0.05
    y = torch.randint(0, 1000, (B,), device=dev, generator=g)
    zs = [torch.randn(B, 256, 1024, device=dev, generator=g)]
    m.force_drop_mask = torch.rand(B, device=dev, generator=g) < 0.1
    lf = SILoss(enc_names=["dinov2-vit-l"], loss_weights={"dinov2-vit-l": 1.0})

    def step(tile, dp, group):
        m.engine().fused_delta = dp != "0"
        monkeypatch.setenv("REED_WGRAD_GROUP", group)
        ops.gemm_force_tile(tile)
        try:
            for p in m.parameters():
                p.grad = None
            m.engine().zero_grad()
            m.engine()._dot_delta.clear()
            out = lf(m, x, dict(y=y), zs=zs, time_input=t.cpu(), noises=noise)
            loss = out["denoising_loss"].mean() + 0.5 * out["proj_loss"]
            loss.backward()
            torch.cuda.synchronize()
            return float(loss.detach()), m._arena.grad.clone()
        finally:
            ops.gemm_force_tile(0)

    L = m._layout

    def worst_rel(g0, g1):
        worst, who = 0.0, None
        for name, (off, shp) in L.seg.items():
            n = int(np.prod(shp))
            if off + n > L.n_train:
                continue
            a, b = g0[off:off + n].double(), g1[off:off + n].double()
            rel = float((a - b).norm() / a.norm().clamp_min(1e-30))
            if rel > worst:
                worst, who = rel, name
        return worst, who, float((g0.double() - g1.double()).norm() / g0.double().norm())

    lb, gb = step(0, "1", "1")
    la, ga = step(0, "0", "1")
    l2, g2 = step(128, "0", "1")
    l3, g3 = step(128, "0", "0")
    assert np.isfinite(lb) and float(gb.abs().max()) > 0 and all(bool(torch.isfinite(v).all()) for v in (ga, g2, g3))
    assert lb == la == l2 == l3, (lb, la, l2, l3)
    # tile choice: bit-invisible at b = 256 too (round 6: the grouped weight gradients no longer cut any tile along K)
    assert torch.equal(ga, g2)
    w32, who32, all32 = worst_rel(g2, g3)                   # grouped vs split-K weight gradients
    wdp, whodp, alldp = worst_rel(ga, gb)                   # delta from the row kernel vs from the dO GEMM's epilogue
    print(f"b=256 plan tie: loss {lb}; split-K vs grouped: worst tensor {w32:.2e} ({who32}), arena {all32:.2e}; "
          f"delta source: worst tensor {wdp:.2e} ({whodp}), arena {alldp:.2e}")
    assert w32 <= 1e-5 and all32 <= 1e-6, (w32, who32, all32)
    assert wdp <= 3e-3 and alldp <= 2e-4, (wdp, whodp, alldp)


def test_c4_xl2_two_encoders_vs_reference_bf16(dev):
    """C4 (BASELINE.json configs[3]): SiT-XL/2 with CLIP-L-shaped image tokens (1024-d, tap after block 8) and a pooled
    text / VLM vector (3584-d, tap after block 16), repa coefficients 1.0 / 0.5, B=4, 2 optimiser steps on injected
    draws. Golden = the reference under bf16 autocast (tools/gen_golden.py: g_xl_c4). Step 1 is a pure forward/backward
    comparison (bar 1e-3); step 2 comes after one AdamW update of every weight by ±lr on deterministic-fill weights,
    where the reference's own loss jumps from 2.4 to 8.5: there the bound is relative."""
    g = load("xl2_c4")
    kw = dict(z_dims=[1024, 3584], z_types=["i", "t"], encoder_depth=8, encoder_depth_text=16)
    m, ema, opt, lf = _hip_trainer("SiT-XL/2", kw, dev, ["clip", "text_embeds_qwenvl_7b"], [1.0, 0.5])
    rec = _run_traj(m, opt, lf, dev, 4, 2, [(1024, "i"), (3584, "t")], True)
    print("HIP :", rec["loss"], rec["proj_loss"], rec["grad_norm"])
    print("REF :", g["loss"], g["proj_loss"], g["grad_norm"])
    assert abs(rec["loss"][0] - g["loss"][0]) <= 1e-3
    assert abs(rec["proj_loss"][0] - g["proj_loss"][0]) <= 1e-3
    np.testing.assert_allclose(rec["grad_norm"][0], g["grad_norm"][0], rtol=3e-2)
    assert abs(rec["loss"][1] - g["loss"][1]) <= 2e-3            # measured 4.2e-4 (loss 8.478)
    np.testing.assert_allclose(rec["proj_loss"][1], g["proj_loss"][1], atol=1e-4)


def test_b2_alignment_trajectory_vs_reference_bf16(dev):
    """SiT-B/2 (12 blocks, 12 heads of 64) with alignment on, B=8, 3 steps: golden = the reference in bf16 autocast and
    fp32 (tools/gen_golden.py: g_b2). Same bars as C2."""
    g = load("b2_align")
    kw = dict(z_dims=[768], z_types=["i"], encoder_depth=4)
    m, ema, opt, lf = _hip_trainer("SiT-B/2", kw, dev, ["dinov2"], [1.0])
    rec = _run_traj(m, opt, lf, dev, 8, 3, [(768, "i")], True)
    d_bf16 = np.abs(np.array(rec["loss"]) - g["bf16.loss"])
    ref_gap = np.abs(g["bf16.loss"] - g["fp32.loss"])
    print("HIP  loss:", rec["loss"], " REF bf16:", g["bf16.loss"], " REF fp32:", g["fp32.loss"])
    assert (d_bf16 <= np.maximum(1e-3, 1.5 * ref_gap)).all(), (d_bf16, ref_gap)
    np.testing.assert_allclose(rec["grad_norm"], g["bf16.grad_norm"], rtol=5e-2)
    np.testing.assert_allclose(rec["proj_loss"], g["bf16.proj_loss"], atol=2e-3)


@pytest.mark.parametrize("input_size,B", [(64, 2), (16, 3), (32, 1)])
def test_inference_other_resolutions_vs_oracle(dev, input_size, B):
    """generate.py --resolution 512 (latent 64x64 -> T = 1024 tokens: four 256-key attention tiles with online softmax,
    M = B*1024 rows) and odd batch sizes: the HIP inference forward against the oracle under bf16 autocast; also the
    Euler sampler with CFG on top of it (same model evaluations, fp64 state)."""
    from oracle import samplers as osamplers
    from reed_amd.samplers import euler_sampler
    cfg = TINY_CASES["hd64"]["cfg"].copy()
    cfg.update(input_size=input_size, num_classes=1000, depth=2)
    m = build_hip_model(cfg, dev, 21).eval()
    P = detfill.fill_state_dict(osit.init_params(cfg), base_seed=21)
    om = osit.OracleModel(P, cfg, autocast_bf16=True, training=False)
    x, _, t, y, _, _ = inputs(B, 4, input_size, 3, [], 0, 1000)
    with torch.no_grad():
        o, z = m(x.to(dev), t.to(dev), y.to(dev))
        ro = om(x, t, y)[0].float()
    assert z is None and o.shape == (B, 4, input_size, input_size)
    assert (o.cpu() - ro).abs().max().item() <= 3e-2 * ro.abs().max().item() + 1e-3
    with torch.no_grad():
        s_hip = euler_sampler(m, x.to(dev), y.to(dev), num_steps=3, heun=True, cfg_scale=1.5).cpu()
        s_ref = osamplers.euler_sampler(om, x, y, num_steps=3, heun=True, cfg_scale=1.5)
    assert s_hip.dtype == torch.float64 and s_hip.shape == x.shape
    assert (s_hip - s_ref).abs().max().item() <= 5e-2 * s_ref.abs().max().item() + 1e-3


def test_gradient_accumulation_matches_full_batch(dev):
    """--gradient-accumulation-steps 2 (reference: accelerate accumulate()): two micro-steps on the
    halves of a batch — loss / 2 each, gradients accumulated in the arena, optimiser + EMA + step counter only on the
    second — must leave the same gradients and weights as one step on the whole batch (same per-sample arithmetic,
    only the fp32 summation split differs)."""
    import copy
    from reed_amd.loss import SILoss
    from reed_amd.optim import FusedAdamWEMA
    from reed_amd.trainer import TrainStep
    c = TINY_CASES["hd64"]
    cfg = c["cfg"]
    T = (cfg["input_size"] // cfg["patch_size"]) ** 2
    x, noise, t, y, drop_u, zs = inputs(8, 4, cfg["input_size"], 31, c["zspec"], T, cfg["num_classes"])
    drop = drop_u < cfg["class_dropout_prob"]
    out = {}
    for accum in (1, 2):
        m = build_hip_model(cfg, dev, 31).train()
        ema = copy.deepcopy(m).requires_grad_(False).eval()
        opt = FusedAdamWEMA(m, ema, lr=1e-3)
        step = TrainStep(m, SILoss(enc_names=c["enc"], loss_weights=dict(zip(c["enc"], c["co"]))), opt, None,
                         proj_coeff=0.5, diffusion_warm_up_steps=0, grad_accum=accum)
        n = 8 // accum
        for k in range(accum):
            sl = slice(k * n, (k + 1) * n)
            m.force_drop_mask = drop[sl]
            res = step(x[sl].to(dev), y[sl].to(dev), [z[sl].to(dev) for z in zs], time_input=t[sl], noises=noise[sl])
            assert ("grad_norm" in res) == (k == accum - 1)
        assert step.global_step == 1 and opt.step_count == 1
        sd = m.state_dict()
        torch.cuda.synchronize()
        out[accum] = (m._arena.grad.clone(), m._arena.master.clone(), ema._arena.master.clone(), float(res["grad_norm"]))
    g1, g2 = out[1][0], out[2][0]
    assert torch.nn.functional.cosine_similarity(g1, g2, dim=0).item() > 0.9995
    torch.testing.assert_close(g2.norm(), g1.norm(), rtol=5e-3, atol=0)
    np.testing.assert_allclose(out[2][3], out[1][3], rtol=5e-3)
    # Adam's first step moves every weight by +-lr: compare where the gradient sign is not in the rounding noise
    big = g1.abs() > 1e-3 * g1.abs().max()
    nt = g1.numel()
    dw = (out[1][1][:nt] - out[2][1][:nt])[big]
    assert (dw.abs() > 1e-6).float().mean().item() < 0.02
    torch.testing.assert_close(out[2][2], out[1][2], atol=1e-6, rtol=0)


def test_fp16_switch_restores_and_trains(dev):
    """The library selection is restored after every forward: a bf16 training step right after an fp16 evaluation of
    another model is bit-identical to one without it; a precision='fp16' model trains too (IEEE-half operands), with
    gradients close to the bf16 ones, and changing the precision between forward and backward is refused."""
    from reed_amd import ops
    from reed_amd.loss import SILoss
    c = TINY_CASES["hd64"]
    cfg = c["cfg"]
    T = (cfg["input_size"] // cfg["patch_size"]) ** 2
    x, noise, t, y, drop_u, zs = inputs(4, 4, cfg["input_size"], 13, c["zspec"], T, cfg["num_classes"])
    lf = SILoss(enc_names=c["enc"], loss_weights=dict(zip(c["enc"], c["co"])))

    def fwd(m):
        m.train()
        m.force_drop_mask = drop_u < 0.1
        for p in m.parameters():
            p.grad = None
        m.engine().zero_grad()
        out = lf(m, x.to(dev), dict(y=y.to(dev)), zs=[z.to(dev) for z in zs], time_input=t, noises=noise)
        return out["denoising_loss"].mean() + 0.5 * out["proj_loss"]

    def step(m):
        fwd(m).backward()
        torch.cuda.synchronize()
        return m._arena.grad.clone()

    m = build_hip_model(cfg, dev, 13)
    g0 = step(m)
    s = build_hip_model(cfg, dev, 14).eval()
    s.precision = "fp16"
    with torch.no_grad():
        o16, _ = s(x.to(dev), t.to(dev), y.to(dev))
    s.precision = "bf16"
    with torch.no_grad():
        ob, _ = s(x.to(dev), t.to(dev), y.to(dev))
    assert ops._PRECISION == "bf16"
    assert torch.isfinite(o16).all() and 0 < (o16 - ob).abs().max().item() < 0.1 * ob.abs().max().item()
    assert torch.equal(step(m), g0)
    m.precision = "fp16"
    g16 = step(m)
    assert ops._PRECISION == "bf16"
    assert torch.isfinite(g16).all() and not torch.equal(g16, g0)
    assert cos(g16.cpu(), g0.cpu()) > 0.999
    total = fwd(m)
    m.precision = "bf16"
    with pytest.raises(RuntimeError, match="precision changed"):
        total.backward()
    assert ops._PRECISION == "bf16"


@pytest.mark.parametrize("name", ["hd64", "xl3"])
def test_fp16_training_gradients_vs_reference(dev, name):
    """--mixed-precision fp16: loss x 1024 backward through the IEEE-half build against the reference under
    autocast(float16) with the same scale (fp16.npz): every parameter's unscaled gradient norm and the element probes."""
    from reed_amd.loss import SILoss
    g = load("fp16")
    c = TINY_CASES[name]
    cfg = c["cfg"]
    T = (cfg["input_size"] // cfg["patch_size"]) ** 2
    x, noise, t, y, drop_u, zs = inputs(4, 4, cfg["input_size"], 11, c["zspec"], T, cfg["num_classes"])
    m = build_hip_model(cfg, dev, 11)
    m.precision = "fp16"
    m.train()
    m.force_drop_mask = drop_u < 0.1
    lf = SILoss(enc_names=c["enc"], loss_weights=dict(zip(c["enc"], c["co"])))
    out = lf(m, x.to(dev), dict(y=y.to(dev)), zs=[z.to(dev) for z in zs], time_input=t, noises=noise)
    total = out["denoising_loss"].mean() + 0.5 * out["proj_loss"]
    (total * 1024.0).backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(total.detach()), float(g[f"{name}.total"]), rtol=FP16_TRAIN_LOSS_BAR)
    np.testing.assert_allclose(out["denoising_loss"].detach().cpu().numpy(), g[f"{name}.denoising_loss"],
                               rtol=FP16_TRAIN_LOSS_BAR)
    worst = [0.0, ""]
    for k, p in m.named_parameters():
        if not p.requires_grad:
            continue
        ref = float(g[f"{name}.gnorm.{k}"])
        nh = p.grad.detach().double().norm().item() / 1024.0
        if ref < 5e-5:
            assert nh < 5e-4, (k, nh, ref)
            continue
        d = abs(nh / ref - 1)
        if d > worst[0]:
            worst = [d, k]
    worst_c = 1.0
    for k in ("final_layer.linear.weight", "blocks.0.attn.qkv.bias", "x_embedder.proj.weight", "projectors.0.4.bias",
              "blocks.1.adaLN_modulation.1.bias"):
        gh = dict(m.named_parameters())[k].grad.detach().cpu() / 1024.0
        worst_c = min(worst_c, cos(gh, torch.from_numpy(g[f"{name}.grad.{k}"])))
    print(f"[fp16 {name}] worst |norm ratio - 1| vs fp16 reference {worst[0]:.5f} ({worst[1]}), worst probe cosine "
          f"{worst_c:.6f}")
    assert worst[0] <= FP16_TRAIN_NORM_BAR, worst
    assert worst_c >= FP16_TRAIN_COS_BAR, worst_c


FP16_TRAIN_LOSS_BAR = 1e-3     # fp16 operands carry 3 more mantissa bits than bf16
FP16_TRAIN_NORM_BAR = 0.001
FP16_TRAIN_COS_BAR = 0.9999


def _fp16_trainer(dev, init_scale, **optkw):
    import copy
    from reed_amd.loss import SILoss
    from reed_amd.models.sit import SiT_models
    from reed_amd.optim import FusedAdamWEMA
    from reed_amd.trainer import TrainStep
    m = SiT_models["SiT-S/2"](z_dims=[768], z_types=["i"], encoder_depth=8)
    detfill.fill_state_dict(m.state_dict(), base_seed=0)
    m = m.to(dev).train()
    m.precision = "fp16"
    ema = copy.deepcopy(m).requires_grad_(False).eval()
    opt = FusedAdamWEMA(m, ema, lr=1e-4, max_grad_norm=1.0, init_scale=init_scale, **optkw)
    lf = SILoss(enc_names=["dinov2"], loss_weights={"dinov2": 1.0})
    ts = TrainStep(m, lf, opt, diffusion_warm_up_steps=0)
    return m, ema, opt, ts


def _fp16_steps(m, ts, opt, dev, steps):
    rec = {"loss": [], "grad_norm": [], "scale": []}
    for s in range(steps):
        x, noise, t, y, drop_u, zs = inputs(8, 4, 32, s, [(768, "i")], 256, 1000)
        m.force_drop_mask = drop_u < 0.1
        r = ts(x.to(dev), y.to(dev), [z.to(dev) for z in zs], time_input=t, noises=noise)
        rec["loss"].append(float(r["loss"])); rec["grad_norm"].append(float(r["grad_norm"]))
        rec["scale"].append(float(opt.scaler_state[0]))
    opt.flush()
    return rec


def test_fp16_training_trajectory_vs_reference(dev):
    """SiT-S/2 + 768-d alignment, B = 8, 6 optimiser steps under --mixed-precision fp16 through TrainStep (scaled loss,
    device-side unscale / clip / step / scale update) against the reference under autocast(float16) + GradScaler."""
    g = load("fp16")
    m, ema, opt, ts = _fp16_trainer(dev, 65536.0)
    rec = _fp16_steps(m, ts, opt, dev, 6)
    print("HIP :", [f"{v:.5f}" for v in rec["loss"]], [f"{v:.4f}" for v in rec["grad_norm"]])
    print("REF :", [f"{v:.5f}" for v in g["s2.loss"]], [f"{v:.4f}" for v in g["s2.grad_norm"]])
    np.testing.assert_allclose(rec["loss"], g["s2.loss"], atol=2e-4)      # measured 1e-5
    np.testing.assert_allclose(rec["grad_norm"], g["s2.grad_norm"], rtol=2e-3)   # measured 1e-4
    assert rec["scale"] == list(g["s2.scale"])
    assert opt.scaler_state.tolist() == [65536.0, 6.0, 0.0, 6.0]
    sd, esd = m.state_dict(), ema.state_dict()
    for k in ("blocks.0.attn.qkv.weight", "final_layer.linear.weight", "t_embedder.mlp.2.bias"):
        np.testing.assert_allclose(sd[k].flatten()[:64].cpu().numpy(), g["s2.w." + k], atol=2e-4)
        np.testing.assert_allclose(esd[k].flatten()[:64].cpu().numpy(), g["s2.ema." + k], atol=1e-5)
    assert {v["step"].item() for v in opt.state_dict()["state"].values()} == {6.0}


def test_fp16_overflow_steps_are_skipped(dev):
    """init_scale 2^40: every one of the 6 steps overflows fp16 in the backward. As GradScaler does: the logged norm is not
    finite, the scale halves, weights / Adam moments / step count stay, the EMA update still runs (and leaves EMA = weights)."""
    g = load("fp16")
    m, ema, opt, ts = _fp16_trainer(dev, 2.0 ** 40)
    w0 = m._arena.master.clone()
    rec = _fp16_steps(m, ts, opt, dev, 6)
    np.testing.assert_allclose(rec["loss"], g["s2_overflow.loss"], atol=2e-3)
    assert not np.isfinite(rec["grad_norm"]).any() and not np.isfinite(g["s2_overflow.grad_norm"]).any()
    assert rec["scale"] == list(g["s2_overflow.scale"])
    assert torch.equal(m._arena.master, w0) and torch.equal(ema._arena.master, w0)
    assert not opt.exp_avg.any() and not opt.exp_avg_sq.any()
    assert opt.scaler_state.tolist() == [2.0 ** 34, 0.0, 1.0, 0.0]
    assert opt.state_dict()["state"] == {}


def test_fp16_scale_grows_and_first_clean_step_matches(dev):
    """growth_interval clean steps double the scale (GradScaler.update); a run that first overflows once (scale 2^17 x 2^23)
    then takes the same first step as one that never did: skipped steps leave no trace in the bias corrections."""
    m, ema, opt, ts = _fp16_trainer(dev, 65536.0, growth_interval=2)
    rec = _fp16_steps(m, ts, opt, dev, 5)
    assert rec["scale"] == [65536.0, 131072.0, 131072.0, 262144.0, 262144.0]
    assert opt.scaler_state.tolist() == [262144.0, 1.0, 0.0, 5.0]
    a = _fp16_trainer(dev, 65536.0)
    b = _fp16_trainer(dev, 2.0 ** 40, backoff_factor=2.0 ** -24)
    x, noise, t, y, drop_u, zs = inputs(8, 4, 32, 0, [(768, "i")], 256, 1000)
    for mm, _, oo, tt in (a, b, b):
        mm.force_drop_mask = drop_u < 0.1
        tt(x.to(dev), y.to(dev), [z.to(dev) for z in zs], time_input=t, noises=noise)
        oo.flush()
    assert b[2].scaler_state.tolist() == [65536.0, 1.0, 0.0, 1.0]
    # both ran the clean step at scale 65536 from the same weights: the same update
    torch.testing.assert_close(b[0]._arena.master, a[0]._arena.master, atol=1e-7, rtol=0)
