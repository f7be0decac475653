"""Pin the oracle (oracle/) against golden vectors produced by the REFERENCE itself (tools/gen_golden.py).
CPU only. Tolerances: index bookkeeping and fp64 tables bit-exact; fp32 forward/backward 1e-5 relative
(same ops, same order up to BLAS scheduling)."""
import os

import numpy as np
import pytest
import torch

from oracle import detfill
from oracle import loss as oloss
from oracle import samplers as osamp
from oracle import sit as osit
from oracle import train_step as otrain

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    p = os.path.join(G, name + ".npz")
    if not os.path.exists(p):
        pytest.skip(f"{name}.npz not generated")
    return np.load(p)


def inputs(B, C=4, HW=32, seed=0, zdims=(), T=256, num_classes=1000):
    """Same recipe as tools/gen_golden.py:inputs."""
    x = detfill.normal((B, C, HW, HW), 1000 + seed)
    noise = detfill.normal((B, C, HW, HW), 2000 + seed)
    t = detfill.uniform((B,), 3000 + seed, 0.02, 0.98)
    y = (detfill.uniform((B,), 4000 + seed, 0.0, 1.0) * num_classes).long().clamp_(0, num_classes - 1)
    drop_u = detfill.uniform((B,), 5000 + seed, 0.0, 1.0)
    zs = [detfill.normal((B, T, z) if kind == "i" else (B, z), 6000 + seed + 17 * j)
          for j, (z, kind) in enumerate(zdims)]
    return x, noise, t, y, drop_u, zs


def tiny_cfg(D=128, heads=2, depth=3, **kw):
    d = dict(input_size=8, patch_size=2, in_channels=4, hidden_size=D, depth=depth, num_heads=heads, num_classes=10,
             z_dims=[128], z_types=["i"], encoder_depth=2, encoder_depth_text=None, projector_dim=128,
             class_dropout_prob=0.1, mlp_ratio=4.0, fused_attn=True, qk_norm=False)
    d.update(kw)
    return d


TINY_CASES = {
    "hd64": dict(cfg=tiny_cfg(), zspec=[(128, "i")], enc=["dinov2"], co=[1.0], hip=True),
    "hd72": dict(cfg=tiny_cfg(D=144, z_dims=[64]), zspec=[(64, "i")], enc=["dinov2"], co=[1.0], hip=False),
    "unfused": dict(cfg=tiny_cfg(fused_attn=False), zspec=[(128, "i")], enc=["dinov2"], co=[1.0], hip=True),
    "qknorm": dict(cfg=tiny_cfg(qk_norm=True), zspec=[(128, "i")], enc=["dinov2"], co=[1.0], hip=True),
    "two_same": dict(cfg=tiny_cfg(z_dims=[128, 256], z_types=["i", "t"]), zspec=[(128, "i"), (256, "t")],
                     enc=["clip", "text_embeds_qwenvl"], co=[1.0, 0.5], hip=True),
    "two_split": dict(cfg=tiny_cfg(z_dims=[128, 256], z_types=["i", "t"], encoder_depth=1, encoder_depth_text=3),
                      zspec=[(128, "i"), (256, "t")], enc=["clip", "text_embeds_qwenvl"], co=[1.0, 0.5], hip=True),
    "patch4": dict(cfg=tiny_cfg(input_size=16, patch_size=4), zspec=[(128, "i")], enc=["dinov2"], co=[1.0], hip=True),
    "xl3": dict(cfg=tiny_cfg(D=1152, heads=16, input_size=16, projector_dim=256), zspec=[(128, "i")],
                enc=["dinov2"], co=[1.0], hip=True),
}


def test_static_tables():
    g = load("static")
    pe = osit.pos_embed_table(384, 16)[0].numpy()
    assert np.array_equal(pe, g["pos_embed_384"])                      # bit-exact
    pe = osit.pos_embed_table(1152, 16)[0].numpy()
    assert np.array_equal(pe[::17], g["pos_embed_1152_rows"])
    assert np.array_equal(pe.astype(np.float64).sum(0), g["pos_embed_1152_sum"])
    assert np.array_equal(osit.pos_embed_table(128, 4)[0].numpy(), g["pos_embed_128_g4"])
    un = osit.unpatchify(torch.arange(256 * 16, dtype=torch.float32).reshape(1, 256, 16), 2, 4).long().numpy()
    assert np.array_equal(un, g["unpatchify_idx"])                      # bit-exact (M8)
    # patchify order (c, pi, pj): conv2d with one-hot weights
    w = torch.zeros(64, 4, 2, 2)
    for k in range(16):
        w.view(64, 16)[k, k] = 1.0
    xin = torch.arange(4 * 32 * 32, dtype=torch.float32).reshape(1, 4, 32, 32)
    pat = torch.nn.functional.conv2d(xin, w, None, stride=2).flatten(2).transpose(1, 2)[0, :, :16].long().numpy()
    assert np.array_equal(pat, g["patchify_idx"])
    s = osit.timestep_sinusoid(torch.from_numpy(g["sinus_t"]), 256).numpy()
    assert np.array_equal(s, g["sinus"])


@pytest.mark.parametrize("name", list(TINY_CASES))
def test_tiny_forward_backward(name):
    g = load("tiny")
    c = TINY_CASES[name]
    cfg = c["cfg"]
    P = detfill.fill_state_dict(osit.init_params(cfg), base_seed=11)
    P = {k: v.requires_grad_(k != "pos_embed") for k, v in P.items()}
    T = (cfg["input_size"] // cfg["patch_size"]) ** 2
    x, noise, t, y, drop_u, zs = inputs(4, 4, cfg["input_size"], 11, c["zspec"], T, cfg["num_classes"])
    model = osit.OracleModel(P, cfg, training=True)
    model.drop_mask = drop_u < cfg["class_dropout_prob"]
    out = oloss.si_loss(model, x, dict(y=y), zs, enc_names=c["enc"],
                        loss_weights=dict(zip(c["enc"], c["co"])), t=t, noise=noise)
    total = out["denoising_loss"].mean() + 0.5 * out["proj_loss"]
    total.backward()
    np.testing.assert_allclose(out["denoising_loss"].detach().numpy(), g[f"{name}.denoising_loss"], rtol=2e-5)
    np.testing.assert_allclose(float(out["proj_loss"]), g[f"{name}.proj_loss"], rtol=2e-5)
    np.testing.assert_allclose(float(out["img_proj_loss"]), g[f"{name}.img_proj_loss"], rtol=2e-5)
    np.testing.assert_allclose(float(out["text_proj_loss"]), g[f"{name}.text_proj_loss"], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(float(total), g[f"{name}.total"], rtol=2e-5)
    nchecked = 0
    for k, v in P.items():
        key = f"{name}.gnorm.{k}"
        if key in g.files:
            np.testing.assert_allclose(float(v.grad.double().norm()), g[key], rtol=2e-4, atol=1e-7)
            nchecked += 1
    assert nchecked >= len(P) - 3
    for k in ("final_layer.linear.weight", "blocks.0.attn.qkv.bias", "x_embedder.proj.weight", "x_embedder.proj.bias",
              "blocks.1.adaLN_modulation.1.bias", "blocks.2.mlp.fc1.bias", "projectors.0.4.bias",
              "final_layer.linear.bias"):
        np.testing.assert_allclose(P[k].grad.numpy(), g[f"{name}.grad.{k}"], rtol=2e-3, atol=2e-6)
    # eval-mode inference
    x, _, t, y, _, _ = inputs(4, 4, cfg["input_size"], 11, [], 0, 10)
    with torch.no_grad():
        o, z = osit.sit_forward({k: v.detach() for k, v in P.items()}, cfg, x, t, y, inference=True)
    assert z is None
    np.testing.assert_allclose(o.numpy(), g[f"{name}.infer"], rtol=1e-4, atol=2e-6)


def test_loss_units():
    g = load("loss_units")
    B = 6
    x, noise, t, y, _, zs = inputs(B, 4, 8, 21, [(32, "i"), (16, "t")], 16, 10)
    # the model's projector outputs are bf16 values (upcast by accelerate): keep the stand-in's bf16-representable, so the
    # HIP SILoss (bf16 projector inputs, tests/test_kernels_gpu.py) sees exactly the values the reference saw
    zt = [detfill.normal((B, 16, 32), 901).bfloat16().float(), detfill.normal((B, 16), 902).bfloat16().float()]
    vel = detfill.normal((B, 4, 8, 8), 903)
    for sched in ["constant", "linear", "cosine", "sigmoid", "loglinear", "cutoff"]:
        for path in ["linear", "cosine"]:
            o = oloss.si_loss(lambda xx, tt, **k: (vel + 0.1 * xx, zt), x, dict(y=y), zs,
                              enc_names=["clip", "text_embeds_qwenvl"],
                              loss_weights={"clip": 1.0, "text_embeds_qwenvl": 0.5}, path_type=path,
                              time_schedule=sched, cutoffs=[0.2, 0.8], t=t, noise=noise)
            for k in ("denoising_loss", "proj_loss", "img_proj_loss", "text_proj_loss"):
                np.testing.assert_allclose(np.asarray(o[k]), g[f"{sched}.{path}.{k}"], rtol=1e-6, atol=1e-7)
    o = oloss.si_loss(lambda xx, tt, **k: (vel, [zt[0]]), x, dict(y=y), [zs[0]], enc_names=["text_embeds_qwenvl"],
                      loss_weights={"text_embeds_qwenvl": 0.0}, time_schedule="linear", t=t, noise=noise)
    np.testing.assert_allclose(float(o["proj_loss"]), g["zero_weight.proj_loss"], rtol=1e-6)
    np.testing.assert_allclose(float(o["img_proj_loss"]), g["zero_weight.img_proj_loss"], rtol=1e-6)
    # the [B] x [B,1,1,1] broadcast quirk: proj = mean(cur) * mean(w)   
    o = oloss.si_loss(lambda xx, tt, **k: (vel, [zt[0]]), x, dict(y=y), [zs[0]], enc_names=["dinov2"],
                      loss_weights={"dinov2": 1.0}, time_schedule="linear", t=t, noise=noise)
    np.testing.assert_allclose(float(o["proj_loss"]), float(o["img_proj_loss"]) * float((1 - t).mean()), rtol=1e-5)
    # lognormal t transform
    rn = detfill.normal((B, 1, 1, 1), 77)
    for path in ["linear", "cosine"]:
        sig = rn.exp()
        tt = sig / (1 + sig) if path == "linear" else 2 / np.pi * torch.atan(sig)
        np.testing.assert_allclose(tt.flatten().numpy(), g[f"lognormal.{path}.t"], rtol=1e-6)


def test_samplers():
    g = load("samplers")
    cfg = tiny_cfg(num_classes=1000)
    P = detfill.fill_state_dict(osit.init_params(cfg), base_seed=5)
    model = osit.OracleModel(P, cfg)
    z = detfill.normal((3, 4, 8, 8), 41)
    y = torch.tensor([3, 500, 999])
    cfgs = {"euler": dict(heun=False, cfg_scale=1.0), "heun": dict(heun=True, cfg_scale=1.0),
            "euler_cfg": dict(heun=False, cfg_scale=2.5), "heun_cfg": dict(heun=True, cfg_scale=1.5),
            "heun_cfg_interval": dict(heun=True, cfg_scale=3.0, guidance_low=0.3, guidance_high=0.75)}
    for name, c in cfgs.items():
        out = osamp.euler_sampler(model, z, y, num_steps=6, **c)
        assert out.dtype == torch.float64
        np.testing.assert_allclose(out.numpy(), g[name], rtol=1e-4, atol=1e-5)
    eps = [detfill.normal((3, 4, 8, 8), 600 + i).double() for i in range(8)]
    for name, c in {"sde": dict(cfg_scale=1.0), "sde_cfg": dict(cfg_scale=2.0, guidance_high=0.9),
                    "sde_cosine": dict(cfg_scale=1.0, path_type="cosine")}.items():
        out = osamp.euler_maruyama_sampler(model, z, y, num_steps=6, noises=eps, **c)
        np.testing.assert_allclose(out.numpy(), g[name], rtol=1e-4, atol=1e-5)


def test_schedules_hand_values():
    """train.py:363-385 (not importable: needs diffusers/wandb) — pinned by hand-evaluated values."""
    assert otrain.repa_weight_decay("constant", 123, 1000) == 1.0
    assert otrain.repa_weight_decay("linear", 250, 1000) == 0.75
    assert otrain.repa_weight_decay("linear", 2000, 1000) == 0.0
    assert abs(otrain.repa_weight_decay("cosine", 500, 1000) - 0.5) < 1e-12
    assert otrain.diffusion_loss_decay("constant", 0, 0, 50000, 400000) == 0.0         # step-0 weight is 0
    assert otrain.diffusion_loss_decay("constant", 25000, 0, 50000, 400000) == 0.5
    assert otrain.diffusion_loss_decay("constant", 50000, 0, 50000, 400000) == 1.0
    assert otrain.diffusion_loss_decay("constant", 5, 10, 100, 1000) == 0.0
    assert otrain.diffusion_loss_decay("linear", 555, 10, 100, 1000) == 1.0 - (555 - 110) / (1000 - 110)
    v = otrain.diffusion_loss_decay("cosine", 555, 10, 100, 1000)                      # precedence quirk §9-7
    assert abs(v - (1.0 + np.cos(np.pi * 445 / 1000 - 110)) / 2) < 1e-12


def _traj(name, model_name, cfgkw, B, steps, zspec, enc, co, tag, autocast, align=True, rtol=2e-4):
    g = load(name)
    cfg = osit.make_config(model_name, **cfgkw)
    P = detfill.fill_state_dict(osit.init_params(cfg), base_seed=0)
    tr = otrain.Trainer(P, cfg, enc if align else [], co if align else [], autocast_bf16=autocast,
                        diffusion_warm_up_steps=0) if align else None
    if not align:
        cfg2 = dict(cfg, z_dims=[], z_types=[])
        P = {k: v for k, v in P.items() if not k.startswith("projectors.")}
        tr = otrain.Trainer(P, cfg2, [], [], autocast_bf16=autocast, diffusion_warm_up_steps=0)
    pre = (tag + ".") if tag else ""
    for s in range(steps):
        x, noise, t, y, drop_u, zs = inputs(B, 4, 32, s, zspec, 256, 1000)
        r = tr.step(x, y, zs if align else [], t=t, noise=noise, drop_mask=drop_u < 0.1)
        np.testing.assert_allclose(r["denoising_loss"], g[pre + "denoising_loss"][s], rtol=rtol)
        np.testing.assert_allclose(r["grad_norm"], g[pre + "grad_norm"][s], rtol=5 * rtol)
        if align:
            np.testing.assert_allclose(r["proj_loss"], g[pre + "proj_loss"][s], rtol=rtol, atol=1e-6)
            np.testing.assert_allclose(r["loss"], g[pre + "loss"][s], rtol=rtol)
    for k in ("blocks.0.attn.qkv.weight", "final_layer.linear.weight", "t_embedder.mlp.2.bias"):
        np.testing.assert_allclose(tr.P[k].detach().flatten()[:64].numpy(), g[pre + "w." + k], rtol=1e-3, atol=2e-6)
    return tr, g


def test_s2_c1_trajectory_prefix():
    """C1 (SiT-S/2, B=64, alignment off): first 2 of the 10 reference steps on CPU (the full 10 run on the GPU box)."""
    g = load("s2_c1")
    cfg = osit.make_config("SiT-S/2", z_dims=[], z_types=[])
    P = detfill.fill_state_dict(osit.init_params(cfg), base_seed=0)
    tr = otrain.Trainer(P, cfg, [], [], diffusion_warm_up_steps=0)
    for s in range(2):
        x, noise, t, y, drop_u, _ = inputs(64, 4, 32, s, [], 256, 1000)
        r = tr.step(x, y, [], t=t, noise=noise, drop_mask=drop_u < 0.1)
        np.testing.assert_allclose(r["denoising_loss"], g["denoising_loss"][s], rtol=2e-4)
        np.testing.assert_allclose(r["grad_norm"], g["grad_norm"][s], rtol=1e-3)


def test_b2_alignment_fp32_and_bf16():
    g = load("b2_align")
    for tag, ac, rtol in (("fp32", False, 2e-4), ("bf16", True, 3e-3)):
        cfg = osit.make_config("SiT-B/2", z_dims=[768], z_types=["i"], encoder_depth=4)
        P = detfill.fill_state_dict(osit.init_params(cfg), base_seed=0)
        tr = otrain.Trainer(P, cfg, ["dinov2"], [1.0], autocast_bf16=ac, diffusion_warm_up_steps=0)
        for s in range(2):
            x, noise, t, y, drop_u, zs = inputs(8, 4, 32, s, [(768, "i")], 256, 1000)
            r = tr.step(x, y, zs, t=t, noise=noise, drop_mask=drop_u < 0.1)
            np.testing.assert_allclose(r["loss"], g[f"{tag}.loss"][s], rtol=rtol)
            np.testing.assert_allclose(r["denoising_loss"], g[f"{tag}.denoising_loss"][s], rtol=rtol)
            np.testing.assert_allclose(r["proj_loss"], g[f"{tag}.proj_loss"][s], rtol=rtol, atol=1e-5)


def test_optim_toy():
    """clip_grad_norm_(1.0) + AdamW(wd) + EMA on three tensors: the arithmetic the fused HIP optimiser mirrors."""
    g = load("optim_toy")
    shapes = [(5, 7), (11,), (3, 4, 2)]
    ps = [detfill.normal(s, 70 + i).double() for i, s in enumerate(shapes)]
    ema = [p.clone() for p in ps]
    m = [torch.zeros_like(p) for p in ps]
    v = [torch.zeros_like(p) for p in ps]
    lr, b1, b2, eps, wd = 1e-2, 0.9, 0.999, 1e-8, 0.01
    for s in range(3):
        gr = [(detfill.normal(sh, 80 + 10 * s + i) * (3.0 if s == 0 else 0.1)).double() for i, sh in enumerate(shapes)]
        norm = torch.sqrt(sum((x.float() ** 2).sum() for x in gr)).double()
        np.testing.assert_allclose(float(norm), g[f"gn{s}"], rtol=1e-5)
        coef = min(1.0, 1.0 / (float(norm) + 1e-6))
        for i in range(3):
            gg = gr[i] * coef
            ps[i] = ps[i] * (1 - lr * wd)
            m[i] = m[i] + (1 - b1) * (gg - m[i])
            v[i] = v[i] * b2 + (1 - b2) * gg * gg
            bc1, bc2 = 1 - b1 ** (s + 1), 1 - b2 ** (s + 1)
            ps[i] = ps[i] - (lr / bc1) * m[i] / (v[i].sqrt() / bc2 ** 0.5 + eps)
            ema[i] = ema[i] * 0.99 + ps[i] * 0.01
            np.testing.assert_allclose(ps[i].numpy(), g[f"p{s}_{i}"], rtol=2e-5, atol=1e-6)
            np.testing.assert_allclose(ema[i].numpy(), g[f"e{s}_{i}"], rtol=2e-5, atol=1e-6)


def test_clip_encoder_oracle_vs_reference():
    """Frozen CLIP image encoder: oracle/clip_vit.py against the outputs of the reference's own
    LayerNorm / ResidualAttentionBlock / Transformer / UpdatedVisionTransformer (clip_vit.py:159-230), fp32 ≤ 2e-5 and
    bf16-autocast within bf16 noise; the 'clip' preprocessing geometry (train.py:53-57) on a fixed ramp."""
    from oracle import clip_vit as oclip
    g = load("clip")
    for tag, cfg, B in (("t2", oclip.make_config(width=128, layers=2, heads=2, patch=14, image=56), 3),
                        ("t3", oclip.make_config(width=256, layers=3, heads=4, patch=14, image=28), 2)):
        P = oclip.fill_params(cfg, base_seed=5)
        x = detfill.normal((B, 3, cfg["image"], cfg["image"]), 77)
        with torch.no_grad():
            o32 = oclip.forward(P, cfg, x)
            o16 = oclip.forward(P, cfg, x, autocast_bf16=True).float()
        ref32, ref16 = torch.from_numpy(g[tag + ".fp32"]), torch.from_numpy(g[tag + ".bf16"])
        assert o32.shape == ref32.shape == (B, (cfg["image"] // 14) ** 2, cfg["width"])
        torch.testing.assert_close(o32, ref32, atol=2e-5, rtol=2e-5)
        torch.testing.assert_close(o16, ref16, atol=2e-2, rtol=2e-2)
    raw = (torch.arange(2 * 3 * 256 * 256) % 251).reshape(2, 3, 256, 256).to(torch.uint8)
    pre = oclip.preprocess(raw)
    mean = torch.tensor(oclip.CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(oclip.CLIP_STD).view(1, 3, 1, 1)
    assert pre.shape == (2, 3, 224, 224)
    torch.testing.assert_close((pre * std + mean)[:, :, ::37, ::41], torch.from_numpy(g["pre.sample"]), atol=1e-5, rtol=1e-5)


def test_vit_towers():
    """oracle/vit_towers.py against the reference's own JEPA class, its MAE forward_features and MoCo-v3 constructor /
    pos-embed (tools/gen_golden.py:g_towers): fp32 to 2e-5, bf16-autocast to bf16 noise; pos-embed tables of the real
    grids and the preprocess geometry of both resample orders exactly / to 1e-6."""
    from oracle import vit_towers as ot
    g = load("towers")
    cases = {"jepa80": (ot.make_config(640, 2, 8, 14, 56, False, True, "jepa"), 3),
             "jepa64": (ot.make_config(256, 3, 4, 14, 56, False, True, "jepa"), 3),
             "mae": (ot.make_config(256, 2, 4, 16, 64, True, False, "learned"), 2),
             "moco": (ot.make_config(256, 2, 4, 16, 64, True, True, "moco"), 2)}
    for tag, (cfg, B) in cases.items():
        P = ot.fill_params(cfg, base_seed=9)
        x = detfill.normal((B, 3, cfg["image"], cfg["image"]), 55)
        with torch.no_grad():
            o32 = ot.forward(P, cfg, x).numpy()
            o16 = ot.forward(P, cfg, x, autocast_bf16=True).float().numpy()
        assert o32.shape == g[tag + ".fp32"].shape
        np.testing.assert_allclose(o32, g[tag + ".fp32"], rtol=2e-5, atol=2e-5)
        sc = np.abs(g[tag + ".fp32"]).max()
        assert np.abs(o16 - g[tag + ".bf16"]).max() <= 2e-2 * sc, tag
    pe = ot.jepa_pos_embed(1280, 16)[0]
    assert np.array_equal(pe[::17].numpy(), g["jepa_pos_1280_rows"])
    assert np.array_equal(pe.double().sum(0).numpy(), g["jepa_pos_1280_sum"])
    pm = ot.moco_pos_embed(768, 16, 16)[0]
    assert np.array_equal(pm[::17].numpy(), g["moco_pos_768_rows"])
    raw = (torch.arange(2 * 3 * 256 * 256) % 251).reshape(2, 3, 256, 256).to(torch.uint8)
    np.testing.assert_allclose(ot.preprocess(raw, "jepa")[:, :, ::37, ::41].numpy(), g["pre.jepa.sample"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ot.preprocess(raw, "mae")[:, :, ::37, ::41].numpy(), g["pre.mae.sample"], rtol=1e-6, atol=1e-6)
    assert ot.preprocess(raw, "mocov3").shape == (2, 3, 256, 256) and ot.preprocess(raw, "dinov2").shape == (2, 3, 224, 224)


def test_dinov2_tower():
    """oracle/vit_towers.py's DINOv2 restatement (class token, learned pos-embed, 0 / 4 register tokens behind the class
    token, LayerScale, final norm, patch tokens out) against transformers' Dinov2Model / Dinov2WithRegistersModel
    (tests/golden/dinov2.npz, tools/gen_golden.py:g_dinov2), and the pos-embed resampling of image/utils.py:99-101."""
    from oracle import vit_towers as ot
    g = load("dinov2")
    for tag, E, H, depth, image, reg, B in (("plain", 128, 2, 2, 56, 0, 3), ("reg4", 256, 4, 3, 28, 4, 2)):
        cfg = ot.make_config(E, depth, H, 14, image, True, True, "learned", ls=True, reg=reg)
        P = ot.fill_params(cfg, base_seed=21)
        x = detfill.normal((B, 3, image, image), 56)
        with torch.no_grad():
            o32 = ot.forward(P, cfg, x).numpy()
            o16 = ot.forward(P, cfg, x, autocast_bf16=True).float().numpy()
        assert o32.shape == g[tag + ".fp32"].shape == (B, (image // 14) ** 2, E)
        np.testing.assert_allclose(o32, g[tag + ".fp32"], rtol=2e-5, atol=2e-5)
        sc = np.abs(g[tag + ".fp32"]).max()
        assert np.abs(o16 - g[tag + ".bf16"]).max() <= 2e-2 * sc, tag
    pe = detfill.normal((1, 1 + 37 * 37, 64), 57)
    np.testing.assert_allclose(ot.resample_abs_pos_embed(pe, (16, 16)).numpy(), g["pos_resample"], rtol=1e-6, atol=1e-6)
    assert ot.resample_abs_pos_embed(pe, (37, 37)) is pe


def test_fp16_autocast_and_grad_scaler():
    """--mixed-precision fp16 (fp16.npz: the reference under autocast(float16) + GradScaler): the oracle's tiny-case
    losses, two steps of the S/2 trajectory, and the skipped-step path (scale halves, weights stay)."""
    g = load("fp16")
    for name in ("hd64", "xl3"):
        case = TINY_CASES[name]
        cfg = case["cfg"]
        P = detfill.fill_state_dict(osit.init_params(cfg), base_seed=11)
        P = {k: v.clone().requires_grad_(k != "pos_embed") for k, v in P.items()}
        T = (cfg["input_size"] // cfg["patch_size"]) ** 2
        x, noise, t, y, drop_u, zs = inputs(4, 4, cfg["input_size"], 11, case["zspec"], T, cfg["num_classes"])
        om = osit.OracleModel(P, cfg, training=True, autocast_dtype=torch.float16)
        om.drop_mask = drop_u < 0.1
        o = oloss.si_loss(om, x, dict(y=y), zs, enc_names=case["enc"], loss_weights={"dinov2": 1.0}, t=t, noise=noise)
        total = o["denoising_loss"].mean() + 0.5 * o["proj_loss"]
        np.testing.assert_allclose(float(total.detach()), g[f"{name}.total"], rtol=1e-3)
        (total * 1024.0).backward()
        for k in ("final_layer.linear.weight", "blocks.0.attn.qkv.bias", "projectors.0.4.bias"):
            ref = g[f"{name}.grad.{k}"]
            got = (P[k].grad / 1024.0).numpy()
            assert np.abs(got - ref).max() <= 2e-2 * np.abs(ref).max() + 1e-7, (name, k)
    cfg = osit.make_config("SiT-S/2", z_dims=[768], z_types=["i"], encoder_depth=8)
    for tag, init, steps in (("s2", 65536.0, 2), ("s2_overflow", 2.0 ** 40, 2)):
        P = detfill.fill_state_dict(osit.init_params(cfg), base_seed=0)
        tr = otrain.Trainer(P, cfg, ["dinov2"], [1.0], diffusion_warm_up_steps=0, autocast_dtype=torch.float16,
                            init_scale=init)
        for s in range(steps):
            x, noise, t, y, drop_u, zs = inputs(8, 4, 32, s, [(768, "i")], 256, 1000)
            r = tr.step(x, y, zs, t=t, noise=noise, drop_mask=drop_u < 0.1)
            np.testing.assert_allclose(r["loss"], g[f"{tag}.loss"][s], rtol=2e-3)
            assert r["scale"] == g[f"{tag}.scale"][s]
            if tag == "s2":
                np.testing.assert_allclose(r["grad_norm"], g[f"{tag}.grad_norm"][s], rtol=1e-2)
            else:
                assert not np.isfinite(r["grad_norm"]) and not np.isfinite(g[f"{tag}.grad_norm"][s])
        if tag == "s2_overflow":   # every step skipped: the weights are the initial ones
            assert torch.equal(tr.P["blocks.0.attn.qkv.weight"].detach(), P["blocks.0.attn.qkv.weight"])
