#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE (ChenyuWang-Monica/REED, image/) in this container.

Runs only where /root/reference exists (the authoring container); the fixtures it writes are committed and are
the only thing that travels. The reference's models/sit.py imports three classes from `timm`, which is not
installed and not vendored (image/requirements.txt:5, unpinned): this script supplies a minimal stand-in for
exactly those classes with timm>=0.9 semantics in a temporary directory on sys.path.
Weights and inputs come from oracle.detfill (hash of the element index), so tests regenerate them instead of
shipping them. Random draws inside the reference (t, noise, label drop) are injected by temporarily replacing
torch.rand / torch.randn_like.

usage: python tools/gen_golden.py [--only NAME ...] [--skip-xl]
"""
import argparse
import contextlib
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import detfill  # noqa: E402

REF = "/root/reference/image"
OUT = os.path.join(ROOT, "tests", "golden")

TIMM_STANDIN = '''
import torch, torch.nn as nn, torch.nn.functional as F
class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, bias=True):
        super().__init__()
        self.patch_size = (patch_size, patch_size)
        self.num_patches = (img_size // patch_size) ** 2
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)
    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)
class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_norm=False):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.fused_attn = True
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.q_norm = nn.LayerNorm(self.head_dim) if qk_norm else nn.Identity()
        self.k_norm = nn.LayerNorm(self.head_dim) if qk_norm else nn.Identity()
        self.proj = nn.Linear(dim, dim)
    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        q, k = self.q_norm(q), self.k_norm(k)
        if self.fused_attn:
            x = F.scaled_dot_product_attention(q, k, v)
        else:
            attn = (q * self.scale) @ k.transpose(-2, -1)
            x = attn.softmax(dim=-1) @ v
        return self.proj(x.transpose(1, 2).reshape(B, N, C))
class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, in_features)
    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))
# --- the container image/models/mae_vit.py and mocov3_vit.py subclass (timm >= 0.9 semantics, inference subset) ---
def _cfg(**kw):
    return dict(kw)
class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))
class VisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4.0, qkv_bias=True, norm_layer=None, embed_layer=PatchEmbed, **kw):
        super().__init__()
        norm_layer = norm_layer or nn.LayerNorm
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = embed_layer(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        self.patch_embed.img_size = (img_size, img_size)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.randn(1, self.patch_embed.num_patches + 1, embed_dim) * .02)
        self.pos_drop = nn.Identity()
        self.patch_drop = nn.Identity()
        self.norm_pre = nn.Identity()
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads, mlp_ratio, qkv_bias, norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
    def _pos_embed(self, x):
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1)
        return self.pos_drop(x + self.pos_embed)
    def forward_features(self, x):
        x = self.patch_embed(x)
        x = self._pos_embed(x)
        x = self.patch_drop(x)
        x = self.norm_pre(x)
        x = self.blocks(x)
        return self.norm(x)
'''


def import_reference():
    d = tempfile.mkdtemp(prefix="timm_standin_")
    os.makedirs(os.path.join(d, "timm", "models"))
    open(os.path.join(d, "timm", "__init__.py"), "w").close()
    open(os.path.join(d, "timm", "models", "__init__.py"), "w").close()
    with open(os.path.join(d, "timm", "models", "vision_transformer.py"), "w") as f:
        f.write(TIMM_STANDIN)
    os.makedirs(os.path.join(d, "timm", "layers"))
    open(os.path.join(d, "timm", "layers", "__init__.py"), "w").close()
    with open(os.path.join(d, "timm", "layers", "helpers.py"), "w") as f:
        f.write("def to_2tuple(x):\n    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)\n")
    sys.path.insert(0, d)
    sys.path.insert(0, REF)
    import loss as ref_loss  # noqa
    import samplers as ref_samplers  # noqa
    from models import sit as ref_sit  # noqa
    return ref_sit, ref_loss, ref_samplers


@contextlib.contextmanager
def inject(t=None, noise=None, drop_u=None, eps_list=None):
    """Replace the reference's random draws: torch.rand((B,1,1,1)) -> t, torch.rand(B) -> drop_u,
    torch.randn_like(x) -> noise (or successive eps_list entries)."""
    o_rand, o_randn_like = torch.rand, torch.randn_like
    it = iter(eps_list) if eps_list is not None else None

    def rand(*size, **kw):
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
        if len(shape) == 4 and t is not None:
            return t.reshape(shape).clone()
        if len(shape) == 1 and drop_u is not None:
            return drop_u.clone()
        return o_rand(*size, **kw)

    def randn_like(x, **kw):
        if it is not None:
            return next(it).to(x.dtype).clone()
        if noise is not None:
            return noise.to(x.dtype).clone()
        return o_randn_like(x, **kw)

    torch.rand, torch.randn_like = rand, randn_like
    try:
        yield
    finally:
        torch.rand, torch.randn_like = o_rand, o_randn_like


def build_ref_model(ref_sit, name, seed=0, **kw):
    if name.startswith("SiT-S"):
        kw.setdefault("decoder_hidden_size", 384)
    m = ref_sit.SiT_models[name](**kw) if name in ref_sit.SiT_models else ref_sit.SiT(**kw)
    detfill.fill_state_dict(m.state_dict(), base_seed=seed)
    return m


def tiny_kwargs(D=128, heads=2, depth=3, **kw):
    d = dict(input_size=8, patch_size=2, in_channels=4, hidden_size=D, decoder_hidden_size=D, depth=depth,
             num_heads=heads, num_classes=10, z_dims=[128], z_types=["i"], encoder_depth=2, projector_dim=128,
             fused_attn=True, qk_norm=False)
    d.update(kw)
    return d


def inputs(B, C=4, HW=32, seed=0, zdims=(), T=256, num_classes=1000):
    x = detfill.normal((B, C, HW, HW), 1000 + seed)
    noise = detfill.normal((B, C, HW, HW), 2000 + seed)
    t = detfill.uniform((B,), 3000 + seed, 0.02, 0.98)
    y = (detfill.uniform((B,), 4000 + seed, 0.0, 1.0) * num_classes).long().clamp_(0, num_classes - 1)
    drop_u = detfill.uniform((B,), 5000 + seed, 0.0, 1.0)
    zs = [detfill.normal((B, T, z) if kind == "i" else (B, z), 6000 + seed + 17 * j)
          for j, (z, kind) in enumerate(zdims)]
    return x, noise, t, y, drop_u, zs


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    p = os.path.join(OUT, name + ".npz")
    np.savez_compressed(p, **out)
    print(f"  wrote {p} ({os.path.getsize(p) / 1024:.1f} KiB)")


# ---------------------------------------------------------------------------------------------
def g_static(ref_sit, ref_loss, ref_samplers):
    """G-a pos-embed, G-b patch index maps, G-c timestep sinusoid."""
    pe384 = ref_sit.get_2d_sincos_pos_embed(384, 16).astype(np.float32)
    pe1152 = ref_sit.get_2d_sincos_pos_embed(1152, 16).astype(np.float32)
    pe128_4 = ref_sit.get_2d_sincos_pos_embed(128, 4).astype(np.float32)
    m = ref_sit.SiT(**tiny_kwargs(input_size=32, D=64, heads=1, depth=1))
    # unpatchify on arange
    T, NO = 256, 16
    un = m.unpatchify(torch.arange(T * NO, dtype=torch.float32).reshape(1, T, NO)).long()
    # patchify order via one-hot conv weights: token feature k picks input element k of the patch
    with torch.no_grad():
        w = torch.zeros(64, 4, 2, 2)
        for k in range(16):
            w.view(64, 16)[k, k] = 1.0
        m.x_embedder.proj.weight.copy_(w)
        m.x_embedder.proj.bias.zero_()
        xin = torch.arange(4 * 32 * 32, dtype=torch.float32).reshape(1, 4, 32, 32)
        pat = m.x_embedder(xin)[0, :, :16].long()  # [T,16]: flat input index feeding (token, k)
    tvals = torch.tensor([0.0, 1e-3, 0.04, 0.25, 0.5, 0.731, 0.999, 1.0])
    sinus = ref_sit.TimestepEmbedder.positional_embedding(tvals, 256)
    save("static", pos_embed_384=pe384, pos_embed_1152_rows=pe1152[::17], pos_embed_1152_sum=pe1152.astype(np.float64).sum(0),
         pos_embed_128_g4=pe128_4, unpatchify_idx=un, patchify_idx=pat, sinus_t=tvals, sinus=sinus)


def run_fwd_bwd(ref_sit, ref_loss, kw, B, seed, zspec, enc_names, coeffs, autocast=False, path_type="linear",
                time_schedule="constant", train_mode=True):
    m = build_ref_model(ref_sit, "custom", seed=seed, **kw)
    m.train(train_mode)
    T = (kw["input_size"] // kw["patch_size"]) ** 2
    x, noise, t, y, drop_u, zs = inputs(B, kw["in_channels"], kw["input_size"], seed, zspec, T, kw["num_classes"])
    lf = ref_loss.SILoss(path_type=path_type, enc_names=list(enc_names),
                         loss_weights={n: c for n, c in zip(enc_names, coeffs)}, time_schedule=time_schedule)

    def model(xx, tt, **k):
        if autocast:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                o, z = m(xx, tt, **k)
            return o.float(), [a.float() for a in z]
        return m(xx, tt, **k)

    with inject(t=t, noise=noise, drop_u=drop_u):
        out = lf(model, x, dict(y=y), zs=zs)
    total = out["denoising_loss"].mean() + 0.5 * out["proj_loss"]
    total.backward()
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    return m, out, total, grads, (x, noise, t, y, drop_u, zs)


def g_tiny(ref_sit, ref_loss, ref_samplers):
    """G-d / G-g: tiny SiT forward + grads in several structural variants (fp32)."""
    cases = {
        "hd64": dict(kw=tiny_kwargs(), zspec=[(128, "i")], enc=["dinov2"], co=[1.0]),
        "hd72": dict(kw=tiny_kwargs(D=144, z_dims=[64]), zspec=[(64, "i")], enc=["dinov2"], co=[1.0]),
        "unfused": dict(kw=tiny_kwargs(fused_attn=False), zspec=[(128, "i")], enc=["dinov2"], co=[1.0]),
        "qknorm": dict(kw=tiny_kwargs(qk_norm=True), zspec=[(128, "i")], enc=["dinov2"], co=[1.0]),
        "two_same": dict(kw=tiny_kwargs(z_dims=[128, 256], z_types=["i", "t"]), zspec=[(128, "i"), (256, "t")],
                         enc=["clip", "text_embeds_qwenvl"], co=[1.0, 0.5]),
        "two_split": dict(kw=tiny_kwargs(z_dims=[128, 256], z_types=["i", "t"], encoder_depth=1, encoder_depth_text=3),
                          zspec=[(128, "i"), (256, "t")], enc=["clip", "text_embeds_qwenvl"], co=[1.0, 0.5]),
        "patch4": dict(kw=tiny_kwargs(input_size=16, patch_size=4), zspec=[(128, "i")], enc=["dinov2"], co=[1.0]),
        "xl3": dict(kw=tiny_kwargs(D=1152, heads=16, input_size=16, projector_dim=256), zspec=[(128, "i")],
                    enc=["dinov2"], co=[1.0]),
    }
    out = {}
    for name, c in cases.items():
        m, o, total, grads, _ = run_fwd_bwd(ref_sit, ref_loss, c["kw"], 4, 11, c["zspec"], c["enc"], c["co"])
        out[f"{name}.total"] = total
        out[f"{name}.denoising_loss"] = o["denoising_loss"]
        out[f"{name}.proj_loss"] = o["proj_loss"]
        out[f"{name}.img_proj_loss"] = torch.as_tensor(o["img_proj_loss"])
        out[f"{name}.text_proj_loss"] = torch.as_tensor(o["text_proj_loss"])
        for k, g in grads.items():
            out[f"{name}.gnorm.{k}"] = g.double().norm()
        out[f"{name}.grad.final_layer.linear.weight"] = grads["final_layer.linear.weight"]
        out[f"{name}.grad.blocks.0.attn.qkv.bias"] = grads["blocks.0.attn.qkv.bias"]
        out[f"{name}.grad.x_embedder.proj.weight"] = grads["x_embedder.proj.weight"]
        out[f"{name}.grad.x_embedder.proj.bias"] = grads["x_embedder.proj.bias"]
        out[f"{name}.grad.t_embedder.mlp.0.weight"] = grads["t_embedder.mlp.0.weight"][:, ::16]
        out[f"{name}.grad.blocks.1.adaLN_modulation.1.bias"] = grads["blocks.1.adaLN_modulation.1.bias"]
        out[f"{name}.grad.blocks.2.mlp.fc1.bias"] = grads["blocks.2.mlp.fc1.bias"]
        out[f"{name}.grad.projectors.0.4.bias"] = grads["projectors.0.4.bias"]
        out[f"{name}.grad.final_layer.linear.bias"] = grads["final_layer.linear.bias"]
        out[f"{name}.grad.y_embedder.rows"] = grads["y_embedder.embedding_table.weight"][:, :8]
        # eval-mode inference forward
        m.eval()
        x, _, t, y, _, _ = inputs(4, 4, c["kw"]["input_size"], 11, [], 0, 10)
        with torch.no_grad():
            out[f"{name}.infer"] = m(x, t, y)[0]
    save("tiny", **out)


def g_loss_units(ref_sit, ref_loss, ref_samplers):
    """G-h: SILoss over time_schedule x path_type x weighting with a fixed stand-in model."""
    B = 6
    x, noise, t, y, _, zs = inputs(B, 4, 8, 21, [(32, "i"), (16, "t")], 16, 10)
    # the model's projector outputs are bf16 values (upcast by accelerate): keep the stand-in's bf16-representable, so the
    # HIP SILoss (bf16 projector inputs, tests/test_kernels_gpu.py) sees exactly the values the reference saw
    zt = [detfill.normal((B, 16, 32), 901).bfloat16().float(), detfill.normal((B, 16), 902).bfloat16().float()]
    vel = detfill.normal((B, 4, 8, 8), 903)

    def model(xx, tt, **k):
        return vel + 0.1 * xx, zt

    out = {}
    for sched in ["constant", "linear", "cosine", "sigmoid", "loglinear", "cutoff"]:
        for path in ["linear", "cosine"]:
            lf = ref_loss.SILoss(path_type=path, enc_names=["clip", "text_embeds_qwenvl"],
                                 loss_weights={"clip": 1.0, "text_embeds_qwenvl": 0.5}, time_schedule=sched,
                                 cutoffs=[0.2, 0.8])
            with inject(t=t, noise=noise):
                o = lf(model, x, dict(y=y), zs=zs)
            for k in ("denoising_loss", "proj_loss", "img_proj_loss", "text_proj_loss"):
                out[f"{sched}.{path}.{k}"] = torch.as_tensor(o[k])
    # zero-weight encoder branch + single-encoder keying
    lf = ref_loss.SILoss(enc_names=["text_embeds_qwenvl"], loss_weights={"text_embeds_qwenvl": 0.0}, time_schedule="linear")
    with inject(t=t, noise=noise):
        o = lf(lambda xx, tt, **k: (vel, [zt[0]]), x, dict(y=y), zs=[zs[0]])
    out["zero_weight.proj_loss"] = o["proj_loss"]
    out["zero_weight.img_proj_loss"] = torch.as_tensor(o["img_proj_loss"])
    # lognormal time sampling transform (weighting) with injected normal draws
    rn = detfill.normal((B, 1, 1, 1), 77)
    o_randn = torch.randn
    for path in ["linear", "cosine"]:
        cap = {}
        torch.randn = lambda *s, **k: rn.clone()
        try:
            lf = ref_loss.SILoss(path_type=path, weighting="lognormal", enc_names=["dinov2"], loss_weights={"dinov2": 1.0})
            with inject(noise=noise):
                lf(lambda xx, tt, **k: (cap.setdefault("t", tt), (vel, [zt[0]]))[1], x, dict(y=y), zs=[zs[0]])
        finally:
            torch.randn = o_randn
        out[f"lognormal.{path}.t"] = cap["t"]
    save("loss_units", **out)


def g_samplers(ref_sit, ref_loss, ref_samplers):
    """G-i: Euler / Heun / Euler-Maruyama with and without (interval) CFG on a tiny SiT, fp32 model, fp64 state."""
    kw = tiny_kwargs(num_classes=1000)
    m = build_ref_model(ref_sit, "custom", seed=5, **kw).eval()
    n = 3
    z = detfill.normal((n, 4, 8, 8), 41)
    y = torch.tensor([3, 500, 999])
    out = {}
    cfgs = {"euler": dict(heun=False, cfg_scale=1.0), "heun": dict(heun=True, cfg_scale=1.0),
            "euler_cfg": dict(heun=False, cfg_scale=2.5), "heun_cfg": dict(heun=True, cfg_scale=1.5),
            "heun_cfg_interval": dict(heun=True, cfg_scale=3.0, guidance_low=0.3, guidance_high=0.75)}
    for name, c in cfgs.items():
        out[name] = ref_samplers.euler_sampler(m, z, y, num_steps=6, **c)
    eps = [detfill.normal((n, 4, 8, 8), 600 + i).double() for i in range(8)]
    for name, c in {"sde": dict(cfg_scale=1.0), "sde_cfg": dict(cfg_scale=2.0, guidance_high=0.9),
                    "sde_cosine": dict(cfg_scale=1.0, path_type="cosine")}.items():
        with inject(eps_list=eps):
            out[name] = ref_samplers.euler_maruyama_sampler(m, z, y, num_steps=6, **c)
    save("samplers", **out)


GRAD_PROBES_XL = [f"blocks.{i}.{n}.weight" for i in (0, 8, 27) for n in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2")] + [
    "blocks.8.adaLN_modulation.1.weight", "blocks.27.adaLN_modulation.1.bias", "projectors.0.4.weight", "projectors.0.0.weight",
    "final_layer.linear.weight", "final_layer.adaLN_modulation.1.weight", "x_embedder.proj.weight", "t_embedder.mlp.0.weight",
    "y_embedder.embedding_table.weight", "blocks.13.attn.qkv.bias"]


def grad_probe(g):
    """(norm, 64 elements spread over the whole tensor) of one gradient — the per-tensor probes of the XL/2 goldens."""
    f = g.detach().flatten()
    return f.double().norm(), f[:: max(1, f.numel() // 64)][:64].clone()


def ref_train_traj(ref_sit, ref_loss, model_name, kw, B, steps, zspec, enc_names, coeffs, autocast, proj_coeff=0.5,
                   align=True, seed=0, grad_probes=None, ac_dtype=torch.bfloat16, scaler=None):
    """Reference-equivalent optimisation steps (train.py:387-412): SILoss -> combine -> backward -> clip -> AdamW -> EMA."""
    m = build_ref_model(ref_sit, model_name, seed=seed, **kw)
    m.train()
    ema = {k: v.detach().clone() for k, v in m.named_parameters()}
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, betas=(0.9, 0.999), weight_decay=0.0, eps=1e-8)
    lf = ref_loss.SILoss(enc_names=list(enc_names), loss_weights={n: c for n, c in zip(enc_names, coeffs)})
    T = (kw.get("input_size", 32) // 2) ** 2
    rec = {k: [] for k in ("loss", "denoising_loss", "proj_loss", "grad_norm", "scale")}

    def model(xx, tt, **k):
        if autocast:
            with torch.autocast("cpu", dtype=ac_dtype):
                o, z = m(xx, tt, **k)
            return o.float(), [a.float() for a in z]
        return m(xx, tt, **k)

    for s in range(steps):
        t0 = time.time()
        x, noise, t, y, drop_u, zs = inputs(B, 4, kw.get("input_size", 32), 100 * seed + s, zspec, T, kw.get("num_classes", 1000))
        with inject(t=t, noise=noise, drop_u=drop_u):
            o = lf(model, x, dict(y=y), zs=zs)
        den = o["denoising_loss"].mean()
        proj = o["proj_loss"].mean()
        total = den * 1.0 + (proj * proj_coeff * 1.0 if align else 0.0)
        opt.zero_grad(set_to_none=True)
        if scaler is not None:   # accelerate fp16: scaler.scale(loss).backward(); unscale_ before clip_grad_norm_ (train.py:401-407)
            scaler.scale(total).backward()
            scaler.unscale_(opt)
        else:
            total.backward()
        if grad_probes is not None and s == 0:   # step-1 gradients, before clipping
            named = dict(m.named_parameters())
            for k in grad_probes[0]:
                grad_probes[1]["gnorm." + k], grad_probes[1]["gslice." + k] = grad_probe(named[k].grad)
        gn = torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        if scaler is not None:
            scaler.step(opt)          # skipped when unscale_ found inf / nan
            scaler.update()
        else:
            opt.step()
        with torch.no_grad():
            for k, p in m.named_parameters():
                ema[k].mul_(0.9999).add_(p.data, alpha=1 - 0.9999)
        rec["loss"].append(float(total)); rec["denoising_loss"].append(float(den))
        rec["proj_loss"].append(float(proj)); rec["grad_norm"].append(float(gn))
        rec["scale"].append(scaler.get_scale() if scaler is not None else 1.0)
        print(f"    step {s}: loss {float(total):.6f} den {float(den):.6f} proj {float(proj):.6f} gn {float(gn):.4f} ({time.time() - t0:.1f}s)")
    sd = m.state_dict()
    probe = {k: sd[k].flatten()[:64].clone() for k in ("blocks.0.attn.qkv.weight", "final_layer.linear.weight", "t_embedder.mlp.2.bias")}
    return rec, probe, {k: ema[k].flatten()[:64].clone() for k in probe}


def g_s2(ref_sit, ref_loss, ref_samplers):
    """G-e: C1 — SiT-S/2, B=64, 10 fp32 steps, alignment contribution zeroed (denoising loss only)."""
    kw = dict(input_size=32, num_classes=1000, z_dims=[768], z_types=["i"], encoder_depth=8, fused_attn=True, qk_norm=False)
    rec, probe, ema = ref_train_traj(ref_sit, ref_loss, "SiT-S/2", kw, 64, 10, [(768, "i")], ["dinov2"], [1.0], False, align=False)
    save("s2_c1", **{k: np.array(v) for k, v in rec.items()}, **{"w." + k: v for k, v in probe.items()},
         **{"ema." + k: v for k, v in ema.items()})


def g_b2(ref_sit, ref_loss, ref_samplers):
    """SiT-B/2-shaped mid-size check with alignment on (bf16 autocast and fp32), B=8, 3 steps."""
    kw = dict(input_size=32, num_classes=1000, z_dims=[768], z_types=["i"], encoder_depth=4, fused_attn=True, qk_norm=False)
    out = {}
    for tag, ac in (("fp32", False), ("bf16", True)):
        rec, probe, _ = ref_train_traj(ref_sit, ref_loss, "SiT-B/2", kw, 8, 3, [(768, "i")], ["dinov2"], [1.0], ac)
        out.update({f"{tag}.{k}": np.array(v) for k, v in rec.items()})
        out.update({f"{tag}.w.{k}": v for k, v in probe.items()})
    save("b2_align", **out)


def g_xl(ref_sit, ref_loss, ref_samplers):
    """G-f: C2 — SiT-XL/2 + 1024-d projector (DINOv2-L-shaped targets), B=8, 5 steps, fp32 and bf16-autocast."""
    kw = dict(input_size=32, num_classes=1000, z_dims=[1024], z_types=["i"], encoder_depth=8, fused_attn=True, qk_norm=False)
    out = {}
    for tag, ac in (("bf16", True), ("fp32", False)):
        print(f"  XL/2 {tag}")
        gp = {}
        rec, probe, _ = ref_train_traj(ref_sit, ref_loss, "SiT-XL/2", kw, 8, 5, [(1024, "i")], ["dinov2"], [1.0], ac,
                                       grad_probes=(GRAD_PROBES_XL, gp))
        out.update({f"{tag}.{k}": np.array(v) for k, v in rec.items()})
        out.update({f"{tag}.w.{k}": v for k, v in probe.items()})
        out.update({f"{tag}.{k}": v for k, v in gp.items()})
    save("xl2_c2", **out)


def g_xl_gnorms(ref_sit, ref_loss, ref_samplers):
    """C2, fp32, step 1 only: the norm of EVERY parameter's gradient (the 22 probes of xl2_c2 leave the biases, the other blocks'
    adaLN and t_embedder.mlp.2 unpinned) and the clip norm."""
    kw = dict(input_size=32, num_classes=1000, z_dims=[1024], z_types=["i"], encoder_depth=8, fused_attn=True, qk_norm=False)
    names = [k for k, p in build_ref_model(ref_sit, "SiT-XL/2", seed=0, **kw).named_parameters() if p.requires_grad]
    gp = {}
    rec, _, _ = ref_train_traj(ref_sit, ref_loss, "SiT-XL/2", kw, 8, 1, [(1024, "i")], ["dinov2"], [1.0], False, grad_probes=(names, gp))
    save("xl2_c2_gnorms", grad_norm=np.array(rec["grad_norm"]), loss=np.array(rec["loss"]),
         **{k: v for k, v in gp.items() if k.startswith("gnorm.")})


def g_xl_infer(ref_sit, ref_loss, ref_samplers):
    """C5 at its real size: one CFG-doubled evaluation of SiT-XL/2 in eval mode as samplers.py:66-78 issues it
    ([x; x], labels [y; 1000]), n = 2, fp32 and bf16-autocast; plus 3 Heun steps with CFG 1.5 on the fp32 model
    (5 evaluations at batch 4, fp64 state)."""
    kw = dict(input_size=32, num_classes=1000, z_dims=[1024], z_types=["i"], encoder_depth=8, fused_attn=True, qk_norm=False,
              use_cfg=True)
    m = build_ref_model(ref_sit, "SiT-XL/2", seed=0, **kw).eval()
    x, _, _, y, _, _ = inputs(2, 4, 32, 77, [], 256, 1000)
    xx, yy = torch.cat([x, x]), torch.cat([y, torch.tensor([1000, 1000])])
    out = {}
    with torch.no_grad():
        for tv in (0.9, 0.35):
            tt = torch.full((4,), tv)
            out[f"fp32.t{tv}"] = m(xx, tt, yy)[0]
            with torch.autocast("cpu", dtype=torch.bfloat16):
                out[f"bf16.t{tv}"] = m(xx, tt, yy)[0].float()
        out["heun3_cfg"] = ref_samplers.euler_sampler(m, x, y, num_steps=3, heun=True, cfg_scale=1.5)
    save("xl2_infer", **out)


def g_samplers_long(ref_sit, ref_loss, ref_samplers):
    """Long-horizon drift pin: SiT-S/2 (fp32 reference model), n = 2, 50-step Heun with CFG 1.5 over the whole interval
    (99 evaluations at batch 4). The state fed to every 9th evaluation and the final latents are recorded."""
    kw = dict(input_size=32, num_classes=1000, z_dims=[768], z_types=["i"], encoder_depth=8, fused_attn=True, qk_norm=False,
              use_cfg=True)
    m = build_ref_model(ref_sit, "SiT-S/2", seed=3, **kw).eval()
    z = detfill.normal((2, 4, 32, 32), 91)
    y = torch.tensor([17, 833])
    states = []

    def model(xx, tt, **kw):
        states.append(xx[:2].detach().clone())
        return m(xx, tt, **kw)

    out = {"final": ref_samplers.euler_sampler(model, z, y, num_steps=50, heun=True, cfg_scale=1.5)}
    out["n_evals"] = np.array(len(states))
    out["states"] = torch.stack(states[::9])
    save("samplers_long", **out)


def g_samplers_long_xl(ref_sit, ref_loss, ref_samplers):
    """The same pin at C5's real model size: SiT-XL/2 (fp32 reference model), n = 2, 25-step Heun with
    CFG 1.5 over the whole interval (49 evaluations at batch 4). The state fed to every 6th evaluation and the final latents."""
    kw = dict(input_size=32, num_classes=1000, z_dims=[1024], z_types=["i"], encoder_depth=8, fused_attn=True, qk_norm=False,
              use_cfg=True)
    m = build_ref_model(ref_sit, "SiT-XL/2", seed=0, **kw).eval()
    z = detfill.normal((2, 4, 32, 32), 191)
    y = torch.tensor([207, 980])
    states = []

    def model(xx, tt, **kw):
        states.append(xx[:2].detach().clone())
        return m(xx, tt, **kw)

    with torch.no_grad():
        out = {"final": ref_samplers.euler_sampler(model, z, y, num_steps=25, heun=True, cfg_scale=1.5)}
    out["n_evals"] = np.array(len(states))
    out["states"] = torch.stack(states[::6])
    save("samplers_long_xl", **out)


def g_fp16(ref_sit, ref_loss, ref_samplers):
    """--mixed-precision fp16 (the reference's default and README recipe): the reference under torch.autocast(float16)
    with torch.amp.GradScaler at accelerate's defaults (init 65536, x2 after 2000 clean steps, x0.5 on overflow).
    (a) tiny cases: loss and every unscaled gradient norm + element probes at a fixed scale of 1024; (b) SiT-S/2 + 768-d
    alignment, B = 8, 6 optimiser steps: per-step loss, unscaled grad-norm, scale after the step, weight / EMA probes;
    (c) the same with init_scale 2^40: the first steps overflow and are skipped (scale halves, weights and Adam state
    untouched, EMA still updated) until the scaled gradients fit fp16."""
    out = {}
    cases = {"hd64": dict(kw=tiny_kwargs(), zspec=[(128, "i")]),
             "xl3": dict(kw=tiny_kwargs(D=1152, heads=16, input_size=16, projector_dim=256), zspec=[(128, "i")])}
    for name, c in cases.items():
        kw = c["kw"]
        m = build_ref_model(ref_sit, "custom", seed=11, **kw)
        m.train()
        T = (kw["input_size"] // kw["patch_size"]) ** 2
        x, noise, t, y, drop_u, zs = inputs(4, 4, kw["input_size"], 11, c["zspec"], T, kw["num_classes"])
        lf = ref_loss.SILoss(enc_names=["dinov2"], loss_weights={"dinov2": 1.0})

        def model(xx, tt, **k):
            with torch.autocast("cpu", dtype=torch.float16):
                o, z = m(xx, tt, **k)
            return o.float(), [a.float() for a in z]
        with inject(t=t, noise=noise, drop_u=drop_u):
            o = lf(model, x, dict(y=y), zs=zs)
        total = o["denoising_loss"].mean() + 0.5 * o["proj_loss"]
        (total * 1024.0).backward()
        out[f"{name}.total"], out[f"{name}.denoising_loss"], out[f"{name}.proj_loss"] = total, o["denoising_loss"], o["proj_loss"]
        for k, p in m.named_parameters():
            if p.grad is not None:
                out[f"{name}.gnorm.{k}"] = (p.grad / 1024.0).double().norm()
        for k in ("final_layer.linear.weight", "blocks.0.attn.qkv.bias", "x_embedder.proj.weight", "projectors.0.4.bias",
                  "blocks.1.adaLN_modulation.1.bias"):
            out[f"{name}.grad.{k}"] = dict(m.named_parameters())[k].grad / 1024.0
    kw = dict(input_size=32, num_classes=1000, z_dims=[768], z_types=["i"], encoder_depth=8, fused_attn=True, qk_norm=False)
    for tag, init in (("s2", 65536.0), ("s2_overflow", 2.0 ** 40)):
        sc = torch.amp.GradScaler("cpu", init_scale=init)
        rec, probe, ema = ref_train_traj(ref_sit, ref_loss, "SiT-S/2", kw, 8, 6, [(768, "i")], ["dinov2"], [1.0], True,
                                         ac_dtype=torch.float16, scaler=sc)
        out.update({f"{tag}.{k}": np.array(v) for k, v in rec.items()})
        out.update({f"{tag}.w.{k}": v for k, v in probe.items()})
        out.update({f"{tag}.ema.{k}": v for k, v in ema.items()})
    save("fp16", **out)


def g_xl_c4(ref_sit, ref_loss, ref_samplers):
    """C4 — SiT-XL/2, CLIP-L image tokens (1024) at block 8 + pooled text vector (3584) at block 16, bf16, B=4, 2 steps."""
    kw = dict(input_size=32, num_classes=1000, z_dims=[1024, 3584], z_types=["i", "t"], encoder_depth=8,
              encoder_depth_text=16, fused_attn=True, qk_norm=False)
    rec, probe, _ = ref_train_traj(ref_sit, ref_loss, "SiT-XL/2", kw, 4, 2, [(1024, "i"), (3584, "t")],
                                   ["clip", "text_embeds_qwenvl_7b"], [1.0, 0.5], True)
    save("xl2_c4", **{k: np.array(v) for k, v in rec.items()}, **{"w." + k: v for k, v in probe.items()})


def g_clip(ref_sit, ref_loss, ref_samplers):
    """Frozen CLIP image encoder: the reference's own LayerNorm / ResidualAttentionBlock / Transformer
    / UpdatedVisionTransformer.forward (image/models/clip_vit.py:159-230) on a container that holds what openai/CLIP's
    VisionTransformer holds (that class itself lives in the un-vendored `clip` package; the module's top-level
    `import clip` is satisfied by an empty stand-in). fp32 and bf16-autocast outputs for two tiny configurations."""
    import types
    sys.modules.setdefault("clip", types.ModuleType("clip"))
    from models import clip_vit as ref_clip
    from oracle import clip_vit as oclip
    out = {}
    for tag, cfg, B in (("t2", oclip.make_config(width=128, layers=2, heads=2, patch=14, image=56), 3),
                        ("t3", oclip.make_config(width=256, layers=3, heads=4, patch=14, image=28), 2)):
        P = oclip.fill_params(cfg, base_seed=5)
        vis = torch.nn.Module()
        vis.conv1 = torch.nn.Conv2d(3, cfg["width"], cfg["patch"], cfg["patch"], bias=False)
        vis.class_embedding = torch.nn.Parameter(P["class_embedding"].clone())
        vis.positional_embedding = torch.nn.Parameter(P["positional_embedding"].clone())
        vis.ln_pre = ref_clip.LayerNorm(cfg["width"])
        vis.transformer = ref_clip.Transformer(cfg["width"], cfg["layers"], cfg["heads"])
        missing = vis.load_state_dict({k: v for k, v in P.items()}, strict=True)
        enc = ref_clip.UpdatedVisionTransformer(vis).eval()
        x = detfill.normal((B, 3, cfg["image"], cfg["image"]), 77)
        with torch.no_grad():
            out[tag + ".fp32"] = enc(x).numpy()
            with torch.autocast("cpu", dtype=torch.bfloat16):
                out[tag + ".bf16"] = enc(x).float().numpy()
    # preprocess_raw_image (train.py:53-57, 'clip' branch) cannot be imported (train.py needs diffusers/wandb): its
    # three torch calls are restated in oracle.clip_vit.preprocess; pin the bicubic geometry on a fixed ramp
    raw = (torch.arange(2 * 3 * 256 * 256) % 251).reshape(2, 3, 256, 256).to(torch.uint8)
    xr = raw.float() / 255.
    xr = torch.nn.functional.interpolate(xr, 224, mode='bicubic')
    out["pre.sample"] = xr[:, :, ::37, ::41].numpy()
    save("clip", **out)


def g_towers(ref_sit, ref_loss, ref_samplers):
    """Frozen ViT towers other than CLIP.  JEPA: the reference's own VisionTransformer
    (image/models/jepa.py, self-contained), head_dim 80 and 64.  MAE: the reference's own forward_features
    (mae_vit.py:33-48) and MoCo-v3: its own sin-cos pos-embed builder and constructor (mocov3_vit.py:52-101), both over the
    stand-in for timm's VisionTransformer container / Block above.  fp32 and bf16-autocast outputs; pos-embed tables of the
    real ViT-H/14 and ViT-B/16 grids; preprocess_raw_image (train.py:53-74 cannot be imported: its torch calls are
    restated in oracle.vit_towers.preprocess) pinned on a fixed ramp for both orders of normalise / resample."""
    from functools import partial
    from models import jepa as ref_jepa
    from models import mae_vit as ref_mae
    from models import mocov3_vit as ref_moco
    from oracle import vit_towers as ot
    ln = partial(torch.nn.LayerNorm, eps=1e-6)
    out = {}

    def run(tag, model, cfg, fwd, B):
        P = ot.fill_params(cfg, base_seed=9)
        sd = model.state_dict()
        keep = {k: v for k, v in P.items() if k in sd and not (k == "pos_embed" and cfg["pos"] != "learned")}
        assert torch.equal(sd["pos_embed"].float(), P["pos_embed"]) or cfg["pos"] == "learned", tag   # the reference's own table
        missing = model.load_state_dict(keep, strict=False)
        assert not [k for k in missing.missing_keys if k not in ("pos_embed", "head.weight", "head.bias")], missing
        model.eval()
        x = detfill.normal((B, 3, cfg["image"], cfg["image"]), 55)
        with torch.no_grad():
            out[tag + ".fp32"] = fwd(x).numpy()
            with torch.autocast("cpu", dtype=torch.bfloat16):
                out[tag + ".bf16"] = fwd(x).float().numpy()

    for tag, E, H, depth in (("jepa80", 640, 8, 2), ("jepa64", 256, 4, 3)):
        cfg = ot.make_config(E, depth, H, 14, 56, False, True, "jepa")
        m = ref_jepa.VisionTransformer(img_size=[56], patch_size=14, embed_dim=E, depth=depth, num_heads=H, mlp_ratio=4,
                                       qkv_bias=True, norm_layer=ln)
        run(tag, m, cfg, m.forward, 3)
    cfg = ot.make_config(256, 2, 4, 16, 64, True, False, "learned")
    m = ref_mae.VisionTransformer(num_classes=0, img_size=64, patch_size=16, embed_dim=256, depth=2, num_heads=4, mlp_ratio=4,
                                  qkv_bias=True, norm_layer=ln)
    run("mae", m, cfg, m.forward_features, 2)
    cfg = ot.make_config(256, 2, 4, 16, 64, True, True, "moco")
    m = ref_moco.VisionTransformerMoCo(img_size=64, patch_size=16, embed_dim=256, depth=2, num_heads=4, mlp_ratio=4,
                                       qkv_bias=True, norm_layer=ln)
    run("moco", m, cfg, lambda x: m.forward_features(x)[:, 1:], 2)
    # the real grids' tables
    pe = torch.from_numpy(ref_jepa.get_2d_sincos_pos_embed(1280, 16, cls_token=False)).float()
    out["jepa_pos_1280_rows"], out["jepa_pos_1280_sum"] = pe[::17].numpy(), pe.double().sum(0).numpy()
    mb = ref_moco.vit_base()
    out["moco_pos_768_rows"], out["moco_pos_768_sum"] = mb.pos_embed[0, ::17].detach().numpy(), mb.pos_embed[0].double().sum(0).detach().numpy()
    # preprocess geometry: /255 -> normalise -> bicubic (dinov2 / jepa order) on a fixed ramp, and the no-resample branch
    raw = (torch.arange(2 * 3 * 256 * 256) % 251).reshape(2, 3, 256, 256).to(torch.uint8)
    xr = raw.float() / 255.
    mean = torch.tensor(ot.IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(ot.IMAGENET_STD).view(1, 3, 1, 1)
    xn = (xr - mean) / std
    out["pre.jepa.sample"] = torch.nn.functional.interpolate(xn, 224, mode="bicubic")[:, :, ::37, ::41].numpy()
    out["pre.mae.sample"] = xn[:, :, ::37, ::41].numpy()
    save("towers", **out)


def g_dinov2(ref_sit, ref_loss, ref_samplers):
    """DINOv2 (image/utils.py:92-104 loads it from torch.hub, which cannot be reached here; the class is not in the
    reference tree).  The oracle restates the published model; these vectors come from transformers' Dinov2Model /
    Dinov2WithRegistersModel — an independent port of the hub model installed in this image — loaded with the oracle's
    parameters under the hub's names (qkv split into query / key / value).  fp32 and bf16-autocast patch tokens
    (= forward_features(...)['x_norm_patchtokens'], train.py:356), without and with 4 register tokens; plus timm's
    resample_abs_pos_embed geometry (utils.py:99-101: 37 x 37 -> 16 x 16, bicubic, antialias) on a fixed table, restated
    from timm.layers.pos_embed (timm is not installed)."""
    from transformers import Dinov2Config, Dinov2Model, Dinov2WithRegistersConfig, Dinov2WithRegistersModel
    from oracle import vit_towers as ot
    out = {}
    for tag, E, H, depth, image, reg, B in (("plain", 128, 2, 2, 56, 0, 3), ("reg4", 256, 4, 3, 28, 4, 2)):
        cfg = ot.make_config(E, depth, H, 14, image, True, True, "learned", ls=True, reg=reg)
        P = ot.fill_params(cfg, base_seed=21)
        kw = dict(hidden_size=E, num_hidden_layers=depth, num_attention_heads=H, mlp_ratio=4, image_size=image, patch_size=14,
                  layer_norm_eps=1e-6, qkv_bias=True, hidden_act="gelu", use_swiglu_ffn=False)
        m = (Dinov2WithRegistersModel(Dinov2WithRegistersConfig(num_register_tokens=reg, **kw)) if reg
             else Dinov2Model(Dinov2Config(**kw)))
        sd = {"embeddings.cls_token": P["cls_token"], "embeddings.mask_token": torch.zeros(1, E),
              "embeddings.position_embeddings": P["pos_embed"],
              "embeddings.patch_embeddings.projection.weight": P["patch_embed.proj.weight"],
              "embeddings.patch_embeddings.projection.bias": P["patch_embed.proj.bias"],
              "layernorm.weight": P["norm.weight"], "layernorm.bias": P["norm.bias"]}
        if reg:
            sd["embeddings.register_tokens"] = P["register_tokens"]
        for i in range(depth):
            b, h = f"blocks.{i}.", f"encoder.layer.{i}."
            for j, nm in enumerate(("query", "key", "value")):
                sd[h + f"attention.attention.{nm}.weight"] = P[b + "attn.qkv.weight"][j * E:(j + 1) * E]
                sd[h + f"attention.attention.{nm}.bias"] = P[b + "attn.qkv.bias"][j * E:(j + 1) * E]
            for src, dst in (("norm1", "norm1"), ("norm2", "norm2"), ("attn.proj", "attention.output.dense"), ("mlp.fc1", "mlp.fc1"),
                             ("mlp.fc2", "mlp.fc2")):
                sd[h + dst + ".weight"], sd[h + dst + ".bias"] = P[b + src + ".weight"], P[b + src + ".bias"]
            sd[h + "layer_scale1.lambda1"], sd[h + "layer_scale2.lambda1"] = P[b + "ls1.gamma"], P[b + "ls2.gamma"]
        r = m.load_state_dict(sd, strict=True)
        m.eval()
        x = detfill.normal((B, 3, image, image), 56)
        with torch.no_grad():
            out[tag + ".fp32"] = m(pixel_values=x).last_hidden_state[:, 1 + reg:].numpy()
            with torch.autocast("cpu", dtype=torch.bfloat16):
                out[tag + ".bf16"] = m(pixel_values=x).last_hidden_state[:, 1 + reg:].float().numpy()
    pe = detfill.normal((1, 1 + 37 * 37, 64), 57)
    g = pe[:, 1:].reshape(1, 37, 37, 64).permute(0, 3, 1, 2)
    g = torch.nn.functional.interpolate(g, size=(16, 16), mode="bicubic", antialias=True).permute(0, 2, 3, 1).reshape(1, 256, 64)
    out["pos_resample"] = torch.cat([pe[:, :1], g], 1).numpy()
    save("dinov2", **out)


def make_tiny_dataset(root, n=6, text_dim=16):
    """Deterministic tiny dataset in the reference's on-disk format (image/dataset.py:18-85; written by
    preprocessing/dataset_tools.py): images/XXXXX/imgNNNNNNNN.png, vae-sd/XXXXX/img-mean-std-NNNNNNNN.npy,
    vae-sd/dataset.json, text_embeds_t/XXXXX/imgNNNNNNNN.npy. Used by the generator (reference loader) and by the tests
    (build loaders), so both read identical bytes."""
    import json
    import PIL.Image
    labels = []
    for i in range(n):
        sub = f"{i // 4:05d}"
        for d in ("images", "vae-sd", "text_embeds_t"):
            os.makedirs(os.path.join(root, d, sub), exist_ok=True)
        img = (detfill.uniform((64, 64, 3), 1000 + i, 0, 256).numpy()).astype(np.uint8)
        PIL.Image.fromarray(img).save(os.path.join(root, "images", sub, f"img{i:08d}.png"))
        np.save(os.path.join(root, "vae-sd", sub, f"img-mean-std-{i:08d}.npy"), detfill.normal((8, 8, 8), 2000 + i).numpy())
        np.save(os.path.join(root, "text_embeds_t", sub, f"img{i:08d}.npy"), detfill.normal((text_dim,), 3000 + i).numpy())
        labels.append([f"{sub}/img-mean-std-{i:08d}.npy", int((i * 7) % 10)])
    with open(os.path.join(root, "vae-sd", "dataset.json"), "w") as f:
        json.dump({"labels": labels[::-1]}, f)   # file order differs from sorted order on purpose


def g_dataset(ref_sit, ref_loss, ref_samplers):
    """image/dataset.py:18-85 CustomDataset (the module imports here: torch, numpy, PIL only) on the tiny dataset above,
    with and without text embeddings: every field of every item."""
    import dataset as ref_dataset
    root = tempfile.mkdtemp(prefix="tinyds_")
    make_tiny_dataset(root)
    out = {}
    for tag, td in (("plain", None), ("text", "text_embeds_t")):
        ds = ref_dataset.CustomDataset(root, text_embeds_dir=td)
        out[tag + ".len"] = np.array(len(ds))
        for i in range(len(ds)):
            im, mo, la, tx = ds[i]
            out[f"{tag}.{i}.image"], out[f"{tag}.{i}.moments"] = im.numpy(), mo.numpy()
            out[f"{tag}.{i}.label"], out[f"{tag}.{i}.text"] = la.numpy(), tx.numpy()
    save("dataset", **out)


def g_sched(ref_sit, ref_loss, ref_samplers):
    """G-j: optimiser toy (clip + AdamW + EMA on a 3-tensor toy) — schedules are pure python in train.py's main()
    (not importable: needs diffusers/wandb), so they are pinned by hand-derived values in tests instead."""
    ps = [torch.nn.Parameter(detfill.normal(s, 70 + i)) for i, s in enumerate([(5, 7), (11,), (3, 4, 2)])]
    ema = [p.detach().clone() for p in ps]
    opt = torch.optim.AdamW(ps, lr=1e-2, betas=(0.9, 0.999), weight_decay=0.01, eps=1e-8)
    out = {}
    for s in range(3):
        for i, p in enumerate(ps):
            p.grad = detfill.normal(tuple(p.shape), 80 + 10 * s + i) * (3.0 if s == 0 else 0.1)
        gn = torch.nn.utils.clip_grad_norm_(ps, 1.0)
        opt.step()
        with torch.no_grad():
            for e, p in zip(ema, ps):
                e.mul_(0.99).add_(p.data, alpha=0.01)
        out[f"gn{s}"] = gn
        for i, p in enumerate(ps):
            out[f"p{s}_{i}"] = p.detach().clone()
            out[f"e{s}_{i}"] = ema[i].clone()
    save("optim_toy", **out)


def g_init(ref_sit, ref_loss, ref_samplers):
    """Reference weight init (sit.py:217-254) for a fixed torch seed: probes of several tensors."""
    out = {}
    for tag, name, kw in (("s2", "SiT-S/2", dict(decoder_hidden_size=384, z_dims=[768], fused_attn=True, qk_norm=False)),
                          ("tiny", "custom", tiny_kwargs(z_dims=[128, 256], z_types=["i", "t"]))):
        torch.manual_seed(1234)
        m = ref_sit.SiT_models[name](**kw) if name in ref_sit.SiT_models else ref_sit.SiT(**kw)
        sd = m.state_dict()
        out[f"{tag}.keys"] = np.array(sorted(sd.keys()))
        out[f"{tag}.shapes"] = np.array([str(tuple(sd[k].shape)) for k in sorted(sd.keys())])
        for k in ("x_embedder.proj.weight", "t_embedder.mlp.0.weight", "t_embedder.mlp.2.weight",
                  "y_embedder.embedding_table.weight", "blocks.0.attn.qkv.weight", "blocks.2.mlp.fc2.weight",
                  "projectors.0.4.weight", "blocks.1.adaLN_modulation.1.weight", "final_layer.linear.weight"):
            out[f"{tag}.{k}"] = sd[k].flatten()[:32].clone()
            out[f"{tag}.sum.{k}"] = sd[k].double().sum()
    save("init", **out)


ALL = {"init": g_init, "static": g_static, "tiny": g_tiny, "loss_units": g_loss_units, "samplers": g_samplers, "optim_toy": g_sched,
       "s2_c1": g_s2, "b2_align": g_b2, "xl2_c2": g_xl, "xl2_c2_gnorms": g_xl_gnorms, "xl2_c4": g_xl_c4, "xl2_infer": g_xl_infer, "samplers_long": g_samplers_long, "samplers_long_xl": g_samplers_long_xl, "fp16": g_fp16, "clip": g_clip, "dataset": g_dataset, "towers": g_towers, "dinov2": g_dinov2}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--skip-xl", action="store_true")
    a = ap.parse_args()
    torch.set_num_threads(8)
    mods = import_reference()
    for name, fn in ALL.items():
        if a.only and name not in a.only:
            continue
        if a.skip_xl and (name.startswith("xl2") or name.endswith("_xl")):
            continue
        print(f"[gen_golden] {name}")
        t0 = time.time()
        fn(*mods)
        print(f"  done in {time.time() - t0:.1f}s")
