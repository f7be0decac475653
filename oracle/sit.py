"""Functional CPU restatement of the SiT model (reference: image/models/sit.py; timm ≥0.9
PatchEmbed/Attention/Mlp semantics as pinned in SURVEY.md §8c). TEST INFRASTRUCTURE — see oracle/__init__.py.

Everything is written over a flat {state_dict key: tensor} mapping with torch.nn.functional ops, so the same
code runs in fp32 or under torch.autocast('cpu', bfloat16) exactly like the reference modules do, and
torch.autograd provides the backward."""
import math

import numpy as np
import torch
import torch.nn.functional as F

# preset table: image/models/sit.py:373-407 (S presets get decoder == hidden: SURVEY.md §9-1)
PRESETS = {
    "XL": dict(depth=28, hidden_size=1152, num_heads=16),
    "L": dict(depth=24, hidden_size=1024, num_heads=16),
    "B": dict(depth=12, hidden_size=768, num_heads=12),
    "S": dict(depth=12, hidden_size=384, num_heads=6),
}


def make_config(name="SiT-XL/2", input_size=32, in_channels=4, num_classes=1000, class_dropout_prob=0.1,
                encoder_depth=8, encoder_depth_text=None, z_dims=(768,), z_types=("i",), projector_dim=2048,
                mlp_ratio=4.0, qk_norm=False, fused_attn=True, **over):
    size, patch = name.replace("SiT-", "").split("/")
    cfg = dict(PRESETS[size])
    cfg.update(patch_size=int(patch), input_size=input_size, in_channels=in_channels, num_classes=num_classes,
               class_dropout_prob=class_dropout_prob, encoder_depth=encoder_depth,
               encoder_depth_text=encoder_depth_text, z_dims=list(z_dims), z_types=list(z_types),
               projector_dim=projector_dim, mlp_ratio=mlp_ratio, qk_norm=qk_norm, fused_attn=fused_attn)
    cfg.update(over)
    return cfg


# ---- sin-cos position table (sit.py:319-366): float64 numpy math, column index first ----
def sincos_1d(dim, pos):
    omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def pos_embed_table(dim, grid):
    gh = np.arange(grid, dtype=np.float32)
    gw = np.arange(grid, dtype=np.float32)
    mesh = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid, grid)  # mesh[0] = column index
    emb = np.concatenate([sincos_1d(dim // 2, mesh[0]), sincos_1d(dim // 2, mesh[1])], axis=1)
    return torch.from_numpy(emb).float().unsqueeze(0)


def param_shapes(cfg):
    """Ordered {key: shape} in the reference's construction order (sit.py:198-215)."""
    D, p, C = cfg["hidden_size"], cfg["patch_size"], cfg["in_channels"]
    T = (cfg["input_size"] // p) ** 2
    hd = D // cfg["num_heads"]
    Hm = int(D * cfg["mlp_ratio"])
    sh = {}
    sh["x_embedder.proj.weight"] = (D, C, p, p)
    sh["x_embedder.proj.bias"] = (D,)
    sh["t_embedder.mlp.0.weight"] = (D, 256)
    sh["t_embedder.mlp.0.bias"] = (D,)
    sh["t_embedder.mlp.2.weight"] = (D, D)
    sh["t_embedder.mlp.2.bias"] = (D,)
    rows = cfg["num_classes"] + (1 if cfg["class_dropout_prob"] > 0 else 0)
    sh["y_embedder.embedding_table.weight"] = (rows, D)
    sh["pos_embed"] = (1, T, D)
    for i in range(cfg["depth"]):
        b = f"blocks.{i}."
        sh[b + "attn.qkv.weight"] = (3 * D, D)
        sh[b + "attn.qkv.bias"] = (3 * D,)
        if cfg["qk_norm"]:
            for n in ("q_norm", "k_norm"):
                sh[b + f"attn.{n}.weight"] = (hd,)
                sh[b + f"attn.{n}.bias"] = (hd,)
        sh[b + "attn.proj.weight"] = (D, D)
        sh[b + "attn.proj.bias"] = (D,)
        sh[b + "mlp.fc1.weight"] = (Hm, D)
        sh[b + "mlp.fc1.bias"] = (Hm,)
        sh[b + "mlp.fc2.weight"] = (D, Hm)
        sh[b + "mlp.fc2.bias"] = (D,)
        sh[b + "adaLN_modulation.1.weight"] = (6 * D, D)
        sh[b + "adaLN_modulation.1.bias"] = (6 * D,)
    P = cfg["projector_dim"]
    for j, z in enumerate(cfg["z_dims"]):
        b = f"projectors.{j}."
        sh[b + "0.weight"], sh[b + "0.bias"] = (P, D), (P,)
        sh[b + "2.weight"], sh[b + "2.bias"] = (P, P), (P,)
        sh[b + "4.weight"], sh[b + "4.bias"] = (z, P), (z,)
    sh["final_layer.linear.weight"] = (p * p * C, D)
    sh["final_layer.linear.bias"] = (p * p * C,)
    sh["final_layer.adaLN_modulation.1.weight"] = (2 * D, D)
    sh["final_layer.adaLN_modulation.1.bias"] = (2 * D,)
    return sh


def init_params(cfg, dtype=torch.float32):
    """Zero-filled state (plus the sin-cos table); fill with oracle.detfill.fill_state_dict."""
    sd = {k: torch.zeros(s, dtype=dtype) for k, s in param_shapes(cfg).items()}
    D, p = cfg["hidden_size"], cfg["patch_size"]
    sd["pos_embed"] = pos_embed_table(D, cfg["input_size"] // p).to(dtype)
    return sd


def timestep_sinusoid(t, dim=256, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(t.device)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)  # cos first (sit.py:61)


def modulate(x, shift, scale):
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def attention(P, pre, x, H, qk_norm, fused):
    B, N, C = x.shape
    hd = C // H
    qkv = F.linear(x, P[pre + "qkv.weight"], P[pre + "qkv.bias"]).reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    if qk_norm:
        q = F.layer_norm(q, (hd,), P[pre + "q_norm.weight"], P[pre + "q_norm.bias"], 1e-5)
        k = F.layer_norm(k, (hd,), P[pre + "k_norm.weight"], P[pre + "k_norm.bias"], 1e-5)
    if fused:
        o = F.scaled_dot_product_attention(q, k, v)
    else:
        a = (q * hd ** -0.5) @ k.transpose(-2, -1)
        o = a.softmax(dim=-1) @ v
    o = o.transpose(1, 2).reshape(B, N, C)
    return F.linear(o, P[pre + "proj.weight"], P[pre + "proj.bias"])


def block(P, i, x, c, cfg):
    b = f"blocks.{i}."
    D = x.shape[-1]
    mod = F.linear(F.silu(c), P[b + "adaLN_modulation.1.weight"], P[b + "adaLN_modulation.1.bias"])
    s1, c1, g1, s2, c2, g2 = mod.chunk(6, dim=-1)
    h = modulate(F.layer_norm(x, (D,), None, None, 1e-6), s1, c1)
    x = x + g1.unsqueeze(1) * attention(P, b + "attn.", h, cfg["num_heads"], cfg["qk_norm"], cfg["fused_attn"])
    h = modulate(F.layer_norm(x, (D,), None, None, 1e-6), s2, c2)
    h = F.linear(h, P[b + "mlp.fc1.weight"], P[b + "mlp.fc1.bias"])
    h = F.gelu(h, approximate="tanh")
    h = F.linear(h, P[b + "mlp.fc2.weight"], P[b + "mlp.fc2.bias"])
    return x + g2.unsqueeze(1) * h


def projector(P, j, x):
    b = f"projectors.{j}."
    x = F.silu(F.linear(x, P[b + "0.weight"], P[b + "0.bias"]))
    x = F.silu(F.linear(x, P[b + "2.weight"], P[b + "2.bias"]))
    return F.linear(x, P[b + "4.weight"], P[b + "4.bias"])


def unpatchify(x, p, C):
    N, T, _ = x.shape
    h = w = int(T ** 0.5)
    x = x.reshape(N, h, w, p, p, C).permute(0, 5, 1, 3, 2, 4)  # n c h p w q
    return x.reshape(N, C, h * p, w * p)


def sit_forward(P, cfg, x, t, y, inference=True, training=False, drop_mask=None):
    """Returns (velocity [N,C,H,W], zs list or None). `drop_mask` (bool [N]) replaces the device RNG draw of
    LabelEmbedder.token_drop (sit.py:84-93) so both sides of a parity test drop the same labels; in training
    mode with drop_mask=None labels are dropped with torch.rand like the reference."""
    D, p, C = cfg["hidden_size"], cfg["patch_size"], cfg["in_channels"]
    x = F.conv2d(x, P["x_embedder.proj.weight"], P["x_embedder.proj.bias"], stride=p).flatten(2).transpose(1, 2)
    x = x + P["pos_embed"]
    N, T, _ = x.shape
    te = timestep_sinusoid(t).to(t.dtype)
    te = F.linear(F.silu(F.linear(te, P["t_embedder.mlp.0.weight"], P["t_embedder.mlp.0.bias"])),
                  P["t_embedder.mlp.2.weight"], P["t_embedder.mlp.2.bias"])
    if cfg["class_dropout_prob"] > 0 and (training or drop_mask is not None):
        if drop_mask is None:
            drop_mask = torch.rand(y.shape[0], device=y.device) < cfg["class_dropout_prob"]
        y = torch.where(drop_mask, cfg["num_classes"], y)
    c = te + F.embedding(y, P["y_embedder.embedding_table.weight"])
    ed, edt = cfg["encoder_depth"], cfg["encoder_depth_text"]
    split = edt is not None and edt != ed
    zs, z_img, z_txt = None, None, None
    for i in range(cfg["depth"]):
        x = block(P, i, x, c, cfg)
        if inference:
            continue
        if i + 1 == ed:
            if not split:
                zs = [projector(P, j, x.reshape(-1, D)).reshape(N, T, -1) if zt == "i" else projector(P, j, x.mean(dim=1))
                      for j, zt in enumerate(cfg["z_types"])]
            else:
                for j, zt in enumerate(cfg["z_types"]):
                    if zt == "i":
                        z_img = projector(P, j, x.reshape(-1, D)).reshape(N, T, -1)
        if split and i + 1 == edt:
            for j, zt in enumerate(cfg["z_types"]):
                if zt == "t":
                    z_txt = projector(P, j, x.mean(dim=1))
    if not inference and split:
        zs = [z_img, z_txt]
    sh, sc = F.linear(F.silu(c), P["final_layer.adaLN_modulation.1.weight"],
                      P["final_layer.adaLN_modulation.1.bias"]).chunk(2, dim=-1)
    x = modulate(F.layer_norm(x, (D,), None, None, 1e-6), sh, sc)
    x = F.linear(x, P["final_layer.linear.weight"], P["final_layer.linear.bias"])
    return unpatchify(x, p, C), zs


class OracleModel:
    """Callable with the reference's model signature, for SILoss / the samplers. `autocast_bf16=True` mirrors
    accelerate's mixed_precision='bf16' wrapping (autocast forward + outputs converted to fp32); `autocast_dtype`
    (torch.float16: the reference's default --mixed-precision, train.py:458) picks the autocast type."""

    def __init__(self, P, cfg, autocast_bf16=False, training=False, autocast_dtype=None):
        if autocast_dtype is not None:
            autocast_bf16 = True
        self.autocast_dtype = autocast_dtype or torch.bfloat16
        self.P, self.cfg, self.autocast_bf16, self.training = P, cfg, autocast_bf16, training
        self.drop_mask = None
        self.in_channels = cfg["in_channels"]

    def __call__(self, x, t, y, inference=True):
        if self.autocast_bf16:
            with torch.autocast("cpu", dtype=self.autocast_dtype):
                out, zs = sit_forward(self.P, self.cfg, x, t, y, inference, self.training, self.drop_mask)
            out = out.float()
            zs = None if zs is None else [z.float() for z in zs]
            return out, zs
        return sit_forward(self.P, self.cfg, x, t, y, inference, self.training, self.drop_mask)
