"""ORACLE (test infrastructure, never on the product path): CPU restatement of the frozen ViT target encoders other than
CLIP that image/utils.py:load_encoders can build, as image/train.py:351-357 runs them under autocast:

  jepa    image/models/jepa.py:376-466 VisionTransformer (own PatchEmbed :221-236, Block :200-218, Attention :173-197,
          MLP :154-170; fixed 2-D sin-cos pos-embed :69-130, no class token, final LayerNorm) — vit_huge, patch 14,
          224 x 224 (utils.py:149-160);
  mae     image/models/mae_vit.py:20-48: forward_features = patch_embed -> [cls | x] + pos_embed -> blocks -> x[:, 1:]
          (NO final norm) over timm's VisionTransformer — vit_large_patch16 at 256 x 256 (utils.py:133-147);
  mocov3  image/models/mocov3_vit.py:52-101: timm's VisionTransformer with the fixed sin-cos pos-embed of :78-95 (class
          position zero), forward_features incl. the final norm, class token dropped by train.py:355 — vit_base /
          vit_large at 256 x 256 (utils.py:73-82).

  dinov2  image/utils.py:92-104: torch.hub facebookresearch/dinov2 `dinov2_vit{s,b,l}14[_reg]` (class DinoVisionTransformer, NOT
          in the reference tree; restated from the published model: dinov2/models/vision_transformer.py prepare_tokens_with_masks
          / forward_features, layers/block.py, layers/layer_scale.py): patch 14, [cls | patches] + pos_embed (learned 37 x 37
          grid, resampled to 16 x 16 by utils.py:99-101), then 4 register tokens inserted behind the class token in the _reg
          models, blocks x = x + gamma1 * attn(norm1 x); x = x + gamma2 * mlp(norm2 x) (LayerScale), final LayerNorm;
          train.py:356 takes ['x_norm_patchtokens'] = normed tokens without class / register tokens.  Pinned against
          transformers' Dinov2Model / Dinov2WithRegistersModel (an independent port of the hub model; tools/gen_golden.py:g_dinov2)
          since the hub class cannot be fetched here.

timm is un-vendored and unpinned (image/requirements.txt:5); its VisionTransformer semantics restated here (>= 0.9): Block =
x + attn(norm1 x); x + mlp(norm2 x), LayerNorm(eps 1e-6 as the reference's constructors pass), Attention with qkv bias and
softmax(q k^T / sqrt(hd)) v, Mlp fc1 -> nn.GELU() (exact) -> fc2, no LayerScale / drop-path at inference.
Pinned by tests/golden/towers.npz: the JEPA cases are outputs of the reference's own class; the MAE case runs the
reference's own forward_features and the MoCo case its own pos-embed builder over a stand-in for timm's container and
Block (tools/gen_golden.py:g_towers) — the same caveat as the three timm layers under SiT.
"""
import numpy as np
import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)   # timm.data.IMAGENET_DEFAULT_MEAN / STD (image/train.py:28)
IMAGENET_STD = (0.229, 0.224, 0.225)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

TOWERS = {   # the configurations image/utils.py:55-164 can name
    "jepa-vit-h": dict(embed=1280, depth=32, heads=16, patch=14, image=224, cls=False, final_norm=True, pos="jepa"),
    "mocov3-vit-b": dict(embed=768, depth=12, heads=12, patch=16, image=256, cls=True, final_norm=True, pos="moco"),
    "mocov3-vit-l": dict(embed=1024, depth=24, heads=16, patch=16, image=256, cls=True, final_norm=True, pos="moco"),
    "mae-vit-l": dict(embed=1024, depth=24, heads=16, patch=16, image=256, cls=True, final_norm=False, pos="learned"),
    "dinov2-vit-b": dict(embed=768, depth=12, heads=12, patch=14, image=224, cls=True, final_norm=True, pos="learned", ls=True),
    "dinov2-vit-l": dict(embed=1024, depth=24, heads=16, patch=14, image=224, cls=True, final_norm=True, pos="learned", ls=True),
    "dinov2reg-vit-l": dict(embed=1024, depth=24, heads=16, patch=14, image=224, cls=True, final_norm=True, pos="learned",
                            ls=True, reg=4),
}


def make_config(embed, depth, heads, patch, image, cls, final_norm, pos, ls=False, reg=0):
    return dict(embed=embed, depth=depth, heads=heads, patch=patch, image=image, cls=cls, final_norm=final_norm, pos=pos,
                ls=ls, reg=reg)


def param_shapes(cfg):
    E, P = cfg["embed"], cfg["patch"]
    T = (cfg["image"] // P) ** 2 + (1 if cfg["cls"] else 0)
    s = {"patch_embed.proj.weight": (E, 3, P, P), "patch_embed.proj.bias": (E,), "pos_embed": (1, T, E)}
    if cfg["cls"]:
        s["cls_token"] = (1, 1, E)
    if cfg.get("reg"):
        s["register_tokens"] = (1, cfg["reg"], E)
    for i in range(cfg["depth"]):
        b = f"blocks.{i}."
        s.update({b + "norm1.weight": (E,), b + "norm1.bias": (E,), b + "attn.qkv.weight": (3 * E, E),
                  b + "attn.qkv.bias": (3 * E,), b + "attn.proj.weight": (E, E), b + "attn.proj.bias": (E,),
                  b + "norm2.weight": (E,), b + "norm2.bias": (E,), b + "mlp.fc1.weight": (4 * E, E),
                  b + "mlp.fc1.bias": (4 * E,), b + "mlp.fc2.weight": (E, 4 * E), b + "mlp.fc2.bias": (E,)})
        if cfg.get("ls"):
            s.update({b + "ls1.gamma": (E,), b + "ls2.gamma": (E,)})
    s.update({"norm.weight": (E,), "norm.bias": (E,)})
    return s


def _sincos_1d(embed_dim, pos):   # jepa.py:111-130
    omega = np.arange(embed_dim // 2, dtype=float)
    omega /= embed_dim / 2.
    omega = 1. / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def jepa_pos_embed(embed_dim, grid_size):
    """jepa.py:69-95 get_2d_sincos_pos_embed(cls_token=False): meshgrid(w, h) with w first, halves = (grid[0], grid[1])."""
    gh = np.arange(grid_size, dtype=float)
    gw = np.arange(grid_size, dtype=float)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape([2, 1, grid_size, grid_size])
    emb = np.concatenate([_sincos_1d(embed_dim // 2, grid[0]), _sincos_1d(embed_dim // 2, grid[1])], axis=1)
    return torch.from_numpy(emb).float().unsqueeze(0)


def moco_pos_embed(embed_dim, h, w, temperature=10000.):
    """mocov3_vit.py:78-95: [sin(w), cos(w), sin(h), cos(h)] of fp32 grids (meshgrid(w, h), 'ij'), class position zero."""
    grid_w, grid_h = torch.meshgrid(torch.arange(w, dtype=torch.float32), torch.arange(h, dtype=torch.float32), indexing="ij")
    pos_dim = embed_dim // 4
    omega = 1. / (temperature ** (torch.arange(pos_dim, dtype=torch.float32) / pos_dim))
    out_w = torch.einsum("m,d->md", [grid_w.flatten(), omega])
    out_h = torch.einsum("m,d->md", [grid_h.flatten(), omega])
    pos = torch.cat([torch.sin(out_w), torch.cos(out_w), torch.sin(out_h), torch.cos(out_h)], dim=1)[None]
    return torch.cat([torch.zeros(1, 1, embed_dim), pos], dim=1)


def fill_params(cfg, base_seed=0):
    """Deterministic non-trivial weights (oracle.detfill); the fixed pos-embeds are the reference's own tables."""
    from . import detfill
    P = {}
    G = cfg["image"] // cfg["patch"]
    for name, shp in param_shapes(cfg).items():
        seed = detfill.name_seed("tower." + name, base_seed)
        if name == "pos_embed":
            if cfg["pos"] == "jepa":
                v = jepa_pos_embed(cfg["embed"], G)
            elif cfg["pos"] == "moco":
                v = moco_pos_embed(cfg["embed"], G, G)
            else:
                v = detfill.uniform(shp, seed, -0.1, 0.1)
        elif name.endswith("bias"):
            v = detfill.uniform(shp, seed, -0.05, 0.05)
        elif "norm" in name:
            v = 1.0 + detfill.uniform(shp, seed, -0.1, 0.1)
        elif name in ("cls_token", "register_tokens"):
            v = detfill.uniform(shp, seed, -0.1, 0.1)
        elif name.endswith("gamma"):    # LayerScale: trained DINOv2 gammas spread over ~[0.01, 3]
            v = 0.6 + detfill.uniform(shp, seed, -0.5, 0.5)
        else:
            fan_in = int(np.prod(shp[1:]))
            v = detfill.uniform(shp, seed, -(3.0 / fan_in) ** 0.5, (3.0 / fan_in) ** 0.5)
        P[name] = v
    return P


def preprocess(raw_u8, enc_type):
    """image/train.py:53-74 preprocess_raw_image, every branch."""
    x = raw_u8.float()
    res = x.shape[-1]

    def norm(t, mean, std):   # torchvision.transforms.Normalize
        m = torch.tensor(mean, dtype=t.dtype, device=t.device).view(1, 3, 1, 1)
        s = torch.tensor(std, dtype=t.dtype, device=t.device).view(1, 3, 1, 1)
        return (t - m) / s
    if "clip" in enc_type:
        x = x / 255.
        x = F.interpolate(x, 224 * (res // 256), mode="bicubic")
        x = norm(x, CLIP_MEAN, CLIP_STD)
    elif "mocov3" in enc_type or "mae" in enc_type or "dinov1" in enc_type:
        x = norm(x / 255., IMAGENET_MEAN, IMAGENET_STD)
    elif "dinov2" in enc_type or "jepa" in enc_type:
        x = norm(x / 255., IMAGENET_MEAN, IMAGENET_STD)
        x = F.interpolate(x, 224 * (res // 256), mode="bicubic")
    return x


def forward(P, cfg, x, autocast_bf16=False):
    """x: preprocessed images f32 [B,3,S,S] -> patch tokens [B, (S/patch)^2, embed] without the class token."""
    E, H = cfg["embed"], cfg["heads"]
    hd = E // H
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast_bf16):
        x = F.conv2d(x, P["patch_embed.proj.weight"], P["patch_embed.proj.bias"], stride=cfg["patch"])
        x = x.flatten(2).transpose(1, 2)
        if cfg["cls"]:
            x = torch.cat((P["cls_token"].expand(x.shape[0], -1, -1), x), dim=1)
        x = x + P["pos_embed"]
        if cfg.get("reg"):   # registers go in behind the class token AFTER the position embedding was added
            x = torch.cat((x[:, :1], P["register_tokens"].expand(x.shape[0], -1, -1), x[:, 1:]), dim=1)
        g = (lambda k: P[k]) if cfg.get("ls") else (lambda k: 1.0)
        for i in range(cfg["depth"]):
            b = f"blocks.{i}."
            h = F.layer_norm(x, (E,), P[b + "norm1.weight"], P[b + "norm1.bias"], 1e-6)
            B, N, _ = h.shape
            qkv = F.linear(h, P[b + "attn.qkv.weight"], P[b + "attn.qkv.bias"]).reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4)
            q, k, v = qkv[0], qkv[1], qkv[2]
            a = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
            a = (a @ v).transpose(1, 2).reshape(B, N, E)
            x = x + F.linear(a, P[b + "attn.proj.weight"], P[b + "attn.proj.bias"]) * g(b + "ls1.gamma")
            h = F.layer_norm(x, (E,), P[b + "norm2.weight"], P[b + "norm2.bias"], 1e-6)
            u = F.gelu(F.linear(h, P[b + "mlp.fc1.weight"], P[b + "mlp.fc1.bias"]))
            x = x + F.linear(u, P[b + "mlp.fc2.weight"], P[b + "mlp.fc2.bias"]) * g(b + "ls2.gamma")
        if cfg["final_norm"]:
            x = F.layer_norm(x, (E,), P["norm.weight"], P["norm.bias"], 1e-6)
        npre = (1 if cfg["cls"] else 0) + cfg.get("reg", 0)
        return x[:, npre:] if npre else x


def resample_abs_pos_embed(posemb, new_size, num_prefix_tokens=1):
    """timm.layers.pos_embed.resample_abs_pos_embed as image/utils.py:99-101,140-146 call it (bicubic, antialias=True)."""
    pre, grid = posemb[:, :num_prefix_tokens], posemb[:, num_prefix_tokens:]
    hw = int(round(grid.shape[1] ** 0.5))
    if (hw, hw) == tuple(new_size):
        return posemb
    g = grid.reshape(1, hw, hw, -1).permute(0, 3, 1, 2).float()
    g = F.interpolate(g, size=tuple(new_size), mode="bicubic", antialias=True)
    g = g.permute(0, 2, 3, 1).reshape(1, -1, posemb.shape[-1]).to(posemb.dtype)
    return torch.cat([pre, g], dim=1)
