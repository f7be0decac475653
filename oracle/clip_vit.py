"""ORACLE (test infrastructure, never on the product path): CPU restatement of the frozen CLIP image encoder forward
that image/train.py:351-357 runs every step to produce the alignment targets.

Follows image/models/clip_vit.py: LayerNorm (:159-165, fp32 compute, input dtype out), QuickGELU (:168-170),
ResidualAttentionBlock (:173-195: x + MHA(ln_1 x); x + c_proj(QuickGELU(c_fc(ln_2 x)))), Transformer (:197-205) and
UpdatedVisionTransformer.forward (:213-230: conv1 -> [class | patches] + positional embedding -> ln_pre ->
transformer in LND layout -> drop the class token; no ln_post, no projection). The VisionTransformer container itself
lives in the un-vendored `clip` package (openai/CLIP, git HEAD per image/requirements.txt:10): its parameter names are
kept (conv1.weight, class_embedding, positional_embedding, ln_pre.*, transformer.resblocks.{i}.*), so that
`clip.load(...)[0].visual.state_dict()` loads. Pinned by tests/golden/clip.npz, produced by the reference's own classes
(tools/gen_golden.py: g_clip).
"""
import numpy as np
import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # image/train.py:33 (CLIP_DEFAULT_MEAN / STD)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def make_config(width=1024, layers=24, heads=16, patch=14, image=224):
    return dict(width=width, layers=layers, heads=heads, patch=patch, image=image)


def param_shapes(cfg):
    W, L, P = cfg["width"], cfg["layers"], cfg["patch"]
    T = (cfg["image"] // P) ** 2 + 1
    s = {"conv1.weight": (W, 3, P, P), "class_embedding": (W,), "positional_embedding": (T, W),
         "ln_pre.weight": (W,), "ln_pre.bias": (W,)}
    for i in range(L):
        b = f"transformer.resblocks.{i}."
        s.update({b + "attn.in_proj_weight": (3 * W, W), b + "attn.in_proj_bias": (3 * W,),
                  b + "attn.out_proj.weight": (W, W), b + "attn.out_proj.bias": (W,),
                  b + "ln_1.weight": (W,), b + "ln_1.bias": (W,),
                  b + "mlp.c_fc.weight": (4 * W, W), b + "mlp.c_fc.bias": (4 * W,),
                  b + "mlp.c_proj.weight": (W, 4 * W), b + "mlp.c_proj.bias": (W,),
                  b + "ln_2.weight": (W,), b + "ln_2.bias": (W,)})
    return s


def fill_params(cfg, base_seed=0):
    """Deterministic non-trivial weights (oracle.detfill): the same on the reference, the oracle and the HIP side."""
    from . import detfill
    P = {}
    for name, shp in param_shapes(cfg).items():
        seed = detfill.name_seed("clip." + name, base_seed)
        if name.endswith("bias"):
            v = detfill.uniform(shp, seed, -0.05, 0.05)
        elif name.endswith("ln_pre.weight") or ".ln_" in name and name.endswith("weight"):
            v = 1.0 + detfill.uniform(shp, seed, -0.1, 0.1)
        elif name in ("class_embedding", "positional_embedding"):
            v = detfill.uniform(shp, seed, -0.1, 0.1)
        else:
            fan_in = int(np.prod(shp[1:]))
            a = (3.0 / fan_in) ** 0.5
            v = detfill.uniform(shp, seed, -a, a)
        P[name] = v
    return P


def preprocess(raw_u8):
    """image/train.py:53-57 ('clip' branch): /255, bicubic to 224 * (res // 256), CLIP mean/std."""
    x = raw_u8.float() / 255.0
    res = x.shape[-1]
    x = F.interpolate(x, 224 * (res // 256), mode="bicubic")
    mean = torch.tensor(CLIP_MEAN, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
    return (x - mean) / std


def _ln(x, w, b):   # clip_vit.py:159-165
    return F.layer_norm(x.type(torch.float32), (x.shape[-1],), w, b, 1e-5).type(x.dtype)


def forward(P, cfg, x, autocast_bf16=False):
    """x: normalised images f32 [B,3,S,S] -> patch tokens [B, (S/patch)^2, width] (f32, or bf16 under autocast)."""
    W, H = cfg["width"], cfg["heads"]
    with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast_bf16):
        x = F.conv2d(x, P["conv1.weight"], stride=cfg["patch"])                 # :214
        x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)                # :215-216
        cls = P["class_embedding"].to(x.dtype) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=x.dtype)
        x = torch.cat([cls, x], dim=1)                                            # :217
        x = x + P["positional_embedding"].to(x.dtype)                             # :218
        x = _ln(x, P["ln_pre.weight"], P["ln_pre.bias"])                          # :219
        x = x.permute(1, 0, 2)                                                    # NLD -> LND
        for i in range(cfg["layers"]):
            b = f"transformer.resblocks.{i}."
            h = _ln(x, P[b + "ln_1.weight"], P[b + "ln_1.bias"])
            a = F.multi_head_attention_forward(                                    # nn.MultiheadAttention.forward(h, h, h)
                h, h, h, W, H, P[b + "attn.in_proj_weight"], P[b + "attn.in_proj_bias"], None, None, False, 0.0,
                P[b + "attn.out_proj.weight"], P[b + "attn.out_proj.bias"], training=False, need_weights=False)[0]
            x = x + a                                                             # :192
            h = _ln(x, P[b + "ln_2.weight"], P[b + "ln_2.bias"])
            u = F.linear(h, P[b + "mlp.c_fc.weight"], P[b + "mlp.c_fc.bias"])
            u = u * torch.sigmoid(1.702 * u)                                      # QuickGELU
            x = x + F.linear(u, P[b + "mlp.c_proj.weight"], P[b + "mlp.c_proj.bias"])   # :193
        return x.permute(1, 0, 2)[:, 1:]                                          # :223
