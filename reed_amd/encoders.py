"""Frozen target encoders on the GPU, forward only.

`ClipVisionEncoder` is the MI355X counterpart of the reference's CLIP branch of `load_encoders`
(image/utils.py:123-131: `clip.load("ViT-L/14").visual` wrapped in `UpdatedVisionTransformer`,
image/models/clip_vit.py:208-230) together with `preprocess_raw_image` (image/train.py:53-57): raw uint8 images ->
patch tokens [B, 256, width] without the class token, no ln_post, no projection — the `zs` that `SILoss` aligns
the SiT projector to. Numerics = the reference under `accelerator.autocast()` with bf16 (train.py:351-357): bf16
linears / conv / attention with fp32 accumulation, LayerNorm in fp32 on bf16 rows, bf16 residual stream.

Parameter names are openai/CLIP's (`conv1.weight`, `class_embedding`, `positional_embedding`, `ln_pre.*`,
`transformer.resblocks.{i}.attn.in_proj_weight` ...), so `clip.load(...)[0].visual.state_dict()` loads with
`strict=False` (its `ln_post.*` / `proj` are unused here, as in the reference). No weights ship: there is no network
in this build; `train.py --enc-type clip-vit-L --encoder-ckpt file.pt` takes them from the user.

Every contraction is a `reed_gemm` launch (patch-embedding conv as im2col + GEMM; in_proj; out_proj with the bf16
residual epilogue; c_fc with the QuickGELU epilogue; c_proj with the residual epilogue), attention is
`reed_attention_fwd` (head_dim 64, T = 257: one 256-key tile + a ragged 1-key tile, online softmax), the row passes are
csrc/encoder.hip. No CPU path.
"""
import torch
from torch import nn

from . import ops
from .ops import EPI_BF16, EPI_GATE_RES, EPI_GELU_ERF, EPI_LS_RES, EPI_QGELU, EPI_RES_BF16, NT

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
IMAGENET_MEAN = (0.485, 0.456, 0.406)   # timm.data.IMAGENET_DEFAULT_MEAN / STD (image/train.py:28)
IMAGENET_STD = (0.229, 0.224, 0.225)


def preprocess_raw_image(raw, enc_type):
    """image/train.py:53-74, every branch, as ONE HIP pass (csrc/encoder.hip: preprocess_image_kernel): raw uint8
    [B,3,R,R] on the GPU -> fp32 [B,3,S,S].  'clip': /255 -> bicubic to 224 (R // 256) -> CLIP mean/std; 'mocov3' / 'mae' /
    'dinov1': /255 -> ImageNet mean/std, no resampling; 'dinov2' / 'jepa': /255 -> ImageNet mean/std -> bicubic."""
    ops.require_cuda(raw, "raw images")
    if raw.dtype != torch.uint8:
        raise TypeError("preprocess_raw_image: raw images must be uint8 (the dataset's format, image/dataset.py:66-71)")
    raw = raw.contiguous()
    B, R = raw.shape[0], raw.shape[-1]
    if "clip" in enc_type:
        S, mean, std, order = 224 * (R // 256), CLIP_MEAN, CLIP_STD, 0
    elif "mocov3" in enc_type or "mae" in enc_type or "dinov1" in enc_type:
        S, mean, std, order = R, IMAGENET_MEAN, IMAGENET_STD, 1
    elif "dinov2" in enc_type or "jepa" in enc_type:
        S, mean, std, order = 224 * (R // 256), IMAGENET_MEAN, IMAGENET_STD, 1
    else:
        raise ValueError(f"preprocess_raw_image: unknown encoder type {enc_type!r}")
    out = torch.empty(B, 3, S, S, dtype=torch.float32, device=raw.device)
    ops.preprocess_image(raw, out, B, R, S, mean, std, order)
    return out

CLIP_CONFIGS = {   # openai/CLIP vision towers the reference can name (utils.py:127: f"ViT-{model_config}/14")
    "L": dict(width=1024, layers=24, heads=16, patch=14, image=224),
    "B": dict(width=768, layers=12, heads=12, patch=14, image=224),   # (openai ships B/16 and B/32; kept for shape tests)
}


class ClipVisionEncoder(nn.Module):
    def __init__(self, width=1024, layers=24, heads=16, patch=14, image=224):
        super().__init__()
        if width % 128 or (width // heads) != 64:
            raise ValueError("ClipVisionEncoder: width must be a multiple of 128 with head_dim 64 (CLIP ViT-B/L towers)")
        self.width, self.layers, self.heads, self.patch, self.image = width, layers, heads, patch, image
        self.embed_dim = width
        G = image // patch
        self.tokens = G * G + 1
        self.kp = (3 * patch * patch + 63) // 64 * 64   # GEMM K multiple: 588 -> 640
        sc = width ** -0.5
        self.conv1 = nn.Conv2d(3, width, patch, patch, bias=False)
        self.class_embedding = nn.Parameter(sc * torch.randn(width))
        self.positional_embedding = nn.Parameter(sc * torch.randn(self.tokens, width))
        self.ln_pre = nn.LayerNorm(width)
        blocks = []
        for _ in range(layers):
            b = nn.Module()
            b.attn = nn.Module()
            b.attn.in_proj_weight = nn.Parameter(torch.randn(3 * width, width) * sc)
            b.attn.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
            b.attn.out_proj = nn.Linear(width, width)
            b.ln_1 = nn.LayerNorm(width)
            b.mlp = nn.Module()
            b.mlp.c_fc = nn.Linear(width, 4 * width)
            b.mlp.c_proj = nn.Linear(4 * width, width)
            b.ln_2 = nn.LayerNorm(width)
            blocks.append(b)
        self.transformer = nn.Module()
        self.transformer.resblocks = nn.ModuleList(blocks)
        self.requires_grad_(False)
        self._bf = None   # bf16 operand copies of the GEMM weights, built on first use / after load_state_dict

    # ---- weights -------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd, strict=False):
        sd = {k: v for k, v in sd.items() if not (k.startswith("ln_post") or k == "proj")}
        r = super().load_state_dict(sd, strict=strict)
        self._bf = None
        return r

    def _apply(self, fn, recurse=True):
        self._bf = None
        return super()._apply(fn, recurse)

    def _operands(self):
        if self._bf is None:
            dev = self.conv1.weight.device
            w = torch.zeros(self.width, self.kp, dtype=torch.bfloat16, device=dev)
            w[:, :3 * self.patch * self.patch] = self.conv1.weight.detach().reshape(self.width, -1).to(torch.bfloat16)
            bf = lambda t: t.detach().to(torch.bfloat16).contiguous()  # noqa: E731
            f32 = lambda t: t.detach().float().contiguous()            # noqa: E731
            blocks = []
            for b in self.transformer.resblocks:
                blocks.append(dict(
                    in_w=bf(b.attn.in_proj_weight), in_b=bf(b.attn.in_proj_bias),
                    out_w=bf(b.attn.out_proj.weight), out_b=bf(b.attn.out_proj.bias),
                    fc_w=bf(b.mlp.c_fc.weight), fc_b=bf(b.mlp.c_fc.bias),
                    pj_w=bf(b.mlp.c_proj.weight), pj_b=bf(b.mlp.c_proj.bias),
                    ln1=(f32(b.ln_1.weight), f32(b.ln_1.bias)), ln2=(f32(b.ln_2.weight), f32(b.ln_2.bias))))
            self._bf = dict(conv=w, cls=f32(self.class_embedding), pos=f32(self.positional_embedding),
                            ln_pre=(f32(self.ln_pre.weight), f32(self.ln_pre.bias)), blocks=blocks)
        return self._bf

    # ---- forward -------------------------------------------------------------------------------------------------
    @staticmethod
    def preprocess(raw):
        """image/train.py:53-57, 'clip' branch: uint8 [B,3,R,R] -> /255 -> bicubic to 224·(R//256) -> CLIP mean/std."""
        return preprocess_raw_image(raw, "clip")

    @torch.no_grad()
    def forward(self, x):
        """x: normalised images f32 [B,3,image,image] on the GPU -> bf16 [B, tokens-1, width]."""
        ops.require_cuda(x, "images")
        W, H, T, P = self.width, self.heads, self.tokens, self.patch
        if x.shape[1] != 3 or x.shape[-1] != self.image or x.shape[-2] != self.image:
            raise ValueError(f"ClipVisionEncoder: input {tuple(x.shape)} is not (B,3,{self.image},{self.image})")
        B = x.shape[0]
        dev = x.device
        w = self._operands()
        bf = lambda *s: torch.empty(s, dtype=torch.bfloat16, device=dev)  # noqa: E731
        x = x.contiguous().float()
        Mp, M = B * (T - 1), B * T
        cols = bf(Mp, self.kp)
        ops.clip_im2col(x, cols, B, self.image, P, self.kp)
        patches = bf(Mp, W)
        ops.gemm(NT, EPI_BF16, cols, w["conv"], Mp, W, self.kp, patches, self.kp, self.kp, W)
        tok = bf(M, W)
        ops.clip_tokens(patches, w["cls"], w["pos"], tok, B, T, W)
        xa, xb = bf(M, W), bf(M, W)
        ops.ln_affine_bf16(tok, w["ln_pre"][0], w["ln_pre"][1], xa, M, W)
        h, qkv, o, u = bf(M, W), bf(M, 3 * W), bf(M, W), bf(M, 4 * W)
        for blk in w["blocks"]:
            ops.ln_affine_bf16(xa, blk["ln1"][0], blk["ln1"][1], h, M, W)
            ops.gemm(NT, EPI_BF16, h, blk["in_w"], M, 3 * W, W, qkv, W, W, 3 * W, bias=blk["in_b"])
            ops.attention_fwd(qkv, o, None, B, T, H, 64)
            ops.gemm(NT, EPI_RES_BF16, o, blk["out_w"], M, W, W, xb, W, W, W, R=xa, ldr=W, bias=blk["out_b"])
            ops.ln_affine_bf16(xb, blk["ln2"][0], blk["ln2"][1], h, M, W)
            ops.gemm(NT, EPI_QGELU, h, blk["fc_w"], M, 4 * W, W, None, W, W, 4 * W, C2=u, ldc2=4 * W, bias=blk["fc_b"])
            ops.gemm(NT, EPI_RES_BF16, u, blk["pj_w"], M, W, 4 * W, xa, 4 * W, 4 * W, W, R=xb, ldr=W, bias=blk["pj_b"])
        return xa.view(B, T, W)[:, 1:]

    def forward_features(self, x):
        """The call train.py makes (utils.py:130: encoder.forward_features = encoder.forward)."""
        return self.forward(x)

    def encode_raw(self, raw_u8):
        return self.forward(self.preprocess(raw_u8))


# ------------------------------------------------------------------------------------------------------------------
# The other towers image/utils.py:load_encoders can name: plain pre-LN ViTs.
VIT_TOWERS = {
    # enc-type "jepa-vit-h": models.jepa.vit_huge(img_size=[224, 224], patch_size=14) (utils.py:149-160)
    "jepa-vit-h": dict(embed=1280, depth=32, heads=16, patch=14, image=224, cls=False, final_norm=True),
    # "mocov3-vit-{b,l}": mocov3_vit.vit_base / vit_large, img_size 256, patch 16 (utils.py:73-82; vit_small has head_dim 32)
    "mocov3-vit-b": dict(embed=768, depth=12, heads=12, patch=16, image=256, cls=True, final_norm=True),
    "mocov3-vit-l": dict(embed=1024, depth=24, heads=16, patch=16, image=256, cls=True, final_norm=True),
    # "mae-vit-l": mae_vit.vit_large_patch16(img_size=256): forward_features WITHOUT the final norm (mae_vit.py:33-48)
    "mae-vit-l": dict(embed=1024, depth=24, heads=16, patch=16, image=256, cls=True, final_norm=False),
    # "dinov2-vit-{s,b,l}" / "dinov2reg-vit-{s,b,l}": torch.hub facebookresearch/dinov2 dinov2_vit{s,b,l}14[_reg] (utils.py:92-104):
    # patch 14 at 224 x 224 (preprocess_raw_image resizes), class token, 0 / 4 register tokens, LayerScale, final norm; the
    # learned 37 x 37 pos_embed resampled to 16 x 16 by the loader as utils.py:99-101 does
    "dinov2-vit-s": dict(embed=384, depth=12, heads=6, patch=14, image=224, cls=True, final_norm=True, layerscale=True),
    "dinov2-vit-b": dict(embed=768, depth=12, heads=12, patch=14, image=224, cls=True, final_norm=True, layerscale=True),
    "dinov2-vit-l": dict(embed=1024, depth=24, heads=16, patch=14, image=224, cls=True, final_norm=True, layerscale=True),
    "dinov2reg-vit-s": dict(embed=384, depth=12, heads=6, patch=14, image=224, cls=True, final_norm=True, layerscale=True,
                            registers=4),
    "dinov2reg-vit-b": dict(embed=768, depth=12, heads=12, patch=14, image=224, cls=True, final_norm=True, layerscale=True,
                            registers=4),
    "dinov2reg-vit-l": dict(embed=1024, depth=24, heads=16, patch=14, image=224, cls=True, final_norm=True, layerscale=True,
                            registers=4),
}


class VitEncoder(nn.Module):
    """Forward-only counterpart of the reference's I-JEPA / MAE / MoCo-v3 target encoders (image/models/jepa.py:376-466,
    mae_vit.py:20-48, mocov3_vit.py:52-101 over timm's VisionTransformer): patch-embedding conv (im2col + GEMM with bias)
    -> [class token |] patches + pos_embed in fp32 -> depth x {LayerNorm(eps 1e-6, affine) -> qkv GEMM -> attention (head_dim
    64 or 80) -> proj GEMM + fp32 residual; LayerNorm -> fc1 GEMM + exact GELU -> fc2 GEMM + fp32 residual} -> [final
    LayerNorm] -> patch tokens without the class token, fp32.  Numerics = the reference under accelerator.autocast() with
    bf16 (train.py:351-357): bf16 GEMM operands and linear outputs, fp32 residual stream / LayerNorm, as SiT's.
    Parameter names are the reference's / timm's (patch_embed.proj.*, cls_token, pos_embed, blocks.{i}.norm1.*,
    attn.qkv.*, attn.proj.*, norm2.*, mlp.fc1.*, mlp.fc2.*, norm.*), so the checkpoints utils.py loads load here.  No
    weights ship; no CPU path.
    DINOv2 (utils.py:92-104 loads it from torch.hub; the class is facebookresearch/dinov2's DinoVisionTransformer, not in the
    reference tree) is the same tower plus LayerScale (blocks.{i}.ls1.gamma / ls2.gamma: x + gamma * branch(x), the fp32
    gamma times the bf16 branch output in fp32) and, in the *_reg models, 4 register tokens inserted behind the class token
    after the position embedding was added (register_tokens); forward_features(...)['x_norm_patchtokens'] (train.py:356) =
    the final-normed tokens without class and register tokens, which is what forward() returns."""

    def __init__(self, embed=1024, depth=24, heads=16, patch=16, image=256, cls=True, final_norm=True, layerscale=False,
                 registers=0):
        super().__init__()
        hd = embed // heads
        if embed % 128 or hd not in (64, 80) or heads * hd != embed:
            raise ValueError(f"VitEncoder: embed={embed} / heads={heads}: need embed % 128 == 0 and head_dim 64 or 80")
        self.embed_dim = self.embed = embed
        self.depth, self.heads, self.hd, self.patch, self.image = depth, heads, hd, patch, image
        self.has_cls, self.final_norm = bool(cls), bool(final_norm)
        self.layerscale, self.registers = bool(layerscale), int(registers)
        if self.registers and not cls:
            raise ValueError("VitEncoder: register tokens follow a class token")
        G = image // patch
        self.npatch = G * G
        self.nprefix = (1 if cls else 0) + self.registers
        self.tokens = self.npatch + self.nprefix
        if hd == 80 and self.tokens > 256:
            raise ValueError("VitEncoder: head_dim 80 is built for <= 256 tokens (I-JEPA ViT-H/14 at 224)")
        self.kp = (3 * patch * patch + 63) // 64 * 64
        self.patch_embed = nn.Module()
        self.patch_embed.proj = nn.Conv2d(3, embed, patch, patch)
        if cls:
            self.cls_token = nn.Parameter(torch.zeros(1, 1, embed))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.tokens - self.registers, embed), requires_grad=False)
        if self.registers:
            self.register_tokens = nn.Parameter(torch.zeros(1, self.registers, embed))
        blocks = []
        for _ in range(depth):
            b = nn.Module()
            b.norm1, b.norm2 = nn.LayerNorm(embed, eps=1e-6), nn.LayerNorm(embed, eps=1e-6)
            b.attn = nn.Module()
            b.attn.qkv, b.attn.proj = nn.Linear(embed, 3 * embed), nn.Linear(embed, embed)
            b.mlp = nn.Module()
            b.mlp.fc1, b.mlp.fc2 = nn.Linear(embed, 4 * embed), nn.Linear(4 * embed, embed)
            if self.layerscale:
                b.ls1, b.ls2 = nn.Module(), nn.Module()
                b.ls1.gamma, b.ls2.gamma = nn.Parameter(torch.ones(embed)), nn.Parameter(torch.ones(embed))
            blocks.append(b)
        self.blocks = nn.ModuleList(blocks)
        self.norm = nn.LayerNorm(embed, eps=1e-6)
        self.requires_grad_(False)
        self._bf = None

    def load_state_dict(self, sd, strict=False):
        sd = {k: v for k, v in sd.items() if not k.startswith(("head.", "fc_norm.")) and k != "mask_token"}
        r = super().load_state_dict(sd, strict=strict)
        self._bf = None
        return r

    def _apply(self, fn, recurse=True):
        self._bf = None
        return super()._apply(fn, recurse)

    def _operands(self):
        if self._bf is None:
            dev = self.pos_embed.device
            E = self.embed
            w = torch.zeros(E, self.kp, dtype=torch.bfloat16, device=dev)
            w[:, :3 * self.patch * self.patch] = self.patch_embed.proj.weight.detach().reshape(E, -1).to(torch.bfloat16)
            bf = lambda t: t.detach().to(torch.bfloat16).contiguous()  # noqa: E731
            f32 = lambda t: t.detach().float().contiguous()            # noqa: E731
            blocks = [dict(qkv_w=bf(b.attn.qkv.weight), qkv_b=bf(b.attn.qkv.bias), proj_w=bf(b.attn.proj.weight),
                           proj_b=bf(b.attn.proj.bias), fc1_w=bf(b.mlp.fc1.weight), fc1_b=bf(b.mlp.fc1.bias),
                           fc2_w=bf(b.mlp.fc2.weight), fc2_b=bf(b.mlp.fc2.bias),
                           n1=(f32(b.norm1.weight), f32(b.norm1.bias)), n2=(f32(b.norm2.weight), f32(b.norm2.bias)),
                           ls1=f32(b.ls1.gamma) if self.layerscale else None,
                           ls2=f32(b.ls2.gamma) if self.layerscale else None)
                      for b in self.blocks]
            pos, cls = f32(self.pos_embed[0]), f32(self.cls_token.reshape(1, E)) if self.has_cls else None
            if self.registers:   # prefix rows = [class token, registers]; the registers take no position embedding
                cls = torch.cat([cls, f32(self.register_tokens[0])], 0).contiguous()
                pos = torch.cat([pos[:1], torch.zeros(self.registers, E, device=dev), pos[1:]], 0).contiguous()
            self._bf = dict(conv=w, conv_b=bf(self.patch_embed.proj.bias), pos=pos, cls=cls,
                            norm=(f32(self.norm.weight), f32(self.norm.bias)), blocks=blocks,
                            ones=torch.ones(E, dtype=torch.bfloat16, device=dev))
        return self._bf

    @torch.no_grad()
    def forward(self, x):
        """x: preprocessed images f32 [B,3,image,image] on the GPU -> f32 [B, patches, embed] (class token dropped)."""
        ops.require_cuda(x, "images")
        E, H, T, P = self.embed, self.heads, self.tokens, self.patch
        if x.shape[1] != 3 or x.shape[-1] != self.image or x.shape[-2] != self.image:
            raise ValueError(f"VitEncoder: input {tuple(x.shape)} is not (B,3,{self.image},{self.image})")
        B = x.shape[0]
        dev = x.device
        w = self._operands()
        bf = lambda *s: torch.empty(s, dtype=torch.bfloat16, device=dev)   # noqa: E731
        f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)    # noqa: E731
        x = x.contiguous().float()
        Mp, M = B * self.npatch, B * T
        cols = bf(Mp, self.kp)
        ops.clip_im2col(x, cols, B, self.image, P, self.kp)
        patches = bf(Mp, E)
        ops.gemm(NT, EPI_BF16, cols, w["conv"], Mp, E, self.kp, patches, self.kp, self.kp, E, bias=w["conv_b"])
        xa, xb = f32(M, E), f32(M, E)
        ops.vit_tokens(patches, w["cls"], w["pos"], xa, B, T, E, nprefix=self.nprefix)
        h, qkv, o, u = bf(M, E), bf(M, 3 * E), bf(M, E), bf(M, 4 * E)
        one = w["ones"]
        # x + bf16(branch) in fp32: the gate-residual epilogue with a gate of ones; with LayerScale x + gamma * bf16(branch)
        res = (lambda g: dict(gate=g)) if self.layerscale else (lambda g: dict(gate=one, ldgate=0, rows_per_gate=T))
        epi_res = EPI_LS_RES if self.layerscale else EPI_GATE_RES
        for blk in w["blocks"]:
            ops.ln_affine_f32(xa, blk["n1"][0], blk["n1"][1], h, False, M, E)
            ops.gemm(NT, EPI_BF16, h, blk["qkv_w"], M, 3 * E, E, qkv, E, E, 3 * E, bias=blk["qkv_b"])
            ops.attention_fwd(qkv, o, None, B, T, H, self.hd)
            ops.gemm(NT, epi_res, o, blk["proj_w"], M, E, E, xb, E, E, E, R=xa, ldr=E, bias=blk["proj_b"], **res(blk["ls1"]))
            ops.ln_affine_f32(xb, blk["n2"][0], blk["n2"][1], h, False, M, E)
            ops.gemm(NT, EPI_GELU_ERF, h, blk["fc1_w"], M, 4 * E, E, None, E, E, 4 * E, C2=u, ldc2=4 * E, bias=blk["fc1_b"])
            ops.gemm(NT, epi_res, u, blk["fc2_w"], M, E, 4 * E, xa, 4 * E, 4 * E, E, R=xb, ldr=E, bias=blk["fc2_b"],
                     **res(blk["ls2"]))
        if self.final_norm:
            ops.ln_affine_f32(xa, w["norm"][0], w["norm"][1], xb, True, M, E)
            xa = xb
        out = xa.view(B, T, E)
        return out[:, self.nprefix:] if self.nprefix else out

    def forward_features(self, x):
        return self.forward(x)

    enc_type = "jepa"   # set by load_vit_encoder: selects the preprocess_raw_image branch

    def encode_raw(self, raw_u8):
        return self.forward(preprocess_raw_image(raw_u8, self.enc_type))


def resample_abs_pos_embed(posemb, new_size, num_prefix_tokens=1):
    """timm.layers.pos_embed.resample_abs_pos_embed as image/utils.py:99-101,140-146 call it: the grid part of a learned
    [1, prefix + g*g, D] table to new_size by bicubic interpolation with antialiasing, prefix rows kept.  Checkpoint
    conversion at load time (CPU torch), not part of the per-step path."""
    pre, grid = posemb[:, :num_prefix_tokens], posemb[:, num_prefix_tokens:]
    hw = int(round(grid.shape[1] ** 0.5))
    if (hw, hw) == tuple(new_size) or hw * hw != grid.shape[1]:
        return posemb
    g = grid.reshape(1, hw, hw, -1).permute(0, 3, 1, 2).float()
    g = torch.nn.functional.interpolate(g, size=tuple(new_size), mode="bicubic", antialias=True)
    g = g.permute(0, 2, 3, 1).reshape(1, -1, posemb.shape[-1]).to(posemb.dtype)
    return torch.cat([pre, g], dim=1)


_MOCO_MISNAMED = (("blocks.13.norm13", "norm13", "norm1"), ("blocks.13.mlp.fc13", "fc13", "fc1"),
                  ("blocks.14.norm14", "norm14", "norm2"), ("blocks.14.mlp.fc14", "fc14", "fc2"))


def mocov3_key(k):
    """Key of a published MoCo-v3 checkpoint -> tower parameter name, or None for a tensor the tower does not hold
    (image/utils.py:27-52 fix_mocov3_state_dict): only `module.base_encoder.*` survives (the momentum encoder and the
    predictor are dropped; a plain state dict without that prefix is taken as it is), the projection head (`head.*`,
    `fc.*`) is dropped, and the four mis-named entries of the ViT-L file are repaired — there `blocks.13.norm1 / mlp.fc1`
    are stored as `norm13 / fc13` and `blocks.14.norm2 / mlp.fc2` as `norm14 / fc14`."""
    pre = "module.base_encoder."
    if k.startswith(pre):
        k = k[len(pre):]
    elif k.startswith("module."):
        return None
    for where, bad, good in _MOCO_MISNAMED:
        if where in k:
            k = k.replace(bad, good)
    if "head" in k or k.split(".")[0] == "fc":
        return None
    return k


def load_vit_encoder(enc_type, ckpt_path, device):
    """`jepa-vit-h`, `mocov3-vit-{b,l}`, `mae-vit-l` of image/utils.py:73-82,133-160 from the checkpoint files the reference
    names (ckpts/ijepa_vith.pth: state_dict['encoder'] with a 'module.' prefix; ckpts/mocov3_vit{b,l}.pth: ['state_dict']
    with 'module.base_encoder.' (fix_mocov3_state_dict, utils.py:27-52); ckpts/mae_vitl.pth: ['model']) or a plain state
    dict; `dinov2[reg]-vit-{s,b,l}` of utils.py:92-104 from the torch.hub checkpoint file (dinov2_vit{s,b,l}14[_reg4]_pretrain.pth,
    a plain state dict).  A learned pos_embed of another grid (DINOv2: 37 x 37, MAE: 14 x 14) is resampled to the tower's
    as utils.py:99-101,140-146 do with timm's resample_abs_pos_embed (bicubic, antialias; a load-time torch call on the CPU)."""
    cfg = VIT_TOWERS[enc_type]
    enc = VitEncoder(**cfg)
    sd = torch.load(ckpt_path, map_location="cpu")
    for key in ("encoder", "state_dict", "model"):
        if isinstance(sd, dict) and key in sd and isinstance(sd[key], dict):
            sd = sd[key]
            break
    out = {}
    moco = enc_type.startswith("mocov3")
    for k, v in sd.items():
        if moco:
            k = mocov3_key(k)
            if k is None:
                continue
        elif k.startswith("module."):
            k = k[len("module."):]
        out[k] = v.float() if torch.is_floating_point(v) else v
    if "pos_embed" in out and out["pos_embed"].shape != enc.pos_embed.shape:
        G = enc.image // enc.patch
        out["pos_embed"] = resample_abs_pos_embed(out["pos_embed"], (G, G), 1 if enc.has_cls else 0)
        if out["pos_embed"].shape != enc.pos_embed.shape:
            raise RuntimeError(f"{ckpt_path}: pos_embed {tuple(out['pos_embed'].shape)} does not resample to "
                               f"{tuple(enc.pos_embed.shape)}")
    missing, unexpected = enc.load_state_dict(out, strict=False)
    missing = [k for k in missing if not (k.startswith("norm.") and not enc.final_norm)]
    if missing:
        raise RuntimeError(f"checkpoint {ckpt_path} lacks {missing[:4]}...")
    enc.enc_type = enc_type.split("-")[0]
    return enc.to(device).eval()


def load_clip_encoder(model_config, ckpt_path, device):
    """`clip-vit-{L}` of image/utils.py:123-131 from a user-supplied state dict (`clip.load(...)[0].visual.state_dict()`
    or the full CLIP state dict, whose `visual.` prefix is stripped)."""
    cfg = CLIP_CONFIGS[model_config]
    enc = ClipVisionEncoder(**cfg)
    sd = torch.load(ckpt_path, map_location="cpu")
    sd = sd.get("state_dict", sd) if isinstance(sd, dict) else sd
    if any(k.startswith("visual.") for k in sd):
        sd = {k[len("visual."):]: v for k, v in sd.items() if k.startswith("visual.")}
    sd = {k: (v.float() if torch.is_floating_point(v) else v) for k, v in sd.items()}
    missing, unexpected = enc.load_state_dict(sd, strict=False)
    if missing:
        raise RuntimeError(f"CLIP checkpoint {ckpt_path} lacks {missing[:4]}...")
    return enc.to(device).eval()
