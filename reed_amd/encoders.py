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
from .ops import EPI_BF16, EPI_QGELU, EPI_RES_BF16, NT

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

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
        x = raw.float() / 255.0
        res = x.shape[-1]
        x = torch.nn.functional.interpolate(x, 224 * (res // 256), mode="bicubic")
        mean = torch.tensor(CLIP_MEAN, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
        std = torch.tensor(CLIP_STD, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
        return (x - mean) / std

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
