"""SD-VAE decoder (the `vae.decode(...)` of image/generate.py:87,156 and image/train.py:446-447) without the diffusers package.

The reference decodes sampled latents with `diffusers.models.AutoencoderKL.from_pretrained("stabilityai/sd-vae-ft-{ema,mse}")`.
That class lives in a third-party dependency which is neither vendored in the reference nor installed in this image
(requirements: diffusers, pinned by the reference's environment file), so this module RESTATES the published architecture of
its decoder half — `AutoencoderKL.decode` = `post_quant_conv` (1x1) followed by `Decoder`: conv_in, a mid block (ResNet,
single-head self-attention over the 32x32 positions, ResNet), four up blocks of three ResNet blocks (channels 512, 512, 256, 128;
nearest x2 upsampling + 3x3 conv after the first three), GroupNorm(32, eps 1e-6) + SiLU + conv_out — with the state-dict key
names of the published checkpoints, so that `diffusion_pytorch_model.safetensors` / `.bin` of sd-vae-ft-ema / -mse loads
directly (`--vae-ckpt` of generate.py). Parity: UNPINNED — there is no diffusers and no checkpoint in this container to compare
against; `oracle/vae.py` restates the same published algorithm a second, independent way and the tests hold the two together
on random weights (tests/test_host_cpu.py, tests/test_vae_gpu.py).

Two forms of the same network over one set of parameters.  `decode(z)` is the product: z on the GPU -> HIP kernels only
(`_HipDecode`: fp32 NHWC activations = token matrices, every convolution / Linear a `reed_gemm` call, GroupNorm statistics,
GroupNorm-apply + SiLU + padding + nearest-x2 + im2col in one pass, softmax rows: csrc/vae.hip; no torch / MIOpen operator, and it
raises without the library).  `decode_torch(z)` is the module tree's plain torch forward, kept for the CPU-side cross-check of
the two restatements and as the GPU tests' second witness; nothing in the product calls it.
"""
import os
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

SD_VAE_CONFIG = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                     norm_num_groups=32, scaling_factor=0.18215)


class _Resnet(nn.Module):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class _Attention(nn.Module):
    """Single-head self-attention over the spatial positions (diffusers `Attention` with heads = 1, residual connection)."""

    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c)])

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x).view(b, c, h * w).transpose(1, 2)          # [b, hw, c]
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        p = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * (c ** -0.5), dim=-1)
        o = self.to_out[0](torch.bmm(p, v))
        return x + o.transpose(1, 2).reshape(b, c, h, w)


class _Mid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(c, c, groups), _Resnet(c, c, groups)])
        self.attentions = nn.ModuleList([_Attention(c, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _Upsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _Up(nn.Module):
    def __init__(self, cin, cout, n, groups, upsample):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if i == 0 else cout, cout, groups) for i in range(n)])
        self.upsamplers = nn.ModuleList([_Upsample(cout)]) if upsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return x if self.upsamplers is None else self.upsamplers[0](x)


class _Decoder(nn.Module):
    def __init__(self, latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups):
        super().__init__()
        rev = list(reversed(block_out_channels))
        self.conv_in = nn.Conv2d(latent_channels, rev[0], 3, padding=1)
        self.mid_block = _Mid(rev[0], norm_num_groups)
        ups, prev = [], rev[0]
        for i, c in enumerate(rev):
            ups.append(_Up(prev, c, layers_per_block + 1, norm_num_groups, upsample=i < len(rev) - 1))
            prev = c
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, rev[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(rev[-1], out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for u in self.up_blocks:
            x = u(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class SDVAEDecoder(nn.Module):
    """`decode(z)` == `AutoencoderKL.decode(z).sample`: z [B, 4, h, w] (already divided by the 0.18215 latent scale, as the
    reference's call sites do) -> images [B, 3, 8h, 8w] in [-1, 1]."""

    def __init__(self, latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 norm_num_groups=32, scaling_factor=0.18215):
        super().__init__()
        self.scaling_factor = scaling_factor
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.decoder = _Decoder(latent_channels, out_channels, tuple(block_out_channels), layers_per_block, norm_num_groups)

    @torch.no_grad()
    def decode_torch(self, z):
        """The module tree evaluated with torch operators (tests only: the CPU cross-check against oracle/vae.py)."""
        return self.decoder(self.post_quant_conv(z))

    @torch.no_grad()
    def decode(self, z, precision="fp32"):
        """z [B, 4, h, w] on the GPU -> images f32 [B, 3, 8h, 8w], on the HIP kernels.  precision = the GEMM operand type:
        "fp32" (default: exact fp32 products, the reference's `--no-tf32` arithmetic and a superset of its TF32 default),
        "fp16" / "bf16" (the 16-bit MFMA kernels; channel counts must then be multiples of 128, as the published config's are)."""
        ops.require_cuda(z, "latents")
        if getattr(self, "_hip", None) is None:
            self._hip = _HipDecode(self)
        return self._hip.decode(z, precision)

    forward = decode


def _round_up(n, m):
    return (n + m - 1) // m * m


class _HipDecode:
    """`SDVAEDecoder.decode` on the library.  Activations: fp32 NHWC, i.e. the row-major matrix [B*H*W, C].  A convolution is
    reed_conv_rows (its row operand: GroupNorm apply + SiLU + zero padding + nearest x2 + the 9 taps, one pass) followed by
    reed_gemm NT with epilogue 6 (fp32 + bias; `accumulate` adds onto the residual in place); the rows are produced in chunks of
    at most `WS_BYTES` so the operand never exceeds that, whatever the batch.  Weights are repacked once per precision to
    [Cout, 9*Cin] in (ky, kx, ci) order, K / N zero-padded to the kernels' granules (K 64, N 128 for the 16-bit builds; 4 / 4
    for fp32)."""
    WS_BYTES = 1 << 30

    def __init__(self, vae):
        self.vae = vae
        self.packs = {}
        self.bias32 = {}
        self.ws = None
        self.gn_ws = None

    # ---- weights ----
    def _pack(self, name, mod, prec):
        key = (name, prec)
        w, b = mod.weight, mod.bias
        ver = (w.data_ptr(), w._version, b.data_ptr(), b._version)
        hit = self.packs.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        hd = ops.half_dtype(prec)
        km, nm = (4, 4) if prec == "fp32" else (64, 128)
        w2 = w.detach().float()
        w2 = w2.permute(0, 2, 3, 1).reshape(w2.shape[0], -1) if w2.ndim == 4 else w2
        co, k = w2.shape
        wm = torch.zeros(_round_up(co, nm), _round_up(k, km), dtype=hd, device=w.device)
        wm[:co, :k] = w2.to(hd)
        bm = torch.zeros(wm.shape[0], dtype=hd, device=w.device)
        bm[:co] = b.detach().float().to(hd)
        self.bias32[key] = torch.zeros(wm.shape[0], dtype=torch.float32, device=w.device)
        self.bias32[key][:co] = b.detach().float()
        pack = (wm, bm, co, k)
        self.packs[key] = (ver, pack)
        return pack

    def _workspace(self, nbytes, dev):
        if self.ws is None or self.ws.numel() < nbytes or self.ws.device != dev:
            self.ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return self.ws

    # ---- passes ----
    def _table(self, x, norm):
        """f32 [B, 3, C]: per channel (group mean, rstd * gamma, beta) of GroupNorm(x)"""
        B, H, W, C = x.shape
        tb = torch.empty(B, 3, C, dtype=torch.float32, device=x.device)
        self.gn_ws = ops.groupnorm_stats(x, B, H * W, C, norm.num_groups, norm.eps, ws=self.gn_ws,
                                         gamma=norm.weight.detach().float().contiguous(), beta=norm.bias.detach().float().contiguous(),
                                         table=tb)
        return tb

    def _conv(self, x, name, mod, prec, taps, norm=None, silu=False, up=False, out=None, accumulate=False):
        """x f32 [B, Hi, Wi, C] -> f32 [B, Ho, Wo, Cout_padded] (+= when accumulate) = conv(act(norm(upsample(x))))"""
        B, Hi, Wi, C = x.shape
        wm, bm, co, k = self._pack(name, mod, prec)
        assert k == taps * C, (name, k, taps, C)
        npad, kcols = wm.shape
        Ho, Wo = (Hi * 2, Wi * 2) if up else (Hi, Wi)
        M = B * Ho * Wo
        if out is None:
            out = torch.empty(B, Ho, Wo, npad, dtype=torch.float32, device=x.device)
        assert out.shape == (B, Ho, Wo, npad) and out.is_contiguous()
        tb = self._table(x, norm) if norm is not None else None
        es = ops.half_dtype(prec).itemsize
        if taps == 9 and C % (4 if prec == "fp32" else 64) == 0:
            # the activation once in the operand type (norm + SiLU + rounding: 6 B / element, 8 with fp32 operands), then the
            # implicit GEMM (csrc/conv.hip; fp32 operands: csrc/gemm_f32.hip with the window gather in its staging loads)
            per = Hi * Wi * C * es
            bc = max(1, min(B, 0x7FFF0000 // per))
            act = self._workspace(bc * per, x.device)
            for b0 in range(0, B, bc):
                nb = min(bc, B - b0)
                ops.conv_rows(x, act, B, Hi, Wi, C, 1, b0 * Hi * Wi, nb * Hi * Wi, C, C, table=tb, silu=silu)
                ops.conv3x3(act, wm, self.bias32[(name, prec)], out.data_ptr() + b0 * Ho * Wo * npad * 4, npad, nb, Hi, Wi, C, npad,
                            upsample=up, accumulate=accumulate)
            return out
        direct = prec == "fp32" and taps == 1 and norm is None and not silu and not up and kcols == C
        chunk = M if direct else min(M, max(1024, (self.WS_BYTES // (kcols * es)) // 1024 * 1024))
        cols = None if direct else self._workspace(chunk * kcols * es, x.device)
        for r0 in range(0, M, chunk):
            n = min(chunk, M - r0)
            if direct:
                P = x.data_ptr() + r0 * C * 4
            else:
                ops.conv_rows(x, cols, B, Hi, Wi, C, taps, r0, n, kcols, kcols, table=tb, silu=silu, upsample=up)
                P = cols
            ops.gemm(ops.NT, ops.EPI_F32, P, wm, n, npad, kcols, out.data_ptr() + r0 * npad * 4, kcols, kcols, npad, bias=bm,
                     accumulate=accumulate)
        return out

    def _resnet(self, x, name, r, prec):
        h = self._conv(x, name + ".conv1", r.conv1, prec, 9, norm=r.norm1, silu=True)
        if r.conv_shortcut is not None:
            x = self._conv(x, name + ".conv_shortcut", r.conv_shortcut, prec, 1)
        return self._conv(h, name + ".conv2", r.conv2, prec, 9, norm=r.norm2, silu=True, out=x, accumulate=True)   # x += ...

    def _attention(self, x, name, a, prec):
        B, H, W, C = x.shape
        T, hd = H * W, ops.half_dtype(prec)
        src = tuple((p.data_ptr(), p._version) for p in (a.to_q.weight, a.to_k.weight, a.to_v.weight, a.to_q.bias, a.to_k.bias, a.to_v.bias))
        if getattr(a, "_qkv_src", None) != src:      # q, k, v as one [3C, C] contraction
            a._qkv = types.SimpleNamespace(weight=torch.cat([a.to_q.weight, a.to_k.weight, a.to_v.weight]).detach(),
                                           bias=torch.cat([a.to_q.bias, a.to_k.bias, a.to_v.bias]).detach())
            a._qkv_src = src
        wm, bm, co, k = self._pack(name + ".qkv", a._qkv, prec)
        kcols = wm.shape[1]
        assert wm.shape[0] == 3 * C, "attention width must be a multiple of the GEMM's column granule"
        t = torch.empty(B * T, kcols, dtype=hd, device=x.device)
        ops.conv_rows(x, t, B, H, W, C, 1, 0, B * T, kcols, kcols, table=self._table(x, a.group_norm))
        Tp = _round_up(T, 4 if prec == "fp32" else 128)      # rows of an image's q / k / v block: the score GEMM's N and the
        qkv = (torch.zeros if Tp != T else torch.empty)(B * Tp, 3 * C, dtype=hd, device=x.device)   # P V GEMM's K granule
        es = hd.itemsize
        if Tp == T:
            ops.gemm(ops.NT, ops.EPI_BF16, t, wm, B * T, 3 * C, kcols, qkv, kcols, kcols, 3 * C, bias=bm)
        else:
            for b in range(B):
                ops.gemm(ops.NT, ops.EPI_BF16, t.data_ptr() + b * T * kcols * es, wm, T, 3 * C, kcols,
                         qkv.data_ptr() + b * Tp * 3 * C * es, kcols, kcols, 3 * C, bias=bm)
        s = torch.empty(T, Tp, dtype=torch.float32, device=x.device)
        pm = torch.zeros(T, Tp, dtype=hd, device=x.device)          # pad columns stay zero
        o = torch.empty(B * T, C, dtype=hd, device=x.device)
        for b in range(B):
            q = qkv.data_ptr() + b * Tp * 3 * C * es
            ops.gemm(ops.NT, ops.EPI_F32, q, q + C * es, T, Tp, C, s, 3 * C, 3 * C, Tp)                     # S = Q K^T
            ops.softmax_rows(s, Tp, pm, Tp, T, T, C ** -0.5)
            ops.gemm(ops.NN, ops.EPI_BF16, pm, q + 2 * C * es, T, C, Tp, o.data_ptr() + b * T * C * es, Tp, 3 * C, C)   # O = P V
        wo, bo, _, ko = self._pack(name + ".to_out.0", a.to_out[0], prec)
        assert wo.shape == (C, C)
        ops.gemm(ops.NT, ops.EPI_F32, o, wo, B * T, C, C, x, C, C, C, bias=bo, accumulate=True)           # x += out(O)
        return x

    def decode(self, z, precision="fp32"):
        v, d = self.vae, self.vae.decoder
        prev = ops.use(precision)
        try:
            x = z.detach().float().permute(0, 2, 3, 1).contiguous()                       # NHWC
            if x.shape[-1] % 4:
                raise ValueError("latent channels must be a multiple of 4")
            lc = x.shape[-1]
            x = self._conv(x, "post_quant_conv", v.post_quant_conv, precision, 1)[..., :lc].contiguous()
            x = self._conv(x, "decoder.conv_in", d.conv_in, precision, 9)
            if x.shape[-1] != d.conv_in.out_channels:
                raise ValueError(f"precision {precision!r} needs channel counts that are multiples of 128")
            m = d.mid_block
            x = self._resnet(x, "decoder.mid_block.resnets.0", m.resnets[0], precision)
            x = self._attention(x, "decoder.mid_block.attentions.0", m.attentions[0], precision)
            x = self._resnet(x, "decoder.mid_block.resnets.1", m.resnets[1], precision)
            for i, u in enumerate(d.up_blocks):
                for j, r in enumerate(u.resnets):
                    x = self._resnet(x, f"decoder.up_blocks.{i}.resnets.{j}", r, precision)
                if u.upsamplers is not None:
                    x = self._conv(x, f"decoder.up_blocks.{i}.upsamplers.0.conv", u.upsamplers[0].conv, precision, 9, up=True)
            y = self._conv(x, "decoder.conv_out", d.conv_out, precision, 9, norm=d.conv_norm_out, silu=True)
            return y[..., :d.conv_out.out_channels].permute(0, 3, 1, 2).contiguous()
        finally:
            ops.use(prev)


# attention parameters of the checkpoints published before diffusers renamed them (its loader converts these names too)
_LEGACY_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def decoder_state_dict(sd):
    """The decoder half of an AutoencoderKL state dict in this module's key names: drops encoder.* / quant_conv.*, maps the
    legacy attention names, and squeezes 1x1-conv attention weights [C, C, 1, 1] of very old exports to Linear weights."""
    out = {}
    for k, v in sd.items():
        if not (k.startswith("decoder.") or k.startswith("post_quant_conv.")):
            continue
        parts = k.split(".")
        if "attentions" in parts:
            for old, new in _LEGACY_ATTN.items():
                if old in parts:
                    i = parts.index(old)
                    parts[i:i + 1] = new.split(".")
                    break
            if parts[-1] == "weight" and v.ndim == 4 and "group_norm" not in parts:
                v = v[:, :, 0, 0]
        out[".".join(parts)] = v
    return out


def load_sd_vae_decoder(path, device="cpu", dtype=torch.float32, **config):
    """path: a diffusers model directory (diffusion_pytorch_model.safetensors or .bin inside), or such a file."""
    if os.path.isdir(path):
        for name in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.bin"):
            if os.path.exists(os.path.join(path, name)):
                path = os.path.join(path, name)
                break
        else:
            raise FileNotFoundError(f"{path}: no diffusion_pytorch_model.safetensors / .bin inside")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)
        sd = sd.get("state_dict", sd)
    vae = SDVAEDecoder(**{**SD_VAE_CONFIG, **config})
    vae.load_state_dict(decoder_state_dict(sd), strict=True)
    return vae.to(device=device, dtype=dtype).eval()
