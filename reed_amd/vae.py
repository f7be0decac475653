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
on random weights (tests/test_host_cpu.py). SURVEY.md §8f N4: one decode per 499-998 SiT evaluations, < 1 % of the sampling
wall clock, so it runs on torch's own convolution kernels (MIOpen) — plumbing, not a hot path.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

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
    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))

    forward = decode


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
