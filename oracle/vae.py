"""CPU restatement (numpy, fp64) of the SD-VAE decoder the reference calls through diffusers (image/generate.py:87,156:
`AutoencoderKL.from_pretrained("stabilityai/sd-vae-ft-ema").decode(z).sample`; image/train.py:446-447 for previews).

TEST INFRASTRUCTURE ONLY (tests/ and nothing else may import this). PARITY UNPINNED: the algorithm lives in a third-party
dependency (diffusers, `AutoencoderKL` / `Decoder` / `UNetMidBlock2D` / `UpDecoderBlock2D` / `ResnetBlock2D` / `Attention`) that
is neither vendored under /root/reference nor installed here, and no checkpoint is available offline, so there is no golden
vector to pin it with. It follows the published architecture for the sd-vae-ft config (latent 4, blocks 128-256-512-512, two
layers per block, 32 groups, eps 1e-6) and is written independently of reed_amd/vae.py (walks the checkpoint's key names,
explicit im2col convolution, explicit statistics) so that the two restatements check each other.
"""
import numpy as np


def _conv(x, w, b, pad):
    B, C, H, W = x.shape
    O, _, kh, kw = w.shape
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    cols = np.empty((B, C * kh * kw, H * W))
    r = 0
    for c in range(C):
        for i in range(kh):
            for j in range(kw):
                cols[:, r] = xp[:, c, i:i + H, j:j + W].reshape(B, -1)
                r += 1
    return (np.einsum("ok,bkp->bop", w.reshape(O, -1), cols) + b[None, :, None]).reshape(B, O, H, W)


def _gn(x, g, b, groups, eps=1e-6):
    B, C, H, W = x.shape
    y = x.reshape(B, groups, -1)
    y = (y - y.mean(-1, keepdims=True)) / np.sqrt(y.var(-1, keepdims=True) + eps)
    return y.reshape(B, C, H, W) * g[None, :, None, None] + b[None, :, None, None]


def _silu(x):
    return x / (1.0 + np.exp(-x))


class Decoder:
    def __init__(self, sd, groups=32):
        self.p = {k: np.asarray(v, dtype=np.float64) for k, v in sd.items()}
        self.groups = groups

    def conv(self, x, name, pad):
        return _conv(x, self.p[name + ".weight"], self.p[name + ".bias"], pad)

    def norm(self, x, name):
        return _gn(x, self.p[name + ".weight"], self.p[name + ".bias"], self.groups)

    def resnet(self, x, name):
        h = self.conv(_silu(self.norm(x, name + ".norm1")), name + ".conv1", 1)
        h = self.conv(_silu(self.norm(h, name + ".norm2")), name + ".conv2", 1)
        if name + ".conv_shortcut.weight" in self.p:
            x = self.conv(x, name + ".conv_shortcut", 0)
        return x + h

    def attention(self, x, name):
        B, C, H, W = x.shape
        t = self.norm(x, name + ".group_norm").reshape(B, C, H * W).transpose(0, 2, 1)
        lin = lambda n: t @ self.p[f"{name}.{n}.weight"].T + self.p[f"{name}.{n}.bias"]   # noqa: E731
        q, k, v = lin("to_q"), lin("to_k"), lin("to_v")
        s = q @ k.transpose(0, 2, 1) / np.sqrt(C)
        s = np.exp(s - s.max(-1, keepdims=True))
        o = (s / s.sum(-1, keepdims=True)) @ v
        o = o @ self.p[name + ".to_out.0.weight"].T + self.p[name + ".to_out.0.bias"]
        return x + o.transpose(0, 2, 1).reshape(B, C, H, W)

    def decode(self, z):
        x = self.conv(np.asarray(z, dtype=np.float64), "post_quant_conv", 0)
        x = self.conv(x, "decoder.conv_in", 1)
        x = self.resnet(x, "decoder.mid_block.resnets.0")
        x = self.attention(x, "decoder.mid_block.attentions.0")
        x = self.resnet(x, "decoder.mid_block.resnets.1")
        i = 0
        while f"decoder.up_blocks.{i}.resnets.0.norm1.weight" in self.p:
            j = 0
            while f"decoder.up_blocks.{i}.resnets.{j}.norm1.weight" in self.p:
                x = self.resnet(x, f"decoder.up_blocks.{i}.resnets.{j}")
                j += 1
            if f"decoder.up_blocks.{i}.upsamplers.0.conv.weight" in self.p:
                x = x.repeat(2, axis=2).repeat(2, axis=3)      # nearest-neighbour x2
                x = self.conv(x, f"decoder.up_blocks.{i}.upsamplers.0.conv", 1)
            i += 1
        return self.conv(_silu(self.norm(x, "decoder.conv_norm_out")), "decoder.conv_out", 1)
