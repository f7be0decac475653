"""CPU restatement of SILoss (reference: image/loss.py:7-18,21-64,118-151,153-237). TEST INFRASTRUCTURE.

`si_loss(...)` takes the random draws (t, noise) as explicit arguments — the reference draws t on the CPU
generator and noise on the device generator (loss.py:159,172), which cannot be matched across devices, so parity
tests inject them. With t=None / noise=None it draws like the reference."""
import math

import numpy as np
import torch
import torch.nn.functional as F

IMAGE_ENCODERS = ["dinov2", "mocov3", "clip", "mae", "jepa"]  # loss.py:5


def mean_flat(x):
    return torch.mean(x, dim=list(range(1, x.ndim)))


def interpolant(t, path_type):
    if path_type == "linear":
        return 1 - t, t, -1, 1
    if path_type == "cosine":
        h = np.pi / 2
        return torch.cos(t * h), torch.sin(t * h), -h * torch.sin(t * h), h * torch.cos(t * h)
    raise NotImplementedError(path_type)


def time_weight(t, base_weight=1.0, schedule="constant", cutoffs=(0.0, 1.0)):
    if schedule == "linear":
        s = 1 - t
    elif schedule == "cosine":
        s = 0.5 * (1 + torch.cos(math.pi * t))
    elif schedule == "sigmoid":
        s = 1 / (1 + torch.exp((t - 0.5) * 10))
    elif schedule == "constant":
        s = torch.ones_like(t)
    elif schedule == "loglinear":
        s = 1 - torch.log(t + 1)
    elif schedule == "cutoff":
        s = torch.ones_like(t)
        s[t < cutoffs[0]] = 0
        s[t > cutoffs[1]] = 0
    else:
        raise ValueError(schedule)
    return base_weight * s


def sample_t(n, weighting, path_type, generator=None):
    if weighting == "uniform":
        return torch.rand((n, 1, 1, 1), generator=generator)
    if weighting == "lognormal":
        sigma = torch.randn((n, 1, 1, 1), generator=generator).exp()
        return sigma / (1 + sigma) if path_type == "linear" else 2 / np.pi * torch.atan(sigma)
    raise ValueError(weighting)


def si_loss(model, images, model_kwargs, zs, *, enc_names, loss_weights, path_type="linear", weighting="uniform",
            time_schedule="constant", cutoffs=(0.0, 1.0), t=None, noise=None):
    if t is None:
        t = sample_t(images.shape[0], weighting, path_type)
    t = t.reshape(-1, 1, 1, 1).to(device=images.device, dtype=images.dtype)
    if noise is None:
        noise = torch.randn_like(images)
    a, s, da, ds = interpolant(t, path_type)
    x_t = a * images + s * noise
    target = da * images + ds * noise
    out, zs_tilde = model(x_t, t.flatten(), **dict(model_kwargs or {}), inference=False)
    denoising = mean_flat((out - target) ** 2)
    proj = 0.0
    acc = {"image": [0.0, 0], "text": [0.0, 0]}
    for z, zt, name in zip(zs, zs_tilde, enc_names):
        w = loss_weights.get(name, 1.0)
        wts = time_weight(t, w, time_schedule, cutoffs)
        zt = F.normalize(zt, dim=-1)
        z = F.normalize(z, dim=-1)
        key = "image" if (name in IMAGE_ENCODERS or len(enc_names) == 1) else "text"
        if z.ndim == 2:
            z, zt = z.unsqueeze(1), zt.unsqueeze(1)
        if w == 0.0:
            wts = torch.ones_like(wts)
        cur = -(z * zt).sum(dim=-1).mean(dim=-1)          # [B]
        proj = proj + (cur * wts).mean()                   # [B] x [B,1,1,1] broadcast, as the reference
        acc[key][0] = acc[key][0] + cur.mean()
        acc[key][1] += 1
    return {"denoising_loss": denoising, "proj_loss": proj,
            "img_proj_loss": acc["image"][0] / max(1, acc["image"][1]),
            "text_proj_loss": acc["text"][0] / max(1, acc["text"][1])}
