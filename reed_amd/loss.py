"""SILoss — drop-in for the reference's image/loss.py (same constructor, same __call__ signature and return dict),
with the arithmetic on HIP kernels: interpolant / v-target, per-sample MSE, cosine alignment of the projector
outputs against the frozen-encoder features, forward and backward (reed_amd/csrc/loss.hip).

O(batch) bookkeeping (time weights, means over the batch) stays in torch on B-element tensors, written with the
reference's own broadcasting so that its [B] x [B,1,1,1] quirk (loss.py:221-222) is reproduced.
Extra keyword arguments `time_input=` / `noises=` inject the random draws (parity tests); otherwise t is drawn on
the CPU generator and noise on the device generator exactly where the reference draws them (loss.py:159,172).
"""
import math

import numpy as np
import torch

from . import ops

IMAGE_ENCODERS = ["dinov2", "mocov3", "clip", "mae", "jepa"]


def _to_device_async(t, device):
    """fp32 copy of `t` on `device` without stalling the host.  The reference draws the time steps with the HOST generator
    (loss.py:159-170) and so do we — but a pageable host-to-device copy blocks the host until the stream has drained, i.e. one
    full host/GPU synchronisation per training step (measured: the forward then starts with an empty queue every step and is
    enqueue-bound at b <= 64).  Through pinned staging memory (torch's caching host allocator) the copy is asynchronous."""
    t = t.to(torch.float32)
    if t.device.type == "cpu":
        t = t.pin_memory()
    return t.to(device, non_blocking=True)


def mean_flat(x):
    return torch.mean(x, dim=list(range(1, len(x.size()))))


def sum_flat(x):
    return torch.sum(x, dim=list(range(1, len(x.size()))))


class _MSE(torch.autograd.Function):
    """mean_flat((out - target)**2) -> [B]  (loss.py:186)"""

    @staticmethod
    def forward(ctx, out, target):
        B = out.shape[0]
        per = out.numel() // B
        loss = torch.empty(B, dtype=torch.float32, device=out.device)
        ops.mse_fwd(out, target, loss, B, per)
        ctx.save_for_backward(out, target)
        return loss

    @staticmethod
    def backward(ctx, g):
        out, target = ctx.saved_tensors
        B = out.shape[0]
        dout = torch.empty_like(out)
        ops.mse_bwd(out, target, g.contiguous().float(), dout, B, out.numel() // B)
        return dout, None


class _Cosine(torch.autograd.Function):
    """-(normalize(z) * normalize(zt)).sum(-1).mean(-1) -> [B]  (loss.py:207-221)"""

    @staticmethod
    def forward(ctx, zt, z):
        B = zt.shape[0]
        T = zt.shape[1] if zt.ndim == 3 else 1
        Z = zt.shape[-1]
        loss = torch.empty(B, dtype=torch.float32, device=zt.device)
        rowdot = torch.empty(B * T, dtype=torch.float32, device=zt.device)
        prev = ops.use(ops.precision_of(zt.dtype))   # the projector output's 16-bit type picks the build
        try:
            ops.cosine_fwd(zt, z, rowdot, loss, B, T, Z)
        finally:
            ops.use(prev)
        ctx.save_for_backward(zt, z)
        ctx.dims = (B, T, Z)
        return loss

    @staticmethod
    def backward(ctx, g):
        zt, z = ctx.saved_tensors
        B, T, Z = ctx.dims
        dzt = torch.empty_like(zt)
        prev = ops.use(ops.precision_of(zt.dtype))
        try:
            ops.cosine_bwd(zt, z, g.contiguous().float(), dzt, B, T, Z)
        finally:
            ops.use(prev)
        return dzt, None


class SILoss:
    def __init__(self, prediction="v", path_type="linear", weighting="uniform", encoders=[], enc_names=[],
                 loss_weights={"dinov2": 1.0, "t5": 1.0}, time_schedule="constant", cutoffs=[0.0, 1.0],
                 accelerator=None, latents_scale=None, latents_bias=None):
        self.prediction = prediction
        self.weighting = weighting
        self.path_type = path_type
        self.encoders = encoders
        self.enc_names = enc_names
        self.accelerator = accelerator
        self.latents_scale = latents_scale
        self.latents_bias = latents_bias
        self.loss_weights = loss_weights
        self.time_schedule = time_schedule
        self.cutoffs = cutoffs
        assert len(loss_weights) == len(enc_names), "Loss weights must be provided for each encoder."
        if prediction != "v":
            raise NotImplementedError("only v-prediction is supported (as in the reference)")
        if path_type not in ("linear", "cosine"):
            raise NotImplementedError(path_type)

    def interpolant(self, t):
        if self.path_type == "linear":
            return 1 - t, t, -1, 1
        h = np.pi / 2
        return torch.cos(t * h), torch.sin(t * h), -h * torch.sin(t * h), h * torch.cos(t * h)

    def time_weight(self, t, base_weight=1.0, schedule="constant", cutoffs=[0.0, 1.0]):
        if schedule == "linear":
            scale = 1 - t
        elif schedule == "cosine":
            scale = 0.5 * (1 + torch.cos(math.pi * t))
        elif schedule == "sigmoid":
            scale = 1 / (1 + torch.exp((t - 0.5) * 10))
        elif schedule == "constant":
            scale = torch.ones_like(t)
        elif schedule == "loglinear":
            scale = 1 - torch.log(t + 1)
        elif schedule == "cutoff":
            scale = torch.ones_like(t)
            scale[t < cutoffs[0]] = 0
            scale[t > cutoffs[1]] = 0
        else:
            raise ValueError("Invalid schedule. Choose from 'linear', 'cosine', 'sigmoid'.")
        return base_weight * scale

    def __call__(self, model, images, model_kwargs=None, zs=None, **kwargs):
        if model_kwargs is None:
            model_kwargs = {}
        ops.require_cuda(images, "images")
        B = images.shape[0]
        time_input = kwargs.get("time_input")
        if time_input is None:
            if self.weighting == "uniform":
                time_input = torch.rand((B, 1, 1, 1))
            elif self.weighting == "lognormal":
                sigma = torch.randn((B, 1, 1, 1)).exp()
                time_input = sigma / (1 + sigma) if self.path_type == "linear" else 2 / np.pi * torch.atan(sigma)
            else:
                raise ValueError(self.weighting)
        time_input = _to_device_async(time_input.reshape(B, 1, 1, 1), images.device)
        noises = kwargs.get("noises")
        images = images.contiguous().float()
        if noises is None:
            noises = torch.randn_like(images)
        noises = _to_device_async(noises, images.device).contiguous()

        model_input = torch.empty_like(images)
        model_target = torch.empty_like(images)
        tflat = time_input.flatten().contiguous()
        ops.interpolant(images, noises, tflat, model_input, model_target, B, images.numel() // B,
                        0 if self.path_type == "linear" else 1)
        model_kwargs = dict(model_kwargs)
        model_kwargs["inference"] = False
        model_output, zs_tilde = model(model_input, tflat, **model_kwargs)
        denoising_loss = _MSE.apply(model_output.contiguous().float(), model_target)

        proj_loss = 0.0
        acc = {"image": {"loss": 0.0, "count": 0}, "text": {"loss": 0.0, "count": 0}}
        zs = zs or []
        save = kwargs.get("save_projloss", False)
        if save:
            loss_saver = {"image": torch.zeros(B, device=images.device), "text": torch.zeros(B, device=images.device),
                          "time": tflat}
        for z, z_tilde, enc_name in zip(zs, zs_tilde or [], self.enc_names):
            w = self.loss_weights.get(enc_name, 1.0)
            wts = self.time_weight(time_input, w, self.time_schedule, self.cutoffs)
            key = "image" if enc_name in IMAGE_ENCODERS or len(self.enc_names) == 1 else "text"
            if z.ndim == 2:
                assert key == "text", "Only text encoders should have 2D embeddings."
                assert z_tilde.ndim == 2, "Pooling to 2D to align with text embeddings."
            if w == 0.0:
                wts = torch.ones_like(wts)
            if z_tilde.dtype not in (torch.bfloat16, torch.float16) and getattr(model, "precision", None) != "fp32":
                z_tilde = z_tilde.to(torch.bfloat16)   # foreign model returning fp32 projector outputs (a reed_amd model
                #                                          at precision "fp32" returns fp32 on purpose: --mixed-precision no)
            curr_loss = _Cosine.apply(z_tilde.contiguous(), z.to(images.device).contiguous().float())  # [B]
            weighted_loss = (curr_loss * wts).mean()  # [B] x [B,1,1,1] broadcast, as the reference
            proj_loss = proj_loss + weighted_loss
            acc[key]["loss"] = acc[key]["loss"] + curr_loss.mean()
            acc[key]["count"] += 1
            if save:
                loss_saver[key] += curr_loss.detach()
        img_proj_loss = acc["image"]["loss"] / max(1, acc["image"]["count"])
        text_proj_loss = acc["text"]["loss"] / max(1, acc["text"]["count"])
        out = {"denoising_loss": denoising_loss, "proj_loss": proj_loss, "img_proj_loss": img_proj_loss,
               "text_proj_loss": text_proj_loss}
        if save:
            out["loss_saver"] = loss_saver
        return out
