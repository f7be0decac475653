"""CPU restatement of one optimisation step of image/train.py (:84-105 sample_posterior / update_ema,
:363-385 schedules, :387-412 loss combine, clip, AdamW, EMA). TEST INFRASTRUCTURE; also the "port" CPU baseline
timed by bench.py."""
import math

import numpy as np
import torch

from . import loss as oloss
from . import sit as osit


def sample_posterior(moments, scale=0.18215, bias=0.0, noise=None):
    mean, std = torch.chunk(moments, 2, dim=1)
    if noise is None:
        noise = torch.randn_like(mean)
    return (mean + std * noise) * scale + bias


def repa_weight_decay(kind, step, repa_steps):
    if kind == "constant":
        return 1.0
    if kind == "linear":
        return max(1.0 - step / repa_steps, 0.0)
    if kind == "cosine":
        return max((1.0 + np.cos(np.pi * step / repa_steps)) / 2, 0.0)
    raise NotImplementedError(kind)


def diffusion_loss_decay(kind, step, start, warm, max_steps):
    top = warm + start
    if step < start:
        return 0.0
    if start <= step < top:
        return (step - start) / warm
    if kind == "constant":
        return 1.0
    if kind == "linear":
        return 1.0 - (step - top) / (max_steps - top)
    if kind == "cosine":  # operator precedence exactly as train.py:383
        return (1.0 + np.cos(np.pi * (step - top) / max_steps - top)) / 2
    raise NotImplementedError(kind)


class Trainer:
    """Holds params (requires_grad leaves), EMA copy and a torch AdamW, and runs reference-equivalent steps."""

    def __init__(self, P, cfg, enc_names, repa_coeff, *, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.0, eps=1e-8,
                 max_grad_norm=1.0, proj_coeff=0.5, path_type="linear", weighting="uniform",
                 time_schedule="constant", cutoffs=(0.0, 1.0), autocast_bf16=False, ema_decay=0.9999,
                 repa_decay="constant", repa_steps=400000, start_diffusion_steps=0, diffusion_warm_up_steps=50000,
                 diffusion_decay="constant", max_train_steps=400000, autocast_dtype=None, init_scale=None):
        self.cfg = cfg
        self.P = {k: v.clone().requires_grad_(k != "pos_embed") for k, v in P.items()}
        self.ema = {k: v.detach().clone() for k, v in P.items()}
        self.opt = torch.optim.AdamW([v for k, v in self.P.items() if k != "pos_embed"], lr=lr, betas=betas,
                                     weight_decay=weight_decay, eps=eps)
        self.model = osit.OracleModel(self.P, cfg, autocast_bf16=autocast_bf16, training=True,
                                      autocast_dtype=autocast_dtype)
        # --mixed-precision fp16: accelerate wraps backward / clip / step in torch's GradScaler at its defaults
        # (accelerator.backward -> scaler.scale(loss).backward(); clip_grad_norm_ -> unscale_; optimizer.step ->
        # scaler.step + update), train.py:401-408
        self.scaler = None if init_scale is None else torch.amp.GradScaler("cpu", init_scale=init_scale)
        self.enc_names = list(enc_names)
        self.loss_weights = {n: repa_coeff[i] for i, n in enumerate(enc_names)}
        self.k = dict(path_type=path_type, weighting=weighting, time_schedule=time_schedule, cutoffs=cutoffs)
        self.max_grad_norm, self.proj_coeff, self.ema_decay = max_grad_norm, proj_coeff, ema_decay
        self.sched = (repa_decay, repa_steps, start_diffusion_steps, diffusion_warm_up_steps, diffusion_decay,
                      max_train_steps)
        self.step_idx = 0

    def step(self, x, labels, zs, t=None, noise=None, drop_mask=None):
        rd, rs, sd, wu, dd, mx = self.sched
        w_repa = repa_weight_decay(rd, self.step_idx, rs)
        w_diff = diffusion_loss_decay(dd, self.step_idx, sd, wu, mx)
        self.model.drop_mask = drop_mask
        if self.enc_names:
            out = oloss.si_loss(self.model, x, dict(y=labels), zs, enc_names=self.enc_names,
                                loss_weights=self.loss_weights, t=t, noise=noise, **self.k)
            proj = out["proj_loss"]
        else:  # alignment off: denoising loss only
            out = oloss.si_loss(self.model, x, dict(y=labels), [], enc_names=[], loss_weights={}, t=t, noise=noise,
                                **self.k)
            proj = torch.zeros(())
        den = out["denoising_loss"].mean()
        proj_mean = proj.mean() if torch.is_tensor(proj) else torch.tensor(float(proj))
        total = den * w_diff + proj_mean * self.proj_coeff * w_repa
        self.opt.zero_grad(set_to_none=True)
        if self.scaler is not None:
            self.scaler.scale(total).backward()
            self.scaler.unscale_(self.opt)
        else:
            total.backward()
        gn = torch.nn.utils.clip_grad_norm_([v for k, v in self.P.items() if k != "pos_embed"], self.max_grad_norm)
        if self.scaler is not None:
            self.scaler.step(self.opt)   # skipped when unscale_ saw inf / nan
            self.scaler.update()
        else:
            self.opt.step()
        with torch.no_grad():
            for k, v in self.P.items():
                self.ema[k].mul_(self.ema_decay).add_(v.detach(), alpha=1 - self.ema_decay)
        self.step_idx += 1
        return {"loss": float(total), "denoising_loss": float(den), "proj_loss": float(proj_mean),
                "grad_norm": float(gn), "scale": self.scaler.get_scale() if self.scaler is not None else 1.0,
                "img_proj_loss": float(out["img_proj_loss"]),
                "text_proj_loss": float(out["text_proj_loss"])}
