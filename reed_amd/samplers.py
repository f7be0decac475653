"""Euler / Heun ODE and Euler-Maruyama SDE samplers — drop-in for the reference's image/samplers.py (same
positional and keyword signatures; unknown keywords such as the `prediction=` that the reference's own train.py
passes (train.py:434-445) are accepted and ignored). The fp64 state arithmetic runs in
reed_amd/csrc/sampler.hip in the reference's operation order; the model is evaluated in fp32.
"""
import torch

from . import ops


def _check(latents):
    ops.require_cuda(latents, "latents")
    if latents.dtype != torch.float32:
        raise TypeError("latents must be float32 (the model is evaluated in the latents' dtype; the HIP model "
                        "takes float32 inputs)")


def _check_labels(model, y, cfg_scale):
    """The reference's nn.Embedding raises on a label outside the table (sit.py:98) and the samplers hard-code the null
    class id 1000 (samplers.py:59,120). One host check per sampler call (not per evaluation)."""
    rows = getattr(getattr(model, "engine", lambda: None)(), "table_rows", None) if hasattr(model, "engine") else None
    if rows is None:
        return
    lo, hi = int(y.min()), int(y.max())
    if lo < 0 or hi >= rows:
        raise IndexError(f"class labels in [{lo}, {hi}] but the model's embedding table has {rows} rows")
    if cfg_scale > 1.0 and rows <= 1000:
        raise IndexError(f"classifier-free guidance uses the null class id 1000 (samplers.py:59) but the model's "
                         f"embedding table has only {rows} rows (num_classes={model.num_classes}, "
                         f"class_dropout_prob={model.class_dropout_prob})")


def _model_out(model, xin, rows, t_cur, y_cur):
    t_in = torch.full((rows,), float(t_cur), dtype=torch.float64, device=xin.device).to(torch.float32)
    out = model(xin, t_in, y=y_cur)[0]
    return out.contiguous().float()


def euler_sampler(model, latents, y, num_steps=20, heun=False, cfg_scale=1.0, guidance_low=0.0, guidance_high=1.0,
                  path_type="linear", **_ignored):
    _check(latents)
    _check_labels(model, y, cfg_scale)
    n = y.size(0)
    if cfg_scale > 1.0:
        y_null = torch.tensor([1000] * n, device=y.device)  # hard-coded null id (samplers.py:59)
    t_steps = torch.linspace(1, 0, num_steps + 1, dtype=torch.float64)
    x_next = latents.to(torch.float64).contiguous()
    nel = x_next.numel()
    shape2 = (2 * latents.shape[0],) + tuple(latents.shape[1:])
    with torch.no_grad():
        for i, (t_cur, t_next) in enumerate(zip(t_steps[:-1], t_steps[1:])):
            x_cur = x_next
            guided = bool(cfg_scale > 1.0 and t_cur <= guidance_high and t_cur >= guidance_low)
            y_cur = torch.cat([y, y_null], dim=0) if guided else y
            rows = 2 * n if guided else n
            xin = torch.empty(shape2 if guided else latents.shape, dtype=torch.float32, device=latents.device)
            ops.sampler_input(x_cur, xin, nel, guided)
            d = _model_out(model, xin, rows, t_cur, y_cur)
            dt = float(t_next - t_cur)
            x_next = torch.empty_like(x_cur)
            second = heun and (i < num_steps - 1)
            d_store = torch.empty_like(x_cur) if second else None
            ops.sampler_update(x_cur, d, None, d_store, x_next, nel, guided, cfg_scale, dt, 1.0, 0.0)
            if second:
                ops.sampler_input(x_next, xin, nel, guided)
                d2 = _model_out(model, xin, rows, t_next, y_cur)  # guidance test still uses t_cur (samplers.py:84-90)
                x_heun = torch.empty_like(x_cur)
                ops.sampler_update(x_cur, d2, d_store, None, x_heun, nel, guided, cfg_scale, dt, 0.5, 0.5)
                x_next = x_heun
    return x_next


def euler_maruyama_sampler(model, latents, y, num_steps=20, heun=False, cfg_scale=1.0, guidance_low=0.0,
                           guidance_high=1.0, path_type="linear", noises=None, **_ignored):
    _check(latents)
    _check_labels(model, y, cfg_scale)
    n = y.size(0)
    if cfg_scale > 1.0:
        y_null = torch.tensor([1000] * n, device=y.device)
    pt = {"linear": 0, "cosine": 1}[path_type]
    t_steps = torch.linspace(1.0, 0.04, num_steps, dtype=torch.float64)
    t_steps = torch.cat([t_steps, torch.tensor([0.0], dtype=torch.float64)])
    x_next = latents.to(torch.float64).contiguous()
    nel = x_next.numel()
    shape2 = (2 * latents.shape[0],) + tuple(latents.shape[1:])

    def step(x_cur, t_cur, t_next, eps, last):
        guided = bool(cfg_scale > 1.0 and t_cur <= guidance_high and t_cur >= guidance_low)
        y_cur = torch.cat([y, y_null], dim=0) if guided else y
        rows = 2 * n if guided else n
        xin = torch.empty(shape2 if guided else latents.shape, dtype=torch.float32, device=latents.device)
        ops.sampler_input(x_cur, xin, nel, guided)
        v = _model_out(model, xin, rows, t_cur, y_cur)
        out = torch.empty_like(x_cur)
        ops.sde_update(x_cur, v, eps, out, nel, guided, cfg_scale, float(t_cur), float(t_next - t_cur), pt, last)
        return out

    with torch.no_grad():
        for i, (t_cur, t_next) in enumerate(zip(t_steps[:-2], t_steps[1:-1])):
            eps = noises[i].to(device=latents.device, dtype=torch.float64).contiguous() if noises is not None \
                else torch.randn_like(x_next)
            x_next = step(x_next, t_cur, t_next, eps, False)
        x_next = step(x_next, t_steps[-2], t_steps[-1], None, True)
    return x_next
