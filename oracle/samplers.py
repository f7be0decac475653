"""CPU restatement of the samplers (reference: image/samplers.py:5-43,46-104,107-187). TEST INFRASTRUCTURE.
fp64 state, model evaluated in the latents' dtype; `noises` injects the Euler-Maruyama draws for parity."""
import numpy as np
import torch


def _score(v, x, t, path_type):
    t = t.view(-1, *([1] * (x.ndim - 1)))
    if path_type == "linear":
        a, da, s, ds = 1 - t, -torch.ones_like(x), t, torch.ones_like(x)
    elif path_type == "cosine":
        h = np.pi / 2
        a, s, da, ds = torch.cos(t * h), torch.sin(t * h), -h * torch.sin(t * h), h * torch.cos(t * h)
    else:
        raise NotImplementedError(path_type)
    r = a / da
    var = s ** 2 - r * ds * s
    return (r * v - x) / var


def _eval(model, x, y, y_null, t, dtype, guided):
    if guided:
        xin, yin = torch.cat([x, x], 0), torch.cat([y, y_null], 0)
    else:
        xin, yin = x, y
    tin = torch.ones(xin.size(0), dtype=torch.float64, device=x.device) * t
    return model(xin.to(dtype), tin.to(dtype), y=yin)[0].to(torch.float64), xin, tin


def euler_sampler(model, latents, y, num_steps=20, heun=False, cfg_scale=1.0, guidance_low=0.0,
                  guidance_high=1.0, path_type="linear", **_):
    y_null = torch.full((y.size(0),), 1000, device=y.device, dtype=y.dtype)  # hard-coded null id (samplers.py:59)
    dtype = latents.dtype
    ts = torch.linspace(1, 0, num_steps + 1, dtype=torch.float64)
    x_next = latents.to(torch.float64)
    with torch.no_grad():
        for i, (tc, tn) in enumerate(zip(ts[:-1], ts[1:])):
            x_cur = x_next
            guided = cfg_scale > 1.0 and guidance_low <= tc <= guidance_high
            d, _, _ = _eval(model, x_cur, y, y_null, tc, dtype, guided)
            if guided:
                dc, du = d.chunk(2)
                d = du + cfg_scale * (dc - du)
            x_next = x_cur + (tn - tc) * d
            if heun and i < num_steps - 1:
                d2, _, _ = _eval(model, x_next, y, y_null, tn, dtype, guided)  # t_cur interval test
                if guided:
                    dc, du = d2.chunk(2)
                    d2 = du + cfg_scale * (dc - du)
                x_next = x_cur + (tn - tc) * (0.5 * d + 0.5 * d2)
    return x_next


def euler_maruyama_sampler(model, latents, y, num_steps=20, heun=False, cfg_scale=1.0, guidance_low=0.0,
                           guidance_high=1.0, path_type="linear", noises=None, **_):
    y_null = torch.full((y.size(0),), 1000, device=y.device, dtype=y.dtype)
    dtype = latents.dtype
    ts = torch.cat([torch.linspace(1.0, 0.04, num_steps, dtype=torch.float64), torch.tensor([0.0], dtype=torch.float64)])
    x_next = latents.to(torch.float64)

    def drift(x_cur, tc):
        guided = cfg_scale > 1.0 and guidance_low <= tc <= guidance_high
        v, xin, tin = _eval(model, x_cur, y, y_null, tc, dtype, guided)
        d = v - 0.5 * (2 * tc) * _score(v, xin, tin, path_type)
        if guided:
            dc, du = d.chunk(2)
            d = du + cfg_scale * (dc - du)
        return d

    with torch.no_grad():
        for i, (tc, tn) in enumerate(zip(ts[:-2], ts[1:-1])):
            dt = tn - tc
            x_cur = x_next
            eps = noises[i].to(torch.float64) if noises is not None else torch.randn_like(x_cur)
            deps = eps * torch.sqrt(torch.abs(dt))
            x_next = x_cur + drift(x_cur, tc) * dt + torch.sqrt(2 * tc) * deps
        tc, tn = ts[-2], ts[-1]
        x_cur = x_next
        return x_cur + (tn - tc) * drift(x_cur, tc)
