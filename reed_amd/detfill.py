"""Library-RNG-free deterministic tensors for the bench's `loss_vs_ref` leg (BASELINE.json's second number): the committed
fixture tests/golden/xl2_c2.npz holds the reference's per-step losses for weights and inputs that are pure functions of
(tensor name | shape, seed) — a 64-bit mix of the element index — so the HIP path can rebuild them on the GPU box without weight
files and without anything of the reference or of oracle/ (the product imports neither).  The recipe is the one
tools/gen_golden.py wrote the fixtures with; tests/test_host_cpu.py::test_detfill_matches_the_fixture_recipe pins this module to it
bit for bit."""
import zlib

import numpy as np
import torch

_K = (np.uint64(0x9E3779B97F4A7C15), np.uint64(0xBF58476D1CE4E5B9), np.uint64(0x94D049BB133111EB))


def _mix(n, seed):
    """splitmix64 finaliser of (index + 1) * golden ratio + seed"""
    with np.errstate(over="ignore"):
        h = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * _K[0] + np.uint64(seed)
        for shift, mul in ((30, _K[1]), (27, _K[2])):
            h ^= h >> np.uint64(shift)
            h *= mul
        h ^= h >> np.uint64(31)
    return h


def _unit24(n, seed):
    return (_mix(n, seed) >> np.uint64(40)).astype(np.float64)   # the top 24 bits


def uniform(shape, seed, lo=-1.0, hi=1.0):
    n = int(np.prod(shape)) if len(shape) else 1
    u = _unit24(n, seed) / float(1 << 24)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32)).reshape(shape)


def normal(shape, seed):
    n = int(np.prod(shape)) if len(shape) else 1
    r = np.sqrt(-2.0 * np.log((_unit24(n, seed) + 1.0) / float((1 << 24) + 1)))
    th = 2.0 * np.pi * (_unit24(n, seed ^ 0x5DEECE66D) / float(1 << 24))
    return torch.from_numpy((r * np.cos(th)).astype(np.float32)).reshape(shape)


def fill_model(state_dict, base_seed=0):
    """Weights U(-a, a), a = gain sqrt(3 / fan_in) (gain 0.5 for the adaLN and final-layer matrices, so that every block does
    real work), biases and the label table U(-0.05, 0.05), 1-D affine weights 1 + U(-0.1, 0.1); pos_embed as constructed."""
    for name, t in state_dict.items():
        if name == "pos_embed":
            continue
        seed = (zlib.crc32(name.encode()) + 7919 * base_seed) & 0x7FFFFFFF
        if name.endswith("bias") or "embedding_table" in name:
            v = uniform(tuple(t.shape), seed, -0.05, 0.05)
        elif name.endswith("weight") and t.ndim >= 2:
            gain = 0.5 if ("adaLN_modulation" in name or name.startswith("final_layer.linear")) else 1.0
            a = gain * (3.0 / int(np.prod(t.shape[1:]))) ** 0.5
            v = uniform(tuple(t.shape), seed, -a, a)
        elif name.endswith("weight"):
            v = 1.0 + uniform(tuple(t.shape), seed, -0.1, 0.1)
        else:
            continue
        with torch.no_grad():
            t.copy_(v.to(t.dtype))
    return state_dict


def step_inputs(B, seed, z_dims, C=4, HW=32, T=256, num_classes=1000):
    """(x, noise, t, labels, label-drop uniforms, zs) of injected step `seed` (tools/gen_golden.py:inputs)."""
    x = normal((B, C, HW, HW), 1000 + seed)
    noise = normal((B, C, HW, HW), 2000 + seed)
    t = uniform((B,), 3000 + seed, 0.02, 0.98)
    y = (uniform((B,), 4000 + seed, 0.0, 1.0) * num_classes).long().clamp_(0, num_classes - 1)
    drop_u = uniform((B,), 5000 + seed, 0.0, 1.0)
    zs = [normal((B, T, z), 6000 + seed + 17 * j) for j, z in enumerate(z_dims)]
    return x, noise, t, y, drop_u, zs
