"""Deterministic, library-RNG-free tensor fill (splitmix64 hash of the element index), used to give the
reference model (tools/gen_golden.py), the oracle and the HIP model identical non-trivial weights and
inputs on any machine without shipping weight files."""
import zlib

import numpy as np
import torch

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def _splitmix(n, seed):
    with np.errstate(over="ignore"):
        x = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * _M1 + np.uint64(seed)
        x ^= x >> np.uint64(30)
        x *= _M2
        x ^= x >> np.uint64(27)
        x *= _M3
        x ^= x >> np.uint64(31)
    return x


def uniform(shape, seed, lo=-1.0, hi=1.0):
    """float32 tensor, uniform in [lo, hi), a pure function of (shape, seed)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (_splitmix(n, seed) >> np.uint64(40)).astype(np.float64) / float(1 << 24)  # [0,1)
    return torch.from_numpy((lo + (hi - lo) * u).astype(np.float32)).reshape(shape)


def normal(shape, seed):
    """float32 ~N(0,1) via Box-Muller on two hashed uniforms."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = ((_splitmix(n, seed) >> np.uint64(40)).astype(np.float64) + 1.0) / float((1 << 24) + 1)
    u2 = (_splitmix(n, seed ^ 0x5DEECE66D) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy(z.astype(np.float32)).reshape(shape)


def name_seed(name, base=0):
    return (zlib.crc32(name.encode()) + 7919 * base) & 0x7FFFFFFF


def fill_state_dict(sd, base_seed=0, adaln_gain=0.5, final_gain=0.5):
    """In-place fill of a SiT state_dict (reference key names). pos_embed is left as constructed.
    Weights ~ U(-a, a) with a = gain*sqrt(3/fan_in) (unit-variance-preserving), biases ~ U(-0.05, 0.05);
    adaLN and final-layer weights are non-zero so that every block contributes (at the reference's own
    init they are zero and the network is the identity: SURVEY.md §3.4)."""
    for name, t in sd.items():
        if name == "pos_embed":
            continue
        seed = name_seed(name, base_seed)
        if name.endswith("bias"):
            v = uniform(tuple(t.shape), seed, -0.05, 0.05)
        elif "embedding_table" in name:
            v = uniform(tuple(t.shape), seed, -0.05, 0.05)
        elif name.endswith("weight") and t.ndim >= 2:
            fan_in = int(np.prod(t.shape[1:]))
            gain = 1.0
            if "adaLN_modulation" in name:
                gain = adaln_gain
            elif name.startswith("final_layer.linear"):
                gain = final_gain
            a = gain * (3.0 / fan_in) ** 0.5
            v = uniform(tuple(t.shape), seed, -a, a)
        elif name.endswith("weight"):  # 1-D (q_norm / k_norm affine)
            v = 1.0 + uniform(tuple(t.shape), seed, -0.1, 0.1)
        else:
            continue
        with torch.no_grad():
            t.copy_(v.to(t.dtype))
    return sd
