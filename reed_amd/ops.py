"""Thin Python wrappers over the C ABI (include/reed_hip.h): tensors -> raw device pointers.

PyTorch is used for device memory and streams only; every function here launches hand-written
HIP kernels on torch's current stream and raises if the library is unavailable.
"""
import ctypes

import torch

from . import _lib

NT, NN, TN = 0, 1, 2
EPI_BF16, EPI_GELU, EPI_SILU, EPI_GATE_RES, EPI_DGELU, EPI_DSILU, EPI_F32, EPI_ADDF32_RB, EPI_ATOMIC_F32 = range(9)


def _p(t):
    if t is None:
        return None
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _req(t, dtype, name):
    if not t.is_cuda:
        raise RuntimeError(f"reed_amd: {name} must be a CUDA(HIP) tensor; the hot path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"reed_amd: {name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"reed_amd: {name} must be contiguous")


def gemm(layout, epi, P, Q, M, N, K, C, ldp, ldq, ldc, C2=None, ldc2=0, R=None, ldr=0, bias=None,
         gate=None, ldgate=0, rows_per_gate=1, dbias=None, accumulate=False, split_k=1,
         slab_stride=0):
    L = _lib.load()
    _lib.check(L.reed_gemm(layout, epi, _p(P), ldp, _p(Q), ldq, M, N, K, _p(C), ldc, _p(C2), ldc2,
                           _p(R), ldr, _p(bias), _p(gate), ldgate, rows_per_gate, _p(dbias),
                           int(accumulate), split_k, slab_stride, _stream()), "gemm")


def linear_fwd(x, w, bias, out, epi=EPI_BF16, act_out=None, R=None, gate=None, ldgate=0,
               rows_per_gate=1, y_out=None):
    """out = epilogue(x @ w^T + bias). x bf16 [M,K], w bf16 [N,K]."""
    M, K = x.shape
    N = w.shape[0]
    C2 = act_out if act_out is not None else y_out
    gemm(NT, epi, x, w, M, N, K, out, x.stride(0), w.stride(0), out.stride(0) if out is not None else 0,
         C2=C2, ldc2=C2.stride(0) if C2 is not None else 0, R=R, ldr=R.stride(0) if R is not None else 0,
         bias=bias, gate=gate, ldgate=ldgate, rows_per_gate=rows_per_gate)


def linear_dgrad(dy, w, dx, epi=EPI_BF16, R=None):
    """dx = epilogue(dy @ w). dy bf16 [M,N], w bf16 [N,K] -> dx [M,K]."""
    M, N = dy.shape
    K = w.shape[1]
    gemm(NN, epi, dy, w, M, K, N, dx, dy.stride(0), w.stride(0), dx.stride(0), R=R,
         ldr=R.stride(0) if R is not None else 0)


def linear_wgrad(dy, x, dw, dbias=None, accumulate=False, split_k=1):
    """dw f32 [N,K] (+)= dy^T x ; dbias f32 [N] (+)= colsum(dy). dy bf16 [M,N], x bf16 [M,K]."""
    Mtok, N = dy.shape
    K = x.shape[1]
    epi = EPI_ATOMIC_F32 if split_k > 1 else EPI_F32
    gemm(TN, epi, dy, x, N, K, Mtok, dw, dy.stride(0), x.stride(0), dw.stride(0), dbias=dbias,
         accumulate=accumulate, split_k=split_k)


def attention_fwd(qkv, o, lse, B, T, H, hd):
    L = _lib.load()
    _lib.check(L.reed_attention_fwd(_p(qkv), _p(o), _p(lse), B, T, H, hd, _stream()), "attention_fwd")


def attention_bwd(qkv, o, do, lse, dqkv, B, T, H, hd):
    L = _lib.load()
    _lib.check(L.reed_attention_bwd(_p(qkv), _p(o), _p(do), _p(lse), _p(dqkv), B, T, H, hd, _stream()),
               "attention_bwd")
