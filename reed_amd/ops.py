"""Thin Python wrappers over the C ABI (include/reed_hip.h): tensors -> raw device pointers.

PyTorch is used for device memory and streams only; every function here launches hand-written
HIP kernels on torch's current stream and raises if the library is unavailable.
Pointer arguments may be torch tensors or raw integer device addresses (for sub-views of arenas).
"""
import ctypes

import math

import torch

from . import _lib

NT, NN, TN = 0, 1, 2
TN_TALL, TN_WIDE = 3, 4   # TN on the 256x128 / 128x256 tile of csrc/gemm_tn.hip (weight gradients)
(EPI_BF16, EPI_GELU, EPI_SILU, EPI_GATE_RES, EPI_DGELU, EPI_DSILU, EPI_F32, EPI_ADDF32_RB, EPI_ATOMIC_F32, EPI_QGELU,
 EPI_RES_BF16, EPI_GELU_ERF, EPI_LS_RES, EPI_BF16_DOT, EPI_GELU_G, EPI_SILU_G, EPI_MUL) = range(17)


def _p(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def _stream():
    # the raw hipStream_t of torch's current stream: the C entry point, not torch.cuda.current_stream() (which builds a
    # Stream object through several Python layers: ~8 us x 500 launches per step of host time)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


# Which build of the library the calls go to: "bf16" (libreed_hip.so) or "fp16" (libreed_hip_f16.so, the same sources with
# IEEE-half operands: the sampling path).  The engine selects it per forward from the model's `precision`; torch tensors
# handed to the kernels must then be torch.float16 where the bf16 build takes torch.bfloat16 (half_dtype()).
_PRECISION = "bf16"


def use(precision):
    """Select the library build for the following calls; returns the previous selection."""
    global _PRECISION
    if precision not in _lib.PRECISIONS:
        raise ValueError(f"precision {precision!r}: one of {_lib.PRECISIONS}")
    prev, _PRECISION = _PRECISION, precision
    return prev


_HALF_DTYPE = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}
_PREC_OF_DTYPE = {v: k for k, v in _HALF_DTYPE.items()}


def half_dtype(precision=None):
    """torch dtype of the build's operand / activation arrays ("the 16-bit type"; float32 in the fp32-operand build)."""
    return _HALF_DTYPE[precision or _PRECISION]


def precision_of(dtype):
    """The build whose operand type is `dtype` (a projector output's dtype names the build that produced it)."""
    return _PREC_OF_DTYPE[dtype]


def _call(name, *args):
    L = _lib.load(_PRECISION)
    _lib.check(getattr(L, name)(*args), name, L)


def require_cuda(t, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"reed_amd: {name} is on {t.device}; the SiT hot path runs only on an AMD GPU through "
                           "libreed_hip.so (no CPU / PyTorch fallback)")


# ---------------- GEMM ----------------
# Measurement hook (bench.py only): gemm_probe(layout, epi, M, N, K) -> key or None; for a key the launch is bracketed
# by an event pair on the launch stream and (key, start, end) appended to gemm_probe_log. None (the default) costs one
# attribute test per launch.
gemm_probe = None
gemm_probe_log = []
wgrad_group_probe = None   # the same for the grouped weight-gradient launch: wgrad_group_probe(tokens, [(n_out, k_in), ...])


def gemm(layout, epi, P, Q, M, N, K, C, ldp, ldq, ldc, C2=None, ldc2=0, R=None, ldr=0, bias=None,
         gate=None, ldgate=0, rows_per_gate=1, dbias=None, accumulate=False, split_k=1,
         slab_stride=0):
    key = gemm_probe(layout, epi, M, N, K) if gemm_probe is not None else None
    if key is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _call("reed_gemm", layout, epi, _p(P), ldp, _p(Q), ldq, M, N, K, _p(C), ldc, _p(C2), ldc2,
          _p(R), ldr, _p(bias), _p(gate), ldgate, rows_per_gate, _p(dbias), int(accumulate), split_k,
          slab_stride, _stream())
    if key is not None:
        e1.record()
        gemm_probe_log.append((key, e0, e1))


def linear_fwd(x, w, bias, out, epi=EPI_BF16, act_out=None, R=None, gate=None, ldgate=0,
               rows_per_gate=1, y_out=None, M=None, N=None, K=None, ldx=None, ldw=None, ldo=None):
    """out = epilogue(x @ w^T + bias). x bf16 [M,K], w bf16 [N,K]."""
    if M is None:
        M, K = x.shape
        N = w.shape[0]
        ldx, ldw = x.stride(0), w.stride(0)
        ldo = out.stride(0) if out is not None else 0
    C2 = act_out if act_out is not None else y_out
    gemm(NT, epi, x, w, M, N, K, out, ldx, ldw, ldo,
         C2=C2, ldc2=(N if isinstance(C2, int) else C2.stride(0)) if C2 is not None else 0,
         R=R, ldr=(N if isinstance(R, int) else R.stride(0)) if R is not None else 0,
         bias=bias, gate=gate, ldgate=ldgate, rows_per_gate=rows_per_gate)


def linear_dgrad(dy, w, dx, epi=EPI_BF16, R=None, M=None, N=None, K=None, ldw=None):
    """dx = epilogue(dy @ w). dy bf16 [M,N], w bf16 [N,K] -> dx [M,K]."""
    if M is None:
        M, N = dy.shape
        K = w.shape[1]
        ldw = w.stride(0)
    gemm(NN, epi, dy, w, M, K, N, dx, N, ldw, K, R=R, ldr=K if R is not None else 0)


def linear_wgrad(dy, x, dw, dbias=None, accumulate=False, split_k=1, Mtok=None, N=None, K=None, ws=None, lay=TN):
    """dw f32 [N,K] (+)= dy^T x ; dbias f32 [N] (+)= colsum(dy). dy bf16 [Mtok,N], x bf16 [Mtok,K].
    split_k>1 goes through slabs in `ws` (f32, >= split_k*(N*K+N) floats) and a deterministic reduce."""
    if Mtok is None:
        Mtok, N = dy.shape
        K = x.shape[1]
    if split_k <= 1:
        gemm(lay, EPI_F32, dy, x, N, K, Mtok, dw, N, K, K, dbias=dbias, accumulate=accumulate)
        return
    slab = N * K
    stride = slab + N if dbias is not None else slab   # slab z = [dW slice z | dbias slice z]
    if ws is None:
        ws = torch.empty(split_k * stride, dtype=torch.float32, device=dw.device if torch.is_tensor(dw) else "cuda")
    wsp = ws if isinstance(ws, int) else ws.data_ptr()
    gemm(lay, EPI_F32, dy, x, N, K, Mtok, wsp, N, K, K, dbias=wsp + 4 * slab if dbias is not None else None,
         accumulate=False, split_k=split_k, slab_stride=stride)
    # the launcher may round the split count down; it reports nothing back, so recompute it the same way
    ksteps = (Mtok + 63) // 64
    per = (ksteps + split_k - 1) // split_k
    eff = (ksteps + per - 1) // per
    if dbias is not None and _p(dbias) == _p(dw) + 4 * slab:   # bias right behind its weight (the gradient arena): one pass
        reduce_slabs(wsp, stride, eff, dw, stride, accumulate)
        return
    reduce_slabs(wsp, stride, eff, dw, slab, accumulate)
    if dbias is not None:
        reduce_slabs(wsp + 4 * slab, stride, eff, dbias, N, accumulate)


def colsum_ws_floats(M, N):
    return ((M + 255) // 256) * N


def colsum_bf16(x, ld, ws, out, M, N, accumulate=False):
    _call("reed_colsum_bf16", _p(x), ld, _p(ws), _p(out), M, N, int(accumulate), _stream())


_FORCED_TILE = 0


def gemm_forced_tile():
    return _FORCED_TILE


def gemm_force_tile(tile):
    """0 = heuristic; 128 / 256 (eight waves) / 257 (four 128x128 waves, one tile per workgroup) / 258 (the same, persistent
    form wherever it applies) / 144 / 64 (the skinny one-wave tiles) = force that GEMM kernel
    where it applies; 259 = the heuristic plus the column split of csrc/gemm.hip (tests and A/B timing).  Set in both builds of the library."""
    global _FORCED_TILE
    _FORCED_TILE = int(tile)
    for prec in ("bf16", "fp16"):
        _lib.load(prec).reed_gemm_force_tile(int(tile))
    if "fp32" in _lib.loaded():
        _lib.load("fp32").reed_gemm_force_tile(int(tile))


WGRAD_SLOTS = 512  # resident 128x128 blocks: 256 CUs x 2 (64 KiB LDS, <=128 VGPRs... see gemm.hip launch bounds)
_CU_RESERVE = 0


def set_cu_reserve(n):
    """Plan the GEMM grids for n fewer CUs (RCCL's channels hold CUs while a gradient bucket is in flight)."""
    global _CU_RESERVE
    _CU_RESERVE = max(0, int(n))
    for prec in ("bf16", "fp16") + (("fp32",) if "fp32" in _lib.loaded() else ()):
        _lib.load(prec).reed_set_cu_reserve(_CU_RESERVE)


_COMM_FORMS = __import__("os").environ.get("REED_COMM_FORMS", "1") != "0"   # REED_COMM_FORMS=0: off from the start (A/B on a node)


def set_comm_forms(on):
    """Whether set_concurrent_comm(True) selects the kernel forms that degrade gracefully beside a collective (the default) or leaves
    the single-GPU forms in place (one workgroup per CU for its whole run: faster while no RCCL channel holds a CU, + 40-60 % per
    kernel while one does — profiles/r4_wgrad_under_cu_hog.txt, r4_kernels_under_cu_hog.txt).  Which wins depends on how long the
    buckets are in flight on the node at hand: TrainStep.plan_tuning measures both."""
    global _COMM_FORMS
    _COMM_FORMS = bool(on)


def comm_forms():
    return _COMM_FORMS


def cu_reserve():
    return _CU_RESERVE


def set_concurrent_comm(on):
    """Collectives run beside the GEMMs from now on (data-parallel training): the library keeps to kernels that degrade
    gracefully when RCCL's channels hold CUs (csrc/gemm256.hip:reed_set_concurrent_comm)."""
    for prec in ("bf16", "fp16") + (("fp32",) if "fp32" in _lib.loaded() else ()):
        _lib.load(prec).reed_set_concurrent_comm(1 if (on and _COMM_FORMS) else 0)


def wgrad_slots():
    """Workgroup slots the weight-gradient planning fills: 2 per CU the heuristics plan for."""
    if torch.cuda.is_available():
        return 2 * int(_lib.load().reed_planning_cus())
    return WGRAD_SLOTS - 2 * _CU_RESERVE
WGRAD_SPLIT_MAX = 8


WGRAD_TILES = ((TN, 128, 128, 1.0), (TN_TALL, 256, 128, 1.10), (TN_WIDE, 128, 256, 1.10))   # layout, rows, cols, rate


def wgrad_group_blocks(shapes):
    """Workgroups reed_wgrad_group launches for [(n_out, k_in), ...]: per problem the smaller of its 256x128 / 128x256 tile
    counts, each run padded to a multiple of 8 (csrc/gemm_tn.hip)."""
    tot = 0
    for n_out, k_in in shapes:
        if k_in % 128 or n_out % 16:
            return None
        tall = ((n_out + 255) // 256) * (k_in // 128)
        wide = ((n_out + 127) // 128) * (k_in // 256) if k_in % 256 == 0 else 1 << 30
        tot += (min(tall, wide) + 7) // 8 * 8
    return tot


def wgrad_group_fits(shapes, min_fill=0.85):
    """True when the block's weight gradients should go out as ONE launch without split-K: their tiles fill one round of
    workgroup slots to at least min_fill (SiT-XL/2: exactly 512 of 512; SiT-L/2 fills 384 = 75 % and measured 2 % slower
    than its wave-quantised split-K plan; smaller models are far below)."""
    import os
    if os.environ.get("REED_WGRAD_GROUP", "1") == "0" or _PRECISION == "fp32":   # the grouped launch is a 16-bit MFMA kernel
        return False
    n = wgrad_group_blocks(shapes)
    slots = wgrad_slots()
    return n is not None and len(shapes) <= 4 and min_fill * slots <= n <= slots


def wgrad_group_deal(shapes, cus=256, precision="bf16"):
    """How the one-workgroup-per-CU form of the grouped weight gradients deals [(n_out, k_in, has_bias), ...] over `cus` CUs
    (reed_wgrad_group_deal: host arithmetic, runs without a GPU).  Returns None where the form does not apply, else a list of 8
    lists (one per XCD) of items {p, mode, row, col, rows, cols, bias}: row / col / rows / cols in elements of dw; mode 0 = a
    256 x 256 tile, 6 = 384 x 128 with the bias gradient of its rows, 7 = 128 x 384, 8 = the bias gradient of `rows` rows only."""
    n = len(shapes)
    ip = ctypes.c_int * n
    items = (ctypes.c_uint * 256)()
    L = _lib.load(precision)
    T = L.reed_wgrad_group_deal(n, ctypes.cast(ip(*[int(q[0]) for q in shapes]), ctypes.c_void_p),
                                ctypes.cast(ip(*[int(q[1]) for q in shapes]), ctypes.c_void_p),
                                ctypes.cast(ip(*[int(bool(q[2])) for q in shapes]), ctypes.c_void_p), int(cus),
                                ctypes.cast(items, ctypes.c_void_p))
    if T <= 0:
        return None
    wpx, out = cus // 8, []
    for x in range(8):
        run = []
        for j in range(wpx):
            it = items[x * wpx + j]
            if it == 0xFFFFFFFF:
                continue
            mode, mu, nu = (it >> 2) & 15, (it >> 6) & 255, (it >> 14) & 255
            vm, vn = ((it >> 22) & 3) + 1, ((it >> 24) & 3) + 1
            rows = {0: 256, 6: 128 * vm, 7: 128, 8: 128 * vm}[mode]
            cols = {0: 256, 6: 128, 7: 128 * vn, 8: 0}[mode]
            run.append(dict(p=it & 3, mode=mode, row=128 * mu, col=128 * nu, rows=rows, cols=cols, bias=(it >> 26) & 1))
        out.append(run)
    assert sum(len(r) for r in out) == T
    return out


def wgrad_group(problems, tokens, accumulate=False):
    """problems: [(dy [tokens, n_out], x [tokens, k_in], dw f32 [n_out, k_in], dbias f32 [n_out] | None, n_out, k_in), ...]
    (tensors or raw device addresses) -> one launch of the 256x128 / 128x256 TN tiles, no split-K (reed_wgrad_group).
    Returns False, with nothing launched, when the device has too few workgroup slots for one round."""
    n = len(problems)
    vp, ip = ctypes.c_void_p * n, ctypes.c_int * n
    dy = vp(*[_p(q[0]) for q in problems])
    x = vp(*[_p(q[1]) for q in problems])
    dw = vp(*[_p(q[2]) for q in problems])
    db = vp(*[_p(q[3]) for q in problems])
    no = ip(*[int(q[4]) for q in problems])
    ki = ip(*[int(q[5]) for q in problems])
    cast = lambda a: ctypes.cast(a, ctypes.c_void_p)  # noqa: E731
    L = _lib.load(_PRECISION)
    key = wgrad_group_probe(tokens, [(int(q[4]), int(q[5])) for q in problems]) if wgrad_group_probe is not None else None
    if key is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = L.reed_wgrad_group(n, cast(dy), cast(x), cast(dw), cast(db), cast(no), cast(ki), int(tokens), int(accumulate),
                            _stream())
    if rc == 1002:
        return False
    _lib.check(rc, "reed_wgrad_group", L)
    if key is not None:
        e1.record()
        gemm_probe_log.append((key, e0, e1))
    return True


def plan_wgrad(Mtok, N, K):
    """(layout, split_k) for dw[N,K] = dy[Mtok,N]^T x[Mtok,K]: which TN kernel and how many K slices.
    Tiles: gemm.hip's 128x128 (64x64 per wave) or gemm_tn.hip's 256x128 / 128x256 (128x64 / 64x128 per wave: 0.375
    instead of 0.5 LDS fragment reads per MFMA, measured ~1.10x per flop — tools/wgrad_sweep.py); all keep two
    workgroups per CU, i.e. 512 resident blocks. Split-K is chosen for WAVE QUANTISATION: tiles*split should fill a
    whole number of rounds of those 512 blocks (SiT-XL/2 at b=256: fc1 162 tiles of 256x128 x 3, fc2 162 of 128x256
    x 3, qkv 126 of 256x128 x 4 with 3.6 % of the 14th tile row empty, proj 81 of 128x128 x 6), and among equally
    good choices the one whose K slice is closest to 256 K-tiles (16384 tokens) wins: long enough to amortise the
    prologue/epilogue + slab traffic, short enough that two rounds overlap their tails."""
    ktiles = (Mtok + 63) // 64
    slots = wgrad_slots()
    best, best_key = (TN, 1), None
    for lay, bm, bn, rate in WGRAD_TILES:
        if K % bn or N % 128:
            continue
        tiles = ((N + bm - 1) // bm) * (K // bn)
        fill = (N * K) / (tiles * bm * bn)          # rows of a ragged last tile row are computed and dropped
        for s in range(1, WGRAD_SPLIT_MAX + 1):
            if s > 1 and ktiles // s < 32:
                break
            blocks = tiles * s
            eff = blocks / (((blocks + slots - 1) // slots) * slots)
            dist = abs(math.log((ktiles / s) / 256.0))
            key = (-round(rate * fill * eff / 0.03), dist)   # 3 % buckets of estimated throughput, then slice length
            if best_key is None or key < best_key:
                best, best_key = (lay, s), key
    return best


def reduce_slabs(slabs, stride, n, out, count, accumulate=False):
    _call("reed_reduce_slabs", _p(slabs), stride, n, _p(out), count, int(accumulate), _stream())


# ---------------- row kernels ----------------
def ln_modulate_fwd(x, shift, scale, ldmod, h, mean, rstd, M, D, T, eps=1e-6):
    _call("reed_ln_modulate_fwd", _p(x), _p(shift), _p(scale), ldmod, _p(h), _p(mean), _p(rstd), M, D, T, eps, _stream())


def ln_modulate_bwd(dh, x, mean, rstd, scale, ldmod, dx, part, M, D, T):
    _call("reed_ln_modulate_bwd", _p(dh), _p(x), _p(mean), _p(rstd), _p(scale), ldmod, _p(dx), _p(part), M, D, T, _stream())


def ln_modulate_bwd_gate(dh, x, mean, rstd, scale, ldmod, dx, part, y, gate, ldgate, dy, part_g, part_dy, M, D, T):
    _call("reed_ln_modulate_bwd_gate", _p(dh), _p(x), _p(mean), _p(rstd), _p(scale), ldmod, _p(dx), _p(part), _p(y),
          _p(gate), ldgate, _p(dy), _p(part_g), _p(part_dy), M, D, T, _stream())


def gate_bwd(dx, y, gate, ldgate, dy, part, M, D, T, part_dy=None):
    _call("reed_gate_bwd", _p(dx), _p(y), _p(gate), ldgate, _p(dy), _p(part), _p(part_dy), M, D, T, _stream())


def transpose_bf16(src, dst, R, C):
    _call("reed_transpose_bf16", _p(src), _p(dst), R, C, _stream())


def rowsum_f32(part, R, out, N, accumulate=False, ws=None):
    _call("reed_rowsum_f32", _p(part), R, _p(ws), _p(out), N, int(accumulate), _stream())


def reduce_mod_parts(parts, dmod, lddmod, B, D, chunks):
    """parts: list of (ptr, stride, dmod column offset)."""
    n = len(parts)
    ptrs = (ctypes.c_void_p * n)(*[_p(p[0]) for p in parts])
    strides = (ctypes.c_int64 * n)(*[p[1] for p in parts])
    offs = (ctypes.c_int64 * n)(*[p[2] for p in parts])
    _call("reed_reduce_mod_parts", ptrs, strides, offs, n, _p(dmod), lddmod, B, D, chunks, _stream())


def token_mean_fwd(x, out, B, T, D):
    _call("reed_token_mean_fwd", _p(x), _p(out), B, T, D, _stream())


def token_mean_bwd(dmean, dx, B, T, D):
    _call("reed_token_mean_bwd", _p(dmean), _p(dx), B, T, D, _stream())


def cast_bf16(src, dst, n):
    _call("reed_cast_bf16", _p(src), _p(dst), n, _stream())


# ---------------- attention ----------------
def attention_fwd(qkv, o, lse, B, T, H, hd):
    _call("reed_attention_fwd", _p(qkv), _p(o), _p(lse), B, T, H, hd, _stream())


def attention_bwd(qkv, o, do, lse, dqkv, B, T, H, hd, ws=None):
    """ws: f32 workspace of attention_bwd_ws_floats(B, T, H) elements -> the persistent backward (delta by a row kernel in
    front); None -> the workspace-free kernels."""
    if ws is None:
        _call("reed_attention_bwd", _p(qkv), _p(o), _p(do), _p(lse), _p(dqkv), B, T, H, hd, _stream())
    else:
        _call("reed_attention_bwd_ws", _p(qkv), _p(o), _p(do), _p(lse), _p(dqkv), _p(ws), B, T, H, hd, _stream())


def attention_bwd_dp(qkv, do, lse, dpart, dqkv, ws, B, T, H, hd):
    """The persistent backward with delta from the partial dot products of gemm epilogue 13 (dpart f32 [H, 1|2, B*T])."""
    _call("reed_attention_bwd_dp", _p(qkv), _p(do), _p(lse), _p(dpart), _p(dqkv), _p(ws), B, T, H, hd, _stream())


def dgrad_with_head_dots(dy, w, dx, o, dpart, M, N, K, hd, wt=None):
    """dx bf16 [M, K] = dy [M, N] @ w [N, K] and dpart = per-row, per-head partial dot products of dx with o (epilogue 13).
    wt: the transposed copy w^T [K, N] — the same product as an NT GEMM (both operands k-contiguous).
    False (nothing launched) where this shape's GEMM kernel has no such epilogue: store plainly instead."""
    L = _lib.load(_PRECISION)
    if wt is not None:
        rc = L.reed_gemm(NT, EPI_BF16_DOT, _p(dy), N, _p(wt), N, M, K, N, _p(dx), K, _p(dpart), 0, _p(o), K, None, None, 0, hd, None, 0,
                         1, 0, _stream())
    else:
        rc = L.reed_gemm(NN, EPI_BF16_DOT, _p(dy), N, _p(w), K, M, K, N, _p(dx), K, _p(dpart), 0, _p(o), K, None, None, 0, hd, None, 0,
                         1, 0, _stream())
    if rc == 1002:
        return False
    _lib.check(rc, "reed_gemm", L)
    return True


def attention_bwd_ws_floats(B, T, H):
    return int(_lib.load(_PRECISION).reed_attention_bwd_ws_floats(B, T, H))


def qk_norm_fwd(qkv, qw, qb, kw, kb, out, stats, M, H, hd, eps=1e-5):
    _call("reed_qk_norm_fwd", _p(qkv), _p(qw), _p(qb), _p(kw), _p(kb), _p(out), _p(stats), M, H, hd, eps, _stream())


def qk_norm_part_floats(M, H, hd):
    return int(_lib.load().reed_qk_norm_bwd_part_floats(M, H, hd))


def qk_norm_bwd(dn, qkv, stats, qw, kw, dpre, part, M, H, hd):
    _call("reed_qk_norm_bwd", _p(dn), _p(qkv), _p(stats), _p(qw), _p(kw), _p(dpre), _p(part), M, H, hd, _stream())


# ---------------- embedders / final layer ----------------
def patch_embed_fwd(x, w, bias, pos, tokens, B, C, HW, P, D):
    _call("reed_patch_embed_fwd", _p(x), _p(w), _p(bias), _p(pos), _p(tokens), B, C, HW, P, D, _stream())


def patchify_bf16(x, out, B, C, HW, P, order):
    _call("reed_patchify_bf16", _p(x), _p(out), B, C, HW, P, order, _stream())


def smallk_ws_floats(Dw, KS):
    return int(_lib.load().reed_smallk_wgrad_ws_floats(Dw, KS))


def smallk_wgrad(wide, wide_is_f32, small, ws, out, colsum_wide, colsum_small, M, Dw, KS, layout, accumulate):
    _call("reed_smallk_wgrad", _p(wide), int(wide_is_f32), _p(small), _p(ws), _p(out), _p(colsum_wide),
          _p(colsum_small), M, Dw, KS, layout, int(accumulate), _stream())


def timestep_sinusoid(t, out, B, dim=256, max_period=10000.0):
    _call("reed_timestep_sinusoid", _p(t), _p(out), B, dim, max_period, _stream())


def label_cond(labels, drop, num_classes, table, t_emb, labels_out, c, silu_c, B, D, table_rows=None, err=None):
    if table_rows is None:
        table_rows = num_classes + 1
    _call("reed_label_cond", _p(labels), _p(drop), num_classes, table_rows, _p(table), _p(t_emb), _p(labels_out), _p(c),
          _p(silu_c), _p(err), B, D, _stream())


def label_cond_bwd(dsilu_c, c, labels_eff, dt_emb, dtable, B, D):
    _call("reed_label_cond_bwd", _p(dsilu_c), _p(c), _p(labels_eff), _p(dt_emb), _p(dtable), B, D, _stream())


def final_layer_fwd(x, shift, scale, ldmod, w, bias, out, mean, rstd, B, T, D, C, P, eps=1e-6):
    _call("reed_final_layer_fwd", _p(x), _p(shift), _p(scale), ldmod, _p(w), _p(bias), _p(out), _p(mean), _p(rstd),
          B, T, D, C, P, eps, _stream())


def final_layer_bwd_rows(dout, x, mean, rstd, shift, scale, ldmod, w, hbuf, dlin, dh, B, T, D, C, P):
    _call("reed_final_layer_bwd_rows", _p(dout), _p(x), _p(mean), _p(rstd), _p(shift), _p(scale), ldmod, _p(w),
          _p(hbuf), _p(dlin), _p(dh), B, T, D, C, P, _stream())


# ---------------- frozen CLIP image encoder (forward only) ----------------
def clip_im2col(img, out, B, S, P, Kp):
    _call("reed_clip_im2col", _p(img), _p(out), B, S, P, Kp, _stream())


def clip_tokens(patches, cls, pos, out, B, T, D):
    _call("reed_clip_tokens", _p(patches), _p(cls), _p(pos), _p(out), B, T, D, _stream())


def ln_affine_bf16(x, w, b, out, M, D, eps=1e-5):
    _call("reed_ln_affine_bf16", _p(x), _p(w), _p(b), _p(out), M, D, eps, _stream())


# ---------------- loss ----------------
def sample_posterior(moments, eps, out, B, half, scale, bias):
    _call("reed_sample_posterior", _p(moments), _p(eps), _p(out), B, half, float(scale), float(bias), _stream())


def interpolant(x, noise, t, xt, target, B, per, path_type):
    _call("reed_interpolant", _p(x), _p(noise), _p(t), _p(xt), _p(target), B, per, path_type, _stream())


def mse_fwd(out, target, loss, B, per):
    _call("reed_mse_fwd", _p(out), _p(target), _p(loss), B, per, _stream())


def mse_bwd(out, target, gscale, dout, B, per):
    _call("reed_mse_bwd", _p(out), _p(target), _p(gscale), _p(dout), B, per, _stream())


def cosine_fwd(zt, z, rowdot, loss, B, T, Z):
    _call("reed_cosine_fwd", _p(zt), _p(z), _p(rowdot), _p(loss), B, T, Z, _stream())


def cosine_bwd(zt, z, gscale, dzt, B, T, Z):
    _call("reed_cosine_bwd", _p(zt), _p(z), _p(gscale), _p(dzt), B, T, Z, _stream())


# ---------------- frozen ViT towers (encoders.py) ----------------
def ln_affine_f32(x, w, b, out, out_is_f32, M, D, ldo=None, eps=1e-6):
    _call("reed_ln_affine_f32", _p(x), _p(w), _p(b), _p(out), int(out_is_f32), M, D, D if ldo is None else ldo, eps, _stream())


def vit_tokens(patches, cls, pos, out, B, T, D, nprefix=None):
    """cls: f32 [nprefix, D] prefix rows (class token [+ register tokens]) or None."""
    if nprefix is None:
        nprefix = 0 if cls is None else 1
    _call("reed_vit_tokens", _p(patches), _p(cls), int(nprefix), _p(pos), _p(out), B, T, D, _stream())


def preprocess_image(raw_u8, out, B, R, S, mean, std, order):
    """image/train.py:53-74 on the device: uint8 [B,3,R,R] -> f32 [B,3,S,S]; order 0 = /255, bicubic, normalise (clip);
    order 1 = /255, normalise, bicubic (dinov2 / jepa); S == R: no resampling (mocov3 / mae)."""
    m = (ctypes.c_float * 3)(*mean)
    sd = (ctypes.c_float * 3)(*std)
    _call("reed_preprocess_image", _p(raw_u8), _p(out), B, R, S, ctypes.cast(m, ctypes.c_void_p), ctypes.cast(sd, ctypes.c_void_p),
          order, _stream())


# ---------------- optimiser ----------------
def grad_sqnorm(g, n, partial, nblocks):
    _call("reed_grad_sqnorm", _p(g), n, _p(partial), nblocks, _stream())


def clip_finalize(partial, nblocks, max_norm, norm_clip):
    _call("reed_clip_finalize", _p(partial), nblocks, max_norm, _p(norm_clip), _stream())


def clip_finalize_scaled(partial, nblocks, max_norm, norm_clip, scaler_state, growth=2.0, backoff=0.5, interval=2000):
    _call("reed_clip_finalize_scaled", _p(partial), nblocks, max_norm, _p(norm_clip), _p(scaler_state), growth, backoff,
          float(interval), _stream())


def adamw_ema(p, g, m, v, ema, shadow, n_train, n_total, norm_clip, lr, beta1, beta2, eps, wd, bc1, bc2, ema_decay,
              scaler_state=None):
    _call("reed_adamw_ema", _p(p), _p(g), _p(m), _p(v), _p(ema), _p(shadow), n_train, n_total, _p(norm_clip),
          _p(scaler_state), lr, beta1, beta2, eps, wd, bc1, bc2, ema_decay, _stream())


# ---------------- samplers ----------------
def sampler_input(x, out, n, dup):
    _call("reed_sampler_input", _p(x), _p(out), n, int(dup), _stream())


def sampler_update(x_cur, model_out, d_prev, d_store, x_next, n, cfg, cfg_scale, dt, w0, w1):
    _call("reed_sampler_update", _p(x_cur), _p(model_out), _p(d_prev), _p(d_store), _p(x_next), n, int(cfg),
          float(cfg_scale), float(dt), float(w0), float(w1), _stream())


def sde_update(x_cur, model_out, eps, x_next, n, cfg, cfg_scale, t_cur, dt, path_type, last):
    _call("reed_sde_update", _p(x_cur), _p(model_out), _p(eps), _p(x_next), n, int(cfg), float(cfg_scale),
          float(t_cur), float(dt), path_type, int(last), _stream())


# ---------------- SD-VAE decoder passes (csrc/vae.hip) ----------------
def groupnorm_stats(x, B, hw, C, G, eps, stats=None, ws=None, gamma=None, beta=None, table=None):
    """stats f32 [B, G, 2] = (mean, rstd) of x f32 [B, hw, C] per (image, group); table f32 [B, 3, C] = the per-channel
    (mean, rstd * gamma, beta) that conv_rows applies.  Returns the fp64 workspace for re-use."""
    L = _lib.load(_PRECISION)
    need = int(L.reed_groupnorm_ws_doubles(B, hw, C))
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float64, device=x.device)
    _call("reed_groupnorm_stats", _p(x), B, hw, C, G, float(eps), _p(gamma), _p(beta), _p(ws), _p(stats), _p(table), _stream())
    return ws


def conv_rows(x, out, B, Hi, Wi, C, taps, row0, nrows, kcols, ldo, table=None, silu=False, upsample=False):
    """out (operand type) [nrows, ldo] = the rows [row0, row0 + nrows) of the convolution's GEMM operand (reed_conv_rows)."""
    _call("reed_conv_rows", _p(x), _p(table), B, Hi, Wi, C, int(silu), int(upsample), taps, row0, nrows, kcols, _p(out), ldo,
          _stream())


def softmax_rows(s, lds, p, ldp, rows, cols, scale):
    _call("reed_softmax_rows", _p(s), lds, _p(p), ldp, rows, cols, float(scale), _stream())


def conv3x3(a, w, bias, out, ldc, B, Hi, Wi, C, N, upsample=False, accumulate=False):
    """out f32 [B*Ho*Wo, ldc] (+)= conv3x3(a operand-type NHWC [B, Hi, Wi, C]; w [N, 9C]) + bias f32 (reed_conv3x3, 16-bit builds)."""
    _call("reed_conv3x3", _p(a), _p(w), _p(bias), _p(out), ldc, B, Hi, Wi, C, N, int(upsample), int(accumulate), _stream())
