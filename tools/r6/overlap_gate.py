#!/usr/bin/env python
"""Round 6, verdict item 4 — the gate for running an HBM-bound kernel and an MFMA-bound kernel at the same time on partitioned CUs.
Half a local batch each (128 images): a plain dgrad GEMM <NN, plain store> of 128 images (dgrad fc1: [32768, 4608] x [4608, 1152])
in its persistent one-workgroup-per-CU form on 256 - R CUs (reed_set_cu_reserve(R)) on stream A, beside, on stream B, the row
kernel ln_mod_bwd2 (LayerNorm + modulate backward with the next gate's backward riding along, 18 B/element) or the attention forward
(persistent, grid = R workgroups through the same reserve) of the other 128 images — against the two run back to back on the
whole chip.  The GEMM's workgroups hold their CUs' whole register files, so the second kernel's workgroups can only land on the R
CUs the GEMM's grid left free.
usage: python tools/r6/overlap_gate.py  ->  the table of profiles/r6_overlap_gate.txt"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reed_amd import ops  # noqa: E402

dev = torch.device("cuda")
B, T, D, Hm, H, hd = 128, 256, 1152, 4608, 16, 72
M = B * T
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)  # noqa: E731
f32 = lambda *s: torch.randn(*s, device=dev)  # noqa: E731
# the GEMM: dx[M, D] = dy[M, Hm] W[Hm, D]
dy, w, dxg = bf(M, Hm), bf(Hm, D), torch.empty(M, D, dtype=torch.bfloat16, device=dev)
# the row kernel
dh, x, mean, rstd = bf(M, D), f32(M, D), f32(M), torch.rand(M, device=dev) + 0.5
mod = bf(B, 6 * D)
dx, part = f32(M, D), torch.empty(M // 16, 2, D, device=dev)
y, dyo, pg = bf(M, D), torch.empty(M, D, dtype=torch.bfloat16, device=dev), torch.empty(M // 16, D, device=dev)
# attention forward
qkv, o, lse = bf(B, T, 3, H, hd), torch.empty(B, T, H, hd, dtype=torch.bfloat16, device=dev), torch.empty(B, H, T, device=dev)

sA, sB = torch.cuda.Stream(), torch.cuda.Stream()


def gemm():
    ops.linear_dgrad(dy, w, dxg)


def row():
    ops.ln_modulate_bwd_gate(dh, x, mean, rstd, mod[:, D:], 6 * D, dx, part, y, mod[:, 2 * D:], 6 * D, dyo, pg, None, M, D, T)


def attn():
    ops.attention_fwd(qkv, o, lse, B, T, H, hd)


def timed(fn, n=20):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)


def pair(side, R):
    """GEMM on 256 - R CUs on stream A beside `side` on stream B; wall time from a common start to the later end (us), and each
    kernel's own time inside the pair."""
    cur = torch.cuda.current_stream()
    walls, tg, ts_ = [], [], []
    for _ in range(20):
        st, ga0, ga1, sb0, sb1, en = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        st.record(cur)
        sA.wait_event(st)
        sB.wait_event(st)
        with torch.cuda.stream(sA):
            ops.set_cu_reserve(R)
            ga0.record(sA)
            gemm()
            ga1.record(sA)
        with torch.cuda.stream(sB):
            ops.set_cu_reserve(256 - R)      # the attention forward's persistent grid = R workgroups (row kernels: plain grids)
            sb0.record(sB)
            side()
            sb1.record(sB)
        ops.set_cu_reserve(0)
        cur.wait_event(ga1)
        cur.wait_event(sb1)
        en.record(cur)
        torch.cuda.synchronize()
        walls.append(st.elapsed_time(en) * 1e3)
        tg.append(ga0.elapsed_time(ga1) * 1e3)
        ts_.append(sb0.elapsed_time(sb1) * 1e3)
    return statistics.median(walls), statistics.median(tg), statistics.median(ts_)


ops.gemm_force_tile(258)     # the persistent form wherever it applies (at 128 images the heuristic would pick the one-shot grid)
for f in (gemm, row, attn):
    for _ in range(5):
        f()
torch.cuda.synchronize()
flop = 2.0 * M * D * Hm
g0, r0, a0 = timed(gemm), timed(row), timed(attn)
print(f"alone on 256 CUs (us): GEMM <NN, plain> of 128 images {g0:.1f} ({flop / g0 / 1e6:.0f} TFLOP/s), ln_mod_bwd2 {r0:.1f} "
      f"({M * D * 18 / r0 / 1e6:.2f} TB/s at 18 B/element), attention forward {a0:.1f}")
for name, side, s0 in (("ln_mod_bwd2", row, r0), ("attn_fwd", attn, a0)):
    both = timed(lambda: (gemm(), side()))
    print(f"-- {name}: back to back on the whole chip {both:.1f} us (sum of the two alone {g0 + s0:.1f})")
    for R in (16, 32, 48, 64, 96):
        wall, tg, ts_ = pair(side, R)
        print(f"   R = {R:3d}: GEMM on {256 - R} CUs beside {name} on {R}: wall {wall:.1f} us ({100 * (1 - wall / both):+.1f} % against back to back), "
              f"GEMM {tg:.1f} us ({flop / tg / 1e6:.0f} TFLOP/s), {name} {ts_:.1f} us", flush=True)
ops.gemm_force_tile(0)
ops.set_cu_reserve(0)
