#!/usr/bin/env python
"""Round 6, verdict item 4, second gate — the schedule the engine would run: the local batch in two contiguous halves, a block's
GEMMs and attention of one half on 256 - R CUs (persistent grids through reed_set_cu_reserve) while the LayerNorm + modulate row
kernels of the other half run on the R CUs those grids leave free.  Part 1: what half-size GEMMs cost (bench.time_gemms at b = 256
against b = 128 with and without the reserve).  Part 2: NB forward blocks of SiT-XL/2 at b = 256 through the product entry points
(LN1, qkv, attention, proj + gate + residual, LN2, fc1 + GELU, fc2 + gate + residual) — serial as the engine runs them today against the
two-half schedule on two streams with events for every dependency; both write the same arrays (the halves are row ranges), the
outputs are compared bit for bit.
usage: python tools/r6/half_batch.py  ->  profiles/r6_half_batch_pipeline.txt"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from reed_amd import ops  # noqa: E402

dev = torch.device("cuda")
T, D, Hm, H, hd = 256, 1152, 4608, 16, 72

if "--no-table" not in sys.argv:
    for b, R in ((256, 0), (128, 0), (128, 32), (128, 48)):
        ops.set_cu_reserve(R)
        rows = bench.time_gemms(b, iters=10)[:8]
        ops.set_cu_reserve(0)
        print(f"b={b:3d} reserve {R:2d}: " + " | ".join(f"{r['kernel'].split()[0][0]}{r['kernel'].split()[1]}:{r['ms']:.3f}" for r in rows)
              + f" | sum of the 8 fwd / dgrad GEMMs {sum(r['ms'] for r in rows):.3f} ms" + (f" (x {256 // b} = {256 // b * sum(r['ms'] for r in rows):.3f})" if b != 256 else ""), flush=True)

B, NB = 256, 8
M = B * T
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)  # noqa: E731
e_bf = lambda *s: torch.empty(s, dtype=torch.bfloat16, device=dev)  # noqa: E731
e_f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
w_qkv, b_qkv, w_proj, b_proj, w1, b1, w2, b2 = bf(3 * D, D), bf(3 * D), bf(D, D), bf(D), bf(Hm, D), bf(Hm), bf(D, Hm), bf(D)
mod = bf(B, 6 * D)
x0 = torch.randn(M, D, device=dev)


def buffers():
    return dict(x=[x0.clone()] + [e_f(M, D) for _ in range(2)], xmid=e_f(M, D), h=e_bf(M, D), qkv=e_bf(M, 3 * D), o=e_bf(M, D), lse=e_f(B, H, T),
                y1=e_bf(M, D), h2=e_bf(M, D), a1=e_bf(M, Hm), u=e_bf(M, Hm), y2=e_bf(M, D), mean1=e_f(M), rstd1=e_f(M), mean2=e_f(M), rstd2=e_f(M))


hb = 2


def ops_of(bufs, b0, nb, xin, xout):
    """The seven launches of a block for the samples [b0, b0 + nb): row ranges of the same arrays."""
    r0, m = b0 * T, nb * T
    mp = mod.data_ptr() + hb * b0 * 6 * D
    P = lambda t, w: t.data_ptr() + t.element_size() * r0 * w  # noqa: E731
    lsep = bufs["lse"].data_ptr() + 4 * b0 * H * T
    v = lambda t: t.data_ptr() + t.element_size() * r0  # noqa: E731
    return dict(
        ln1=lambda: ops.ln_modulate_fwd(P(xin, D), mp, mp + hb * D, 6 * D, P(bufs["h"], D), v(bufs["mean1"]), v(bufs["rstd1"]), m, D, T),
        qkv=lambda: ops.gemm(ops.NT, ops.EPI_BF16, P(bufs["h"], D), w_qkv, m, 3 * D, D, P(bufs["qkv"], 3 * D), D, D, 3 * D, bias=b_qkv),
        att=lambda: ops.attention_fwd(P(bufs["qkv"], 3 * D), P(bufs["o"], D), lsep, nb, T, H, hd),
        proj=lambda: ops.gemm(ops.NT, ops.EPI_GATE_RES, P(bufs["o"], D), w_proj, m, D, D, P(bufs["xmid"], D), D, D, D, C2=P(bufs["y1"], D), ldc2=D,
                              R=P(xin, D), ldr=D, bias=b_proj, gate=mp + 2 * hb * D, ldgate=6 * D, rows_per_gate=T),
        ln2=lambda: ops.ln_modulate_fwd(P(bufs["xmid"], D), mp + 3 * hb * D, mp + 4 * hb * D, 6 * D, P(bufs["h2"], D), v(bufs["mean2"]), v(bufs["rstd2"]), m, D, T),
        fc1=lambda: ops.gemm(ops.NT, ops.EPI_GELU_G, P(bufs["h2"], D), w1, m, Hm, D, P(bufs["a1"], Hm), D, D, Hm, C2=P(bufs["u"], Hm), ldc2=Hm, bias=b1),
        fc2=lambda: ops.gemm(ops.NT, ops.EPI_GATE_RES, P(bufs["u"], Hm), w2, m, D, Hm, P(xout, D), Hm, Hm, D, C2=P(bufs["y2"], D), ldc2=D,
                             R=P(bufs["xmid"], D), ldr=D, bias=b2, gate=mp + 5 * hb * D, ldgate=6 * D, rows_per_gate=T),
    )


def serial(bufs):
    for i in range(NB):
        o = ops_of(bufs, 0, B, bufs["x"][i % 3], bufs["x"][(i + 1) % 3])
        for k in ("ln1", "qkv", "att", "proj", "ln2", "fc1", "fc2"):
            o[k]()
    return bufs["x"][NB % 3]


sG, sR = torch.cuda.Stream(), torch.cuda.Stream()


def pipelined(bufs, R, att_on_rows=False, force=0):
    """G stream: qkv, (attention,) proj, fc1, fc2 of half 0, then of half 1, block after block, on 256 - R CUs; R stream: the row
    kernels, each behind the event of what it reads.  att_on_rows: the attention forward on the row stream (R workgroups)."""
    cur = torch.cuda.current_stream()
    st = torch.cuda.Event()
    st.record(cur)
    sG.wait_event(st)
    sR.wait_event(st)
    nh = B // 2
    ev = {}

    def run(stream, key, fn, waits, reserve):
        with torch.cuda.stream(stream):
            for w in waits:
                if w in ev:
                    stream.wait_event(ev[w])
            ops.set_cu_reserve(reserve)
            fn()
            e = torch.cuda.Event()
            e.record(stream)
            ev[key] = e

    if force:
        ops.gemm_force_tile(force)
    for i in range(NB):
        hs = [ops_of(bufs, h * nh, nh, bufs["x"][i % 3], bufs["x"][(i + 1) % 3]) for h in (0, 1)]
        for h in (0, 1):
            run(sR, ("ln1", i, h), hs[h]["ln1"], [("fc2", i - 1, h)], 0)
        for h in (0, 1):
            run(sG, ("qkv", i, h), hs[h]["qkv"], [("ln1", i, h)], R)
            if att_on_rows:
                run(sR, ("att", i, h), hs[h]["att"], [("qkv", i, h)], 256 - R)
            else:
                run(sG, ("att", i, h), hs[h]["att"], [], R)
            run(sG, ("proj", i, h), hs[h]["proj"], [("att", i, h)], R)
            run(sR, ("ln2", i, h), hs[h]["ln2"], [("proj", i, h)], 0)
        for h in (0, 1):
            run(sG, ("fc1", i, h), hs[h]["fc1"], [("ln2", i, h)], R)
            run(sG, ("fc2", i, h), hs[h]["fc2"], [], R)
    ops.set_cu_reserve(0)
    ops.gemm_force_tile(0)
    cur.wait_stream(sG)
    cur.wait_stream(sR)
    return bufs["x"][NB % 3]


def timed(fn, n=8):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts), min(ts)


bs, bp = buffers(), buffers()
ref = serial(bs).clone()
torch.cuda.synchronize()
t_ser = timed(lambda: serial(bs))
print(f"{NB} forward blocks at b = {B}, serial (the engine today): median {t_ser[0]:.3f} ms, min {t_ser[1]:.3f} ms = {t_ser[0] / NB:.4f} ms per block", flush=True)
for R, att_rows, force in ((0, False, 0), (32, False, 0), (32, False, 258), (48, False, 258), (32, True, 258), (64, True, 258), (16, False, 258), (24, False, 258)):
    out = pipelined(bp, R, att_rows, force).clone()
    torch.cuda.synchronize()
    same = torch.equal(out, ref)
    t = timed(lambda: pipelined(bp, R, att_rows, force))
    print(f"two halves, reserve {R:2d}, attention on the {'row' if att_rows else 'GEMM'} stream, force_tile {force}: median {t[0]:.3f} ms, min {t[1]:.3f} ms "
          f"({100 * (t_ser[0] / t[0] - 1):+.1f} % against serial); output bit-identical: {same}", flush=True)
