#!/usr/bin/env python
"""bench.py — SiT-XL/2 ImageNet-256 train images/sec on MI355X (BASELINE.json metric), HIP path only.

  python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU over RCCL. Either the caller starts the ranks (`python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...`: RANK / LOCAL_RANK / WORLD_SIZE in the environment), or — with no WORLD_SIZE in the environment —
this process starts them itself: BEFORE any GPU call it spawns that very command as a child process (never exec), relays
rank 0's single JSON line to its own stdout and returns the child's exit code; with fewer than N visible devices it exits
non-zero with a one-line message.

A "step" is one full optimisation step of image/train.py on one batch of synthetic inputs already resident in HBM:
sample_posterior -> SILoss (interpolant, SiT-XL/2 forward with the 1024-d DINOv2-L-shaped projector tap, MSE +
cosine alignment) -> backward -> [RCCL gradient all-reduce, overlapped] -> clip_grad_norm_(1.0) -> AdamW -> EMA.
Global batch 256 is sharded over the N ranks (b = 256/N per GPU, train.py:263); scaling = "strong".
Rank 0 prints ONE JSON line; `roofline` is the dominant kernel (the bf16 MFMA GEMM) timed live with events on the
launch stream; `cpu_baseline` is the oracle (CPU restatement, fp32) timed on this box's host cores at N=1.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_IMG_STEP = 724.97e9   # 241.66 GFLOP forward x 3 (XL/2, z=1024 projector)
PEAK_BF16 = 2.5e15             # dense MFMA peak, MI355X_MICROARCH.md
PEAK_HBM = 8.0e12


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--global-batch", type=int, default=256)
    ap.add_argument("--model", type=str, default="SiT-XL/2")
    ap.add_argument("--z-dim", type=int, default=1024)
    ap.add_argument("--mixed-precision", choices=["bf16", "fp16", "fp32"], default="bf16",
                    help="operand type of the step (BASELINE's configuration is bf16; fp16 = the reference CLI's default, IEEE-half "
                         "operands with dynamic loss scaling)")
    ap.add_argument("--save-act-grad", choices=["auto", "0", "1"], default="auto",
                    help="the activation backward (engine.save_act_grad): auto = by tokens per GPU, 1 = saved derivative, 0 = recomputed")
    ap.add_argument("--dgrad-nt", choices=["auto", "0", "1"], default="auto",
                    help="the blocks' input gradients as NT GEMMs on a transposed copy of the weights (engine.dgrad_nt): auto = by tokens per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-table", action="store_true")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the C4 (text + image alignment), C5 (sampler) and N2 (frozen encoder) legs the N = 1 run appends")
    ap.add_argument("--no-vae-leg", action="store_true", help="skip the SD-VAE decode leg (SURVEY.md N4) the N = 1 run appends")
    ap.add_argument("--no-loss-vs-ref", action="store_true", help="skip the loss-vs-reference leg (the C2 fixture's 5 injected steps) the N = 1 run appends")
    ap.add_argument("--no-c3-leg", action="store_true", help="skip the b = 32 per-GPU leg (the 8-GPU shape) the N = 1 run appends")
    ap.add_argument("--launch-timeout", type=float, default=2400.0,
                    help="seconds the self-launcher lets its ranks run before it ends them (a hung collective must not hang the caller)")
    ap.add_argument("--tuned-timeout", type=float, default=300.0,
                    help="N > 1: seconds the second timed region (the tuned plan) may take before the plain-plan record is printed instead")
    ap.add_argument("--plain-file", type=str, default=None, help=argparse.SUPPRESS)        # launcher -> rank 0: where to leave the plain record
    ap.add_argument("--test-tuner-fail", choices=["hang", "raise"], default=None, help=argparse.SUPPRESS)   # tests: a tuner that fails on first contact
    return ap.parse_args(argv)


# Every run-time switch of the package (DESIGN.md §4 "Switches"); anything else that starts with REED_ is a typo or a switch of an
# older round, and a bench record measured under a plan nobody asked for is worse than no record: main() refuses to run.
KNOWN_ENV = {
    "REED_HIP_LIB",          # _lib.py: another build of the library (same-box A/B)
    "REED_WGRAD_W4",         # csrc/gemm256w.hip: 0 = gemm_tn.hip's grouped weight gradients everywhere, 1 = the four-wave form beside collectives too
    "REED_GEMM_COLSPLIT",    # csrc/gemm.hip: 1 = column split of the 2.25-round GEMMs (b = 32 per GPU: fc1 forward, fc2 input gradient); measured equal, off
    "REED_GEMM288",          # csrc/gemm288.hip: 1 = the heuristic may take the 256x288 kernel (measured equal to the 256x144 kernel: off)
    "REED_WGRAD_GROUP",      # ops.py: 0 = per-GEMM split-K weight gradients (the plan the goldens pin)
    "REED_WGRAD_STREAM",     # engine.py: 0 / 1 / auto — the weight gradients on a second stream
    "REED_OPT_OVERLAP",      # optim.py: 0 = the optimiser pass on the main stream
    "REED_COMM", "REED_COMM_ALGO", "REED_COMM_CUS", "REED_COMM_FORMS", "REED_ADA_GATHER", "REED_OPT_SHARD",   # the N > 1 plan (parallel.py, trainer.py)
    "REED_FORCE_REDUCER",    # the N > 1 code path at world 1
    "REED_BENCH_REHEARSE",   # bench.py: gloo rehearsal of an N > 1 run on fewer GPUs
    "REED_BENCH_TUNED",      # bench.py: 0 = no second (tuned) timed region
    "REED_ATTN_FWD_DBG", "REED_ATTN_KSP_DBG",   # diagnosis builds of csrc/attention.hip only (-DREED_ATTN_DIAG)
}


def check_env():
    bad = sorted(k for k in os.environ if k.startswith("REED_") and k not in KNOWN_ENV)
    if bad:
        raise SystemExit(f"bench.py: unknown switch(es) in the environment: {', '.join(bad)} (known: {', '.join(sorted(KNOWN_ENV))})")


def launcher_argv(args, argv, port):
    """The child command of the self-launcher: torch.distributed.run with one rank per GPU of this node, rendezvous on
    127.0.0.1 (the container hostname may not resolve), this script and its own arguments unchanged."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


# REED_BENCH_REHEARSE=gloo: a rehearsal of the N > 1 run on a box with fewer GPUs — the ranks share the visible devices
# (LOCAL_RANK modulo their count) and the collectives go over gloo (RCCL refuses two ranks on one device).  Everything else
# is the real path: self-launch, rendezvous, the reducer's buckets fired from backward, the barriers, MAX over ranks, the one
# JSON line.  The record says "rehearsal" and its value is not a measurement.
REHEARSE = os.environ.get("REED_BENCH_REHEARSE", "") == "gloo"


def _free_port():
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` with no ranks started by the caller (image/README.md:23 uses `accelerate launch`, which does
    the same): count the devices WITHOUT initialising the GPU (torch.cuda.device_count() does not, on this image), then run
    the ranks as a child process group and relay rank 0's JSON line.  Nothing here touches HIP, so the parent stays exec-safe."""
    import signal
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not REHEARSE:
        print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) visible on this node", file=sys.stderr, flush=True)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this image
    # rank 0 leaves the record of its FIRST timed region (the plain plan) here the moment it has it: if the ranks then have to
    # be abandoned (a tuner that hangs on first contact with RCCL and takes the in-process watchdog with it), this parent —
    # which never touched the GPU — ends the group and still prints that record
    plain_file = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"reed_bench_plain.{os.getpid()}.json")
    cmd = launcher_argv(args, list(argv) + ["--plain-file", plain_file], _free_port())
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True, text=True)
    timed_out = False
    try:
        out, _ = p.communicate(timeout=args.launch_timeout)
    except subprocess.TimeoutExpired:
        timed_out = True
        try:
            os.killpg(p.pid, signal.SIGKILL)     # exactly the process group this function started
        except ProcessLookupError:
            pass
        out, _ = p.communicate()
        print(f"bench.py: ranks still running after {args.launch_timeout:.0f} s - ended", file=sys.stderr, flush=True)
    line = None
    for ln in out.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    rc = 3 if timed_out else p.returncode
    if line is None and os.path.exists(plain_file):
        try:
            with open(plain_file) as f:
                rec = json.load(f)
            rec.setdefault("plans", {})["tuned"] = {"error": "ranks abandoned by the launcher" + (" (timeout)" if timed_out else f" (exit code {p.returncode})")}
            line, rc = json.dumps(rec), 0
        except Exception as e:
            print(f"bench.py: could not read the plain-plan record: {e!r}", file=sys.stderr, flush=True)
    try:
        os.unlink(plain_file)
    except OSError:
        pass
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        print("bench.py: the ranks exited without a result line", file=sys.stderr, flush=True)
        return 4
    return rc


def random_fill(model, seed):
    """Random-init weights of the architecture, adaLN/final layers non-zero so every block does real work
    (at the reference's exact init all gates are 0 and most gradients vanish: SURVEY.md §8d)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    A, L = model._arena, model._layout
    for name, (off, shp) in L.seg.items():
        if name == "pos_embed":
            continue
        n = 1
        for s in shp:
            n *= s
        v = A.master[off:off + n]
        if name.endswith("bias"):
            v.uniform_(-0.02, 0.02, generator=g)
        elif "embedding_table" in name:
            v.normal_(0, 0.02, generator=g)
        else:
            fan_in = n // shp[0]
            gain = 0.5 if ("adaLN" in name or name.startswith("final_layer.linear")) else 1.0
            a = gain * (3.0 / fan_in) ** 0.5
            v.uniform_(-a, a, generator=g)
    A.shadow_version = -1


# HBM-side bytes per launch of the dominant kernel (the block's grouped weight gradients, b = 256): NOT measured by this run — bench.py
# cannot read PMC counters. The number is the committed rocprofv3 --pmc pass named in TRAFFIC_SOURCE (separate FETCH_SIZE /
# WRITE_SIZE passes, FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md), of the same kernel build.
TRAFFIC_B256 = None
TRAFFIC_SOURCE = None
try:
    with open(os.path.join(ROOT, "profiles", "r6_traffic.json")) as _f:
        _t = json.load(_f)
        TRAFFIC_B256, TRAFFIC_SOURCE = _t["wgrad_group_b256_bytes_per_launch"], _t["source"]
except Exception:
    pass


def time_attention(b, H=16, hd=72, T=256, iters=10):
    """The attention kernels of a block at the step's shape, launched in isolation after the timed region: us per launch, algorithmic
    TFLOP/s (4 T^2 hd forward, 10 T^2 hd backward per head) and the HBM roofline that bounds them — forward reads qkv and writes o
    (4 M D bf16), the backward in the step's form (delta = rowsum(dO o) taken from the dO GEMM's epilogue 13) reads qkv and dO and
    writes dqkv (7 M D bf16) — as TB/s and the fraction of 8 TB/s."""
    from reed_amd import ops
    dev = torch.device("cuda")
    M, D = b * T, H * hd
    qkv = (torch.randn(M, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
    o = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
    do = torch.randn(M, D, device=dev).to(torch.bfloat16)
    dqkv = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(b, H, T, device=dev)
    ws = torch.empty(ops.attention_bwd_ws_floats(b, T, H), device=dev)
    dpart = torch.empty(H, 2 if hd == 72 else 1, M, device=dev)
    w = (torch.randn(D, D, device=dev) / D ** 0.5).to(torch.bfloat16)
    dy = torch.randn(M, D, device=dev).to(torch.bfloat16)
    ops.attention_fwd(qkv, o, lse, b, T, H, hd)
    step_form = bool(ops.dgrad_with_head_dots(dy, w, do, o, dpart, M, D, D, hd))

    def t(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3

    tf = t(lambda: ops.attention_fwd(qkv, o, lse, b, T, H, hd))
    if step_form:
        tb = t(lambda: ops.attention_bwd_dp(qkv, do, lse, dpart, dqkv, ws, b, T, H, hd))
    else:
        tb = t(lambda: ops.attention_bwd(qkv, o, do, lse, dqkv, b, T, H, hd, ws=ws))
    bf_, bb_ = 4.0 * M * D * 2, (7.0 if step_form else 8.0) * M * D * 2
    ff, fb = 4.0 * T * T * hd * b * H, 10.0 * T * T * hd * b * H
    rec = lambda sec, byt, fl: {"us": round(sec * 1e6, 1), "bytes": int(byt), "TB_per_s": round(byt / sec / 1e12, 3),  # noqa: E731
                                "frac_of_8_TB_per_s": round(byt / sec / 8e12, 4), "tflops": round(fl / sec / 1e12, 1)}
    return {"shape": {"batch": b, "tokens": T, "heads": H, "head_dim": hd}, "bound": "hbm", "peak_TB_per_s": 8.0,
            "forward": rec(tf, bf_, ff), "backward": rec(tb, bb_, fb),
            "backward_form": "delta from the dO GEMM's epilogue 13 (the step's form)" if step_form else "delta by a row kernel",
            "measured": "launched in isolation after the timed region; in-step launch times: profiles/r6_bench_n1_kernel_stats.csv"}


def time_gemms(b, D=1152, Hm=4608, T=256, iters=8, act_grad=True):
    """Per-shape timing of the block GEMMs through the SAME entry points and kernel-selection logic the engine uses
    (events on the launch stream): forward NT with fused epilogues, dgrad NN on the weight shadow, wgrad TN through
    ops.plan_wgrad (wave-quantised split-K slabs + deterministic reduce, bias gradient fused as a ones-MFMA)."""
    from reed_amd import ops
    dev = torch.device("cuda")
    M = b * T
    # the epilogues the step uses (engine.save_act_grad, round 5): fc1 saves GELU'(pre) and the fc2 dgrad multiplies by it
    epi_gelu, epi_dgelu = (ops.EPI_GELU_G, ops.EPI_MUL) if act_grad else (ops.EPI_GELU, ops.EPI_DGELU)
    bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).to(torch.bfloat16)  # noqa: E731
    x, w_qkv, w_proj, w1, w2 = bf(M, D), bf(3 * D, D), bf(D, D), bf(Hm, D), bf(D, Hm)
    big, big2 = bf(M, Hm), bf(M, Hm)
    o3 = bf(M, 3 * D)
    ybuf = bf(M, D)
    xo = torch.empty(M, D, device=dev)
    xi = torch.randn(M, D, device=dev)
    gate = bf(b, 6 * D)
    gw = torch.empty(Hm * D, device=dev)
    gb = torch.empty(Hm, device=dev)
    bias = bf(Hm)
    ws = torch.empty(8 * (Hm * D + Hm), device=dev)

    def wgrad(dy, xx, N, K, with_bias=True):
        lay, split = ops.plan_wgrad(M, N, K)
        ops.linear_wgrad(dy, xx, gw, dbias=gb if with_bias else None, split_k=split, Mtok=M, N=N, K=K, ws=ws, lay=lay)

    def dgrad(epi, dy, w, N, K, out, **kw):
        ops.gemm(ops.NN, epi, dy, w, M, K, N, out, N, K, K, **kw)

    cases = [
        ("fwd qkv  NT bias", 2.0 * M * 3 * D * D, lambda: ops.linear_fwd(x, w_qkv, bias[:3 * D], o3)),
        ("fwd proj NT gate+res", 2.0 * M * D * D, lambda: ops.linear_fwd(x, w_proj, bias[:D], xo, epi=ops.EPI_GATE_RES, R=xi, gate=gate, ldgate=6 * D, rows_per_gate=T, y_out=ybuf)),
        ("fwd fc1  NT gelu", 2.0 * M * Hm * D, lambda: ops.linear_fwd(x, w1, bias, big, epi=epi_gelu, act_out=big2)),
        ("fwd fc2  NT gate+res", 2.0 * M * Hm * D, lambda: ops.linear_fwd(big, w2, bias[:D], xo, epi=ops.EPI_GATE_RES, R=xi, gate=gate, ldgate=6 * D, rows_per_gate=T, y_out=ybuf)),
        ("dgrad fc2 NN dgelu", 2.0 * M * Hm * D, lambda: dgrad(epi_dgelu, x, w2, D, Hm, big, R=big2, ldr=Hm)),
        ("dgrad fc1 NN", 2.0 * M * Hm * D, lambda: dgrad(ops.EPI_BF16, big, w1, Hm, D, ybuf)),
        ("dgrad proj NN", 2.0 * M * D * D, lambda: dgrad(ops.EPI_BF16, x, w_proj, D, D, ybuf)),
        ("dgrad qkv NN", 2.0 * M * 3 * D * D, lambda: dgrad(ops.EPI_BF16, o3, w_qkv, 3 * D, D, ybuf)),
        ("wgrad fc1 TN +dbias", 2.0 * M * Hm * D, lambda: wgrad(big, x, Hm, D)),
        ("wgrad fc2 TN", 2.0 * M * Hm * D, lambda: wgrad(x, big, D, Hm, False)),
        ("wgrad proj TN", 2.0 * M * D * D, lambda: wgrad(x, ybuf, D, D, False)),
        ("wgrad qkv TN +dbias", 2.0 * M * 3 * D * D, lambda: wgrad(o3, x, 3 * D, D)),
    ]
    for _, _, fn in cases:   # warm the clocks and the code objects before timing anything
        fn()
    rows = []
    for name, flop, fn in cases:
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        rows.append({"kernel": name, "ms": round(ms, 4), "tflops": round(flop / ms / 1e9, 1)})
    return rows


def cpu_baseline():
    """The oracle (CPU restatement of the reference step, fp32, torch CPU ops) on this box's host cores, BASELINE.md §4
    protocol: SiT-XL/2 + 1024-d projector at B = 8 and SiT-S/2 + 768-d projector at B = 64; one warm-up step, then the
    median of 3 timed steps; all host threads torch gives the process."""
    from oracle import sit as osit
    from oracle import train_step as otrain
    threads = torch.get_num_threads()
    g = torch.Generator().manual_seed(0)

    def run(model, z, B, nrep):
        cfg = osit.make_config(model, z_dims=[z], z_types=["i"])
        P = osit.init_params(cfg)
        for k, v in P.items():
            if k == "pos_embed":
                continue
            fan = v[0].numel() if v.ndim > 1 else 1
            v.copy_((torch.rand(v.shape, generator=g) * 2 - 1) * (0.5 * (3.0 / max(fan, 1)) ** 0.5 if v.ndim > 1 else 0.02))
        tr = otrain.Trainer(P, cfg, ["dinov2"], [1.0], diffusion_warm_up_steps=0)

        def batch(n):
            return (torch.randn(n, 4, 32, 32, generator=g), torch.randint(0, 1000, (n,), generator=g),
                    [torch.randn(n, 256, z, generator=g)])
        tr.step(*batch(2))                      # warm-up (allocator, thread pool)
        ts = []
        for _ in range(nrep):
            x, y, zs = batch(B)
            t0 = time.time()
            tr.step(x, y, zs)
            ts.append(time.time() - t0)
        ts.sort()
        return B / ts[len(ts) // 2], ts

    ips_xl, ts_xl = run("SiT-XL/2", 1024, 8, 3)
    ips_s, ts_s = run("SiT-S/2", 768, 64, 3)
    return {"value": round(ips_xl, 4), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"oracle (CPU restatement of image/train.py step, fp32): SiT-XL/2 + 1024-d projector, B=8, median of 3 "
                      f"steps after a B=2 warm-up ({', '.join(f'{t:.1f}' for t in ts_xl)} s); SiT-S/2 + 768-d projector, "
                      f"B=64: {ips_s:.2f} images/sec ({', '.join(f'{t:.1f}' for t in ts_s)} s); every host thread torch gives the "
                      f"process ({threads}) — more than a B=8 step feeds (the survey's 8-core container ran the same step at 0.41 "
                      f"images/sec): a reported baseline, not a target",
            "s2_b64_images_per_sec": round(ips_s, 3)}


def dp_consistency(reducer, opt, model, world, b):
    """Data-parallel consistency after the timed steps: identical optimiser steps on all-reduced (and factor-gathered)
    gradients must leave every rank with the same parameters bit for bit — two checksums per rank, compared on rank 0."""
    import torch.distributed as dist
    if hasattr(opt, "sync_replicas"):
        opt.sync_replicas()     # (sharded update, REED_OPT_SHARD=1: the owners' master weights to every rank first)
    if hasattr(opt, "flush"):
        opt.flush()
    A = model._arena
    cs = torch.stack([A.master[:A.layout.n_train].double().sum(), A.master[:A.layout.n_train].double().abs().sum()])
    allcs = [torch.zeros_like(cs) for _ in range(world)]
    if world > 1:
        dist.all_gather(allcs, cs)
    else:
        allcs = [cs]
    allcs = torch.stack(allcs).cpu()
    L = A.layout
    ada = L.ada_rows * args_D(model) if reducer.ada_gather else 0
    dp = {"binding": reducer.binding, "adaln_factor_gather": bool(reducer.ada_gather),
          "ranks_bit_identical_params": bool((allcs == allcs[0]).all()),
          "param_checksum": float(allcs[0, 0]),
          "allreduce_bytes_per_step": int(4 * (L.n_train - ada - (L.ada_rows if reducer.ada_gather else 0))),
          "allgather_bytes_per_rank_per_step": int(2 * b * (L.ada_rows + args_D(model))) if reducer.ada_gather else 0}
    return dp


def args_D(model):
    return model.engine().D


def c3_leg(step, dev, z_dim, b=32, steps=10, warmup=3, dp_world1=True):
    """The per-GPU leg of C3 on this one GPU: the same model, optimiser and step at the local batch the 8-GPU run of the
    headline configuration sees (global batch 256 / 8), no communication.  Run after the main timed region, so the driver's
    record carries the 8-GPU-shape number; 8 x this is the upper bound of the 8-GPU run before any all-reduce cost."""
    g = torch.Generator(device=dev).manual_seed(4242)
    mean = torch.randn(b, 4, 32, 32, device=dev, generator=g) * 5.49
    moments = torch.cat([mean, torch.full_like(mean, 0.5)], dim=1)
    labels = torch.randint(0, 1000, (b,), device=dev, generator=g)
    zs = [torch.randn(b, 256, z_dim, device=dev, generator=g)]
    for _ in range(warmup):
        step(None, labels, zs, moments=moments)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step(None, labels, zs, moments=moments)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ips = b * steps / dt
    final_loss = float(res["loss"])
    # The data-parallel CODE PATH at this shape, for real, with one rank (VERDICT round 5, item 6 a): a second process runs this
    # script with REED_FORCE_REDUCER=1 at world 1 — process group on RCCL, the bucketed gradient all-reduce fired from backward on
    # the side stream, the adaLN factor gather, the kernel forms beside collectives: what every rank of the 8-GPU run executes,
    # minus the wire.  Nothing is skipped or left stale.  (The sharded optimiser pass cannot be part of it: optim.py:_shard_plan
    # cuts every update chunk into `world` pieces and is the replicated pass at world 1; its two-rank runs are tests/test_cli_gpu.py's.
    # Round 5's stand-in — every chunk cut to 1 / 8, the rest left stale — is gone with its knob in the optimiser.)
    dp1 = None
    if dp_world1:
        import subprocess
        env = dict(os.environ, REED_FORCE_REDUCER="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 2000))
        env["REED_BENCH_TUNED"] = "0"   # one timed region: the plain plan
        cmd = [sys.executable, os.path.abspath(__file__), "--global-batch", str(b), "--steps", str(steps), "--warmup", str(warmup),
               "--no-cpu-baseline", "--no-kernel-table", "--no-c3-leg", "--no-vae-leg", "--no-config-legs", "--no-loss-vs-ref"]
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
            rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            dp1 = {"images_per_sec_per_gpu": rec["value"], "ms_per_step": rec["ms_per_step"], "step_mfma_frac": rec["step_mfma_frac"],
                   "final_loss": rec["final_loss"], "plan": rec.get("plans", {}).get("plain", {}).get("plan"),
                   "data_parallel": rec.get("data_parallel"),
                   "note": "a second process: this script at world 1 with REED_FORCE_REDUCER=1 — the 8-GPU run's per-rank code path (RCCL "
                           "reducer, gradient buckets from backward on the side stream, adaLN factor gather, kernel forms beside "
                           "collectives, replicated optimiser pass) with one rank: the compute side of the 8-GPU run INCLUDING what "
                           "its collectives' kernels cost the GPU, before any wire time"}
        except Exception as e:  # noqa: BLE001 — a leg, not the number: report and go on
            dp1 = {"error": f"{type(e).__name__}: {e}"[:300]}
    return {"local_batch": b, "steps": steps, "warmup": warmup, "images_per_sec_per_gpu": round(ips, 2),
            "ms_per_step": round(dt / steps * 1e3, 3), "step_mfma_frac": round(ips * FLOP_PER_IMG_STEP / PEAK_BF16, 4),
            "x8_upper_bound_images_per_sec": round(8 * ips, 1), "final_loss": round(final_loss, 5),
            "data_parallel_code_path_at_world_1": dp1,
            "note": "one GPU, no gradient all-reduce: the compute side of the 8-GPU run (b = 256 / 8 per GPU)"}


def c4_leg(dev, b=32, steps=8, warmup=3):
    """BASELINE configs[3] (C4) on this one GPU at its 8-GPU per-GPU shape: SiT-XL/2 with TWO alignment projectors — CLIP-ViT-L
    image tokens (1024-d, tapped at block 8) and a pooled text embedding (768-d, tapped at block 10) — the REED loss with
    coefficients 1.0 / 0.5 (image/train.py `--enc-type clip-vit-L --text-embeds-dir ... --repa-coeff 1.0 0.5
    --encoder-depth-text 10`), local batch 32, no communication."""
    import copy
    from reed_amd.loss import SILoss
    from reed_amd.models.sit import SiT_models
    from reed_amd.optim import FusedAdamWEMA
    from reed_amd.trainer import TrainStep
    model = SiT_models["SiT-XL/2"](z_dims=[1024, 768], z_types=["i", "t"], encoder_depth=8, encoder_depth_text=10).to(dev).train()
    random_fill(model, 4321)
    ema = copy.deepcopy(model).requires_grad_(False).eval()
    opt = FusedAdamWEMA(model, ema, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.0, eps=1e-8, max_grad_norm=1.0)
    lf = SILoss(enc_names=["clip-vit-L", "text_embeds_open_clip"], loss_weights={"clip-vit-L": 1.0, "text_embeds_open_clip": 0.5})
    step = TrainStep(model, lf, opt, None, proj_coeff=1.0, diffusion_warm_up_steps=0)
    g = torch.Generator(device=dev).manual_seed(777)
    mean = torch.randn(b, 4, 32, 32, device=dev, generator=g) * 5.49
    moments = torch.cat([mean, torch.full_like(mean, 0.5)], dim=1)
    labels = torch.randint(0, 1000, (b,), device=dev, generator=g)
    zs = [torch.randn(b, 256, 1024, device=dev, generator=g), torch.randn(b, 768, device=dev, generator=g)]
    for _ in range(warmup):
        step(None, labels, zs, moments=moments)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step(None, labels, zs, moments=moments)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    opt.flush()
    return {"local_batch": b, "steps": steps, "images_per_sec_per_gpu": round(b * steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 3),
            "final_loss": round(float(res["loss"]), 5), "text_proj_loss": round(float(torch.as_tensor(res["text_proj_loss"]).detach()), 5),
            "note": "C4: image + text alignment (two projectors), one GPU at the per-GPU batch of the 8-GPU run, no all-reduce"}


def c5_leg(dev, n=32, num_steps=250):
    """BASELINE configs[4] (C5): generate.py's sampling loop on this GPU at its REAL length — SiT-XL/2, the 250-step Heun ODE
    sampler with classifier-free guidance over the whole interval (2 * 250 - 1 = 499 model evaluations, every one at batch 2n;
    image/samplers.py:46-104), fp16 operands (what generate.py uses under the reference's default --tf32).  One full run is
    timed (about 9 s), nothing is scaled."""
    from reed_amd.models.sit import SiT_models
    from reed_amd.samplers import euler_sampler
    model = SiT_models["SiT-XL/2"](z_dims=[1024], z_types=["i"], encoder_depth=8, use_cfg=True).to(dev).eval()
    random_fill(model, 1234)
    model.precision = "fp16"
    g = torch.Generator(device=dev).manual_seed(55)
    z = torch.randn(n, 4, 32, 32, device=dev, generator=g)
    y = torch.randint(0, 1000, (n,), device=dev, generator=g)
    euler_sampler(model, z, y, num_steps=2, heun=True, cfg_scale=1.5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = euler_sampler(model, z, y, num_steps=num_steps, heun=True, cfg_scale=1.5, guidance_low=0.0, guidance_high=1.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_eval = 2 * num_steps - 1
    per_eval = dt / n_eval
    flop_eval = 237.23e9 * 2 * n
    return {"n_per_gpu": n, "operands": "fp16", "num_steps": num_steps, "evaluations": n_eval, "seconds": round(dt, 3),
            "seconds_per_image": round(dt / n, 4), "ms_per_evaluation": round(per_eval * 1e3, 2),
            "images_per_sec_per_gpu_250_heun_cfg": round(n / dt, 3),
            "mfma_frac": round(flop_eval / per_eval / PEAK_BF16, 4), "finite": bool(torch.isfinite(out).all()),
            "note": f"C5: the whole {num_steps}-step Heun + CFG run ({n_eval} evaluations at batch {2 * n}) timed once; ranks sample "
                    "independent images, N GPUs give N times this"}


def n2_encoder_leg(dev, batch=64, reps=3):
    """SURVEY.md §8f N2 beside the headline: the frozen target encoder the reference runs on the raw images every step
    (image/train.py:351-357) — the CLIP ViT-L/14 tower of the C4 configuration, raw uint8 256x256 images -> preprocess_raw_image
    -> 257 tokens x 24 blocks -> patch-token features, random weights, bf16 operands, at the per-GPU batch of a 4-GPU run."""
    from reed_amd.encoders import CLIP_CONFIGS, ClipVisionEncoder
    cfg = CLIP_CONFIGS["L"]
    enc = ClipVisionEncoder(**cfg)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for n, p in enc.named_parameters():
            if p.ndim >= 2:
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * (3.0 / p[0].numel()) ** 0.5)
            elif "ln_" in n and n.endswith("weight"):
                p.fill_(1.0)
            else:
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * 0.05)
    enc = enc.to(dev).eval()
    raw = torch.randint(0, 256, (batch, 3, 256, 256), dtype=torch.uint8, device=dev)
    for _ in range(2):
        out = enc.encode_raw(raw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = enc.encode_raw(raw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    W, L, T = cfg["width"], cfg["layers"], (cfg["image"] // cfg["patch"]) ** 2 + 1
    flop = 2.0 * (L * (T * 12 * W * W + 2 * T * T * W) + (T - 1) * 3 * cfg["patch"] ** 2 * W)
    # the same tower at batch 256 (one-GPU feature extraction): 64 x 257 tokens leave a quarter-full 65th tile row, 256 x 257 a
    # full one more
    raw4 = torch.randint(0, 256, (256, 3, 256, 256), dtype=torch.uint8, device=dev)
    enc.encode_raw(raw4)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    enc.encode_raw(raw4)
    torch.cuda.synchronize()
    dt4 = time.perf_counter() - t1
    return {"images_per_sec": round(batch / dt, 1), "ms_per_batch": round(dt * 1e3, 2), "batch": batch, "operands": "bf16",
            "gflop_per_image": round(flop / 1e9, 1), "mfma_frac": round(flop * batch / dt / PEAK_BF16, 4),
            "batch256": {"images_per_sec": round(256 / dt4, 1), "mfma_frac": round(flop * 256 / dt4 / PEAK_BF16, 4)},
            "finite": bool(torch.isfinite(out.float()).all()),
            "note": "CLIP ViT-L/14 image tower (reed_amd/encoders.py) incl. the preprocessing pass; runs beside the train step when "
                    "features are not precomputed"}


def loss_vs_ref_leg(dev, steps=5, B=8):
    """BASELINE.json's second number, "loss-vs-ref delta": the C2 fixture tests/golden/xl2_c2.npz — SiT-XL/2 + 1024-d projector,
    B = 8, 5 optimiser steps on injected (x, t, noise, labels, zs) with deterministically filled weights; the per-step total loss
    (image/train.py:396-398) of the reference under bf16 autocast and in fp32, written by tools/gen_golden.py from the imported
    reference — run on the HIP path here, after the timed region.  Nothing of oracle/ or the reference is imported: the fixture
    holds the reference's numbers and reed_amd/detfill.py rebuilds weights and inputs."""
    import copy
    import numpy as np
    from reed_amd import detfill
    from reed_amd.loss import SILoss
    from reed_amd.models.sit import SiT_models
    from reed_amd.optim import FusedAdamWEMA
    g = np.load(os.path.join(ROOT, "tests", "golden", "xl2_c2.npz"))
    m = SiT_models["SiT-XL/2"](z_dims=[1024], z_types=["i"], encoder_depth=8)
    detfill.fill_model(m.state_dict(), base_seed=0)
    m = m.to(dev).train()
    m.engine().save_act_grad = True   # the activation backward of the timed b = 256 step (the engine's choice above 12288 tokens), not B = 8's
    ema = copy.deepcopy(m).requires_grad_(False).eval()
    opt = FusedAdamWEMA(m, ema, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.0, eps=1e-8, max_grad_norm=1.0)
    lf = SILoss(enc_names=["dinov2"], loss_weights={"dinov2": 1.0})
    losses = []
    for s_ in range(steps):
        x, noise, t, y, drop_u, zs = detfill.step_inputs(B, s_, [1024])
        m.force_drop_mask = drop_u < 0.1
        out = lf(m, x.to(dev), dict(y=y.to(dev)), zs=[z.to(dev) for z in zs], time_input=t, noises=noise)
        total = out["denoising_loss"].mean() + 0.5 * out["proj_loss"].mean()
        opt.zero_grad()
        total.backward()
        opt.step()
        losses.append(float(total))
    opt.flush()
    ref_b, ref_f = g["bf16.loss"][:steps], g["fp32.loss"][:steps]
    d_b, d_f = np.abs(np.array(losses) - ref_b), np.abs(np.array(losses) - ref_f)
    return {"max_abs_delta_bf16": float(round(d_b.max(), 7)), "bar": 1e-3, "steps": steps, "batch": B,
            "within_bar": bool((d_b <= 1e-3).all()),
            "hip_loss": [round(v, 6) for v in losses], "reference_bf16_autocast_loss": [round(float(v), 6) for v in ref_b],
            "abs_delta_per_step_bf16": [float(round(v, 7)) for v in d_b],
            "max_abs_delta_vs_fp32_reference": float(round(d_f.max(), 7)),
            "reference_own_bf16_vs_fp32_gap": float(round(np.abs(ref_b - ref_f).max(), 7)),
            "activation_backward": "saved derivative (the timed step's form)",
            "fixture": "tests/golden/xl2_c2.npz (tools/gen_golden.py, the imported reference; image/train.py:396-398)"}


def n4_vae_leg(dev, batch=8, reps=3):
    """SURVEY.md §8f N4 beside the headline: the SD-VAE decoder of generate.py / the previews (published sd-vae-ft configuration,
    random weights, 32x32 latents -> 256x256 images) on the HIP kernels, fp16 operands = what generate.py uses under the
    reference's default --tf32.  After the timed region; ~1 s including building the 49.5 M-parameter module."""
    from reed_amd.vae import SDVAEDecoder
    torch.manual_seed(0)
    dec = SDVAEDecoder()
    for p in dec.parameters():
        p.data.normal_(0, 0.02)
    dec = dec.to(dev)
    z = torch.randn(batch, 4, 32, 32, device=dev)
    dec.decode(z, precision="fp16")              # warm-up: weights packed, workspaces allocated
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        img = dec.decode(z, precision="fp16")
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return {"images_per_sec": round(batch / dt, 1), "ms_per_image": round(dt / batch * 1e3, 3), "operands": "fp16", "batch": batch,
            "finite": bool(torch.isfinite(img).all()),
            "note": "reed_amd/vae.py on csrc/vae.hip + csrc/conv.hip (implicit-GEMM 3x3 convolutions); parity unpinned vs diffusers"}


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    check_env()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))
    # stdout carries exactly ONE line, the JSON record: C libraries write there too (RCCL prints a version banner on stdout
    # at communicator creation whatever NCCL_DEBUG_FILE says), so file descriptor 1 points at stderr until that line is printed
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} in the environment (unset it to let bench.py start "
                         f"its own ranks, or launch with torch.distributed.run --nproc-per-node {args.gpus})")
    if REHEARSE:
        local_rank %= max(1, torch.cuda.device_count())
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} has no GPU (LOCAL_RANK {local_rank}, {torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    # REED_FORCE_REDUCER=1 drives the whole multi-GPU code path (process group, RCCL reducer, barriers, MAX over ranks)
    # with a single rank: the only rehearsal of it a one-GPU box allows
    use_dist = world > 1 or os.environ.get("REED_FORCE_REDUCER", "0") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # this image exports NCCL_DEBUG=VERSION, which makes RCCL print its banner on STDOUT: keep stdout to the one JSON line
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/reed_rccl.%h.%p.log")
        if REHEARSE:
            os.environ["REED_COMM"] = "torch"
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import copy
    from reed_amd.loss import SILoss
    from reed_amd.models.sit import SiT_models
    from reed_amd.optim import FusedAdamWEMA
    from reed_amd.parallel import GradReducer, shard_batch, rank_seed
    from reed_amd.trainer import TrainStep

    b = shard_batch(args.global_batch, world)
    torch.manual_seed(rank_seed(0, rank))
    model = SiT_models[args.model](z_dims=[args.z_dim], z_types=["i"], encoder_depth=8).to(dev).train()
    model.precision = args.mixed_precision
    random_fill(model, 1234)  # identical on every rank (same seed); broadcast below anyway
    model.engine().save_act_grad = {"auto": None, "0": False, "1": True}[args.save_act_grad]
    model.engine().dgrad_nt = {"auto": None, "0": False, "1": True}[args.dgrad_nt]
    ema = copy.deepcopy(model).requires_grad_(False).eval()
    opt = FusedAdamWEMA(model, ema, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.0, eps=1e-8, max_grad_norm=1.0)
    reducer = None
    if world > 1 or os.environ.get("REED_FORCE_REDUCER", "0") == "1":
        reducer = GradReducer(model, rank, world)
        reducer.broadcast_params(0)
    loss_fn = SILoss(enc_names=["dinov2-vit-l"], loss_weights={"dinov2-vit-l": 1.0})
    # Data-parallel runs time the PLAIN plan first — torch's RCCL binding, all-reduce buckets, no CU reserve, replicated
    # optimiser pass, nothing measured at run time — and only then, as a second timed region inside a watchdog, the TUNED plan
    # (run-time measurement of the CU reserve and of the sharded optimiser pass: TrainStep.plan_tuning).  The tuners have never
    # run over RCCL with more than one rank: a failure or a hang there must not cost the number.  `value` is the better of the
    # two regions, both are reported under "plans".  A caller that fixes REED_COMM_CUS / REED_OPT_SHARD / REED_COMM_ALGO gets
    # exactly that plan in the first region and no second one.
    user_fixed = any(k in os.environ for k in ("REED_COMM_CUS", "REED_OPT_SHARD", "REED_COMM_ALGO"))
    want_tuned = reducer is not None and not user_fixed and os.environ.get("REED_BENCH_TUNED", "1") != "0"
    step = TrainStep(model, loss_fn, opt, reducer, proj_coeff=0.5, diffusion_warm_up_steps=0)

    g = torch.Generator(device=dev).manual_seed(100 + rank)
    mean = torch.randn(b, 4, 32, 32, device=dev, generator=g) * 5.49
    moments = torch.cat([mean, torch.full_like(mean, 0.5)], dim=1)
    labels = torch.randint(0, 1000, (b,), device=dev, generator=g)
    zs = [torch.randn(b, 256, args.z_dim, device=dev, generator=g)]

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    from reed_amd import ops
    T_TOK = 256

    def probe(tokens, shapes):
        # the kernel with the largest share of the step (rocprofv3 --stats, profiles/): the four weight gradients of a transformer
        # block in one launch (csrc/gemm256w.hip's four-wave TN form, or csrc/gemm_tn.hip's); key = its algorithmic flop
        if tokens == b * T_TOK and len(shapes) == 4:
            return 2.0 * tokens * sum(n * k for n, k in shapes)
        return None

    def timed_region(n_warm):
        """n_warm untimed steps, then EXACTLY args.steps steps between barrier + synchronize on both sides; MAX over ranks."""
        res = None
        for _ in range(n_warm):
            res = step(None, labels, zs, moments=moments)
        barrier()
        if rank == 0:
            ops.wgrad_group_probe = probe   # event pairs on the launch stream around every launch of the dominant kernel
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = step(None, labels, zs, moments=moments)
        barrier()
        dt = time.perf_counter() - t0
        ops.wgrad_group_probe = None
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        if use_dist:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dom = {}
        for key, e0, e1 in ops.gemm_probe_log:
            dom.setdefault(key, []).append(e0.elapsed_time(e1))
        ops.gemm_probe_log.clear()
        return {"dt": float(tmax.item()), "loss": float(res["loss"]), "dom": dom}

    def plan_record(reg, what):
        return {"plan": what, "images_per_sec": round(args.global_batch * args.steps / reg["dt"], 2),
                "ms_per_step": round(reg["dt"] / args.steps * 1e3, 3), "final_loss": round(reg["loss"], 5)}

    def plan_of_step():
        return (f"{getattr(reducer, 'binding', None)} binding, {getattr(reducer, 'algo', None)} buckets, CU reserve {step.cu_reserve}, "
                f"kernel forms beside collectives {'on' if getattr(step, 'comm_forms', True) else 'off'}, "
                f"{'sharded' if getattr(opt, '_shard', False) else 'replicated'} optimiser pass") if reducer is not None else "single GPU"

    # W untimed warm-up steps — more when the caller asked for a run-time measurement through the environment (trainer.py): that
    # measurement (one settling + three timed steps per candidate) stays out of the timed region
    plain = timed_region(max(args.warmup, step.tune_steps_left() + 1 if step.tune_steps_left() else 0))
    plans = {"plain": plan_record(plain, plan_of_step())}
    chosen = plain

    def base_record(reg):
        ips = args.global_batch * args.steps / reg["dt"]
        return {
            "metric": "SiT-XL/2 ImageNet-256 train images/sec", "value": round(ips, 2), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(reg["dt"] / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": {"bf16": "bf16", "fp16": "f16", "fp32": "f32"}[args.mixed_precision],
            "data": "synthetic (random ImageNet-256 latents / DINOv2-L-shaped features; random-init weights)" +
                    (" — REHEARSAL: ranks share the visible GPUs, collectives over gloo; not a measurement" if REHEARSE else ""),
            "config": {"workload": f"C{'2' if world == 1 else '3'}: {args.model} + {args.z_dim}-d alignment projector (REED loss), "
                                   f"global batch {args.global_batch} (b={b}/GPU), full train step "
                                   "(fwd+bwd+clip+AdamW+EMA), bf16 MFMA / fp32 master",
                       "global_batch": args.global_batch, "local_batch": b, "parallelism": f"dp{world}"},
            "final_loss": round(reg["loss"], 5),
            # the whole step against the MFMA roofline: images/s/GPU x 724.97 GFLOP / 2.5 PFLOP/s (the headline efficiency)
            "step_mfma_frac": round(ips / world * FLOP_PER_IMG_STEP / PEAK_BF16, 4),
        }

    # ---- the watchdogs (timer threads).  The first bounds the TUNED region only: if it fires, rank 0 prints the plain-plan record.  A
    # second, longer one is armed for the diagnosis / teardown of a data-parallel run (the collectives of dp_consistency, the final
    # barrier, destroy_process_group): if it fires before rank 0 has printed its line, rank 0 prints the record of the plan already
    # chosen.  Either way every rank then leaves with os._exit(0) (no exec, no teardown of a communicator that may be wedged)
    import threading
    wd_lock = threading.Lock()
    wd_state = {"printed": False, "phase": "tuned plan"}

    def emit(rec):
        """rank 0: the ONE JSON line on the real stdout."""
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(rec), flush=True)
        os.dup2(2, 1)

    def wd_fire():
        with wd_lock:
            if rank == 0 and not wd_state["printed"]:
                rec = base_record(plain)
                rec["plans"] = dict(plans, tuned=plans.get("tuned") or {"error": f"timeout in: {wd_state['phase']}"})
                rec["plan_in_value"] = "plain"
                rec["watchdog_fired"] = wd_state["phase"]
                emit(rec)
                wd_state["printed"] = True
            print(f"[bench.py] rank {rank}: watchdog fired in: {wd_state['phase']} - leaving", file=sys.stderr, flush=True)
            os._exit(0)

    watchdog = None
    if use_dist:
        if rank == 0 and args.plain_file:
            try:
                rec = base_record(plain)
                rec["plans"], rec["plan_in_value"] = dict(plans), "plain"
                with open(args.plain_file, "w") as f:
                    json.dump(rec, f)
            except OSError:
                pass
    if want_tuned:
        # the watchdog bounds the tuned region only: armed here, cancelled when that region has returned
        watchdog = threading.Timer(args.tuned_timeout, wd_fire)
        watchdog.daemon = True
        watchdog.start()
        try:
            fail = args.test_tuner_fail or ""   # tests: a tuner that hangs / raises on first contact
            step.plan_tuning("auto", "auto" if world > 1 else None, None)
            if fail == "hang":
                while True:
                    time.sleep(1.0)
            if fail == "raise":
                raise RuntimeError("--test-tuner-fail raise")
            tuned = timed_region(step.tune_steps_left() + 1)
            watchdog.cancel()
            plans["tuned"] = plan_record(tuned, plan_of_step())
            # `value` is the plain region unless the tuner kept a DIFFERENT plan and that plan measured faster: two regions of one
            # plan are two samples of it, and their maximum would bias the number upwards
            if plans["tuned"]["plan"] != plans["plain"]["plan"] and tuned["dt"] < plain["dt"]:
                chosen = tuned
        except Exception as e:   # the plain record stands; the ranks may no longer agree on anything: leave without collectives
            plans["tuned"] = {"error": repr(e)}
            if rank == 0:
                with wd_lock:
                    rec = base_record(plain)
                    rec["plans"], rec["plan_in_value"] = plans, "plain"
                    emit(rec)
                    wd_state["printed"] = True
            print(f"[bench.py] rank {rank}: tuned plan failed ({e!r}) - leaving with the plain record", file=sys.stderr, flush=True)
            os._exit(0)
    wd_state["phase"] = "diagnosis / teardown"

    def wd_fire_late():
        with wd_lock:
            if rank == 0 and not wd_state["printed"]:
                rec = base_record(chosen)
                rec["plans"] = plans
                rec["plan_in_value"] = "tuned" if chosen is not plain else "plain"
                rec["watchdog_fired"] = wd_state["phase"]
                emit(rec)
                wd_state["printed"] = True
            print(f"[bench.py] rank {rank}: watchdog fired in: {wd_state['phase']} - leaving", file=sys.stderr, flush=True)
            os._exit(0)

    watchdog_late = None
    if use_dist:
        watchdog_late = threading.Timer(max(600.0, 2.0 * args.tuned_timeout), wd_fire_late)
        watchdog_late.daemon = True
        watchdog_late.start()
    dt = chosen["dt"]
    loss_val = chosen["loss"]
    dom = chosen["dom"]
    model.engine().check_errors()

    # ---- data-parallel diagnosis (after the timed region): bucket plan, bytes, and with a reducer two extra steps with an
    # event pair around every bucket reduction (comm stream) and around reducer.sync() (compute stream: what the step waits)
    L = model._layout
    bk = L.buckets()
    dp = {"world": world, "buckets": len(bk), "gradient_bytes": int(4 * L.n_train),
          "largest_bucket_bytes": int(4 * max(e - b0 for _, (b0, e) in bk)),
          # CUs the GEMM grids leave to RCCL's channels: measured by the first steps (reserve -> ms per step, MAX over ranks)
          "cu_reserve": step.cu_reserve, "cu_reserve_tuning_ms": step.cu_tuning, "comm_forms": getattr(step, "comm_forms", True), "algo_in_use": getattr(reducer, "algo", None),
          "tune_error": step.tune_error, "optimizer_sharded": bool(getattr(opt, "_shard", False)),
          "optimizer_shard_tuning": step.shard_tuning,
          "env": {k: v for k, v in os.environ.items() if k.startswith(("REED_", "NCCL_", "RCCL_"))}}
    if reducer is not None:
        try:
            dp.update(dp_consistency(reducer, opt, model, world, b))
            dp["algo"] = reducer.algo
            reducer.timing = []
            waits = []
            orig_sync = reducer.sync

            def timed_sync():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                orig_sync()
                e1.record()
                waits.append((e0, e1))
            reducer.sync = timed_sync
            for _ in range(2):
                step(None, labels, zs, moments=moments)
            barrier()
            reducer.sync = orig_sync
            tm, reducer.timing = reducer.timing, None
            per = {}
            for name, nbytes, e0, e1 in tm:
                per.setdefault(name, []).append((nbytes, e0.elapsed_time(e1)))
            rows = [{"bucket": n, "bytes": v[-1][0], "ms": round(v[-1][1], 4),
                     "GBps": round(v[-1][0] / max(v[-1][1], 1e-6) / 1e6, 1)} for n, v in per.items()]
            dp["bucket_reductions_last_step"] = rows if len(rows) <= 80 else rows[:80]
            dp["sum_bucket_ms"] = round(sum(r["ms"] for r in rows), 3)
            dp["exposed_wait_at_sync_ms"] = round(waits[-1][0].elapsed_time(waits[-1][1]), 4) if waits else None
        except Exception as e:   # a reported diagnosis, never a reason to lose the throughput line
            dp["error"] = repr(e)

    if rank == 0:
        out = base_record(chosen)
        eng = model.engine()
        # the kernel forms the engine picks by token count (engine.py): part of the plan, so part of the record
        out["kernel_forms"] = {
            "tokens_per_gpu": b * T_TOK,
            "activation_backward": ("saved derivative (epilogues 14 / 16)" if (eng.save_act_grad if eng.save_act_grad is not None else b * T_TOK > eng.SAVE_ACT_GRAD_MIN_TOKENS)
                                    else "recomputed (epilogues 1 / 4)"),
            "activation_backward_rule": f"saved derivative above {eng.SAVE_ACT_GRAD_MIN_TOKENS} tokens per GPU (engine.save_act_grad = {eng.save_act_grad})",
            "input_gradients": ("NT GEMMs on transposed weight copies" if (eng.dgrad_nt if eng.dgrad_nt is not None else (eng.DGRAD_NT_MIN_TOKENS is not None and b * T_TOK >= eng.DGRAD_NT_MIN_TOKENS))
                                else "NN GEMMs on the weights as they are"),
            "input_gradient_rule": f"NT on W^T from {eng.DGRAD_NT_MIN_TOKENS} tokens per GPU (engine.dgrad_nt = {eng.dgrad_nt})",
            "weight_gradients_on_a_second_stream": bool(eng.wgrad_stream if eng.wgrad_stream is not None else b * T_TOK <= eng.WGRAD_STREAM_MAX_TOKENS),
            "weight_gradient_stream_rule": f"second stream up to {eng.WGRAD_STREAM_MAX_TOKENS} tokens per GPU (engine.wgrad_stream = {eng.wgrad_stream})",
            "library": {k: getattr(v, "_name", None) for k, v in __import__("reed_amd._lib", fromlist=["loaded"]).loaded().items()},
        }
        out["data_parallel"] = dp
        if reducer is not None:
            out["plans"] = plans
            out["plan_in_value"] = "tuned" if chosen is not plain else "plain"
        # roofline = the ONE kernel with the largest share of the step, timed live inside the timed region (with or without the
        # isolated kernel table, which only supplies the stand-in when the grouped launch did not run)
        n_l = sum(len(v) for v in dom.values())
        sag = model.engine().save_act_grad
        sag = b * T_TOK > model.engine().SAVE_ACT_GRAD_MIN_TOKENS if sag is None else sag     # the engine's rule (engine.py)
        rows = time_gemms(b, D=args_D(model), Hm=model.engine().Hm, act_grad=sag) if not args.no_kernel_table else None
        if n_l or rows is not None:
            if n_l:
                tot_ms = sum(sum(v) for v in dom.values())
                tot_fl = sum(fl * len(v) for fl, v in dom.items())
                ach = tot_fl / max(tot_ms, 1e-9) / 1e9
                out["roofline"] = {
                    "bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                    "frac": round(ach * 1e12 / PEAK_BF16, 4),
                    "traffic": TRAFFIC_B256 if (b == 256 and args.model == "SiT-XL/2") else None,
                    "traffic_source": TRAFFIC_SOURCE if (b == 256 and args.model == "SiT-XL/2") else None,
                    "kernel": "the grouped weight-gradient launch = the weight (+ bias) gradients of one transformer block's four "
                              "linears (fc2, fc1, proj, qkv: dW = dY^T X over the b*256 tokens) as ONE launch: gemm256w_tn_group_kernel "
                              "(csrc/gemm256w.hip, round 6: one item per CU — 212 tiles of 256^2 + 24 items of 384x128 + 18 of 128x384 whose "
                              "fourth waves form the bias gradients + 2 bias-only items, every one a whole-K sequence at a full tile's "
                              "pace, dealt around tile rows per XCD) or — with REED_WGRAD_W4=0, where that form does not apply, and beside "
                              "gradient buckets in flight (N > 1) — gemm_tn_group_kernel (csrc/gemm_tn.hip: 256x128 / 128x256 tiles, two "
                              "workgroups per CU); algorithmic flop 2 * tokens * sum(n_out * k_in) per launch / event-timed duration "
                              "of every such launch INSIDE the timed region (events on the launch stream; the largest single share "
                              "of the step)",
                    "launches_timed": n_l, "avg_ms_per_launch": round(tot_ms / n_l, 4),
                    "flop_per_launch": next(iter(dom))}
            else:
                # the grouped launch did not run (a CU reserve under data parallelism, a model whose tiles do not fill the
                # slots, REED_WGRAD_GROUP=0): the same gradients go through the per-GEMM split-K path, and the roofline kernel
                # is its largest member, timed in isolation after the timed region (the kernel table below)
                r = max((r for r in rows if r["kernel"].startswith("wgrad")), key=lambda r: r["ms"])
                out["roofline"] = {
                    "bound": "mfma", "achieved": r["tflops"], "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                    "frac": round(r["tflops"] * 1e12 / PEAK_BF16, 4), "traffic": None, "traffic_source": None,
                    "kernel": f"gemm_tn_kernel ({r['kernel']}): the largest weight-gradient GEMM of a block through the split-K slab "
                              "path (the grouped launch was not used in this run), event-timed in ISOLATION after the timed region",
                    "launches_timed": 0, "avg_ms_per_launch": r["ms"],
                    "flop_per_launch": round(r["tflops"] * r["ms"] * 1e9)}
        if rows is not None:
            tot = sum(r["ms"] for r in rows)
            agg = sum(r["tflops"] * r["ms"] for r in rows) / tot
            # the GEMM family launched in ISOLATION after the timed region (12 shapes of one block, time-weighted): an upper
            # view of the kernels, not the step — the step's efficiency is step_mfma_frac
            out["gemm_family_isolated"] = {"tflops": round(agg, 1), "frac": round(agg * 1e12 / PEAK_BF16, 4),
                                           "ms_per_block": round(tot, 4), "table": rows,
                                           "activation_backward": "saved derivative (fc1 epilogue 14, fc2 dgrad epilogue 16)"
                                           if sag else "recomputed (epilogues 1 / 4)"}
        if rows is not None and world == 1 and args.model == "SiT-XL/2":
            try:
                out["attention"] = time_attention(b)
            except Exception as e:   # a reported leg, never a reason to lose the headline line
                out["attention"] = {"error": repr(e)}
        if world == 1 and not args.no_c3_leg and b != 32 and args.model == "SiT-XL/2":
            try:
                out["c3_per_gpu_leg"] = c3_leg(step, dev, args.z_dim)
            except Exception as e:   # a reported leg, never a reason to lose the headline line
                out["c3_per_gpu_leg"] = {"error": repr(e)}
        if world == 1 and not args.no_config_legs and args.model == "SiT-XL/2":
            for key, fn in (("c4_per_gpu_leg", c4_leg), ("c5_sampler_leg", c5_leg), ("n2_encoder_leg", n2_encoder_leg)):
                try:
                    out[key] = fn(dev)
                except Exception as e:
                    out[key] = {"error": repr(e)}
        if world == 1 and args.mixed_precision == "bf16" and not args.no_loss_vs_ref:
            try:
                out["loss_vs_ref"] = loss_vs_ref_leg(dev)
            except Exception as e:
                out["loss_vs_ref"] = {"error": repr(e)}
        if world == 1 and not args.no_vae_leg:
            try:
                out["n4_vae_decode"] = n4_vae_leg(dev)
            except Exception as e:
                out["n4_vae_decode"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "images/sec", "cores": torch.get_num_threads(),
                                       "kind": "port", "sample": f"failed: {e}"}
        with wd_lock:
            emit(out)
            wd_state["printed"] = True
    if use_dist:   # rank 0 is still timing the kernel table: nobody tears a communicator down under it
        torch.cuda.synchronize()
        dist.barrier()
    if reducer is not None:
        reducer.close()
    if use_dist:
        dist.destroy_process_group()
    if watchdog is not None:
        watchdog.cancel()
    if watchdog_late is not None:
        watchdog_late.cancel()


if __name__ == "__main__":
    main()
