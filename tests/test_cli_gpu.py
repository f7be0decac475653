"""CLI round trip on the GPU: train.py counterpart on synthetic latents (C1-style plumbing, alignment off and on),
checkpoint layout, resume, then generate.py counterpart sampling from the EMA checkpoint."""
import glob
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_checkpoint_resume_generate(dev, tmp_path):
    from reed_amd import generate, train
    out = str(tmp_path / "exps")
    common = ["--model", "SiT-S/2", "--output-dir", out, "--mixed-precision", "bf16", "--batch-size", "16",
              "--synthetic", "64", "--num-workers", "0", "--diffusion-warm-up-steps", "0", "--report-to", "none",
              "--checkpointing-steps", "3"]
    a = train.parse_args(["--exp-name", "c1", "--enc-type", "None", "--max-train-steps", "6"] + common)
    save_dir = train.main(a)
    ck = sorted(glob.glob(os.path.join(save_dir, "checkpoints", "*.pt")))
    assert [os.path.basename(c) for c in ck] == ["0000003.pt", "0000006.pt"]
    c = torch.load(ck[-1], map_location="cpu", weights_only=False)
    assert set(c.keys()) == {"model", "ema", "opt", "args", "steps"} and c["steps"] == 6
    assert "blocks.11.mlp.fc2.weight" in c["model"] and "pos_embed" in c["ema"]
    assert len(c["opt"]["state"]) > 0 and c["opt"]["param_groups"][0]["lr"] == 1e-4
    logs = [json.loads(l) for l in open(os.path.join(save_dir, "metrics.jsonl"))]
    assert len(logs) == 6 and all(np.isfinite(r["training_denoising_loss"]) for r in logs)
    assert logs[-1]["training_denoising_loss"] < logs[0]["training_denoising_loss"] + 0.5
    # resume from step 3 and continue to 5
    exp = os.path.basename(save_dir)
    a2 = train.parse_args(["--exp-name", exp, "--enc-type", "None", "--max-train-steps", "5", "--resume-step", "3"] + common)
    train.main(a2)
    # alignment on (image encoder features + pooled text vector), 2 steps
    a3 = train.parse_args(["--exp-name", "c4", "--enc-type", "clip-vit-L", "--text-embeds-dir", "text_embeds_open_clip",
                           "--repa-coeff", "1.0", "0.5", "--encoder-depth-text", "10", "--max-train-steps", "2"] + common)
    d3 = train.main(a3)
    l3 = [json.loads(l) for l in open(os.path.join(d3, "metrics.jsonl"))]
    assert "text_proj_loss" in l3[-1] and np.isfinite(l3[-1]["proj_loss"])
    # generate from the EMA weights
    g = generate.build_parser().parse_args(["--ckpt", ck[-1], "--model", "SiT-S/2", "--sample-dir", str(tmp_path / "samples"),
                                            "--per-proc-batch-size", "4", "--num-fid-samples", "8", "--num-steps", "4",
                                            "--heun", "--cfg-scale", "1.5", "--save-latents"])
    folder = generate.main(g)
    torch.set_grad_enabled(True)
    lat = np.load(folder + "_latents.npz")["arr_0"]
    assert lat.shape == (8, 4, 32, 32) and np.isfinite(lat).all()
    # the same through the in-repo SD-VAE decoder (reed_amd/vae.py) on a local checkpoint in diffusers' layout: PNGs + .npz
    from safetensors.torch import save_file
    from reed_amd.vae import SDVAEDecoder
    torch.manual_seed(1)
    vdir = tmp_path / "sd-vae-ft-ema"
    vdir.mkdir()
    save_file({k: v.contiguous() for k, v in SDVAEDecoder().state_dict().items()}, str(vdir / "diffusion_pytorch_model.safetensors"))
    g2 = generate.build_parser().parse_args(["--ckpt", ck[-1], "--model", "SiT-S/2", "--sample-dir", str(tmp_path / "samples_png"),
                                             "--per-proc-batch-size", "4", "--num-fid-samples", "4", "--num-steps", "2",
                                             "--vae-ckpt", str(vdir)])
    folder2 = generate.main(g2)
    torch.set_grad_enabled(True)
    assert sorted(os.listdir(folder2)) == [f"{i:06d}.png" for i in range(4)]
    arr = np.load(folder2 + ".npz")["arr_0"]
    assert arr.shape == (4, 256, 256, 3) and arr.dtype == np.uint8
    # the reference's in-training previews (train.py:431-454) with that decoder: a grid at step 1 (and every --sampling-steps)
    a4 = train.parse_args(["--exp-name", "prev", "--enc-type", "None", "--max-train-steps", "1", "--vae-ckpt", str(vdir)] + common)
    d4 = train.main(a4)
    assert sorted(os.listdir(os.path.join(d4, "samples"))) == ["0000001.png", "gt_samples.png"]
    from PIL import Image
    grid = np.asarray(Image.open(os.path.join(d4, "samples", "0000001.png")))
    assert grid.shape == (8 * 258 + 2, 8 * 258 + 2, 3)      # 64 previews, 8 per row, 2-pixel gutters


def test_train_default_mixed_precision_is_fp16(dev, tmp_path):
    """No --mixed-precision flag = the reference's default "fp16" (image/train.py:458): IEEE-half operands with the loss
    scaler in the fused optimiser pass; checkpoint / resume keep working (the Adam step count comes from the scaler), the
    EMA checkpoint samples.  "no" (image/train.py:505) trains with fp32 operands through the fp32 build of the library, and
    generate.py --no-tf32 (image/generate.py:41,183) samples with it: the same seed gives latents within fp16's distance of
    the default (TF32-mantissa) run, and never a silent substitution (the model's precision is what the flag says)."""
    from reed_amd import generate, train
    out = str(tmp_path / "exps")
    common = ["--model", "SiT-S/2", "--output-dir", out, "--batch-size", "8", "--synthetic", "32", "--num-workers", "0",
              "--diffusion-warm-up-steps", "0", "--report-to", "none", "--checkpointing-steps", "2", "--enc-type", "dinov2-vit-b"]
    a = train.parse_args(["--exp-name", "h", "--max-train-steps", "4"] + common)
    assert a.mixed_precision == "fp16"
    d = train.main(a)
    logs = [json.loads(l) for l in open(os.path.join(d, "metrics.jsonl"))]
    assert len(logs) == 4 and all(np.isfinite(r["training_denoising_loss"]) and np.isfinite(r["grad_norm"]) for r in logs)
    c = torch.load(os.path.join(d, "checkpoints", "0000004.pt"), map_location="cpu", weights_only=False)
    assert {v["step"].item() for v in c["opt"]["state"].values()} == {4.0}
    a2 = train.parse_args(["--exp-name", os.path.basename(d), "--max-train-steps", "5", "--resume-step", "4"] + common)
    train.main(a2)
    logs = [json.loads(l) for l in open(os.path.join(d, "metrics.jsonl"))]
    assert len(logs) == 5 and np.isfinite(logs[-1]["grad_norm"])
    g = generate.build_parser().parse_args(["--ckpt", os.path.join(d, "checkpoints", "0000004.pt"), "--model", "SiT-S/2",
                                            "--sample-dir", str(tmp_path / "samples"), "--per-proc-batch-size", "4",
                                            "--num-fid-samples", "4", "--num-steps", "3", "--save-latents"])
    folder = generate.main(g)
    torch.set_grad_enabled(True)
    assert np.isfinite(np.load(folder + "_latents.npz")["arr_0"]).all()
    lat16 = np.load(folder + "_latents.npz")["arr_0"]
    d32 = train.main(train.parse_args(["--exp-name", "n", "--max-train-steps", "2", "--mixed-precision", "no"] + common))
    logs = [json.loads(l) for l in open(os.path.join(d32, "metrics.jsonl"))]
    assert len(logs) == 2 and all(np.isfinite(r["training_denoising_loss"]) and np.isfinite(r["grad_norm"]) for r in logs)
    g32 = generate.build_parser().parse_args(["--ckpt", os.path.join(d, "checkpoints", "0000004.pt"), "--model", "SiT-S/2",
                                              "--sample-dir", str(tmp_path / "samples32"), "--per-proc-batch-size", "4",
                                              "--num-fid-samples", "4", "--num-steps", "3", "--save-latents", "--no-tf32"])
    assert generate.sample_precision(g32) == "fp32" and generate.sample_precision(g) == "fp16"
    folder32 = generate.main(g32)
    torch.set_grad_enabled(True)
    lat32 = np.load(folder32 + "_latents.npz")["arr_0"]
    assert np.isfinite(lat32).all() and np.abs(lat32 - lat16).max() < 5e-3 * np.abs(lat32).max()


def test_cu_reserve_is_measured_under_a_reducer(dev):
    """Data-parallel TrainStep (RCCL process group at world 1, forced).  The measurement is opt-in: without
    REED_COMM_CUS nothing is tuned and the bucket form stays "allreduce".  With REED_COMM_CUS=auto the first optimiser steps
    run with a CU reserve of 0 / 16 / 32 for the GEMM grids (one settling + three timed steps each, the median counts), the
    fastest is kept and reported; then the backward without the kernel forms it selects beside collectives ("static":
    ops.set_comm_forms) is measured at that reserve and kept if faster; REED_COMM_ALGO=auto then measures the reduce-scatter + all-gather bucket form the same way;
    REED_COMM_CUS=<n> fixes the reserve; a failure inside the bookkeeping ends the measurement with the safe plan."""
    import copy
    import torch.distributed as dist
    from oracle import detfill
    from reed_amd import _lib, ops
    from reed_amd.loss import SILoss
    from reed_amd.models.sit import SiT
    from reed_amd.optim import FusedAdamWEMA
    from reed_amd.parallel import GradReducer
    from reed_amd.trainer import TrainStep
    os.environ["REED_COMM"] = "torch"
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29741")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    os.environ["REED_FORCE_REDUCER"] = "1"
    os.environ.pop("REED_COMM_CUS", None)
    os.environ.pop("REED_COMM_ALGO", None)
    try:
        m = SiT(input_size=8, hidden_size=128, decoder_hidden_size=128, depth=3, num_heads=2, num_classes=10, z_dims=[128],
                z_types=["i"], encoder_depth=2, projector_dim=128)
        detfill.fill_state_dict(m.state_dict(), base_seed=5)
        m = m.to(dev).train()
        lf = SILoss(enc_names=["dinov2"], loss_weights={"dinov2": 1.0})
        x = detfill.normal((4, 4, 8, 8), 1).to(dev)
        y = torch.tensor([1, 5, 9, 0], device=dev)
        zs = [detfill.normal((4, 16, 128), 4).to(dev)]
        plain = TrainStep(m, lf, FusedAdamWEMA(m, None, lr=1e-4), None, diffusion_warm_up_steps=0)
        assert plain.tune_steps_left() == 0 and plain.cu_reserve == 0
        red0 = GradReducer(m)
        quiet = TrainStep(m, lf, FusedAdamWEMA(m, None, lr=1e-4), red0, diffusion_warm_up_steps=0)
        assert quiet.tune_steps_left() == 0 and quiet.cu_reserve == 0 and red0.algo == "allreduce"   # the default: nothing measured
        red0.close()
        os.environ["REED_COMM_CUS"] = "auto"
        os.environ["REED_COMM_ALGO"] = "auto"
        red = GradReducer(m)
        assert red.algo == "allreduce"
        ts = TrainStep(m, lf, FusedAdamWEMA(m, None, lr=1e-4), red, diffusion_warm_up_steps=0)
        assert ts.tune_steps_left() == 20         # 3 reserves + the backward without the kernel forms it selects beside collectives
        full = _lib.load().reed_planning_cus()    # ("static") + the reduce-scatter / all-gather bucket form, 1 + 3 steps each
        seen, algos, forms = [], [], []
        for _ in range(21):
            algos.append(red.algo)
            forms.append(ops.comm_forms())
            r = ts(x, y, zs)
            seen.append(_lib.load().reed_planning_cus())
        torch.cuda.synchronize()
        assert torch.isfinite(r["loss"]).item() and ts.tune_error is None
        # (read after each step: the twelfth ends the reserve phase)
        assert seen[:11] == [full] * 4 + [full - 16] * 4 + [full - 32] * 3
        assert ts.tune_steps_left() == 0 and set(ts.cu_tuning) == {"0", "16", "32", "static", "rsag"} and ts.cu_reserve in (0, 16, 32)
        assert seen[11] == seen[12] == seen[20] == full - ts.cu_reserve == _lib.load("fp16").reed_planning_cus()
        assert forms[:13] == [True] * 13 and forms[13:16] == [False] * 3 and forms[17:] == [ts.comm_forms] * 4
        assert ts.comm_forms == ops.comm_forms() == (not ts.cu_tuning["static"] < ts.cu_tuning[str(ts.cu_reserve)])
        assert algos[:17] == ["allreduce"] * 17 and algos[17:20] == ["rsag"] * 3 and red.algo in ("allreduce", "rsag")
        assert red.algo == ("rsag" if ts.cu_tuning["rsag"] < min(ts.cu_tuning[str(ts.cu_reserve)], ts.cu_tuning["static"]) else "allreduce")
        ops.set_comm_forms(True)
        red.close()
        # a failure inside the bookkeeping is never fatal: reserve 0, all-reduce buckets, the error kept for the report
        os.environ.pop("REED_COMM_ALGO", None)
        red3 = GradReducer(m)
        ts3 = TrainStep(m, lf, FusedAdamWEMA(m, None, lr=1e-4), red3, diffusion_warm_up_steps=0)
        assert ts3.tune_steps_left() == 16
        ts3._agree = lambda times: (_ for _ in ()).throw(RuntimeError("injected"))
        for _ in range(13):
            r = ts3(x, y, zs)
        torch.cuda.synchronize()
        assert "injected" in ts3.tune_error and ts3.tune_steps_left() == 0 and ts3.cu_reserve == 0 and ts3.comm_forms and ops.comm_forms()
        assert _lib.load().reed_planning_cus() == full and red3.algo == "allreduce" and torch.isfinite(r["loss"]).item()
        red3.close()
        os.environ["REED_COMM_CUS"] = "24"
        red2 = GradReducer(m)
        ts2 = TrainStep(m, lf, FusedAdamWEMA(m, None, lr=1e-4), red2, diffusion_warm_up_steps=0)
        assert ts2.tune_steps_left() == 0 and ts2.cu_reserve == 24 and _lib.load().reed_planning_cus() == full - 24
        red2.close()
    finally:
        os.environ.pop("REED_FORCE_REDUCER", None)
        os.environ.pop("REED_COMM_CUS", None)
        os.environ.pop("REED_COMM_ALGO", None)
        ops.set_cu_reserve(0)
        ops.set_comm_forms(True)


@pytest.mark.parametrize("algo", ["allreduce", "rsag"])
@pytest.mark.parametrize("ada_gather", ["1", "0"])
@pytest.mark.parametrize("binding", ["native", "torch"])
def test_rccl_reducer_world1(dev, binding, ada_gather, algo):
    """Exercise every RCCL entry point (unique id, init, broadcast, bucketed all-reduce(avg) fired from backward on the
    side stream, sync, destroy) with a 1-rank communicator: averaging over one rank must leave gradients unchanged.
    binding="torch": the same bucket plan through torch.distributed's RCCL process group (the default binding, and what
    every rank falls back to when the native communicator cannot be created on any of them). algo="rsag": each bucket as
    ncclReduceScatter + ncclAllGather in place (REED_COMM_ALGO)."""
    import copy
    import torch.distributed as dist
    os.environ["REED_COMM"] = binding
    os.environ["REED_COMM_ALGO"] = algo
    if binding == "torch":
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29741")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from oracle import detfill
    from reed_amd.loss import SILoss
    from reed_amd.models.sit import SiT
    from reed_amd.parallel import GradReducer
    os.environ["REED_FORCE_REDUCER"] = "1"
    os.environ["REED_ADA_GATHER"] = ada_gather   # "1": adaLN gradients through the factor all-gather (parallel.py:gather)
    try:
        def run(with_reducer):
            m = SiT(input_size=8, hidden_size=128, decoder_hidden_size=128, depth=3, num_heads=2, num_classes=10,
                    z_dims=[128], projector_dim=128, encoder_depth=2)
            detfill.fill_state_dict(m.state_dict(), base_seed=3)
            m = m.to(dev).train()
            m.force_drop_mask = torch.tensor([False, True, False, False])
            red = GradReducer(m, rank=0, world=1) if with_reducer else None
            if red:
                assert red.binding == binding and red.ada_gather == (ada_gather == "1") and red.algo == algo
                red.broadcast_params(0)
            lf = SILoss(enc_names=["dinov2"], loss_weights={"dinov2": 1.0})
            x, n = detfill.normal((4, 4, 8, 8), 1).to(dev), detfill.normal((4, 4, 8, 8), 2)
            out = lf(m, x, dict(y=torch.tensor([1, 2, 3, 4], device=dev)), zs=[detfill.normal((4, 16, 128), 4).to(dev)],
                     time_input=detfill.uniform((4,), 3, 0.1, 0.9), noises=n)
            (out["denoising_loss"].mean() + 0.5 * out["proj_loss"]).backward()
            if red:
                red.sync()
            torch.cuda.synchronize()
            g = m._arena.grad.clone()
            if red:
                red.close()
            return g
        g0, g1 = run(False), run(True)
        assert torch.equal(g0, g1)
    finally:
        os.environ.pop("REED_FORCE_REDUCER", None)
        os.environ.pop("REED_ADA_GATHER", None)
        os.environ.pop("REED_COMM", None)
        os.environ.pop("REED_COMM_ALGO", None)
        if binding == "torch" and dist.is_initialized():
            dist.destroy_process_group()


_TWO_RANK_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
from oracle import detfill
from reed_amd.loss import SILoss
from reed_amd.models.sit import SiT
from reed_amd.parallel import GradReducer

def make():
    m = SiT(input_size=8, hidden_size=128, decoder_hidden_size=128, depth=3, num_heads=2, num_classes=10, z_dims=[128],
            projector_dim=128, encoder_depth=2)
    detfill.fill_state_dict(m.state_dict(), base_seed=3)
    return m.to(dev).train()

def backward(m, r):   # rank r's local batch (B = 4) and draws
    m.force_drop_mask = torch.tensor([False, True, False, False])
    lf = SILoss(enc_names=["dinov2"], loss_weights={"dinov2": 1.0})
    x, n = detfill.normal((4, 4, 8, 8), 10 + r).to(dev), detfill.normal((4, 4, 8, 8), 20 + r)
    out = lf(m, x, dict(y=torch.tensor([1, 2, 3, 4], device=dev) + r), zs=[detfill.normal((4, 16, 128), 30 + r).to(dev)],
             time_input=detfill.uniform((4,), 40 + r, 0.1, 0.9), noises=n)
    (out["denoising_loss"].mean() + 0.5 * out["proj_loss"]).backward()

# reference on this rank: the two local gradients one after the other without a reducer, averaged
ref = None
for r in range(world):
    m = make()
    backward(m, r)
    torch.cuda.synchronize()
    g = m._arena.grad.clone()
    ref = g if ref is None else ref + g
ref /= world
# the data-parallel step: this rank's batch, bucketed all-reduce(avg) + adaLN factor gather fired from backward
m = make()
red = GradReducer(m, rank=rank, world=world)
assert red.binding == "torch" and red.ada_gather == (os.environ["REED_ADA_GATHER"] == "1")
red.broadcast_params(0)
backward(m, rank)
red.sync()
torch.cuda.synchronize()
got = m._arena.grad.clone()
L = m._layout
covered = torch.zeros(L.n_train, dtype=torch.bool, device=dev)
for b, e in red.buckets.values(): covered[b:e] = True
torch.testing.assert_close(got[covered], ref[covered], rtol=2e-5, atol=2e-6)
gc = got.cpu()
allg = [torch.empty_like(gc) for _ in range(world)]
dist.all_gather(allg, gc)
assert all(torch.equal(allg[0][covered.cpu()], a[covered.cpu()]) for a in allg), "ranks disagree"
dist.barrier(); dist.destroy_process_group()
print("OK", rank)
'''


def test_sharded_optimizer_gather_on_rccl_world1(dev):
    """The collective of the sharded optimiser pass (optim.py:_gather: ncclAllGather IN PLACE, the input a slice of the output, on
    the bit patterns of a 16-bit array viewed as int32) through torch's RCCL process group with one rank — the call form, dtype
    and aliasing are what world > 1 uses; with one rank the gather must leave the buffer as it is."""
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29741")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl"
    from reed_amd.optim import FusedAdamWEMA
    opt = FusedAdamWEMA.__new__(FusedAdamWEMA)
    opt._rank, opt._world = 0, 1
    sh = torch.randn(4096, device=dev).to(torch.bfloat16)
    want = sh.clone()
    opt._gather(sh.view(torch.int32), 256, 1024)
    torch.cuda.synchronize()
    assert torch.equal(sh, want)


@pytest.mark.parametrize("ada_gather,algo", [("1", "allreduce"), ("0", "allreduce"), ("0", "rsag")])
def test_two_ranks_on_one_gpu(dev, tmp_path, ada_gather, algo):
    """The data-parallel ENGINE path with two real ranks (two processes sharing the one GPU of the box, collectives over
    gloo through the torch binding of GradReducer): buckets fired from backward in completion order, the adaLN
    factor gather with world = 2 (1/2-scaled factors, rank-major K = 2 b product, embed bucket cut short) or the adaLN
    buckets — against the average of the two ranks' local gradients computed without a reducer, and both ranks equal.
    algo="rsag": every bucket as reduce-scatter + all-gather (REED_COMM_ALGO; over gloo the same two-phase structure)."""
    script = tmp_path / "w2.py"
    script.write_text(_TWO_RANK_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29745", WORLD_SIZE="2", REED_COMM="torch",
               REED_ADA_GATHER=ada_gather, REED_COMM_ALGO=algo, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o[-3000:]


_TWO_RANK_TRAIN_WORKER = r'''
import copy, os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
from oracle import detfill
from reed_amd.loss import SILoss
from reed_amd.models.sit import SiT
from reed_amd.optim import FusedAdamWEMA
from reed_amd.parallel import GradReducer
from reed_amd.trainer import TrainStep

def make():
    m = SiT(input_size=8, hidden_size=128, decoder_hidden_size=128, depth=3, num_heads=2, num_classes=10, z_dims=[128],
            projector_dim=128, encoder_depth=2)
    detfill.fill_state_dict(m.state_dict(), base_seed=5)
    m = m.to(dev).train()
    ema = copy.deepcopy(m).requires_grad_(False).eval()
    # eps far above the fp32 noise of a gradient: with 1e-8 Adam turns the last bits of a near-zero gradient (the two runs
    # sum tokens in different orders) into a full +-lr step and the comparison below would be meaningless
    opt = FusedAdamWEMA(m, ema, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.0, eps=1e-2, max_grad_norm=1.0)
    return m, ema, opt

def data(r, step):   # rank r's local batch of 4 at a step
    s = 100 * step + 10 * r
    return (detfill.normal((4, 4, 8, 8), s + 1).to(dev), torch.tensor([1, 2, 3, 4], device=dev) + r,
            [detfill.normal((4, 16, 128), s + 2).to(dev)], detfill.uniform((4,), s + 3, 0.1, 0.9), detfill.normal((4, 4, 8, 8), s + 4))

lf = SILoss(enc_names=["dinov2"], loss_weights={"dinov2": 1.0})
# single process, global batch 8 = [rank 0's samples | rank 1's samples]
m, ema, opt = make()
step = TrainStep(m, lf, opt, None, proj_coeff=0.5, diffusion_warm_up_steps=0)
for k in range(2):
    parts = [data(r, k) for r in range(world)]
    m.force_drop_mask = torch.tensor([False, True, False, False] * world)
    step(torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts]), [torch.cat([p[2][0] for p in parts])],
         time_input=torch.cat([p[3] for p in parts]), noises=torch.cat([p[4] for p in parts]))
opt.flush(); torch.cuda.synchronize()
ref, ref_ema = m._arena.master.clone(), ema._arena.master.clone()
# two ranks, local batch 4 each
m, ema, opt = make()
red = GradReducer(m, rank=rank, world=world)
red.broadcast_params(0)
step = TrainStep(m, lf, opt, red, proj_coeff=0.5, diffusion_warm_up_steps=0)
for k in range(2):
    x, y, zs, t, n = data(rank, k)
    m.force_drop_mask = torch.tensor([False, True, False, False])
    step(x, y, zs, time_input=t, noises=n)
opt.flush(); torch.cuda.synchronize()
got, got_ema = m._arena.master.clone(), ema._arena.master.clone()
nt = m._layout.n_train
torch.testing.assert_close(got[:nt], ref[:nt], rtol=1e-4, atol=3e-6)
torch.testing.assert_close(got_ema[:nt], ref_ema[:nt], rtol=1e-4, atol=3e-6)
allg = [torch.empty(nt) for _ in range(world)]
dist.all_gather(allg, got[:nt].cpu())
assert all(torch.equal(allg[0], a) for a in allg), "parameters differ between the ranks after two steps"
# the sharded update (REED_OPT_SHARD=1): every rank runs the fused pass on its 1 / world of every chunk only, the 16-bit shadows are
# gathered in place; after sync_replicas() master weights, EMA and Adam moments are bit-identical to the replicated run's, on both ranks
os.environ["REED_OPT_SHARD"] = "1"
m2, ema2, opt2 = make()
red2 = GradReducer(m2, rank=rank, world=world)
red2.broadcast_params(0)
step2 = TrainStep(m2, lf, opt2, red2, proj_coeff=0.5, diffusion_warm_up_steps=0)
for k in range(2):
    x, y, zs, t, n = data(rank, k)
    m2.force_drop_mask = torch.tensor([False, True, False, False])
    step2(x, y, zs, time_input=t, noises=n)
assert opt2._shard and {o for _, subs in opt2._shard for _, _, o in subs} >= {0, 1}, "both ranks own pieces"
torch.cuda.synchronize()
assert torch.equal(m2._arena.shadow, m._arena.shadow), "the operand copies differ from the replicated run's"
owned = torch.zeros(m2._layout.n_total, dtype=torch.bool)
for _, subs in opt2._shard:
    for b, e, o in subs:
        if o in (rank, -1): owned[b:e] = True
assert not torch.equal(m2._arena.master, got), "a non-owner's master weights are stale before sync_replicas()"
assert torch.equal(m2._arena.master[owned.to(dev)], got[owned.to(dev)])
opt2.sync_replicas(); torch.cuda.synchronize()
assert torch.equal(m2._arena.master, got) and torch.equal(ema2._arena.master, got_ema)
assert torch.equal(opt2.exp_avg, opt.exp_avg) and torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq)
# REED_OPT_SHARD=auto: TrainStep measures the sharded pass against the replicated one (self-test of the gather, one settling + one
# timed step each) and keeps the faster; whatever it keeps, five steps end with the parameters of five replicated steps
def five(mode):
    os.environ["REED_OPT_SHARD"] = mode
    TrainStep.TUNE_STEPS = 1
    m3, ema3, opt3 = make()
    red3 = GradReducer(m3, rank=rank, world=world)
    red3.broadcast_params(0)
    st3 = TrainStep(m3, lf, opt3, red3, proj_coeff=0.5, diffusion_warm_up_steps=0)
    for k in range(5):
        x, y, zs, t, n = data(rank, k)
        m3.force_drop_mask = torch.tensor([False, True, False, False])
        st3(x, y, zs, time_input=t, noises=n)
    opt3.sync_replicas(); opt3.flush(); torch.cuda.synchronize()
    return m3._arena.master.clone(), ema3._arena.master.clone(), st3
ra, rea, _ = five("0")
sa, sea, st3 = five("auto")
assert st3.tune_steps_left() == 0 and st3.tune_error is None, st3.tune_error
assert st3.shard_tuning is not None and "sharded_ms" in st3.shard_tuning and st3.shard_tuning["kept"] == bool(st3.opt._shard), st3.shard_tuning
kept = torch.tensor([int(st3.shard_tuning["kept"])]); both = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
dist.all_gather(both, kept)
assert all(int(b) == int(kept) for b in both), "the ranks kept different optimiser passes"
assert torch.equal(sa, ra) and torch.equal(sea, rea), "the measured run's parameters differ from five replicated steps"
dist.barrier(); dist.destroy_process_group()
print("OK", rank)
'''


def test_two_ranks_train_steps_match_global_batch(dev, tmp_path):
    """Two optimiser steps of TrainStep with two real ranks (local batch 4 each, gloo through the torch binding, adaLN
    factor gather on) against the same two steps in one process on the global batch of 8: the reference's DDP contract
    (train.py:263,293-295) — parameters and EMA equal to fp32 noise, bit-identical on both ranks.  Then the same two steps with
    the sharded optimiser pass (REED_OPT_SHARD=1): operand copies bit-identical to the replicated run's after every step, a
    non-owner's master stale until sync_replicas(), everything bit-identical after it."""
    script = tmp_path / "w2t.py"
    script.write_text(_TWO_RANK_TRAIN_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29747", WORLD_SIZE="2", REED_COMM="torch",
               REED_ADA_GATHER="1", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o[-3000:]


def test_train_with_on_device_clip_encoder(dev, tmp_path, monkeypatch):
    """train.py counterpart on the reference's on-disk format (images/*.png + vae-sd/*.npy + dataset.json,
    image/dataset.py:18-85) with the frozen CLIP image encoder running on the GPU every step from a user-supplied
    state dict (--encoder-ckpts). A 1-block tower of ViT-L/14 width stands in for the 24-block one."""
    import PIL.Image
    from oracle import clip_vit as oclip
    from reed_amd import encoders, train
    data = tmp_path / "data"
    (data / "images" / "00000").mkdir(parents=True)
    (data / "vae-sd" / "00000").mkdir(parents=True)
    rng = np.random.default_rng(0)
    labels = []
    for i in range(8):
        img = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
        PIL.Image.fromarray(img).save(data / "images" / "00000" / f"img{i:08d}.png")
        mom = np.concatenate([rng.standard_normal((4, 32, 32)) * 5.0, np.full((4, 32, 32), 0.5)]).astype(np.float32)
        np.save(data / "vae-sd" / "00000" / f"img-mean-std-{i:08d}.npy", mom)
        labels.append([f"00000/img-mean-std-{i:08d}.npy", int(i % 5)])
    json.dump({"labels": labels}, open(data / "vae-sd" / "dataset.json", "w"))
    cfg = oclip.make_config(width=1024, layers=1, heads=16, patch=14, image=224)
    monkeypatch.setitem(encoders.CLIP_CONFIGS, "L", cfg)
    sd = {"visual." + k: v for k, v in oclip.fill_params(cfg, base_seed=1).items()}   # a full-CLIP style state dict
    sd["visual.proj"] = torch.zeros(1024, 768)
    ck = str(tmp_path / "clip_visual.pt")
    torch.save(sd, ck)
    a = train.parse_args(["--exp-name", "clip", "--model", "SiT-S/2", "--output-dir", str(tmp_path / "exps"),
                          "--data-dir", str(data), "--enc-type", "clip-vit-L", "--encoder-ckpts", ck,
                          "--mixed-precision", "bf16", "--batch-size", "4", "--num-workers", "0",
                          "--diffusion-warm-up-steps", "0", "--report-to", "none", "--max-train-steps", "3",
                          "--num-classes", "5", "--checkpointing-steps", "100"])
    d = train.main(a)
    logs = [json.loads(l) for l in open(os.path.join(d, "metrics.jsonl"))]
    assert len(logs) == 3 and all(np.isfinite(r["proj_loss"]) and np.isfinite(r["training_denoising_loss"]) for r in logs)
    assert logs[0]["img_proj_loss"] != 0.0
    # the same run from the packed (memory-mapped) form of the dataset: identical items -> identical first-step losses
    from reed_amd.dataset import pack_dataset
    pack_dataset(str(data), str(tmp_path / "packed"))
    a2 = train.parse_args(["--exp-name", "clip_packed", "--model", "SiT-S/2", "--output-dir", str(tmp_path / "exps"),
                           "--packed-dir", str(tmp_path / "packed"), "--enc-type", "clip-vit-L", "--encoder-ckpts", ck,
                           "--mixed-precision", "bf16", "--batch-size", "4", "--num-workers", "0",
                           "--diffusion-warm-up-steps", "0", "--report-to", "none", "--max-train-steps", "3",
                           "--num-classes", "5", "--checkpointing-steps", "100"])
    d2 = train.main(a2)
    logs2 = [json.loads(l) for l in open(os.path.join(d2, "metrics.jsonl"))]
    assert len(logs2) == 3 and all(np.isfinite(r["proj_loss"]) for r in logs2)
    torch.set_grad_enabled(True)


def test_train_with_on_device_jepa_and_mae_towers(dev, tmp_path, monkeypatch):
    """train.py counterpart with THREE frozen image encoders running on the GPU every step (--enc-type jepa-vit-h,mae-vit-l,dinov2reg-vit-b
    --encoder-ckpts a b; image/train.py:182-186,351-357 with the jepa / mae branches of preprocess_raw_image and
    load_encoders): 1-block towers of the real widths stand in; checkpoints in the reference's file layouts
    ({'encoder': {'module.*'}} for I-JEPA, utils.py:153-158; {'model': ...} for MAE, utils.py:137-145)."""
    import PIL.Image
    from oracle import vit_towers as ot
    from reed_amd import encoders, train
    data = tmp_path / "data"
    (data / "images" / "00000").mkdir(parents=True)
    (data / "vae-sd" / "00000").mkdir(parents=True)
    rng = np.random.default_rng(1)
    labels = []
    for i in range(8):
        PIL.Image.fromarray(rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)).save(data / "images" / "00000" / f"img{i:08d}.png")
        mom = np.concatenate([rng.standard_normal((4, 32, 32)) * 5.0, np.full((4, 32, 32), 0.5)]).astype(np.float32)
        np.save(data / "vae-sd" / "00000" / f"img-mean-std-{i:08d}.npy", mom)
        labels.append([f"00000/img-mean-std-{i:08d}.npy", int(i % 5)])
    json.dump({"labels": labels}, open(data / "vae-sd" / "dataset.json", "w"))
    cks = []
    for key, wrap in (("jepa-vit-h", lambda sd: {"encoder": {"module." + k: v for k, v in sd.items()}}),
                      ("mae-vit-l", lambda sd: {"model": sd})):
        kw = dict(encoders.VIT_TOWERS[key], depth=1)
        monkeypatch.setitem(encoders.VIT_TOWERS, key, kw)
        P = ot.fill_params(ot.make_config(pos="jepa" if "jepa" in key else "learned", **kw), base_seed=2)
        path = str(tmp_path / (key + ".pth"))
        torch.save(wrap(P), path)
        cks.append(path)
    # DINOv2 with registers (utils.py:92-104): the torch.hub checkpoint layout — a plain state dict with the 37 x 37 pos_embed
    # (resampled to 16 x 16 at load), mask_token, register_tokens, ls{1,2}.gamma
    kw = dict(encoders.VIT_TOWERS["dinov2reg-vit-b"], depth=1)
    monkeypatch.setitem(encoders.VIT_TOWERS, "dinov2reg-vit-b", kw)
    P = ot.fill_params(ot.make_config(768, 1, 12, 14, 224, True, True, "learned", ls=True, reg=4), base_seed=2)
    P["pos_embed"] = torch.randn(1, 1 + 37 * 37, 768, generator=torch.Generator().manual_seed(1)) * 0.02
    P["mask_token"] = torch.zeros(1, 768)
    cks.append(str(tmp_path / "dinov2_vitb14_reg4_pretrain.pth"))
    torch.save(P, cks[-1])
    a = train.parse_args(["--exp-name", "towers", "--model", "SiT-S/2", "--output-dir", str(tmp_path / "exps"),
                          "--data-dir", str(data), "--enc-type", "jepa-vit-h,mae-vit-l,dinov2reg-vit-b", "--encoder-ckpts", *cks,
                          "--repa-coeff", "1.0", "0.5", "0.5", "--mixed-precision", "bf16", "--batch-size", "4", "--num-workers", "0",
                          "--diffusion-warm-up-steps", "0", "--report-to", "none", "--max-train-steps", "2",
                          "--num-classes", "5", "--checkpointing-steps", "100"])
    d = train.main(a)
    logs = [json.loads(l) for l in open(os.path.join(d, "metrics.jsonl"))]
    assert len(logs) == 2 and all(np.isfinite(r["proj_loss"]) and np.isfinite(r["training_denoising_loss"]) for r in logs)
    assert logs[0]["img_proj_loss"] != 0.0
    torch.set_grad_enabled(True)


def test_bench_prints_one_json_line_with_the_contract_keys(dev):
    """bench.py (the driver's entry): exactly one JSON line on stdout with the contract's keys, `roofline` for the dominant kernel
    timed inside the step, `value` = global batch x steps / measured time, and — in the default form — `cpu_baseline`
    (skipped here for time: its presence is checked through the flag that removes it)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--global-batch", "32", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "step_mfma_frac", "data_parallel"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "images/sec" and d["dtype"] == "bf16" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 32 * 1e3 / d["ms_per_step"]) <= 1e-2 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches_timed", "avg_ms_per_launch"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0 and rf["launches_timed"] == 3 * 28
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.05 < rf["frac"] < 1.0
    assert "cpu_baseline" not in d      # --no-cpu-baseline; the default run adds {"value", "unit", "cores", "kind", "sample"}
    # BASELINE.json's second number: the C2 fixture's 5 injected steps on the HIP path against the reference's recorded losses
    lv = d["loss_vs_ref"]
    assert "error" not in lv, lv
    assert lv["steps"] == 5 and lv["bar"] == 1e-3 and lv["within_bar"] is True and lv["max_abs_delta_bf16"] <= 1e-3, lv
    assert len(lv["hip_loss"]) == 5 and len(lv["reference_bf16_autocast_loss"]) == 5
    # the legs the N = 1 run appends (C4, C5 at its real 499 evaluations, N2, N4) and the kernel forms of the plan
    assert d["c4_per_gpu_leg"]["images_per_sec_per_gpu"] > 0 and d["c5_sampler_leg"].get("finite", True) and "error" not in d["c5_sampler_leg"]
    assert "error" not in d["n2_encoder_leg"] and d["n4_vae_decode"]["finite"] is True
    kf = d["kernel_forms"]
    assert kf["tokens_per_gpu"] == 32 * 256 and kf["activation_backward"].startswith("recomputed") and kf["weight_gradients_on_a_second_stream"] is False


def test_bench_stdout_is_one_line_under_a_reducer(dev):
    """The multi-GPU path of bench.py (process group, RCCL reducer: rehearsed with one rank through REED_FORCE_REDUCER=1): RCCL
    prints a version banner on stdout when the communicator is created — stdout must still be the one JSON line."""
    env = dict(os.environ, REED_FORCE_REDUCER="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--global-batch", "16", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-kernel-table", "--no-config-legs", "--no-vae-leg", "--no-loss-vs-ref"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["data_parallel"]["world"] == 1 and d["data_parallel"]["cu_reserve_tuning_ms"] is not None


def test_bench_self_launch(dev):
    """The driver's command form: `python bench.py --gpus N ...` with NO rank environment.  N = 1 runs in-process and prints one
    JSON line with the C3 per-GPU leg; N = 2 on this one-GPU box must exit non-zero with a one-line message, promptly (no hang,
    no rank started)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    bench = os.path.join(ROOT, "bench.py")
    t0 = time.time()
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 2 and r.stdout.strip() == "" and "only 1 GPU" in r.stderr and time.time() - t0 < 60
    r = subprocess.run([sys.executable, bench, "--gpus", "1", "--steps", "2", "--warmup", "1", "--model", "SiT-XL/2", "--global-batch", "64",
                        "--no-cpu-baseline", "--no-kernel-table", "--no-config-legs", "--no-vae-leg", "--no-loss-vs-ref"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["local_batch"] == 64
    leg = d["c3_per_gpu_leg"]
    assert leg["local_batch"] == 32 and leg["images_per_sec_per_gpu"] > 0 and np.isfinite(leg["final_loss"])
    # (round 6) the leg's second process: the data-parallel code path at world 1 — reducer, buckets, kernel forms beside collectives
    dp1 = leg["data_parallel_code_path_at_world_1"]
    assert "error" not in dp1, dp1
    assert dp1["images_per_sec_per_gpu"] > 0 and dp1["data_parallel"]["world"] == 1 and dp1["data_parallel"]["buckets"] > 0
    assert "torch binding" in dp1["plan"] and "with_optimizer_pass_sharded_8_ways" not in leg


def test_bench_two_ranks_rehearsal(dev):
    """`python bench.py --gpus 2` end to end on a one-GPU box (REED_BENCH_REHEARSE=gloo: both ranks on the one device,
    collectives over gloo): the self-launch, the rendezvous, broadcast of the parameters, the reducer's buckets fired from
    backward with the run-time CU-reserve tuning, the barriers, MAX over ranks and the single relayed JSON line — every line of
    the N > 1 bench path except RCCL itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["REED_BENCH_REHEARSE"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--model", "SiT-S/2",
                        "--global-batch", "16", "--no-cpu-baseline", "--no-kernel-table"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["local_batch"] == 8 and d["config"]["global_batch"] == 16
    assert d["data_parallel"]["world"] == 2 and "REHEARSAL" in d["data"] and np.isfinite(d["final_loss"])
    assert "c3_per_gpu_leg" not in d and d["scaling"] == "strong"


def _rehearse(extra_env, timeout=900, extra_args=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["REED_BENCH_REHEARSE"] = "gloo"
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--model", "SiT-S/2",
                           "--global-batch", "16", "--no-cpu-baseline", "--no-kernel-table"] + list(extra_args), env=env,
                          capture_output=True, text=True, timeout=timeout)


def test_bench_two_ranks_times_the_plain_plan_first(dev):
    """At N > 1 the bench times the plain plan (torch binding, all-reduce buckets, no CU reserve, replicated optimiser pass, no
    run-time measurement) through warm-up and the timed region BEFORE the tuned plan is even scheduled; both are reported, `value`
    is the plain region's unless the tuner kept a DIFFERENT plan that measured faster (two regions of one plan are two samples
    of it: their maximum would bias the number)."""
    r = _rehearse({})
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    pl = d["plans"]
    assert "CU reserve 0" in pl["plain"]["plan"] and "replicated" in pl["plain"]["plan"] and "allreduce" in pl["plain"]["plan"]
    assert pl["plain"]["images_per_sec"] > 0 and pl["tuned"].get("images_per_sec", 0) > 0, pl
    same_plan = pl["plain"]["plan"] == pl["tuned"]["plan"]
    best = pl["plain"]["images_per_sec"] if same_plan else max(pl["plain"]["images_per_sec"], pl["tuned"]["images_per_sec"])
    assert d["value"] == best and d["plan_in_value"] in ("plain", "tuned") and (d["plan_in_value"] == "plain" or not same_plan)
    assert d["data_parallel"]["cu_reserve_tuning_ms"] is not None     # the tuned region did measure something


@pytest.mark.parametrize("fail", ["raise", "hang"])
def test_bench_keeps_the_plain_record_when_the_tuner_fails(dev, fail):
    """A tuner that raises, or hangs (bench.py --test-tuner-fail), after the plain plan was timed: rc 0, exactly one JSON line, the
    plain plan's number in it, the failure named under plans.tuned.  The hang is ended by the in-process watchdog (a timer
    thread: rank 0 prints the held record, every rank leaves with os._exit — no exec, no teardown of a wedged communicator)."""
    r = _rehearse({}, extra_args=["--test-tuner-fail", fail, "--tuned-timeout", "25"])
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["plan_in_value"] == "plain" and d["value"] == d["plans"]["plain"]["images_per_sec"] > 0
    err = d["plans"]["tuned"]["error"]
    assert ("timeout" in err) if fail == "hang" else ("--test-tuner-fail" in err), err
    assert ("watchdog_fired" in d) == (fail == "hang")



def test_bench_attention_record(dev):
    """The bench record's `attention` entry: forward and backward of a block's attention at the step's head
    shape against the HBM roofline that bounds them — bytes, us, TB/s, the fraction of 8 TB/s."""
    sys.path.insert(0, ROOT)
    import bench
    rec = bench.time_attention(8, iters=3)
    assert rec["bound"] == "hbm" and rec["shape"] == {"batch": 8, "tokens": 256, "heads": 16, "head_dim": 72}
    for k in ("forward", "backward"):
        r = rec[k]
        assert r["us"] > 0 and r["bytes"] > 0 and 0 < r["frac_of_8_TB_per_s"] < 1 and abs(r["TB_per_s"] / 8 - r["frac_of_8_TB_per_s"]) < 1e-3
    assert rec["forward"]["bytes"] == 4 * 8 * 256 * 1152 * 2
