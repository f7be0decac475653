"""CPU-only tests of the host side: the C-ABI library loads and exports every symbol of include/reed_hip.h,
the SiT module keeps the reference's state_dict surface and init, arena/bucket bookkeeping, schedules, the
data-parallel plan (gloo, world_size 2), CLI flag surfaces, and loud failure without a GPU."""
import copy
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import sit as osit
from tests.test_oracle_golden import load, tiny_cfg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    from reed_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 40
    # the product library, its IEEE-half build (the sampling path) and its fp32-operand build (--mixed-precision no)
    for prec, kind in (("bf16", 0), ("fp16", 1), ("fp32", 2)):
        lib = _lib.load(prec)
        assert lib._reed_missing == [], lib._reed_missing
        assert lib.reed_version() == 100 and lib.reed_half_kind() == kind
        for name in protos:
            assert hasattr(lib, name)


def test_arg_errors_do_not_need_a_gpu():
    from reed_amd import _lib
    lib = _lib.load()
    rc = lib.reed_attention_fwd(None, None, None, 1, 16, 2, 48, None)
    assert rc == 1001
    assert b"null pointer" in lib.reed_last_error() or b"head_dim" in lib.reed_last_error()
    rc = lib.reed_adamw_ema(1, 1, 1, 1, None, None, 6, 8, None, None, 1e-4, 0.9, 0.999, 1e-8, 0.0, 0.1, 0.1, 0.9999, None)
    assert rc == 1001 and b"multiples of 4" in lib.reed_last_error()


def test_state_dict_surface_matches_reference():
    from reed_amd.models.sit import SiT, SiT_models
    g = load("init")
    m = SiT_models["SiT-S/2"](z_dims=[768])
    sd = m.state_dict()
    assert sorted(sd.keys()) == list(g["s2.keys"])
    assert [str(tuple(sd[k].shape)) for k in sorted(sd.keys())] == list(g["s2.shapes"])
    assert not m.pos_embed.requires_grad and m.blocks[0].attn.qkv.weight.requires_grad
    assert sum(p.numel() for p in m.parameters()) == 39_515_664          # SURVEY.md BASELINE §2
    kw = tiny_cfg(z_dims=[128, 256], z_types=["i", "t"])
    t = SiT(input_size=8, hidden_size=128, decoder_hidden_size=128, depth=3, num_heads=2, num_classes=10,
            z_dims=[128, 256], z_types=["i", "t"], encoder_depth=2, projector_dim=128)
    assert sorted(t.state_dict().keys()) == list(g["tiny.keys"])
    assert list(osit.param_shapes(kw).keys()) == [k for k in osit.param_shapes(kw)]  # oracle uses the same names
    assert set(osit.param_shapes(kw)) == set(t.state_dict().keys())
    assert len(SiT_models) == 12 and "SiT-XL/2" in SiT_models
    # qk_norm=True adds timm's q_norm / k_norm LayerNorm(head_dim) parameters, initialised to (1, 0), adjacent in the arena
    q = SiT(input_size=8, hidden_size=128, decoder_hidden_size=128, depth=2, num_heads=2, num_classes=10, z_dims=[],
            qk_norm=True)
    kq = tiny_cfg(qk_norm=True, depth=2, z_dims=[], z_types=[])
    assert set(osit.param_shapes(kq)) == set(q.state_dict().keys())
    assert float(q.blocks[1].attn.q_norm.weight.sum()) == 64.0 and float(q.blocks[1].attn.k_norm.bias.abs().sum()) == 0.0
    L = q._layout
    assert L.off("blocks.0.attn.q_norm.bias") == L.off("blocks.0.attn.q_norm.weight") + 64
    assert L.off("blocks.0.attn.k_norm.bias") == L.off("blocks.0.attn.q_norm.weight") + 192


def test_init_matches_reference_rng_stream():
    """torch.manual_seed(s); SiT(...) reproduces the reference's initial weights (sit.py:217-254) bit for bit."""
    from reed_amd.models.sit import SiT, SiT_models
    g = load("init")
    for tag, build in (("s2", lambda: SiT_models["SiT-S/2"](z_dims=[768])),
                       ("tiny", lambda: SiT(input_size=8, hidden_size=128, decoder_hidden_size=128, depth=3,
                                            num_heads=2, num_classes=10, z_dims=[128, 256], z_types=["i", "t"],
                                            encoder_depth=2, projector_dim=128))):
        torch.manual_seed(1234)
        sd = build().state_dict()
        for k in ("x_embedder.proj.weight", "t_embedder.mlp.0.weight", "t_embedder.mlp.2.weight",
                  "y_embedder.embedding_table.weight", "blocks.0.attn.qkv.weight", "blocks.2.mlp.fc2.weight",
                  "projectors.0.4.weight", "blocks.1.adaLN_modulation.1.weight", "final_layer.linear.weight"):
            assert np.array_equal(sd[k].flatten()[:32].numpy(), g[f"{tag}.{k}"]), (tag, k)
            assert float(sd[k].double().sum()) == float(g[f"{tag}.sum.{k}"]), (tag, k)


def test_pos_embed_and_unpatchify_bit_exact():
    from reed_amd.models.sit import SiT, get_2d_sincos_pos_embed
    g = load("static")
    assert np.array_equal(get_2d_sincos_pos_embed(384, 16).astype(np.float32), g["pos_embed_384"])
    assert np.array_equal(get_2d_sincos_pos_embed(1152, 16).astype(np.float32)[::17], g["pos_embed_1152_rows"])
    m = SiT(input_size=32, hidden_size=128, decoder_hidden_size=128, depth=1, num_heads=2, z_dims=[])
    un = m.unpatchify(torch.arange(256 * 16, dtype=torch.float32).reshape(1, 256, 16)).long().numpy()
    assert np.array_equal(un, g["unpatchify_idx"])
    assert np.array_equal(m.pos_embed[0].numpy()[:, :64], get_2d_sincos_pos_embed(128, 16).astype(np.float32)[:, :64])


def test_arena_views_deepcopy_and_load():
    from reed_amd.models.sit import SiT
    m = SiT(input_size=8, hidden_size=128, decoder_hidden_size=128, depth=2, num_heads=2, num_classes=10, z_dims=[128],
            projector_dim=128)
    base = m._arena.master
    for n, p in m.named_parameters():
        off = m._layout.off(n)
        assert p.data_ptr() == base.data_ptr() + 4 * off, n
    # adaLN rows of all blocks + final layer are one contiguous [N_all, D] matrix
    L = m._layout
    assert L.ada_rows == 2 * 6 * 128 + 2 * 128
    assert L.off("blocks.1.adaLN_modulation.1.weight") == L.ada_w_off + 6 * 128 * 128
    assert L.off("final_layer.adaLN_modulation.1.weight") == L.ada_w_off + 2 * 6 * 128 * 128
    assert L.n_train % 64 == 0 and L.off("pos_embed") == L.n_train
    e = copy.deepcopy(m)
    assert e._arena.master.data_ptr() != base.data_ptr()
    assert e.blocks[1].mlp.fc1.weight.data_ptr() == e._arena.master.data_ptr() + 4 * L.off("blocks.1.mlp.fc1.weight")
    sd = {k: torch.full_like(v, 0.5) for k, v in m.state_dict().items()}
    e.load_state_dict(sd)
    assert float(e._arena.master[L.off("blocks.0.attn.proj.bias")]) == 0.5
    assert float(m._arena.master[L.off("blocks.0.attn.proj.bias")]) == 0.0
    e.requires_grad_(False)
    assert not any(p.requires_grad for p in e.parameters())


def test_bucket_plan_covers_every_parameter_once():
    from reed_amd.models.sit import SiT_models
    from reed_amd.parallel import check_buckets
    for name, kw in (("SiT-XL/2", dict(z_dims=[1024])), ("SiT-S/2", dict(z_dims=[])),
                     ("SiT-B/2", dict(z_dims=[768, 3584], z_types=["i", "t"], encoder_depth_text=8))):
        torch.manual_seed(0)
        L = SiT_models[name].__call__.__self__ if False else None  # noqa
    from reed_amd.arena import ArenaLayout
    for cfg in (osit.make_config("SiT-XL/2", z_dims=[1024]), osit.make_config("SiT-S/2", z_dims=[], z_types=[]),
                osit.make_config("SiT-B/2", z_dims=[768, 3584], z_types=["i", "t"])):
        shapes = osit.param_shapes(cfg)
        L = ArenaLayout(shapes, cfg["depth"], len(cfg["z_dims"]))
        bk = check_buckets(L)
        names = [b[0] for b in bk]
        d = cfg["depth"]
        assert names[:4] == ["final", f"ada{d}", f"block{d - 1}", f"ada{d - 1}"] and names[-1] == "embed"
        covered = sum(e - b for _, (b, e) in bk)
        assert covered <= L.n_train
    # XL/2: 683,472,144 trainable elements = the gradient all-reduce payload
    cfg = osit.make_config("SiT-XL/2", z_dims=[1024])
    shapes = osit.param_shapes(cfg)
    assert sum(int(np.prod(s)) for k, s in shapes.items() if k != "pos_embed") == 683_472_144


def test_schedules_match_oracle():
    from oracle import train_step as otrain
    from reed_amd import trainer
    for kind in ("constant", "linear", "cosine"):
        for s in (0, 1, 999, 400000, 500000):
            assert trainer.repa_weight_decay(kind, s, 400000) == otrain.repa_weight_decay(kind, s, 400000)
            for start, warm in ((0, 50000), (1000, 10), (0, 1)):
                if s >= start + warm and kind == "linear" and 400000 - (start + warm) == 0:
                    continue
                assert trainer.diffusion_loss_decay(kind, s, start, warm, 400000) == \
                    otrain.diffusion_loss_decay(kind, s, start, warm, 400000)


def test_index_and_seed_rules():
    from reed_amd.parallel import rank_seed, sample_index, sample_seed, shard_batch
    assert shard_batch(256, 8) == 32 and shard_batch(256, 1) == 256 and shard_batch(100, 8) == 12
    assert rank_seed(0, 3) == 3
    assert sample_seed(2, 8, 5) == 21
    # indices over 2 iterations of n=3 on 4 ranks are a permutation of 0..23 (generate.py:164)
    idx = sorted(sample_index(i, 4, r, it * 12) for it in range(2) for r in range(4) for i in range(3))
    assert idx == list(range(24))


def test_cli_flag_surfaces():
    from reed_amd import generate, train
    a = train.parse_args(["--exp-name", "x", "--model", "SiT-XL/2"])
    ref_defaults = dict(output_dir="exps", logging_dir="logs", report_to="wandb", sampling_steps=10000, resume_step=0,
                        num_classes=1000, encoder_depth=8, encoder_depth_text=None, fused_attn=True, qk_norm=False,
                        data_dir="../data/imagenet256", resolution=256, batch_size=256, allow_tf32=False,
                        mixed_precision="fp16", epochs=1400, max_train_steps=400000, checkpointing_steps=50000,
                        gradient_accumulation_steps=1, learning_rate=1e-4, adam_beta1=0.9, adam_beta2=0.999,
                        adam_weight_decay=0.0, adam_epsilon=1e-8, max_grad_norm=1.0, seed=0, num_workers=4,
                        path_type="linear", prediction="v", cfg_prob=0.1, enc_type="dinov2-vit-b", proj_coeff=0.5,
                        weighting="uniform", legacy=False, time_schedule="constant", repa_coeff=[1.0],
                        cutoffs=[0.0, 1.0], cfg=True, text_embeds_dir=None, repa_weight_decay="constant",
                        repa_steps=400000, start_diffusion_steps=0, diffusion_warm_up_steps=50000,
                        diffusion_decay="constant")
    for k, v in ref_defaults.items():
        assert getattr(a, k) == v, k
    a = train.parse_args(["--exp-name", "x", "--no-fused-attn", "--qk-norm", "--repa-coeff", "1.0", "0.5", "--no-cfg"])
    assert a.fused_attn is False and a.qk_norm is True and a.repa_coeff == [1.0, 0.5] and a.cfg is False
    assert train.encoder_specs("dinov2-vit-l,clip-vit-L") == (["dinov2", "clip"], [1024, 1024])
    assert train.encoder_specs("None") == ([], [])
    g = generate.build_parser().parse_args([])
    gd = dict(global_seed=0, tf32=True, ckpt=None, sample_dir="samples", model="SiT-XL/2", num_classes=1000,
              encoder_depth=8, resolution=256, fused_attn=False, qk_norm=False, vae="ema", per_proc_batch_size=32,
              num_fid_samples=50000, mode="ode", cfg_scale=1.5, projector_embed_dims="768", path_type="linear",
              num_steps=50, heun=False, guidance_low=0.0, guidance_high=1.0, legacy=False, prediction="v")
    for k, v in gd.items():
        assert getattr(g, k) == v, k
    g.ckpt = "exps/run/checkpoints/0400000.pt"
    assert generate.folder_name(g) == "SiT-XL-2-0400000-size-256-vae-ema-cfg-1.5-seed-0-ode"
    g.guidance_high = 0.7
    assert generate.folder_name(g).endswith("-ode-cfg-high-0.7")
    sd = {"decoder_blocks.2.attn.qkv.weight": 1, "blocks.0.x": 2}
    assert generate.load_legacy_checkpoints(sd, 8) == {"blocks.10.attn.qkv.weight": 1, "blocks.0.x": 2}


def test_no_cpu_fallback():
    from reed_amd.loss import SILoss
    from reed_amd.models.sit import SiT
    from reed_amd.samplers import euler_sampler
    m = SiT(input_size=8, hidden_size=128, decoder_hidden_size=128, depth=1, num_heads=2, num_classes=10, z_dims=[])
    x, t, y = torch.zeros(2, 4, 8, 8), torch.zeros(2), torch.zeros(2, dtype=torch.long)
    with pytest.raises(RuntimeError, match="no CPU"):
        m(x, t, y)
    with pytest.raises(RuntimeError, match="no CPU"):
        SILoss(enc_names=[], loss_weights={})(m, x, dict(y=y), zs=[])
    with pytest.raises(RuntimeError, match="no CPU"):
        euler_sampler(m, x, y, num_steps=2)
    with pytest.raises(ValueError, match="decoder_hidden_size"):
        SiT(hidden_size=384, decoder_hidden_size=768, num_heads=6)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under reed_amd/ may import it."""
    for root, _d, files in os.walk(os.path.join(ROOT, "reed_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


_GLOO_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from oracle import sit as osit
from reed_amd.arena import ArenaLayout
from reed_amd.parallel import TorchDistGradReducer, shard_batch, rank_seed
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
cfg = osit.make_config("SiT-S/2", z_dims=[768])
L = ArenaLayout(osit.param_shapes(cfg), cfg["depth"], 1)
torch.manual_seed(rank_seed(0, rank))
grad = torch.randn(L.n_train)
mine = grad.clone()
red = TorchDistGradReducer(L, grad, world)
d = cfg["depth"]
order = ["final", f"ada{d}"]
for i in reversed(range(d)):
    order += [f"block{i}", f"ada{i}"]
    if i + 1 == cfg["encoder_depth"]: order.append("projectors")
order.append("embed")
for name in order:            # the order in which Engine.backward fires buckets
    red.ready(name)
red.sync()
# reference result: plain all-reduce(avg) of the whole arena
ref = mine.clone(); dist.all_reduce(ref); ref /= world
covered = torch.zeros(L.n_train, dtype=torch.bool)
for b, e in red.buckets.values(): covered[b:e] = True
assert torch.equal(grad[covered], ref[covered]), "bucketed all-reduce(avg) != whole-arena all-reduce(avg)"
assert torch.equal(grad[~covered], mine[~covered])
for name, (off, shp) in L.seg.items():
    if name != "pos_embed": assert covered[off], name
assert shard_batch(256, world) == 128
dist.barrier(); dist.destroy_process_group()
print("OK", rank)
'''


def test_gloo_world2_bucketed_allreduce(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o


_GLOO_FACTOR_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from oracle import sit as osit
from reed_amd.arena import ArenaLayout
from reed_amd.parallel import TorchDistGradReducer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
cfg = osit.make_config("SiT-S/2", z_dims=[768])
L = ArenaLayout(osit.param_shapes(cfg), cfg["depth"], 1)
D, d, Nall, B = cfg["hidden_size"], cfg["depth"], L.ada_rows, 4
torch.manual_seed(100 + rank)
dmod = torch.randn(B, Nall).to(torch.bfloat16)          # this rank's factors (what Engine.backward holds)
silu = torch.randn(B, D).to(torch.bfloat16)
grad = torch.randn(L.n_train)
# reference: the all-reduce path — local adaLN gradient, then all-reduce(avg) of every bucket
ref = grad.clone()
ref[L.ada_w_off:L.ada_w_off + Nall * D] = (dmod.float().t() @ silu.float()).reshape(-1)
ref[L.ada_b_off:L.ada_b_off + Nall] = dmod.float().sum(0)
dist.all_reduce(ref); ref /= world
# factor path, in the engine's order: gather silu(c); per block (backward order) scale + pack + gather the dmod rows and
# fire the block's bucket; at the end the global-batch products, then the embed bucket WITHOUT the adaLN biases
red = TorchDistGradReducer(L, grad, world)
def gather(x):
    out = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(out, x.contiguous())
    return torch.cat(out, 0)                           # rank-major [world * b, ...]
s_all = gather(silu)
recv = {}
red.ready("final")
for i in reversed(range(d + 1)):
    c0, rows = i * 6 * D, (6 * D if i < d else 2 * D)
    recv[i] = gather(dmod[:, c0:c0 + rows] * (1.0 / world))
    if i < d:
        red.ready(f"block{i}")
        if i + 1 == cfg["encoder_depth"]: red.ready("projectors")
for i in reversed(range(d + 1)):
    c0, rows = i * 6 * D, (6 * D if i < d else 2 * D)
    assert recv[i].dtype == torch.bfloat16 and recv[i].shape == (world * B, rows)
    grad[L.ada_w_off + c0 * D:L.ada_w_off + (c0 + rows) * D] = (recv[i].float().t() @ s_all.float()).reshape(-1)
    grad[L.ada_b_off + c0:L.ada_b_off + c0 + rows] = recv[i].float().sum(0)
eb, ee = red.buckets["embed"]
assert eb == L.ada_b_off
dist.all_reduce(grad[L.ada_b_off + Nall:ee]); grad[L.ada_b_off + Nall:ee] /= world
covered = torch.zeros(L.n_train, dtype=torch.bool)
for b, e in red.buckets.values(): covered[b:e] = True
# 1/world is a power of two: the scaled bf16 factors are exact, so the two paths differ by fp32 summation order only
torch.testing.assert_close(grad[covered], ref[covered], rtol=1e-5, atol=1e-4)
every = torch.cat([torch.empty_like(grad) for _ in range(world)]).view(world, -1)
dist.all_gather(list(every.unbind(0)), grad)
assert all(torch.equal(every[0][covered], every[r][covered]) for r in range(world)), "ranks disagree"
dist.barrier(); dist.destroy_process_group()
print("OK", rank)
'''


def test_gloo_world2_adaln_factor_gather(tmp_path):
    """The factor path of Engine.backward (reed_amd/parallel.py:GradReducer.gather): all-gathering the 1/world-scaled
    bf16 dmod rows and silu(c) and forming the global-batch product on every rank == all-reduce(avg) of the per-rank
    adaLN gradients (weights and biases), every other bucket reduced as before, identical results on every rank."""
    script = tmp_path / "wf.py"
    script.write_text(_GLOO_FACTOR_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29733", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o


def test_update_chunks_cover_arena_in_forward_order():
    """ArenaLayout.update_chunks (overlapped optimiser): the chunks tile [0, n_total) exactly once, start with the
    embedders + the head of the adaLN matrix, visit the blocks in forward order with the adaLN tail behind the head
    blocks and the projectors right behind the tap block, and end with the final layer (+ the frozen pos_embed, which
    only the EMA pass touches)."""
    from reed_amd.models.sit import SiT
    m = SiT(input_size=8, hidden_size=128, decoder_hidden_size=128, depth=6, num_heads=2, num_classes=10, z_dims=[128],
            z_types=["i"], encoder_depth=2, projector_dim=128)
    L = m._layout
    ch = L.update_chunks([2])
    names = [n for n, _, _ in ch]
    assert names == ["embed", "ada_head", "block0", "block1", "projectors", "block2", "block3", "ada_tail", "block4",
                     "block5", "final"]
    spans = sorted((b, e) for _, b, e in ch)
    assert spans[0][0] == 0 and spans[-1][1] == L.n_total
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    bk = dict(L.buckets())
    by = {n: (b, e) for n, b, e in ch}
    for n in ("block0", "block5", "projectors", "final"):   # every such chunk contains its all-reduce bucket
        assert by[n][0] <= bk[n][0] and bk[n][1] <= by[n][1]
    D = 128
    assert by["ada_head"] == (0, 4 * 6 * D * D) and by["ada_tail"] == (4 * 6 * D * D, L.ada_b_off)
    assert L.update_chunks(())[-1][0] == "final"
    m2 = SiT(input_size=8, hidden_size=128, decoder_hidden_size=128, depth=3, num_heads=2, num_classes=10, z_dims=[],
             z_types=[], encoder_depth=2, projector_dim=128)
    n2 = [n for n, _, _ in m2._layout.update_chunks(())]
    assert n2 == ["embed", "ada_head", "block0", "block1", "block2", "ada_tail", "final"]


def _tiny_dataset(tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(ROOT, "tools", "gen_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    root = str(tmp_path / "tinyds")
    os.makedirs(root)
    mod.make_tiny_dataset(root)
    return root


def test_dataset_and_packed_dataset_match_reference_bit_exact(tmp_path):
    """Data path: `CustomDataset` on the reference's on-disk format returns, item by item and field
    by field, exactly what the reference's own loader returned (tests/golden/dataset.npz, image/dataset.py:18-85 run by
    tools/gen_golden.py on the same bytes): image bytes, moments, label from the un-sorted dataset.json, text embedding
    or its zeros_like(moments) stand-in. `pack_dataset` + `PackedDataset` return the same items again, with and without
    images / text / precomputed features."""
    from reed_amd.dataset import CustomDataset, PackedDataset, pack_dataset
    g = np.load(os.path.join(ROOT, "tests", "golden", "dataset.npz"))
    root = _tiny_dataset(tmp_path)
    for tag, td in (("plain", None), ("text", "text_embeds_t")):
        ds = CustomDataset(root, text_embeds_dir=td)
        meta = pack_dataset(root, str(tmp_path / f"packed_{tag}"), text_embeds_dir=td)
        pk = PackedDataset(str(tmp_path / f"packed_{tag}"))
        assert len(ds) == len(pk) == int(g[tag + ".len"]) == meta["n"]
        for i in range(len(ds)):
            for item in (ds[i], pk[i]):
                im, mo, la, tx = item[:4]
                assert im.dtype == torch.uint8 and np.array_equal(im.numpy(), g[f"{tag}.{i}.image"])
                assert mo.dtype == torch.float32 and np.array_equal(mo.numpy(), g[f"{tag}.{i}.moments"])
                assert la.dtype == torch.int64 and int(la) == int(g[f"{tag}.{i}.label"])
                assert np.array_equal(tx.numpy(), g[f"{tag}.{i}.text"]) and tx.dtype == torch.float32
    # precomputed features, no images: packed == unpacked
    import shutil
    fz = os.path.join(root, "feat_a")
    for i in range(6):
        os.makedirs(os.path.join(fz, f"{i // 4:05d}"), exist_ok=True)
        np.save(os.path.join(fz, f"{i // 4:05d}", f"img{i:08d}.npy"), np.full((4, 8), float(i), dtype=np.float32))
    ds = CustomDataset(root, features_dirs=["feat_a"], need_images=False)
    pack_dataset(root, str(tmp_path / "packed_z"), features_dirs=["feat_a"], with_images=False)
    pk = PackedDataset(str(tmp_path / "packed_z"))
    assert "images" not in pk.arr and pk.zkeys == ["z0"]
    for i in range(6):
        a, b = ds[i], pk[i]
        assert len(a) == len(b) == 5 and a[0].numel() == b[0].numel() == 0
        for x, y in zip(a[1:], b[1:]):
            assert x.dtype == y.dtype and torch.equal(x, y)
        assert float(a[4][0, 0]) == float(i)
    shutil.rmtree(root)


def _gloo_rsag_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from reed_amd.parallel import GradReducer
        red = GradReducer.__new__(GradReducer)          # the collective forms only: no model, no GPU
        red.rank, red.world, red._rccl = rank, world, False
        res = {}
        for algo in ("allreduce", "rsag"):
            red.algo = algo
            for n in (1, 7, 4096, 4099):                # below one element per rank, ragged tails, whole multiples
                t = torch.arange(n, dtype=torch.float32) * (rank + 1) + rank
                red._t_allreduce_avg(t)
                res[(algo, n)] = t.clone()
        ret[rank] = res
    finally:
        dist.destroy_process_group()


def test_gloo_world4_reduce_scatter_allgather_form():
    """REED_COMM_ALGO=rsag (each bucket as reduce-scatter + all-gather, tail through all-reduce) on FOUR ranks over gloo:
    same averages as the plain all-reduce form for bucket sizes below / not divisible by / divisible by the world size,
    identical on every rank."""
    import torch.multiprocessing as mp
    world = 4
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=_gloo_rsag_worker, args=(r, world, 29771, ret)) for r in range(world)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(180)
        assert p.exitcode == 0
    for n in (1, 7, 4096, 4099):
        exp = sum((torch.arange(n, dtype=torch.float32) * (r + 1) + r) for r in range(world)) / world
        for r in range(world):
            for algo in ("allreduce", "rsag"):
                torch.testing.assert_close(ret[r][(algo, n)], exp, rtol=1e-6, atol=1e-6)
            assert torch.equal(ret[r][("rsag", n)], ret[0][("rsag", n)])


def test_sd_vae_decoder_two_restatements_agree(tmp_path):
    """reed_amd/vae.py (torch modules, diffusers key names) vs oracle/vae.py (numpy fp64 walk over the checkpoint keys) on
    random weights: a reduced config end to end, then the published sd-vae-ft config's key surface and a round trip through
    a checkpoint file with the legacy attention names. PARITY UNPINNED vs diffusers itself (absent here): both files say so."""
    from oracle import vae as ovae
    from reed_amd import vae as rvae
    torch.manual_seed(0)
    small = rvae.SDVAEDecoder(block_out_channels=(16, 32, 32), layers_per_block=1, norm_num_groups=8).double()
    for p in small.parameters():
        p.data.normal_(0, 0.15)
    z = torch.randn(2, 4, 5, 6, dtype=torch.float64)
    got = small.decode_torch(z).numpy()
    want = ovae.Decoder({k: v.numpy() for k, v in small.state_dict().items()}, groups=8).decode(z.numpy())
    assert got.shape == (2, 3, 20, 24)          # two upsamplers for three blocks: x4
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)
    # the published configuration: key names / shapes of the decoder half of sd-vae-ft-{ema,mse}
    full = rvae.SDVAEDecoder()
    sd = full.state_dict()
    assert sd["decoder.conv_in.weight"].shape == (512, 4, 3, 3) and sd["post_quant_conv.weight"].shape == (4, 4, 1, 1)
    assert sd["decoder.up_blocks.2.resnets.0.conv_shortcut.weight"].shape == (256, 512, 1, 1)
    assert sd["decoder.up_blocks.3.resnets.0.conv_shortcut.weight"].shape == (128, 256, 1, 1)
    assert "decoder.up_blocks.3.upsamplers.0.conv.weight" not in sd and "decoder.up_blocks.2.upsamplers.0.conv.weight" in sd
    assert sd["decoder.mid_block.attentions.0.to_q.weight"].shape == (512, 512)
    assert sd["decoder.conv_out.weight"].shape == (3, 128, 3, 3)
    assert sum(v.numel() for v in sd.values()) == 49_490_199   # decoder 49,490,179 + post_quant_conv 20
    # a checkpoint with encoder keys and the legacy attention names (conv-shaped weights) loads into the same module
    legacy = {"encoder.conv_in.weight": torch.zeros(1), "quant_conv.weight": torch.zeros(1)}
    ren = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
    for k, v in small.state_dict().items():
        for new, old in ren.items():
            if f".attentions.0.{new}." in k:
                k = k.replace(f".{new}.", f".{old}.")
                if k.endswith("weight"):
                    v = v[:, :, None, None]
        legacy[k] = v.float()
    path = str(tmp_path / "diffusion_pytorch_model.bin")
    torch.save(legacy, path)
    re = rvae.load_sd_vae_decoder(str(tmp_path), block_out_channels=(16, 32, 32), layers_per_block=1, norm_num_groups=8)
    np.testing.assert_allclose(re.decode_torch(z.float()).numpy(), got, rtol=2e-4, atol=2e-4)


def test_mocov3_checkpoint_key_repair(tmp_path, monkeypatch):
    """image/utils.py:27-52 fix_mocov3_state_dict: the published MoCo-v3 ViT-L file stores blocks.13.norm1 / mlp.fc1 as norm13 /
    fc13 and blocks.14.norm2 / mlp.fc2 as norm14 / fc14, under `module.base_encoder.`, beside a momentum encoder, a predictor
    and a projection head.  The loader must repair the four names, drop the rest, and fill every tower parameter."""
    from reed_amd import encoders
    assert encoders.mocov3_key("module.base_encoder.blocks.13.norm13.weight") == "blocks.13.norm1.weight"
    assert encoders.mocov3_key("module.base_encoder.blocks.13.mlp.fc13.bias") == "blocks.13.mlp.fc1.bias"
    assert encoders.mocov3_key("module.base_encoder.blocks.14.norm14.bias") == "blocks.14.norm2.bias"
    assert encoders.mocov3_key("module.base_encoder.blocks.14.mlp.fc14.weight") == "blocks.14.mlp.fc2.weight"
    assert encoders.mocov3_key("module.base_encoder.blocks.13.norm2.weight") == "blocks.13.norm2.weight"
    assert encoders.mocov3_key("module.base_encoder.blocks.3.norm1.weight") == "blocks.3.norm1.weight"
    assert encoders.mocov3_key("module.base_encoder.head.0.weight") is None
    assert encoders.mocov3_key("module.momentum_encoder.blocks.0.norm1.weight") is None
    assert encoders.mocov3_key("module.predictor.0.weight") is None
    assert encoders.mocov3_key("blocks.2.attn.qkv.weight") == "blocks.2.attn.qkv.weight"     # a plain state dict passes
    tiny = dict(embed=128, depth=15, heads=2, patch=16, image=64, cls=True, final_norm=True)
    monkeypatch.setitem(encoders.VIT_TOWERS, "mocov3-vit-l", tiny)
    ref = encoders.VitEncoder(**tiny)
    g = torch.Generator().manual_seed(3)
    want = {k: torch.randn(v.shape, generator=g) for k, v in ref.state_dict().items()}
    bad = {"blocks.13.norm1": "blocks.13.norm13", "blocks.13.mlp.fc1": "blocks.13.mlp.fc13",
           "blocks.14.norm2": "blocks.14.norm14", "blocks.14.mlp.fc2": "blocks.14.mlp.fc14"}
    ck = {}
    for k, v in want.items():
        for good, wrong in bad.items():
            if k.startswith(good + "."):
                k = wrong + k[len(good):]
        ck["module.base_encoder." + k] = v
        ck["module.momentum_encoder." + k] = torch.zeros_like(v)
    ck["module.base_encoder.head.0.weight"] = torch.zeros(4, 128)
    ck["module.predictor.0.weight"] = torch.zeros(4, 4)
    path = str(tmp_path / "mocov3_vitl.pth")
    torch.save({"state_dict": ck, "epoch": 300}, path)
    enc = encoders.load_vit_encoder("mocov3-vit-l", path, "cpu")
    got = enc.state_dict()
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k


def test_sharded_optimizer_plan_covers_arena_and_balances():
    """REED_OPT_SHARD (reed_amd/optim.py:_shard_plan): every update chunk cut into `world` equal 4-aligned pieces in rank order
    (what one in-place all-gather completes) plus a replicated tail shorter than 4 * world elements — every element of the arena
    in exactly one piece, chunks in next-forward order, every rank the same share, the same plan on every rank."""
    from reed_amd.models.sit import SiT_models
    from reed_amd.optim import FusedAdamWEMA
    m = SiT_models["SiT-B/2"](z_dims=[768], z_types=["i"], encoder_depth=4)
    L = m._layout
    opt = FusedAdamWEMA.__new__(FusedAdamWEMA)
    opt._chunks = L.update_chunks([4])
    for world in (2, 4, 8):
        plans = [opt._shard_plan(L, world=world, rank=r) for r in range(world)]
        assert all(p == plans[0] for p in plans)
        covered = np.zeros(L.n_total, dtype=np.int32)
        load = [0] * world
        for (name, subs), (cname, cb, ce) in zip(plans[0], opt._chunks):
            assert name == cname and subs[0][0] == cb and subs[-1][1] == ce
            own = [s_ for s_ in subs if s_[2] >= 0]
            assert [o for _, _, o in own] in ([], list(range(world)))
            assert len({e - b for b, e, _ in own}) <= 1                       # equal pieces ...
            assert all(own[i][1] == own[i + 1][0] for i in range(len(own) - 1))   # ... back to back in rank order
            for b, e, o in subs:
                assert b % 4 == 0 and e > b
                covered[b:e] += 1
                if o >= 0:
                    load[o] += e - b
                else:
                    assert e - b < 4 * world + 4 and (b, e, o) == subs[-1]
        assert len(plans[0]) == len(opt._chunks) and (covered == 1).all()
        assert max(load) == min(load) and sum(load) > 0.99 * L.n_total


def test_bench_launcher_argv(monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment starts its own ranks: the child
    command is torch.distributed.run with one rank per GPU, rendezvous on 127.0.0.1, this script and its arguments unchanged; with
    fewer devices than ranks the launcher returns 2 without starting anything (and without touching the GPU)."""
    import bench
    argv = ["--gpus", "8", "--steps", "7", "--warmup", "2"]
    args = bench.parse(argv)
    cmd = bench.launcher_argv(args, argv, 29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    k = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[k + 1:] == argv
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    started = []
    import subprocess
    monkeypatch.setattr(subprocess, "Popen", lambda *a, **k: started.append(a) or (_ for _ in ()).throw(AssertionError("started")))
    assert bench.self_launch(args, argv) == 2 and not started


def test_counted_waits_of_the_attention_forward_match_the_isa():
    """attn_fwd256p_kernel and (round 6) attn_fwd256v_kernel wait with s_waitcnt vmcnt(N) for immediates derived from how many
    vector-memory instructions a wave issues per item: compile csrc/attention.hip to gfx950 ISA (hipcc cross-compiles here) and count
    them in every instantiation — 20 + 7 before the item loop and 15 + 7 per item (attn_fwd256p_kernel), 15 + 8 before, 15 + 7 per
    item and 7 behind the loop (attn_fwd256v_kernel), the waits in the expected order, no scratch traffic (tools/check_attn_isa.py;
    the compiler once merged seven identical prologue stores into one, which left the first item's waits six operations short)."""
    import subprocess
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_attn_isa.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    # hd 64 / 72 / 80 x (T = 256, T < 256) + the V-double-buffered 64 / 72 + the ring backward's own-row registers (hd 64 / 72)
    assert r.stdout.count("ok  ") == 10 and "BAD" not in r.stdout, r.stdout


def test_detfill_matches_the_fixture_recipe():
    """reed_amd/detfill.py (what bench.py's loss_vs_ref leg rebuilds the C2 fixture's weights and inputs with; the product imports
    nothing of oracle/) is the recipe the fixtures were written with (oracle/detfill.py, tools/gen_golden.py:inputs) bit for bit."""
    import torch
    from oracle import detfill as od
    from reed_amd import detfill as pd
    from tests.test_oracle_golden import inputs
    for shape, seed in (((7,), 3), ((4, 5, 6), 12345), ((2, 256, 32), 6017)):
        assert torch.equal(od.uniform(shape, seed, -0.3, 0.7), pd.uniform(shape, seed, -0.3, 0.7))
        assert torch.equal(od.normal(shape, seed), pd.normal(shape, seed))
    names = {"pos_embed": (1, 4, 8), "blocks.0.attn.qkv.weight": (24, 8), "blocks.0.attn.qkv.bias": (24,),
             "blocks.1.adaLN_modulation.1.weight": (48, 8), "final_layer.linear.weight": (16, 8),
             "y_embedder.embedding_table.weight": (11, 8), "blocks.0.attn.q_norm.weight": (4,), "x_embedder.proj.weight": (8, 4, 2, 2)}
    a = od.fill_state_dict({k: torch.zeros(v) for k, v in names.items()}, base_seed=5)
    b = pd.fill_model({k: torch.zeros(v) for k, v in names.items()}, base_seed=5)
    for k in names:
        assert torch.equal(a[k], b[k]), k
    ra = inputs(3, 4, 32, 2, [(64, "i")], 256, 1000)
    rb = pd.step_inputs(3, 2, [64])
    for u, v in zip(ra[:5], rb[:5]):
        assert torch.equal(u, v)
    assert torch.equal(ra[5][0], rb[5][0])


def test_bench_refuses_unknown_switches():
    """A REED_* variable bench.py does not know (a typo, a switch of an older round) must stop the run before anything is timed."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], env=dict(os.environ, REED_GEMM_W4="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "REED_GEMM_W4" in r.stderr and not r.stdout.strip()


@pytest.mark.parametrize("shapes", [
    [(1152, 4608, 1), (4608, 1152, 1), (1152, 1152, 1), (3456, 1152, 1)],     # the SiT-XL/2 block: fc2, fc1, proj, qkv
    [(4608, 1152, 0), (1152, 4608, 1), (3456, 1152, 1), (1152, 1152, 0)],     # another order, some without a bias gradient
    [(1024, 4096, 1), (4096, 1024, 1), (1024, 1024, 1), (3072, 1024, 1)],     # SiT-L/2: no ragged edge at all (192 full tiles)
    [(1280, 5120, 1), (5120, 1280, 1), (1280, 1280, 1)],                      # no ragged edge; bias-only items do not all fit
])
def test_wgrad_group_deal_covers_every_gradient_element_once(shapes):
    """The one-workgroup-per-CU form of reed_wgrad_group (csrc/gemm256w.hip, round 6: the ragged edges as three-unit items that
    keep a full tile's pace instead of K-cut pieces): on the host, without a GPU — every element of every dw is written by
    exactly one item, every row's bias gradient by exactly one item, every XCD gets a contiguous eighth of the sequence and at
    most cus / 8 items, and an item sits on the XCD that holds full tiles of its rows (what the deal is for)."""
    from reed_amd import ops
    deal = ops.wgrad_group_deal(shapes, 256)
    assert deal is not None and len(deal) == 8 and all(len(r) <= 32 for r in deal)
    sizes = [len(r) for r in deal]
    assert max(sizes) - min(sizes) <= 1
    cover = [np.zeros((m // 128, n // 128), dtype=np.int32) for m, n, _ in shapes]
    bias = [np.zeros(m // 16, dtype=np.int32) for m, _, _ in shapes]
    for run in deal:
        for it in run:
            m, n, hb = shapes[it["p"]]
            r0, c0 = it["row"] // 128, it["col"] // 128
            assert it["row"] + it["rows"] <= m and it["col"] + it["cols"] <= n
            if it["mode"] != 8:
                cover[it["p"]][r0:r0 + it["rows"] // 128, c0:c0 + it["cols"] // 128] += 1
            if it["bias"]:
                assert hb
                # mode 0: the first tile column's extra MFMA; 6 / 8: the fourth wave(s) over the item's rows; 7: its 128 rows
                bias[it["p"]][it["row"] // 16:(it["row"] + it["rows"]) // 16] += 1
    for (m, n, hb), c, bsum in zip(shapes, cover, bias):
        assert (c == 1).all()
        assert (bsum == (1 if hb else 0)).all()
    # locality (the XL/2 block): a 384-row item shares an XCD with a full tile of one of its tile rows, a 384-column item with a
    # full tile of one of its tile columns — except where an XCD boundary falls right beside it (at most one item per boundary)
    if shapes[0] == (1152, 4608, 1):
        assert sum(sizes) == 256 and sum(1 for r in deal for it in r if it["mode"] == 0) == 212      # every CU has an item
        assert sum(1 for r in deal for it in r if it["mode"] == 6) == 24 and sum(1 for r in deal for it in r if it["mode"] == 7) == 18
        assert sum(1 for r in deal for it in r if it["mode"] == 8) == 2
        assert not any(it["bias"] for r in deal for it in r if it["mode"] == 0)      # no full tile carries a bias gradient
        lonely = 0
        for run in deal:
            full = [(it["p"], it["row"] // 256, it["col"] // 256) for it in run if it["mode"] == 0]
            for it in run:
                if it["mode"] == 6:
                    rows = {it["row"] // 256, (it["row"] + it["rows"] - 1) // 256}
                    lonely += not any(p == it["p"] and r in rows for p, r, _ in full)
                elif it["mode"] == 7:
                    cols = {it["col"] // 256, (it["col"] + it["cols"] - 1) // 256}
                    lonely += not any(p == it["p"] and c in cols for p, _, c in full)
        assert lonely <= 7
    # a CU reserve (collectives holding CUs) leaves too few CUs for one round: the form does not apply
    assert ops.wgrad_group_deal([(1152, 4608, 1), (4608, 1152, 1), (1152, 1152, 1), (3456, 1152, 1)], 224) is None
    assert ops.wgrad_group_deal([(384, 1536, 1), (1536, 384, 1), (384, 384, 1), (1152, 384, 1)], 256) is None    # SiT-S/2: too thin
