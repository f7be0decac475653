"""train.py — drop-in counterpart of the reference's image/train.py on the MI355X HIP path.

    python -m reed_amd.train --exp-name run --model SiT-XL/2 --enc-type None ...                      (1 GPU)
    python -m torch.distributed.run --nproc-per-node 8 -m reed_amd.train --exp-name run ...           (8 GPUs, RCCL)

Same flag surface as image/train.py:483-555 (every flag name, default and choice kept), same per-step arithmetic
(:349 sample_posterior, :363-385 schedules, :387-412 loss/clip/AdamW/EMA), same checkpoint dict
{"model","ema","opt","args","steps"} (:418-429) and resume rule (:280-291). The accelerate/DDP machinery is replaced
by reed_amd.parallel.GradReducer (RCCL over xGMI, bucketed, overlapped with backward); mixed precision is the
16-bit-MFMA/fp32-master scheme (`--mixed-precision bf16`, or "fp16" = IEEE-half operands + dynamic loss scaling) or, with
`--mixed-precision no`, fp32 operands and activations on the fp32 matrix instruction (the fp32-operand build of the library).

Deliberate fixes of reference defects: `--enc-type None` = alignment off (§9-4); the preview
sampling at step 1 / every --sampling-steps runs only with --vae-ckpt (a local SD-VAE checkpoint, reed_amd/vae.py) and writes
PNG grids instead of wandb images (§9-2); checkpoints and
args.json are written regardless of --report-to (§9-11); gradients are clipped once and the pre-clip norm is
logged (§9-5); unknown --text-embeds-dir names get their width from the first .npy (§9-9).
Additive flags: --features-dirs (precomputed frozen-encoder features, §8f N2), --encoder-ckpts (clip-vit-* encoder run
on the GPU every step from a user-supplied state dict, reed_amd/encoders.py), --synthetic N (random latents),
--log-every.
"""
import argparse
import copy
import datetime
import json
import logging
import os
import time

import numpy as np
import torch

TEXT_Z_DIM_DICT = {'text_embeds_qwenvl': 1536, 'text_embeds_open_clip': 1280, 'text_embeds_qwenvl_7b': 3584,
                   'text_embeds_qwenvl_7b_layer_0': 3584, 'text_embeds_qwenvl_7b_layer_1': 3584,
                   'text_embeds_qwenvl_7b_layer_15': 3584, 'text_embeds_qwenvl_2.5_3B': 2048,
                   'text_embeds_qwenvl_2.5_7B': 3584, 'text_embeds_qwenvl_2.5_7B_layer_15': 3584,
                   'text_embeds_qwenvl_2.5_7B_layer_1': 3584}   # train.py:40-43
ENC_EMBED_DIM = {"s": 384, "b": 768, "l": 1024, "g": 1536, "h": 1280}  # ViT widths by model_config letter


def parse_args(input_args=None):
    parser = argparse.ArgumentParser(description="Training")
    # logging:
    parser.add_argument("--output-dir", type=str, default="exps")
    parser.add_argument("--exp-name", type=str, required=True)
    parser.add_argument("--logging-dir", type=str, default="logs")
    parser.add_argument("--report-to", type=str, default="wandb")
    parser.add_argument("--sampling-steps", type=int, default=10000)
    parser.add_argument("--resume-step", type=int, default=0)
    # model
    parser.add_argument("--model", type=str)
    parser.add_argument("--num-classes", type=int, default=1000)
    parser.add_argument("--encoder-depth", type=int, default=8)
    parser.add_argument("--encoder-depth-text", type=int, default=None)
    parser.add_argument("--fused-attn", action=argparse.BooleanOptionalAction, default=True)
    parser.add_argument("--qk-norm", action=argparse.BooleanOptionalAction, default=False)
    # dataset
    parser.add_argument("--data-dir", type=str, default="../data/imagenet256")
    parser.add_argument("--resolution", type=int, choices=[256], default=256)
    parser.add_argument("--batch-size", type=int, default=256)
    # precision
    parser.add_argument("--allow-tf32", action="store_true")
    parser.add_argument("--mixed-precision", type=str, default="fp16", choices=["no", "fp16", "bf16"])
    # optimization
    parser.add_argument("--epochs", type=int, default=1400)
    parser.add_argument("--max-train-steps", type=int, default=400000)
    parser.add_argument("--checkpointing-steps", type=int, default=50000)
    parser.add_argument("--gradient-accumulation-steps", type=int, default=1)
    parser.add_argument("--learning-rate", type=float, default=1e-4)
    parser.add_argument("--adam-beta1", type=float, default=0.9, help="The beta1 parameter for the Adam optimizer.")
    parser.add_argument("--adam-beta2", type=float, default=0.999, help="The beta2 parameter for the Adam optimizer.")
    parser.add_argument("--adam-weight-decay", type=float, default=0., help="Weight decay to use.")
    parser.add_argument("--adam-epsilon", type=float, default=1e-08, help="Epsilon value for the Adam optimizer")
    parser.add_argument("--max-grad-norm", default=1.0, type=float, help="Max gradient norm.")
    # seed
    parser.add_argument("--seed", type=int, default=0)
    # cpu
    parser.add_argument("--num-workers", type=int, default=4)
    # loss
    parser.add_argument("--path-type", type=str, default="linear", choices=["linear", "cosine"])
    parser.add_argument("--prediction", type=str, default="v", choices=["v"])
    parser.add_argument("--cfg-prob", type=float, default=0.1)
    parser.add_argument("--enc-type", type=str, default='dinov2-vit-b')
    parser.add_argument("--proj-coeff", type=float, default=0.5)
    parser.add_argument("--weighting", default="uniform", type=str, help="Max gradient norm.")
    parser.add_argument("--legacy", action=argparse.BooleanOptionalAction, default=False)
    parser.add_argument("--time-schedule", type=str, default="constant",
                        choices=["constant", "linear", "cosine", "loglinear", "cutoff"])
    parser.add_argument("--repa-coeff", type=float, nargs='+', default=[1.0])
    parser.add_argument("--cutoffs", type=float, nargs='+', default=[0.0, 1.0])
    parser.add_argument("--cfg", action=argparse.BooleanOptionalAction, default=True)
    parser.add_argument("--text-embeds-dir", type=str, default=None)
    parser.add_argument("--repa-weight-decay", type=str, default="constant")
    parser.add_argument("--repa-steps", type=int, default=400000)
    parser.add_argument("--start-diffusion-steps", type=int, default=0)
    parser.add_argument("--diffusion-warm-up-steps", type=int, default=50000)
    parser.add_argument("--diffusion-decay", type=str, default="constant")
    # additive (not in the reference)
    parser.add_argument("--features-dirs", type=str, nargs="*", default=None,
                        help="precomputed frozen-encoder features, one dir per --enc-type entry")
    parser.add_argument("--encoder-ckpts", type=str, nargs="*", default=None,
                        help="state dicts of the frozen image encoders, one per --enc-type entry: the encoder runs on the "
                             "GPU every step as in the reference (clip-vit-* only: reed_amd/encoders.py, SURVEY.md §8f N2)")
    parser.add_argument("--packed-dir", type=str, default=None,
                        help="train from a directory written by `python -m reed_amd.dataset pack` (memory-mapped arrays of "
                             "the same items as --data-dir, SURVEY.md §8f N3)")
    parser.add_argument("--synthetic", type=int, default=0, help="train on N random latents instead of --data-dir (plumbing and throughput runs; 1 MB of encoder features per item still crosses the loader: keep --num-workers >= 4)")
    parser.add_argument("--log-every", type=int, default=1)
    parser.add_argument("--save-act-grad", choices=["auto", "0", "1"], default="auto",
                        help="the activation backward: 1 = one multiply by the derivative the forward saved, 0 = recomputed from the saved "
                             "pre-activation, auto = by tokens per GPU (engine.Engine.SAVE_ACT_GRAD_MIN_TOKENS).  The two differ by one bf16 "
                             "rounding of the derivative: pin it to compare runs of one global batch on different GPU counts bit for bit")
    parser.add_argument("--vae-ckpt", type=str, default=None,
                        help="local sd-vae-ft checkpoint (diffusers layout): turns on the reference's preview sampling at step 1 "
                             "and every --sampling-steps (train.py:431-454), written as PNG grids under <exp>/samples/")
    return parser.parse_args(input_args) if input_args is not None else parser.parse_args()


def encoder_specs(enc_type):
    """'dinov2-vit-b,clip-vit-L' -> ([names], [z_dims]) following utils.py:55-63 naming (type-arch-config)."""
    if enc_type is None or enc_type == "None":
        return [], []
    names, dims = [], []
    for item in enc_type.split(","):
        parts = item.split("-")
        if len(parts) != 3:
            raise ValueError(f"--enc-type entry '{item}' must look like <encoder>-<arch>-<config>, e.g. dinov2-vit-b")
        etype, _arch, cfg = parts
        key = cfg.lower()[0]
        if key not in ENC_EMBED_DIM:
            raise ValueError(f"unknown encoder config '{cfg}' in '{item}'")
        names.append(etype)
        dims.append(ENC_EMBED_DIM[key])
    return names, dims


def text_dim(args):
    d = args.text_embeds_dir
    if d in TEXT_Z_DIM_DICT:
        return TEXT_Z_DIM_DICT[d]
    root = os.path.join(args.data_dir, d)
    for r, _d, files in os.walk(root):
        for f in files:
            if f.endswith(".npy"):
                return int(np.load(os.path.join(r, f)).shape[-1])
    raise KeyError(f"cannot infer the width of --text-embeds-dir {d}")


def create_logger(logging_dir, main):
    logger = logging.getLogger("reed_amd.train")
    logger.setLevel(logging.INFO if main else logging.ERROR)
    if main and not logger.handlers:
        fmt = logging.Formatter('[%(asctime)s] %(message)s', datefmt='%Y-%m-%d %H:%M:%S')
        for h in (logging.StreamHandler(), logging.FileHandler(f"{logging_dir}/log.txt")):
            h.setFormatter(fmt)
            logger.addHandler(h)
    return logger


def main(args):
    import torch.distributed as dist
    from .dataset import CustomDataset, PackedDataset, SyntheticLatents
    from .loss import SILoss
    from .models.sit import SiT_models
    from .optim import FusedAdamWEMA, update_ema
    from .parallel import GradReducer, rank_seed, shard_batch
    from .trainer import TrainStep

    if not torch.cuda.is_available():
        raise RuntimeError("reed_amd.train needs an AMD GPU: the SiT hot path has no CPU fallback")
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=device)
    is_main = rank == 0
    curr_time = datetime.datetime.now().strftime("%Y%m%d_%H%M%S")
    exp_name = args.exp_name if args.resume_step > 0 else f"{args.exp_name}_{curr_time}"
    if world > 1:
        obj = [exp_name]
        dist.broadcast_object_list(obj, src=0)
        exp_name = obj[0]
    save_dir = os.path.join(args.output_dir, exp_name)
    checkpoint_dir = f"{save_dir}/checkpoints"
    if is_main:
        os.makedirs(checkpoint_dir, exist_ok=True)
        with open(os.path.join(save_dir, "args.json"), "w") as f:
            json.dump(vars(args), f, indent=4)
    if world > 1:
        dist.barrier()
    logger = create_logger(save_dir, is_main)
    if args.seed is not None:
        torch.manual_seed(rank_seed(args.seed, rank))
        np.random.seed(rank_seed(args.seed, rank))

    assert args.resolution % 8 == 0
    latent_size = args.resolution // 8
    enc_names, z_dims = encoder_specs(args.enc_type)
    z_types = ["i"] * len(enc_names)
    if args.text_embeds_dir is not None:
        enc_names.append(args.text_embeds_dir)
        z_dims.append(text_dim(args))
        z_types.append("t")
    if enc_names:
        assert len(args.repa_coeff) == len(enc_names), \
            f"Number of alignment loss coefficients {len(args.repa_coeff)} must match the total number of encoders {len(enc_names)}."
    n_img_enc = z_types.count("i")
    encoders = []
    if args.encoder_ckpts:   # on-the-fly frozen encoders, as train.py:182-186,351-357 (DINOv2, CLIP, I-JEPA, MoCo-v3, MAE towers)
        from .encoders import VIT_TOWERS, load_clip_encoder, load_vit_encoder
        if args.features_dirs or args.synthetic:
            raise ValueError("--encoder-ckpts excludes --features-dirs / --synthetic")
        items = args.enc_type.split(",")
        if len(args.encoder_ckpts) != n_img_enc:
            raise ValueError("--encoder-ckpts needs one checkpoint per --enc-type entry")
        for item, path in zip(items, args.encoder_ckpts):
            etype, _arch, cfg = item.split("-")
            key = f"{etype}-vit-{cfg.lower()[0]}"
            if etype == "clip":
                encoders.append(load_clip_encoder(cfg[0].upper(), path, device))
            elif key in VIT_TOWERS:
                encoders.append(load_vit_encoder(key, path, device))
            else:
                raise NotImplementedError(f"on-device frozen encoder '{item}': built are clip-vit-*, {sorted(VIT_TOWERS)} (the "
                                          "towers image/utils.py:55-164 defines, configures or fetches; for others use "
                                          "--features-dirs)")
    if n_img_enc and not args.synthetic and not args.features_dirs and not encoders and not args.packed_dir:
        raise NotImplementedError(
            "this build ships no encoder weights (no network; SURVEY.md §8f N2). Pass --encoder-ckpts <state dict per "
            "clip-vit-* encoder> to run the frozen encoder on the GPU every step, --features-dirs <dir per encoder> with "
            "precomputed [256,z] features, --synthetic N, or --enc-type None.")
    if args.features_dirs and len(args.features_dirs) != n_img_enc:
        raise ValueError("--features-dirs needs one directory per --enc-type entry")

    model = SiT_models[args.model](input_size=latent_size, num_classes=args.num_classes, use_cfg=(args.cfg_prob > 0),
                                   z_dims=z_dims, z_types=z_types, encoder_depth=args.encoder_depth,
                                   encoder_depth_text=args.encoder_depth_text, fused_attn=args.fused_attn,
                                   qk_norm=args.qk_norm).to(device)
    # accelerate's mixed_precision (train.py:141-151): "bf16" = bf16 operands (libreed_hip.so); "fp16" (the reference's default
    # and README recipe) = IEEE-half operands (libreed_hip_f16.so) with dynamic loss scaling (GradScaler defaults) in the
    # fused optimiser pass. Master weights, residual stream, LayerNorm, loss and optimiser state are fp32 either way.
    # "no" (train.py:505) = no autocast at all: fp32 operands and activations on the fp32 matrix instruction
    # (libreed_hip_f32.so, csrc/gemm_f32.hip: 1/16 of the 16-bit MFMA rate — the reference's own fp32 mode is as slow relative
    # to its autocast modes).
    model.precision = {"no": "fp32"}.get(args.mixed_precision, args.mixed_precision)
    ema = copy.deepcopy(model).to(device)
    ema.requires_grad_(False)
    loss_fn = SILoss(prediction=args.prediction, path_type=args.path_type, enc_names=enc_names,
                     weighting=args.weighting, loss_weights={n: args.repa_coeff[i] for i, n in enumerate(enc_names)},
                     time_schedule=args.time_schedule, cutoffs=args.cutoffs, latents_scale=0.18215, latents_bias=0.0)
    logger.info(f"SiT Parameters: {sum(p.numel() for p in model.parameters()):,}")
    logger.info(f"Encoders for Alignment {enc_names}; weights {args.repa_coeff}")

    optimizer = FusedAdamWEMA(model, ema, lr=args.learning_rate, betas=(args.adam_beta1, args.adam_beta2),
                              weight_decay=args.adam_weight_decay, eps=args.adam_epsilon,
                              max_grad_norm=args.max_grad_norm)
    local_batch_size = shard_batch(args.batch_size, world)
    if args.synthetic:
        dataset = SyntheticLatents(args.synthetic, z_dims, z_types, args.num_classes, seed=args.seed or 0,
                                   latent=latent_size)
    elif args.packed_dir:
        dataset = PackedDataset(args.packed_dir)
        if encoders and "images" not in dataset.arr:
            raise ValueError("--encoder-ckpts needs a packed dataset written with images")
        if (args.text_embeds_dir is not None) != ("text" in dataset.arr):
            raise ValueError("--text-embeds-dir and the packed dataset's text field must agree")
        if len(dataset.zkeys) != (0 if encoders else n_img_enc):
            raise ValueError(f"packed dataset holds {len(dataset.zkeys)} feature arrays, --enc-type needs {n_img_enc}")
    else:
        dataset = CustomDataset(args.data_dir, text_embeds_dir=args.text_embeds_dir,
                                features_dirs=args.features_dirs, need_images=bool(encoders))
    # every rank sees the same shuffled order and takes every world-th batch (accelerate BatchSamplerShard, §8a T6)
    gen = torch.Generator().manual_seed(args.seed or 0)
    sampler = torch.utils.data.RandomSampler(dataset, generator=gen)
    batch_sampler = torch.utils.data.BatchSampler(sampler, batch_size=local_batch_size, drop_last=True)

    class Shard(torch.utils.data.Sampler):
        """Every rank yields exactly len(batch_sampler) // world batches per epoch (the tail that does not fill a whole
        round of ranks is dropped), so the ranks stay on the same permutation and leave an epoch together — a rank with
        one batch fewer would leave the others blocked in the bucket all-reduce."""

        def __iter__(self):
            per_rank = len(batch_sampler) // world
            for k, batch in enumerate(batch_sampler):
                if k >= per_rank * world:
                    break
                if k % world == rank:
                    yield batch

        def __len__(self):
            return len(batch_sampler) // world

    loader = torch.utils.data.DataLoader(dataset, batch_sampler=Shard(), num_workers=args.num_workers, pin_memory=True)
    logger.info(f"Dataset contains {len(dataset):,} images")

    update_ema(ema, model, decay=0)
    model.train()
    ema.eval()
    global_step = 0
    if args.resume_step > 0:
        ckpt = torch.load(f"{checkpoint_dir}/{args.resume_step:07d}.pt", map_location="cpu", weights_only=False)
        model.load_state_dict(ckpt["model"])
        ema.load_state_dict(ckpt["ema"])
        optimizer.load_state_dict(ckpt["opt"])
        global_step = ckpt["steps"]
    reducer = None
    if world > 1:
        reducer = GradReducer(model, rank, world)
        reducer.broadcast_params(0)
    model.engine().save_act_grad = {"auto": None, "0": False, "1": True}[args.save_act_grad]
    if is_main:
        eng = model.engine()
        tokens = local_batch_size * eng.T // max(1, args.gradient_accumulation_steps)
        logger.info(f"kernel forms at {tokens} tokens per GPU: activation backward "
                    f"{'saved derivative' if (eng.save_act_grad if eng.save_act_grad is not None else tokens > eng.SAVE_ACT_GRAD_MIN_TOKENS) else 'recomputed'} "
                    f"(--save-act-grad {args.save_act_grad}; auto = saved above {eng.SAVE_ACT_GRAD_MIN_TOKENS} tokens)")
    step_fn = TrainStep(model, loss_fn, optimizer, reducer, proj_coeff=args.proj_coeff,
                        repa_decay=args.repa_weight_decay, repa_steps=args.repa_steps,
                        start_diffusion_steps=args.start_diffusion_steps,
                        diffusion_warm_up_steps=args.diffusion_warm_up_steps, diffusion_decay=args.diffusion_decay,
                        max_train_steps=args.max_train_steps, grad_accum=args.gradient_accumulation_steps)
    step_fn.global_step = global_step
    log_path = os.path.join(save_dir, "metrics.jsonl")
    t_last, n_last = time.time(), global_step
    done = False
    previews = None
    for epoch in range(args.epochs):
        for item in loader:
            _raw, moments, y, textemb = item[:4]
            zs = [z.to(device, non_blocking=True) for z in item[4:]]
            if encoders:   # train.py:351-357: frozen encoder forward on the raw images, no grad
                raw = _raw.to(device, non_blocking=True)
                zs = [enc.encode_raw(raw) for enc in encoders]
            moments = moments.squeeze(dim=1).to(device, non_blocking=True)
            if y.numel() and (int(y.min()) < 0 or int(y.max()) >= args.num_classes):   # on the host, before the copy
                raise IndexError(f"dataset labels in [{int(y.min())}, {int(y.max())}] but --num-classes={args.num_classes}")
            y = y.to(device, non_blocking=True)
            if args.legacy:  # label dropping applied twice (train.py:338-343), kept for reproducibility
                drop_ids = torch.rand(y.shape[0], device=device) < args.cfg_prob
                labels = torch.where(drop_ids, args.num_classes, y)
            else:
                labels = y
            if not args.cfg:
                labels = torch.zeros_like(labels)
            if args.text_embeds_dir is not None:
                zs.append(textemb.to(device, non_blocking=True))
            if args.vae_ckpt and previews is None:
                previews = Previews(args, moments, device, world, save_dir, is_main, latent_size)
            res = step_fn(None, labels, zs, moments=moments)
            if "grad_norm" not in res:
                continue  # accumulation micro-step
            global_step = step_fn.global_step
            want_preview = previews is not None and (global_step == 1 or (global_step % args.sampling_steps == 0 and global_step > 0))
            want_ckpt = global_step % args.checkpointing_steps == 0 and global_step > 0
            if want_preview or want_ckpt:
                optimizer.sync_replicas()   # collective; a no-op unless the update is sharded (REED_OPT_SHARD=1): every rank a full replica
            if want_preview:
                previews(model, global_step)
                logger.info("Generating EMA samples done.")
            if want_ckpt and is_main:
                ckpt = {"model": model.state_dict(), "ema": ema.state_dict(), "opt": optimizer.state_dict(),
                        "args": args, "steps": global_step}
                path = f"{checkpoint_dir}/{global_step:07d}.pt"
                torch.save(ckpt, path)
                logger.info(f"Saved checkpoint to {path}")
            if global_step % args.log_every == 0:
                logs = {"proj_loss": res["proj_loss"], "grad_norm": res["grad_norm"],
                        "training_denoising_loss": res["denoising_loss"], "img_proj_loss": res["img_proj_loss"]}
                if args.text_embeds_dir is not None:
                    logs["text_proj_loss"] = res["text_proj_loss"]
                vals = torch.stack([torch.as_tensor(v, dtype=torch.float32, device=device).reshape(()) for v in logs.values()])
                if world > 1:  # one small all-reduce instead of 4-5 gathers (train.py:456-465)
                    dist.all_reduce(vals)
                    vals /= world
                vals = vals.tolist()  # the only host sync of the step, and only every --log-every steps
                model.engine().check_errors()
                if is_main:
                    now = time.time()
                    ips = (global_step - n_last) * args.batch_size / max(now - t_last, 1e-9)
                    t_last, n_last = now, global_step
                    rec = dict(zip(logs.keys(), vals), step=global_step, images_per_sec=ips)
                    with open(log_path, "a") as f:
                        f.write(json.dumps(rec) + "\n")
                    logger.info(" ".join(f"{k}={v:.5f}" if isinstance(v, float) else f"{k}={v}" for k, v in rec.items()))
            if global_step >= args.max_train_steps:
                done = True
                break
        if done:
            break
    model.eval()
    if world > 1:
        dist.barrier()
    logger.info("Done!")
    if reducer is not None:
        reducer.close()
    if world > 1:
        dist.destroy_process_group()
    return save_dir


def array2grid(x, padding=2):
    """train.py:77-81 (torchvision.utils.make_grid with its defaults: nrow = round(sqrt(N)) images per row, 2-pixel black
    border and gutters) -> uint8 [H, W, 3]."""
    import math
    x = x.clamp(0, 1)
    n, c, h, w = x.shape
    nrow = max(1, round(math.sqrt(n)))
    xmaps, ymaps = min(nrow, n), int(math.ceil(n / nrow))
    grid = x.new_zeros((c, ymaps * (h + padding) + padding, xmaps * (w + padding) + padding))
    for k in range(n):
        r, col = divmod(k, xmaps)
        grid[:, r * (h + padding) + padding:r * (h + padding) + padding + h,
             col * (w + padding) + padding:col * (w + padding) + padding + w] = x[k]
    return grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy()


class Previews:
    """The reference's in-training preview (train.py:315-329,431-454): 64 // world fixed noises and labels per rank, sampled with
    the TRAINING model (50 Euler steps, cfg 4.0), decoded by the SD-VAE and logged as an image grid — here a PNG under
    <exp>/samples/ instead of a wandb image. On only with --vae-ckpt (reed_amd/vae.py; SURVEY.md §9-2: the reference crashes
    without a decoder)."""

    def __init__(self, args, first_moments, device, world, save_dir, is_main, latent_size):
        from .trainer import sample_posterior
        from .vae import load_sd_vae_decoder
        self.vae = load_sd_vae_decoder(args.vae_ckpt, device=device)
        self.args, self.world, self.is_main, self.device = args, world, is_main, device
        self.dir = os.path.join(save_dir, "samples")
        if is_main:
            os.makedirs(self.dir, exist_ok=True)
        n = max(1, 64 // world)
        self.gt_xs = sample_posterior(first_moments[:n].to(device), 0.18215, 0.0)   # (fewer than n if the batch is smaller)
        self.ys = torch.randint(1000, size=(n,), device=device)
        if not args.cfg:
            self.ys = torch.zeros_like(self.ys)
        self.xT = torch.randn((n, 4, latent_size, latent_size), device=device)
        self._gt_done = False

    def _gather(self, x):
        if self.world == 1:
            return x
        import torch.distributed as dist
        out = [torch.empty_like(x) for _ in range(self.world)]
        dist.all_gather(out, x.contiguous())
        return torch.cat(out)

    @torch.no_grad()
    def __call__(self, model, step):
        from PIL import Image
        from .samplers import euler_sampler
        a = self.args
        samples = euler_sampler(model, self.xT, self.ys, num_steps=50, cfg_scale=4.0, guidance_low=0., guidance_high=1.,
                                path_type=a.path_type, heun=False, prediction=a.prediction).to(torch.float32)
        out = self._gather((self.vae.decode(samples / 0.18215) + 1) / 2.)
        gt = None if self._gt_done else self._gather((self.vae.decode(self.gt_xs / 0.18215) + 1) / 2.)
        if self.is_main:
            Image.fromarray(array2grid(out)).save(os.path.join(self.dir, f"{step:07d}.png"))
            if gt is not None:
                Image.fromarray(array2grid(gt)).save(os.path.join(self.dir, "gt_samples.png"))
        self._gt_done = True


if __name__ == "__main__":
    main(parse_args())
