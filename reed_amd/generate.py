"""generate.py — drop-in counterpart of the reference's image/generate.py (FID sampling) on the MI355X HIP path.

    python -m torch.distributed.run --nproc-per-node 8 -m reed_amd.generate --ckpt exps/run/checkpoints/0400000.pt \
        --mode sde --num-steps 250 --cfg-scale 1.275 ...

Same flags as image/generate.py:177-219 (the four undefined `args.*` the reference's main() touches —
repr_detach, repr_residue, strict_conditioning, flex_param_gamma — are dropped: SURVEY.md §9-3), same seed rule
(global_seed*world+rank, :49), same checkpoint handling (ckpt['ema'], drop 'projectors.*', strict=False, :77-85),
same folder name (:92-100) and sample index rule (i*world+rank+total, :164), same .npz packer (:20-34).
Sampling is embarrassingly parallel: ranks only meet at barriers (RCCL via torch.distributed); no tensor collective.

The SD-VAE decoder: with --vae-ckpt <local sd-vae-ft checkpoint> the in-repo restatement of diffusers'
AutoencoderKL decoder runs (reed_amd/vae.py; parity unpinned: no diffusers / weights here to compare with); without it,
`diffusers` and its hub weights are used if importable; --save-latents writes the fp32 latents [N,4,32,32] as
<folder>_latents.npz instead (decode + PNG can then run anywhere).
"""
import argparse
import math
import os

import numpy as np
import torch


def create_npz_from_sample_folder(sample_dir, num=50_000):
    """Builds a single .npz file from a folder of .png samples (generate.py:20-34)."""
    from PIL import Image
    samples = []
    for i in range(num):
        samples.append(np.asarray(Image.open(f"{sample_dir}/{i:06d}.png")).astype(np.uint8))
    samples = np.stack(samples)
    assert samples.shape == (num, samples.shape[1], samples.shape[2], 3)
    npz_path = f"{sample_dir}.npz"
    np.savez(npz_path, arr_0=samples)
    print(f"Saved .npz file to {npz_path} [shape={samples.shape}].")
    return npz_path


def load_legacy_checkpoints(state_dict, encoder_depth):
    """utils.py:207-219: decoder_blocks.i -> blocks.(i + encoder_depth)."""
    new = {}
    for key, value in state_dict.items():
        if "decoder_blocks" in key:
            parts = key.split(".")
            parts[0], parts[1] = "blocks", str(int(parts[1]) + encoder_depth)
            new[".".join(parts)] = value
        else:
            new[key] = value
    return new


def folder_name(args):
    model_string_name = args.model.replace("/", "-")
    ckpt_string_name = os.path.basename(args.ckpt).replace(".pt", "") if args.ckpt else "pretrained"
    name = f"{model_string_name}-{ckpt_string_name}-size-{args.resolution}-vae-{args.vae}-" \
           f"cfg-{args.cfg_scale}-seed-{args.global_seed}-{args.mode}"
    if args.guidance_high != 1.0:
        name += f"-cfg-high-{args.guidance_high}"
    return name


def build_parser():
    from .models.sit import SiT_models
    parser = argparse.ArgumentParser()
    parser.add_argument("--global-seed", type=int, default=0)
    # The reference samples with an fp32 model and, with --tf32 (its default), TF32 matmuls (generate.py:41,183: 10-bit
    # mantissa operands, fp32 accumulation).  The MFMA equivalent at full rate is IEEE-half operands with fp32 accumulation:
    # --tf32 selects the fp16 build of the kernels for the model evaluations (libreed_hip_f16.so).  --no-tf32 is the reference's
    # true-fp32 mode and selects the fp32-operand build (libreed_hip_f32.so: v_mfma_f32_32x32x2_f32, fp32 activations, 1/16 of the
    # 16-bit MFMA rate) — never a silent substitution of narrower arithmetic.  --sample-precision overrides either.
    parser.add_argument("--tf32", action=argparse.BooleanOptionalAction, default=True)
    parser.add_argument("--ckpt", type=str, default=None, help="Optional path to a SiT checkpoint.")
    parser.add_argument("--sample-dir", type=str, default="samples")
    parser.add_argument("--model", type=str, choices=list(SiT_models.keys()), default="SiT-XL/2")
    parser.add_argument("--num-classes", type=int, default=1000)
    parser.add_argument("--encoder-depth", type=int, default=8)
    parser.add_argument("--resolution", type=int, choices=[256, 512], default=256)
    parser.add_argument("--fused-attn", action=argparse.BooleanOptionalAction, default=False)
    parser.add_argument("--qk-norm", action=argparse.BooleanOptionalAction, default=False)
    parser.add_argument("--vae", type=str, choices=["ema", "mse"], default="ema")
    parser.add_argument("--per-proc-batch-size", type=int, default=32)
    parser.add_argument("--num-fid-samples", type=int, default=50_000)
    parser.add_argument("--mode", type=str, default="ode")
    parser.add_argument("--cfg-scale", type=float, default=1.5)
    parser.add_argument("--projector-embed-dims", type=str, default="768")
    parser.add_argument("--path-type", type=str, default="linear", choices=["linear", "cosine"])
    parser.add_argument("--num-steps", type=int, default=50)
    parser.add_argument("--heun", action=argparse.BooleanOptionalAction, default=False)
    parser.add_argument("--guidance-low", type=float, default=0.)
    parser.add_argument("--guidance-high", type=float, default=1.)
    parser.add_argument("--legacy", action=argparse.BooleanOptionalAction, default=False)
    parser.add_argument("--prediction", type=str, default="v", choices=["v"])
    # additive
    parser.add_argument("--save-latents", action="store_true", help="write latents .npz instead of decoding to PNG")
    parser.add_argument("--vae-ckpt", type=str, default=None,
                        help="local sd-vae-ft-{ema,mse} checkpoint (diffusers directory or its diffusion_pytorch_model file): "
                             "decode with the in-repo decoder instead of the diffusers package")
    parser.add_argument("--sample-precision", type=str, choices=["fp16", "bf16", "fp32"], default=None,
                        help="operand type of the model evaluations: fp16 = 10-bit mantissa (the reference's TF32; the default "
                             "with --tf32), fp32 = the reference's --no-tf32 arithmetic (the default with --no-tf32), bf16 = the "
                             "training precision")
    return parser


def sample_precision(args):
    """--sample-precision if given; else what the reference's --tf32 / --no-tf32 means on this hardware (generate.py:41,183)."""
    if getattr(args, "sample_precision", None):
        return args.sample_precision
    return "fp16" if args.tf32 else "fp32"


def decode_latents(vae, z, args):
    """`vae.decode(z).sample` (generate.py:156).  The in-repo decoder runs on the HIP kernels with the operand type the
    reference's flags mean here: fp16 operands + fp32 accumulation and activations under --tf32 (TF32's mantissa; the reference's
    convolutions are TF32 there), exact fp32 under --no-tf32.  A non-finite fp16 result (half saturates at 65504, TF32 does
    not) is decoded again with fp32 operands."""
    from .vae import SDVAEDecoder
    if not isinstance(vae, SDVAEDecoder):
        img = vae.decode(z)
        return getattr(img, "sample", img)        # diffusers returns a DecoderOutput
    prec = "fp32" if sample_precision(args) == "fp32" else "fp16"
    img = vae.decode(z, precision=prec)
    if prec != "fp32" and not bool(torch.isfinite(img).all()):
        import warnings
        warnings.warn("reed_amd.generate: non-finite images from the fp16-operand VAE decoder; decoding this batch with fp32 operands")
        img = vae.decode(z, precision="fp32")
    return img


def finite_or_retry(sampler, kw, model):
    """Run one batch.  IEEE half saturates at 65504 where bf16 and the reference's TF32 do not: a checkpoint with large
    activation outliers would give inf / nan latents silently.  One isfinite reduction per batch (one host
    sync per several hundred model evaluations); a non-finite fp16 batch is re-sampled once with bf16 operands from the same
    latents, labels and device RNG state (the SDE sampler draws noise per step), and anything still non-finite raises."""
    dev = kw["latents"].device
    rng = torch.cuda.get_rng_state(dev)
    samples = sampler(**kw).to(torch.float32)
    if bool(torch.isfinite(samples).all()):
        return samples
    if model.precision != "fp16":
        raise FloatingPointError(f"non-finite latents from the {model.precision} sampler: the checkpoint or the inputs are broken")
    import warnings
    warnings.warn("reed_amd.generate: non-finite latents with fp16 operands (half saturates at 65504); re-sampling this batch "
                  "with bf16 operands")
    model.precision = "bf16"
    after = torch.cuda.get_rng_state(dev)
    torch.cuda.set_rng_state(rng, dev)
    try:
        samples = sampler(**kw).to(torch.float32)
    finally:
        model.precision = "fp16"
        torch.cuda.set_rng_state(after, dev)      # later batches draw what they would have drawn without the retry
    if not bool(torch.isfinite(samples).all()):
        raise FloatingPointError("non-finite latents with fp16 AND bf16 operands: the checkpoint or the inputs are broken")
    return samples


def main(args):
    import torch.distributed as dist
    from .models.sit import SiT_models
    from .parallel import sample_index, sample_seed
    from .samplers import euler_maruyama_sampler, euler_sampler

    assert torch.cuda.is_available(), "Sampling requires an AMD GPU: the SiT hot path has no CPU fallback"
    torch.set_grad_enabled(False)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", rank % max(1, torch.cuda.device_count())))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=device)
    seed = sample_seed(args.global_seed, world, rank)
    torch.manual_seed(seed)
    print(f"Starting rank={rank}, seed={seed}, world_size={world}.")

    latent_size = args.resolution // 8
    model = SiT_models[args.model](input_size=latent_size, num_classes=args.num_classes, use_cfg=True,
                                   z_dims=[], z_types=[], encoder_depth=args.encoder_depth,
                                   fused_attn=args.fused_attn, qk_norm=args.qk_norm).to(device)
    if args.ckpt is None:
        raise ValueError("--ckpt is required: the reference's auto-download of 'last.pt' needs network access")
    state_dict = torch.load(args.ckpt, map_location="cpu", weights_only=False)["ema"]
    if args.legacy:
        state_dict = load_legacy_checkpoints(state_dict, args.encoder_depth)
    for k in list(state_dict.keys()):
        if "projectors" in k:
            state_dict.pop(k)
    model.load_state_dict(state_dict, strict=False)
    model.eval()
    model.precision = sample_precision(args)
    assert args.cfg_scale >= 1.0, "In almost all cases, cfg_scale be >= 1.0"
    if args.cfg_scale > 1.0:
        assert args.num_classes == 1000, "the samplers hard-code the null class id 1000 (samplers.py:59)"

    vae = None
    if not args.save_latents:
        if args.vae_ckpt:     # the in-repo restatement of the decoder (reed_amd/vae.py) on a local sd-vae-ft checkpoint
            from .vae import load_sd_vae_decoder
            vae = load_sd_vae_decoder(args.vae_ckpt, device=device)
        else:
            try:
                from diffusers.models import AutoencoderKL
                vae = AutoencoderKL.from_pretrained(f"stabilityai/sd-vae-ft-{args.vae}").to(device)
            except Exception as e:
                raise RuntimeError("the SD-VAE decoder is unavailable through diffusers (package or weights missing): pass "
                                   "--vae-ckpt <sd-vae-ft-{ema,mse} directory or diffusion_pytorch_model.safetensors> to decode "
                                   "with reed_amd/vae.py, or --save-latents to write latents instead of PNGs") from e

    sample_folder_dir = f"{args.sample_dir}/{folder_name(args)}"
    if rank == 0:
        os.makedirs(sample_folder_dir, exist_ok=True)
        print(f"Saving samples at {sample_folder_dir}")
    if world > 1:
        dist.barrier()
    n = args.per_proc_batch_size
    global_batch_size = n * world
    total_samples = int(math.ceil(args.num_fid_samples / global_batch_size) * global_batch_size)
    assert total_samples % world == 0
    samples_needed_this_gpu = total_samples // world
    assert samples_needed_this_gpu % n == 0
    iterations = samples_needed_this_gpu // n
    if rank == 0:
        print(f"Total number of images that will be sampled: {total_samples}")
        print(f"SiT Parameters: {sum(p.numel() for p in model.parameters()):,}")
    total = 0
    kept = {}
    for _ in range(iterations):
        z = torch.randn(n, model.in_channels, latent_size, latent_size, device=device)
        y = torch.randint(0, args.num_classes, (n,), device=device)
        kw = dict(model=model, latents=z, y=y, num_steps=args.num_steps, heun=args.heun, cfg_scale=args.cfg_scale,
                  guidance_low=args.guidance_low, guidance_high=args.guidance_high, path_type=args.path_type,
                  prediction=args.prediction)
        if args.mode == "sde":
            samples = finite_or_retry(euler_maruyama_sampler, kw, model)
        elif args.mode == "ode":
            samples = finite_or_retry(euler_sampler, kw, model)
        else:
            raise NotImplementedError()
        if vae is not None:
            from PIL import Image
            img = decode_latents(vae, samples / 0.18215, args)
            img = torch.clamp(255. * ((img + 1) / 2.), 0, 255).permute(0, 2, 3, 1).to("cpu", dtype=torch.uint8).numpy()
            for i, s in enumerate(img):
                Image.fromarray(s).save(f"{sample_folder_dir}/{sample_index(i, world, rank, total):06d}.png")
        else:
            lat = samples.cpu().numpy()
            for i in range(n):
                kept[sample_index(i, world, rank, total)] = lat[i]
        total += global_batch_size
    if vae is None:
        np.savez(f"{sample_folder_dir}/latents_rank{rank:03d}.npz", index=np.array(sorted(kept)),
                 latents=np.stack([kept[k] for k in sorted(kept)]))
    if world > 1:
        dist.barrier()
    if rank == 0:
        if vae is not None:
            create_npz_from_sample_folder(sample_folder_dir, args.num_fid_samples)
        else:
            idx, lat = [], []
            for r in range(world):
                d = np.load(f"{sample_folder_dir}/latents_rank{r:03d}.npz")
                idx.append(d["index"]); lat.append(d["latents"])
            idx, lat = np.concatenate(idx), np.concatenate(lat)
            order = np.argsort(idx)
            np.savez(f"{sample_folder_dir}_latents.npz", arr_0=lat[order][:args.num_fid_samples])
            print(f"Saved latents to {sample_folder_dir}_latents.npz")
        print("Done.")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return sample_folder_dir


if __name__ == "__main__":
    main(build_parser().parse_args())
