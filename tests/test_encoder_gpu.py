"""Frozen CLIP image encoder on the HIP path vs the oracle under bf16 autocast (same precision) and
vs the reference's own fp32 output (tests/golden/clip.npz): row kernels exactly / to bf16 resolution, the full tower to
bf16 noise; ViT-L/14-shaped smoke at the real token count (T = 257: ragged attention tile, ragged GEMM rows)."""
import pytest
import torch

from oracle import clip_vit as oclip
from oracle import detfill
from tests.test_oracle_golden import load

pytestmark = pytest.mark.gpu


def bfr(x):
    return x.to(torch.bfloat16).float()


def _hip_encoder(cfg, dev, seed):
    from reed_amd.encoders import ClipVisionEncoder
    enc = ClipVisionEncoder(**cfg)
    enc.load_state_dict(oclip.fill_params(cfg, base_seed=seed), strict=True)
    return enc.to(dev).eval()


def test_row_kernels(dev):
    from reed_amd import ops
    g = torch.Generator().manual_seed(0)
    M, D = 37, 1024
    x = (torch.randn(M, D, generator=g) * 2 + 0.5).to(torch.bfloat16).to(dev)
    w, b = (1 + 0.1 * torch.randn(D, generator=g)).to(dev), (0.1 * torch.randn(D, generator=g)).to(dev)
    out = torch.empty_like(x)
    ops.ln_affine_bf16(x, w, b, out, M, D)
    ref = torch.nn.functional.layer_norm(x.float(), (D,), w, b, 1e-5)
    torch.testing.assert_close(out.float(), bfr(ref), atol=2e-2, rtol=1e-2)
    # im2col: exact bf16 rounding of the gathered pixels, zero padding
    B, S, P, Kp = 2, 28, 14, 640
    img = torch.randn(B, 3, S, S, generator=g).to(dev)
    cols = torch.full((B * 4, Kp), 7.0, dtype=torch.bfloat16, device=dev)
    ops.clip_im2col(img, cols, B, S, P, Kp)
    ref = torch.nn.functional.unfold(img, P, stride=P).transpose(1, 2).reshape(B * 4, 3 * P * P)
    assert torch.equal(cols[:, :588].float(), bfr(ref)) and float(cols[:, 588:].abs().max()) == 0.0
    # tokens: [class | patches] + positional embedding with the eager-mode bf16 roundings
    T, D = 5, 128
    patches = torch.randn(B * (T - 1), D, generator=g).to(torch.bfloat16).to(dev)
    cls, pos = torch.randn(D, generator=g).to(dev), torch.randn(T, D, generator=g).to(dev)
    tok = torch.empty(B, T, D, dtype=torch.bfloat16, device=dev)
    ops.clip_tokens(patches, cls, pos, tok, B, T, D)
    ref = torch.cat([bfr(cls).expand(B, 1, D), patches.float().view(B, T - 1, D)], 1) + bfr(pos)
    assert torch.equal(tok.float(), bfr(ref))


def test_quickgelu_and_residual_epilogues(dev):
    from reed_amd import ops
    g = torch.Generator().manual_seed(1)
    M, N, K = 300, 384, 128
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(torch.bfloat16).to(dev)
    r = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    pre = bfr(x.float() @ w.float().t() + b.float())
    for tile in (0, 128, 256):
        ops.gemm_force_tile(tile)
        act = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        ops.gemm(ops.NT, ops.EPI_QGELU, x, w, M, N, K, None, K, K, N, C2=act, ldc2=N, bias=b)
        ref = bfr(pre * bfr(torch.sigmoid(bfr(1.702 * pre))))
        torch.testing.assert_close(act.float(), ref, atol=2e-2, rtol=2e-2)
        out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        ops.gemm(ops.NT, ops.EPI_RES_BF16, x, w, M, N, K, out, K, K, N, R=r, ldr=N, bias=b)
        torch.testing.assert_close(out.float(), bfr(pre + r.float()), atol=3e-2, rtol=2e-2)
    ops.gemm_force_tile(0)


@pytest.mark.parametrize("tag,cfg,B", [("t2", oclip.make_config(width=128, layers=2, heads=2, patch=14, image=56), 3),
                                       ("t3", oclip.make_config(width=256, layers=3, heads=4, patch=14, image=28), 2)])
def test_tower_vs_oracle_and_reference(dev, tag, cfg, B):
    enc = _hip_encoder(cfg, dev, 5)
    x = detfill.normal((B, 3, cfg["image"], cfg["image"]), 77)
    out = enc(x.to(dev)).float().cpu()
    with torch.no_grad():
        o16 = oclip.forward(oclip.fill_params(cfg, base_seed=5), cfg, x, autocast_bf16=True).float()
    ref32 = torch.from_numpy(load("clip")[tag + ".fp32"])
    assert out.shape == ref32.shape
    scale = ref32.abs().max().item()
    # same-precision oracle: bf16 noise of a 2-3 block tower; reference fp32: the reference's own bf16 gap
    assert (out - o16).abs().max().item() <= 3e-2 * scale
    assert (out - ref32).abs().max().item() <= 4e-2 * scale
    cs = torch.nn.functional.cosine_similarity(out.flatten(), ref32.flatten(), dim=0).item()
    assert cs > 0.9995, cs


def test_vit_l14_shape_and_preprocess(dev):
    """ViT-L/14 width / heads / token count (2 blocks keep the CPU oracle quick): T = 257 exercises the ragged second
    key tile of the attention kernel and ragged GEMM rows (M = B·257); preprocess = train.py:53-57."""
    from reed_amd.encoders import ClipVisionEncoder
    cfg = oclip.make_config(width=1024, layers=2, heads=16, patch=14, image=224)
    enc = _hip_encoder(cfg, dev, 9)
    raw = (torch.arange(2 * 3 * 256 * 256) * 7 % 251).reshape(2, 3, 256, 256).to(torch.uint8)
    pre_hip = ClipVisionEncoder.preprocess(raw.to(dev)).cpu()
    pre_ref = oclip.preprocess(raw)
    torch.testing.assert_close(pre_hip, pre_ref, atol=1e-4, rtol=1e-4)
    out = enc.encode_raw(raw.to(dev)).float().cpu()
    with torch.no_grad():
        o16 = oclip.forward(oclip.fill_params(cfg, base_seed=9), cfg, pre_ref, autocast_bf16=True).float()
    assert out.shape == (2, 256, 1024)
    scale = o16.abs().max().item()
    assert (out - o16).abs().max().item() <= 3e-2 * scale
    assert torch.nn.functional.cosine_similarity(out.flatten(), o16.flatten(), dim=0).item() > 0.9995


# ---------------------------------------------------------------------------------------------------------------------
# The other frozen towers (I-JEPA, MAE, MoCo-v3): reed_amd.encoders.VitEncoder vs the goldens of the reference's classes
TOWER_CASES = {"jepa80": (dict(embed=640, depth=2, heads=8, patch=14, image=56, cls=False, final_norm=True), "jepa", 3),
               "jepa64": (dict(embed=256, depth=3, heads=4, patch=14, image=56, cls=False, final_norm=True), "jepa", 3),
               "mae": (dict(embed=256, depth=2, heads=4, patch=16, image=64, cls=True, final_norm=False), "learned", 2),
               "moco": (dict(embed=256, depth=2, heads=4, patch=16, image=64, cls=True, final_norm=True), "moco", 2)}


@pytest.mark.parametrize("tag", list(TOWER_CASES))
def test_vit_tower_vs_reference(dev, tag):
    """HIP tower (bf16 operands, fp32 residual) against the reference's output under bf16 autocast and in fp32
    (tests/golden/towers.npz: the reference's own I-JEPA class; its MAE forward_features and MoCo-v3 constructor over the
    timm stand-in) and against the same-precision oracle.  head_dim 80 (ViT-H) and 64; with / without class token and
    final norm; T = 16 and 17 tokens."""
    import numpy as np
    from oracle import detfill
    from oracle import vit_towers as ot
    from reed_amd.encoders import VitEncoder
    from tests.test_oracle_golden import load
    g = load("towers")
    kw, pos, B = TOWER_CASES[tag]
    cfg = ot.make_config(pos=pos, **kw)
    P = ot.fill_params(cfg, base_seed=9)
    enc = VitEncoder(**kw)
    enc.load_state_dict(P)
    enc = enc.to(dev).eval()
    x = detfill.normal((B, 3, kw["image"], kw["image"]), 55)
    out = enc(x.to(dev)).float().cpu()
    ref32, ref16 = torch.from_numpy(g[tag + ".fp32"]), torch.from_numpy(g[tag + ".bf16"])
    assert out.shape == ref32.shape
    sc = ref32.abs().max().item()
    e16, e32, eref = (out - ref16).abs().max().item() / sc, (out - ref32).abs().max().item() / sc, (ref16 - ref32).abs().max().item() / sc
    c32 = torch.nn.functional.cosine_similarity(out.flatten(), ref32.flatten(), dim=0).item()
    print(f"tower {tag}: max|HIP - ref_bf16| {e16:.2e}, max|HIP - ref_fp32| {e32:.2e} (reference's own bf16-vs-fp32 {eref:.2e}) of the "
          f"output range; cosine vs fp32 {c32:.6f}")
    assert e32 <= 2.0 * eref + 2e-3 and c32 > 0.9998


@pytest.mark.parametrize("tag,E,H,depth,image,reg,B", [("plain", 128, 2, 2, 56, 0, 3), ("reg4", 256, 4, 3, 28, 4, 2)])
def test_dinov2_tower_vs_hf_port(dev, tag, E, H, depth, image, reg, B):
    """DINOv2 (image/utils.py:92-104; torch.hub class, not in the reference tree) on the HIP tower — LayerScale + fp32 residual
    as the GEMM epilogue 12, register tokens through reed_vit_tokens' prefix rows — against transformers' independent port
    of the model (tests/golden/dinov2.npz) under bf16 autocast and in fp32, hub parameter names."""
    from oracle import detfill
    from oracle import vit_towers as ot
    from reed_amd.encoders import VitEncoder
    from tests.test_oracle_golden import load
    g = load("dinov2")
    cfg = ot.make_config(E, depth, H, 14, image, True, True, "learned", ls=True, reg=reg)
    P = ot.fill_params(cfg, base_seed=21)
    P["mask_token"] = torch.zeros(1, E)          # present in the hub checkpoints, unused at inference
    enc = VitEncoder(embed=E, depth=depth, heads=H, patch=14, image=image, cls=True, final_norm=True, layerscale=True,
                     registers=reg)
    missing, unexpected = enc.load_state_dict(P)
    assert not missing and not unexpected
    enc = enc.to(dev).eval()
    x = detfill.normal((B, 3, image, image), 56)
    out = enc(x.to(dev)).float().cpu()
    ref32, ref16 = torch.from_numpy(g[tag + ".fp32"]), torch.from_numpy(g[tag + ".bf16"])
    assert out.shape == ref32.shape
    sc = ref32.abs().max().item()
    e32, eref = (out - ref32).abs().max().item() / sc, (ref16 - ref32).abs().max().item() / sc
    c32 = torch.nn.functional.cosine_similarity(out.flatten(), ref32.flatten(), dim=0).item()
    print(f"dinov2 {tag}: max|HIP - fp32| {e32:.2e} (the port's own bf16-vs-fp32 {eref:.2e}) of the output range; cosine {c32:.6f}")
    assert e32 <= 2.0 * eref + 2e-3 and c32 > 0.9998


def test_layerscale_residual_epilogue(dev):
    """reed_gemm epilogue 12: C f32 = R f32 + gamma f32[n] * float(bf16(acc + bias)) on the 128^2 and both 256^2 kernels."""
    from reed_amd import ops
    g = torch.Generator().manual_seed(3)
    for M, N, K in ((257 * 3, 384, 384), (4112, 1024, 4096)):
        x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(dev)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
        b = torch.randn(N, generator=g).to(torch.bfloat16).to(dev)
        gamma = (torch.rand(N, generator=g) * 2).to(dev)
        r = torch.randn(M, N, generator=g).to(dev)
        ref = r + gamma * (x.float() @ w.float().t() + b.float()).to(torch.bfloat16).float()
        for tile in (128, 256, 257):
            ops.gemm_force_tile(tile)
            out = torch.full((M, N), float("nan"), device=dev)
            ops.gemm(ops.NT, ops.EPI_LS_RES, x, w, M, N, K, out, K, K, N, R=r, ldr=N, bias=b, gate=gamma)
            ops.gemm_force_tile(0)
            torch.testing.assert_close(out, ref, atol=2e-2, rtol=1e-2)       # one bf16 rounding of the branch x gamma <= 2
            assert (out - ref).abs().mean().item() < 2e-3


def test_preprocess_raw_image_branches(dev):
    """preprocess_raw_image (image/train.py:53-74) as one HIP pass: every branch against the torch restatement
    (F.interpolate bicubic + Normalize) on random uint8 images and against the reference-generated ramp samples."""
    import numpy as np
    from oracle import vit_towers as ot
    from reed_amd.encoders import preprocess_raw_image
    from tests.test_oracle_golden import load
    g = load("towers")
    gc = load("clip")
    raw = torch.randint(0, 256, (3, 3, 256, 256), generator=torch.Generator().manual_seed(4), dtype=torch.uint8)
    for enc in ("clip", "mocov3", "mae", "dinov2", "jepa"):
        got = preprocess_raw_image(raw.to(dev), enc).cpu()
        ref = ot.preprocess(raw, enc)
        assert got.shape == ref.shape, enc
        err = (got - ref).abs().max().item()
        print(f"preprocess {enc}: max abs deviation from torch {err:.2e}")
        assert err <= 2e-5, (enc, err)       # fp32 bicubic taps summed in a different order
    ramp = (torch.arange(2 * 3 * 256 * 256) % 251).reshape(2, 3, 256, 256).to(torch.uint8).to(dev)
    np.testing.assert_allclose(preprocess_raw_image(ramp, "jepa")[:, :, ::37, ::41].cpu().numpy(), g["pre.jepa.sample"], atol=2e-5)
    np.testing.assert_allclose(preprocess_raw_image(ramp, "mae")[:, :, ::37, ::41].cpu().numpy(), g["pre.mae.sample"], atol=2e-6)
    mean = torch.tensor(ot.CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(ot.CLIP_STD).view(1, 3, 1, 1)
    got = preprocess_raw_image(ramp, "clip")[:, :, ::37, ::41].cpu()
    np.testing.assert_allclose((got * std + mean).numpy(), gc["pre.sample"], atol=2e-5)   # golden = before Normalize
    big = torch.randint(0, 256, (1, 3, 512, 512), generator=torch.Generator().manual_seed(5), dtype=torch.uint8)
    assert preprocess_raw_image(big.to(dev), "dinov2").shape == (1, 3, 448, 448)
    assert (preprocess_raw_image(big.to(dev), "jepa").cpu() - ot.preprocess(big, "jepa")).abs().max().item() <= 2e-5


def test_vit_h_tower_shape_and_head_dim_80_attention(dev):
    """I-JEPA ViT-H/14 geometry (embed 1280, 16 heads of 80, 256 tokens) through 2 blocks: shape / finiteness, and the
    head_dim-80 attention kernel against fp32 softmax attention at T = 256 with more (batch, head) items than CUs."""
    from reed_amd import ops
    from reed_amd.encoders import VitEncoder
    from tests.test_attention_gpu import _ref
    B, T, H, hd = 20, 256, 16, 80
    qkv = (torch.randn(B, T, 3, H, hd, generator=torch.Generator().manual_seed(8)) * 1.2).to(torch.bfloat16).to(dev)
    o = torch.full((B, T, H * hd), float("nan"), dtype=torch.bfloat16, device=dev)
    lse = torch.full((B, H, T), float("nan"), device=dev)
    ops.attention_fwd(qkv, o, lse, B, T, H, hd)
    ro, rl = _ref(qkv, B, T, H, hd)
    torch.testing.assert_close(lse, rl, atol=2e-3, rtol=1e-4)
    torch.testing.assert_close(o.float(), ro, atol=2e-2, rtol=2e-2)
    enc = VitEncoder(embed=1280, depth=2, heads=16, patch=14, image=224, cls=False, final_norm=True).to(dev).eval()
    for p in enc.parameters():
        if p.ndim > 1:
            torch.nn.init.normal_(p, std=0.02)
    out = enc(torch.randn(2, 3, 224, 224, device=dev))
    assert out.shape == (2, 256, 1280) and out.dtype == torch.float32 and bool(torch.isfinite(out).all())
