"""Attention parity on the GPU vs a plain PyTorch fp32 softmax(q k^T / sqrt(hd)) v on the same
bf16-rounded q, k, v (timm Attention semantics, qkv layout [B,T,3,H,hd])."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(qkv, B, T, H, hd):
    q, k, v = qkv.float().reshape(B, T, 3, H, hd).permute(2, 0, 3, 1, 4).unbind(0)
    s = (q @ k.transpose(-1, -2)) * hd ** -0.5
    p = s.softmax(-1)
    o = (p @ v).transpose(1, 2).reshape(B, T, H * hd)
    return o, torch.logsumexp(s, -1)


# (24, 256, 16, 72), (37, 256, 16, 64), (70, 100, 6, 72): more (batch, head) items than CUs — the persistent T <= 256
# forward walks several items per workgroup (LDS-DMA ring, counted waits), some workgroups one item more than others;
# 257, 261, 272 (hd 64): the ViT towers' lengths — the last <= 16 query rows go to the row kernel (attn_fwd_rows_kernel),
# 273: one row too many for it
@pytest.mark.parametrize("B,T,H,hd", [(2, 256, 16, 72), (3, 256, 6, 64), (2, 64, 4, 72), (1, 16, 2, 64),
                                       (1, 100, 2, 72), (1, 1024, 2, 72), (1, 320, 3, 64), (24, 256, 16, 72),
                                       (37, 256, 16, 64), (70, 100, 6, 72), (1, 257, 2, 64), (3, 200, 5, 72),
                                       (3, 261, 4, 64), (20, 257, 16, 64), (2, 272, 3, 64), (2, 273, 3, 64)])
def test_attention_fwd(dev, B, T, H, hd):
    from reed_amd import ops
    g = torch.Generator().manual_seed(B * T + hd)
    qkv = (torch.randn(B, T, 3, H, hd, generator=g) * 1.5).to(torch.bfloat16).to(dev)
    o = torch.full((B, T, H * hd), float("nan"), dtype=torch.bfloat16, device=dev)
    lse = torch.full((B, H, T), float("nan"), device=dev)
    ops.attention_fwd(qkv, o, lse, B, T, H, hd)
    ro, rl = _ref(qkv, B, T, H, hd)
    torch.testing.assert_close(lse, rl, atol=2e-3, rtol=1e-4)
    torch.testing.assert_close(o.float(), ro, atol=2e-2, rtol=2e-2)


def test_attention_fwd_exact_pattern(dev):
    """one-hot attention (huge logit on one key) must copy that key's value row exactly:
    catches any k-slot / transposed-read permutation error."""
    from reed_amd import ops
    B, T, H, hd = 1, 256, 2, 72
    qkv = torch.zeros(B, T, 3, H, hd)
    perm = torch.randperm(T, generator=torch.Generator().manual_seed(3))
    # q_t = e_{c(t)} * big ; k_s = e_{c'(s)} so q_t.k_s is big iff s == perm[t]
    for t in range(T):
        qkv[0, t, 0, :, t % hd] = 64.0 * (1 + t // hd)
    for s in range(T):
        qkv[0, s, 1, :, s % hd] = 8.0 * (1 + s // hd)
    vals = (torch.arange(T * hd).reshape(T, hd) % 127).float()
    qkv[0, :, 2, 0] = vals
    qkv[0, :, 2, 1] = vals.flip(0)
    qkv = qkv.to(torch.bfloat16).to(dev)
    o = torch.zeros(B, T, H * hd, dtype=torch.bfloat16, device=dev)
    lse = torch.zeros(B, H, T, device=dev)
    ops.attention_fwd(qkv, o, lse, B, T, H, hd)
    ro, _ = _ref(qkv, B, T, H, hd)
    torch.testing.assert_close(o.float(), ro, atol=1e-2, rtol=1e-2)


# (40, 128, 16, 72), (64, 64, 12, 64), (30, 200, 10, 72): T < 256 with more (batch, head) items than CUs — the tile form of the
# persistent kernel (attn_bwd_ksp_kernel) walks several items per workgroup with the next item's tiles in flight under the current
# one: a stale tile or a miscounted wait in that loop shows up against the independent fp32 reference here
@pytest.mark.parametrize("B,T,H,hd", [(2, 256, 16, 72), (2, 256, 6, 64), (2, 64, 4, 72), (1, 16, 2, 64), (1, 100, 2, 72),
                                       (3, 200, 5, 72), (2, 129, 3, 64), (5, 128, 4, 72), (20, 256, 16, 72),
                                       (40, 128, 16, 72), (64, 64, 12, 64), (30, 200, 10, 72)])
@pytest.mark.parametrize("form", ["ws", "plain"])
def test_attention_bwd(dev, B, T, H, hd, form):
    """form "ws": the persistent key-stationary backward behind reed_attention_bwd_ws (delta by a row kernel, the next item's
    operands in flight under the current one) — what the engine calls; "plain": the workspace-free entry point."""
    from reed_amd import ops
    g = torch.Generator().manual_seed(5 + T)
    qkv = (torch.randn(B, T, 3, H, hd, generator=g)).to(torch.bfloat16).to(dev)
    do = (torch.randn(B, T, H * hd, generator=g)).to(torch.bfloat16).to(dev)
    o = torch.zeros(B, T, H * hd, dtype=torch.bfloat16, device=dev)
    lse = torch.zeros(B, H, T, device=dev)
    ops.attention_fwd(qkv, o, lse, B, T, H, hd)
    dqkv = torch.full_like(qkv, float("nan"))
    ws = torch.full((ops.attention_bwd_ws_floats(B, T, H),), float("nan"), device=dev) if form == "ws" else None
    ops.attention_bwd(qkv, o, do, lse, dqkv, B, T, H, hd, ws=ws)
    q32 = qkv.float().requires_grad_(True)
    ro, _ = _ref(q32, B, T, H, hd)
    ro.backward(do.float())
    ref = q32.grad
    nf = ~torch.isfinite(dqkv.float())
    assert not nf.any(), (int(nf.sum()), nf.nonzero()[:6].tolist())
    for w, name in enumerate("qkv"):
        a, r = dqkv[:, :, w].float(), ref[:, :, w]
        err = (a - r).abs().max().item()
        assert err <= 3e-2 * max(1.0, r.abs().max().item()), (name, err, r.abs().max().item())
        cos = torch.nn.functional.cosine_similarity(a.flatten(), r.flatten(), dim=0).item()
        assert cos > 0.9995, (name, cos)


def test_attention_deterministic_and_forms_agree(dev):
    """Same inputs twice -> identical bits, forward and backward (no atomics, fixed summation order), with more (batch, head)
    items than CUs so that the persistent kernels' item loops, prefetches and counted waits are in play (a tile overwritten
    before its last reader, or read before it landed, shows up as run-to-run differences); and the persistent backward agrees
    with the workspace-free one to bf16 resolution (the same products; delta summed in another order)."""
    from reed_amd import ops
    B, T, H, hd = 40, 256, 16, 72
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(B, T, 3, H, hd, generator=g).to(torch.bfloat16).to(dev)
    do = torch.randn(B, T, H * hd, generator=g).to(torch.bfloat16).to(dev)
    outs = []
    for rep in range(3):
        o = torch.zeros(B, T, H * hd, dtype=torch.bfloat16, device=dev)
        lse = torch.zeros(B, H, T, device=dev)
        dqkv = torch.zeros_like(qkv)
        ops.attention_fwd(qkv, o, lse, B, T, H, hd)
        ws = torch.empty(ops.attention_bwd_ws_floats(B, T, H), device=dev) if rep < 2 else None
        ops.attention_bwd(qkv, o, do, lse, dqkv, B, T, H, hd, ws=ws)
        torch.cuda.synchronize()
        outs.append((o.clone(), lse.clone(), dqkv.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    a, b = outs[0][2].float(), outs[2][2].float()
    assert (a - b).abs().max().item() <= 2e-2 * b.abs().max().item()
    assert torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item() > 0.99999


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("b,H,hd,force,T", [(64, 16, 72, 258, 256), (64, 6, 64, 0, 256), (100, 16, 72, 0, 256), (33, 12, 64, 257, 256),
                                            (256, 16, 72, 0, 256), (128, 16, 72, 258, 128), (50, 12, 64, 257, 64)])
def test_delta_from_the_dgrad_epilogue(dev, prec, b, H, hd, force, T):
    """reed_gemm epilogue 13 + reed_attention_bwd_dp (what the engine runs where the dO GEMM is on the four-wave 256^2 kernel):
    dO = dY W stored exactly as epilogue 0 stores it, the partial dot products dO . O per row and head summing to the row kernel's
    delta (fp32 rounding of another summation order), every slot written exactly once (NaN-filled before), and the attention
    backward through them equal to the workspace form to bf16 resolution; twice -> identical bits.  hd 72: heads straddle the
    64-column strips (two slots); hd 64: one strip per head; b = 100 / 33: a ragged last row tile, N = 768 a ragged column tile;
    force 257 / 258: the one-shot / persistent form of the kernel where the heuristics would take another one; b = 256 is the
    bench's shape; T = 128 / 64: the tile form of the persistent kernel behind the same entry."""
    from reed_amd import ops
    D = H * hd
    M = b * T
    hdt = ops.half_dtype(prec)
    g = torch.Generator().manual_seed(b + hd)
    dy = (torch.randn(M, D, generator=g) * 0.5).to(hdt).to(dev)
    w = (torch.randn(D, D, generator=g) / D ** 0.5).to(hdt).to(dev)
    qkv = torch.randn(b, T, 3, H, hd, generator=g).to(hdt).to(dev)
    prev = ops.use(prec)
    ops.gemm_force_tile(force)
    try:
        o = torch.zeros(b, T, D, dtype=hdt, device=dev)
        lse = torch.zeros(b, H, T, device=dev)
        ops.attention_fwd(qkv, o, lse, b, T, H, hd)
        do0 = torch.empty(M, D, dtype=hdt, device=dev)
        ops.gemm(ops.NN, ops.EPI_BF16, dy, w, M, D, D, do0, D, D, D)
        S = 1 if hd == 64 else 2
        runs = []
        for _ in range(2):
            do1 = torch.full((M, D), float("nan"), dtype=hdt, device=dev)
            dpart = torch.full((H, S, M), float("nan"), device=dev)
            if not ops.dgrad_with_head_dots(dy, w, do1, o, dpart, M, D, D, hd):
                pytest.skip("this shape's GEMM kernel has no head-dot epilogue")
            dq = torch.full_like(qkv, float("nan"))
            ws = torch.full((ops.attention_bwd_ws_floats(b, T, H),), float("nan"), device=dev)
            ops.attention_bwd_dp(qkv, do1, lse, dpart, dq, ws, b, T, H, hd)
            torch.cuda.synchronize()
            runs.append((do1, dpart, dq))
        do1, dpart, dq = runs[0]
        assert all(torch.equal(x, y) for x, y in zip(runs[0], runs[1]))
        assert torch.equal(do1, do0) and torch.isfinite(dpart).all()
        want = (do0.double() * o.view(M, D).double()).view(M, H, hd).sum(-1)
        torch.testing.assert_close(dpart.sum(1).T.double(), want, rtol=2e-5, atol=2e-5 * float(want.abs().max()))
        dq0 = torch.full_like(qkv, float("nan"))
        ops.attention_bwd(qkv, o, do0, lse, dq0, b, T, H, hd, ws=torch.empty_like(ws))
        a, r = dq.float(), dq0.float()
        assert torch.isfinite(a).all() and (a - r).abs().max().item() <= 1e-2 * r.abs().max().item()
    finally:
        ops.gemm_force_tile(0)
        ops.use(prev)


@pytest.mark.parametrize("B,T,H,hd", [(40, 128, 16, 72), (64, 64, 12, 64)])
def test_attention_small_t_item_loop_is_deterministic_and_matches_the_plain_form(dev, B, T, H, hd):
    """T < 256, items > CUs: the persistent tile-form backward twice -> identical bits, and equal to the workspace-free one-shot
    kernel to bf16 resolution (the same products, delta summed in another order)."""
    from reed_amd import ops
    g = torch.Generator().manual_seed(T + hd)
    qkv = torch.randn(B, T, 3, H, hd, generator=g).to(torch.bfloat16).to(dev)
    do = torch.randn(B, T, H * hd, generator=g).to(torch.bfloat16).to(dev)
    o = torch.zeros(B, T, H * hd, dtype=torch.bfloat16, device=dev)
    lse = torch.zeros(B, H, T, device=dev)
    ops.attention_fwd(qkv, o, lse, B, T, H, hd)
    outs = []
    for rep in range(3):
        dqkv = torch.full_like(qkv, float("nan"))
        ws = torch.empty(ops.attention_bwd_ws_floats(B, T, H), device=dev) if rep < 2 else None
        ops.attention_bwd(qkv, o, do, lse, dqkv, B, T, H, hd, ws=ws)
        torch.cuda.synchronize()
        outs.append(dqkv)
    assert torch.equal(outs[0], outs[1])
    a, b = outs[0].float(), outs[2].float()
    assert torch.isfinite(a).all() and (a - b).abs().max().item() <= 2e-2 * b.abs().max().item()
    assert torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item() > 0.99999


@pytest.mark.parametrize("B,T,H,hd", [(80, 256, 16, 72), (40, 256, 16, 72), (70, 128, 16, 72), (33, 256, 12, 64)])
def test_attention_bwd_beside_a_collective_is_the_same_bits(dev, B, T, H, hd):
    """ops.set_concurrent_comm(True) — what the data-parallel backward sets while gradient buckets are in flight — makes the
    persistent backward launch four short item lists per CU instead of one long one (csrc/attention.hip:attention_bwd_persistent;
    with CUs held by RCCL's channels the one-per-CU grid takes + 42 %, profiles/r4_kernels_under_cu_hog.txt).  The items are the
    same and independent: identical bits, with more items than 4 x CUs (1280), fewer (640, 528: lists of 1-3), and the T < 256 kernel."""
    from reed_amd import ops
    ops.set_comm_forms(True)   # (a tuner or REED_COMM_FORMS=0 may have switched them off)
    g = torch.Generator().manual_seed(B + T)
    qkv = torch.randn(B, T, 3, H, hd, generator=g).to(torch.bfloat16).to(dev)
    do = torch.randn(B, T, H * hd, generator=g).to(torch.bfloat16).to(dev)
    o = torch.zeros(B, T, H * hd, dtype=torch.bfloat16, device=dev)
    lse = torch.zeros(B, H, T, device=dev)
    ops.attention_fwd(qkv, o, lse, B, T, H, hd)
    ws = torch.empty(ops.attention_bwd_ws_floats(B, T, H), device=dev)
    outs = []
    for comm in (False, True, True):
        dqkv = torch.full_like(qkv, float("nan"))
        ops.set_concurrent_comm(comm)
        try:
            ops.attention_bwd(qkv, o, do, lse, dqkv, B, T, H, hd, ws=ws)
        finally:
            ops.set_concurrent_comm(False)
        torch.cuda.synchronize()
        outs.append(dqkv)
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])


@pytest.mark.parametrize("B,T,H,hd", [(40, 256, 16, 72), (33, 256, 12, 64), (70, 128, 16, 72)])
@pytest.mark.parametrize("reserve", [32, 96, 200])
def test_attention_grids_follow_the_cu_reserve_with_the_same_bits(dev, B, T, H, hd, reserve):
    """Round 6: the persistent forward and backward size their grids by reed_planning_cus() — the device's CUs minus
    ops.set_cu_reserve (RCCL's channels on a node; a kernel run beside them) — instead of the device's count.  The items are the same
    and independent whatever the grid (224, 160 and 56 workgroups here): identical bits for O, lse and dqkv."""
    from reed_amd import ops
    g = torch.Generator().manual_seed(B + T + reserve)
    qkv = torch.randn(B, T, 3, H, hd, generator=g).to(torch.bfloat16).to(dev)
    do = torch.randn(B, T, H * hd, generator=g).to(torch.bfloat16).to(dev)
    ws = torch.empty(ops.attention_bwd_ws_floats(B, T, H), device=dev)
    outs = []
    for r in (0, reserve, reserve):
        o = torch.full((B, T, H * hd), float("nan"), dtype=torch.bfloat16, device=dev)
        lse = torch.full((B, H, T), float("nan"), device=dev)
        dqkv = torch.full_like(qkv, float("nan"))
        ops.set_cu_reserve(r)
        try:
            ops.attention_fwd(qkv, o, lse, B, T, H, hd)
            ops.attention_bwd(qkv, o, do, lse, dqkv, B, T, H, hd, ws=ws)
        finally:
            ops.set_cu_reserve(0)
        torch.cuda.synchronize()
        outs.append((o, lse, dqkv))
    assert all(torch.isfinite(t.float()).all() for t in outs[0])
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    for a, b in zip(outs[1], outs[2]):
        assert torch.equal(a, b)
