"""GEMM parity on the GPU: HIP bf16-MFMA kernels vs a plain PyTorch fp32 reference of the same
op on bf16-rounded operands (tolerance: fp32 accumulation-order noise + one bf16 output rounding)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[0, 128, 256, 144, 257, 258, 64], autouse=True)
def force_tile(request):
    """Run every GEMM test with the tile heuristic and with each kernel forced (128^2 2-stage, 256^2 pipelined with eight
    waves, 256x144 ring — only NT / NN shapes whose N is a multiple of 144 —, 257 = the 256^2 tile with four 128x128 waves —
    NT / NN with K >= 128 and the epilogues it builds —, 64 = the skinny kernel of 16 x 64 one-wave tiles (csrc/gemm_skinny.hip: NT,
    the tail launch of the ragged-M split); what a kernel does not take falls to the heuristic)."""
    from reed_amd import ops
    ops.gemm_force_tile(request.param)
    yield request.param
    ops.gemm_force_tile(0)


def _bf(x):
    return x.to(torch.bfloat16)


def _gelu(x):
    return torch.nn.functional.gelu(x, approximate="tanh")


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (512, 384, 1152), (200, 256, 128), (8, 128, 256), (1024, 1152, 4608),
                                   (700, 640, 192), (2048, 1152, 1152)])
def test_nt_bias(dev, M, N, K):
    from reed_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    b = _bf(torch.randn(N, generator=g)).to(dev)
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.linear_fwd(x, w, b, out)
    ref = (x.float() @ w.float().t() + b.float())
    torch.testing.assert_close(out.float(), ref.to(torch.bfloat16).float(), atol=2e-2, rtol=2e-2)
    # tighter: compare against fp32 ref with bf16 half-ulp slack
    err = (out.float() - ref).abs().max().item()
    assert err <= ref.abs().max().item() * 2 ** -7, err


def test_nt_asymmetric_identity(dev):
    """A = I against an asymmetric B catches transposed C writes / swapped fragment maps."""
    from reed_amd import ops
    N = K = 128
    M = 128
    x = torch.eye(M, K, device=dev).to(torch.bfloat16)
    w = (torch.arange(N * K, device=dev).reshape(N, K) % 251).float().to(torch.bfloat16)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    ops.linear_fwd(x, w, None, out)
    assert torch.equal(out.float(), w.float().t().contiguous())


@pytest.mark.parametrize("N", [256, 384, 1152])   # 384, 1152: ragged last 256-column tile (re-dealt wave grid)
def test_epilogues(dev, N):
    from reed_amd import ops
    M, K, T = 512, 128, 256
    g = torch.Generator().manual_seed(1)
    x = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.1).to(dev)
    b = _bf(torch.randn(N, generator=g)).to(dev)
    pre_ref = _bf(x.float() @ w.float().t() + b.float())
    # gelu / silu
    for epi, fn in ((ops.EPI_GELU, _gelu), (ops.EPI_SILU, torch.nn.functional.silu)):
        pre = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        act = torch.zeros_like(pre)
        ops.linear_fwd(x, w, b, pre, epi=epi, act_out=act)
        torch.testing.assert_close(pre.float(), pre_ref.float(), atol=2e-2, rtol=2e-2)
        torch.testing.assert_close(act.float(), _bf(fn(pre.float())).float(), atol=1e-2, rtol=1e-2)
    # the derivative-saving forms (round 5): the same activation bit for bit, and act'(pre) where the pre-activation was
    for epi, epi_g, fn in ((ops.EPI_GELU, ops.EPI_GELU_G, _gelu), (ops.EPI_SILU, ops.EPI_SILU_G, torch.nn.functional.silu)):
        pre = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        act, act_g, der = torch.zeros_like(pre), torch.zeros_like(pre), torch.zeros_like(pre)
        ops.linear_fwd(x, w, b, pre, epi=epi, act_out=act)
        ops.linear_fwd(x, w, b, der, epi=epi_g, act_out=act_g)
        assert torch.equal(act, act_g)
        p32 = pre.float().requires_grad_(True)
        fn(p32).sum().backward()
        torch.testing.assert_close(der.float(), _bf(p32.grad).float(), atol=2 ** -7, rtol=2 ** -7)   # one bf16 ulp of a value <= 1.13
        # without the saved array (an inference forward): activation only
        act_n = torch.zeros_like(pre)
        ops.linear_fwd(x, w, b, None, epi=epi_g, act_out=act_n, M=M, N=N, K=K, ldx=K, ldw=K, ldo=N)
        assert torch.equal(act, act_n)
    # gate + residual
    gate = _bf(torch.randn(M // T, 3 * N, generator=g)).to(dev)
    xin = torch.randn(M, N, generator=g).to(dev)
    xout = torch.zeros(M, N, device=dev)
    y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    gv = gate[:, N:2 * N]
    ops.linear_fwd(x, w, b, xout, epi=ops.EPI_GATE_RES, R=xin, gate=gv, ldgate=gate.stride(0),
                   rows_per_gate=T, y_out=y)
    torch.testing.assert_close(y.float(), pre_ref.float(), atol=2e-2, rtol=2e-2)
    ref = xin + _bf(gv.float().repeat_interleave(T, 0) * y.float()).float()
    torch.testing.assert_close(xout, ref, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("M,N,K", [(256, 128, 128), (300, 1152, 384), (8, 256, 1152), (1000, 4608, 1152), (513, 640, 384)])
def test_nn_dgrad(dev, M, N, K):
    """dx[M,K] = dy[M,N] @ w[N,K]"""
    from reed_amd import ops
    g = torch.Generator().manual_seed(7)
    dy = _bf(torch.randn(M, N, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    dx = torch.zeros(M, K, dtype=torch.bfloat16, device=dev)
    ops.linear_dgrad(dy, w, dx)
    ref = dy.float() @ w.float()
    err = (dx.float() - ref).abs().max().item()
    assert err <= ref.abs().max().item() * 2 ** -7, err
    # dgelu epilogue
    pre = _bf(torch.randn(M, K, generator=g)).to(dev)
    dx2 = torch.zeros_like(dx)
    ops.linear_dgrad(dy, w, dx2, epi=ops.EPI_DGELU, R=pre)
    p32 = pre.float().requires_grad_(True)
    _gelu(p32).backward(_bf(ref).float())
    torch.testing.assert_close(dx2.float(), p32.grad, atol=3e-2, rtol=3e-2)
    # the one-multiply form (round 5): exact against its definition bf16(bf16(acc) * R), and — fed the rounded derivative the
    # forward's EPI_GELU_G saves — within one bf16 rounding of the factor of the recomputing epilogue above
    dx3 = torch.zeros_like(dx)
    ops.linear_dgrad(dy, w, dx3, epi=ops.EPI_MUL, R=pre)
    assert torch.equal(dx3, _bf(dx.float() * pre.float()))
    q32 = pre.float().requires_grad_(True)
    _gelu(q32).sum().backward()
    der = _bf(q32.grad)
    ops.linear_dgrad(dy, w, dx3, epi=ops.EPI_MUL, R=der)
    torch.testing.assert_close(dx3.float(), dx2.float(), atol=2 ** -7 * dx2.float().abs().max().item(), rtol=2 ** -6)


def test_nn_asymmetric(dev):
    from reed_amd import ops
    M = N = K = 128
    dy = torch.eye(M, N, device=dev).to(torch.bfloat16)
    w = (torch.arange(N * K, device=dev).reshape(N, K) % 251).float().to(torch.bfloat16)
    dx = torch.zeros(M, K, dtype=torch.bfloat16, device=dev)
    ops.linear_dgrad(dy, w, dx)
    assert torch.equal(dx.float(), w.float())


@pytest.mark.parametrize("Mtok,N,K,split", [(256, 128, 128, 1), (1000, 384, 256, 1), (8, 128, 128, 1), (2048, 256, 128, 4),
                                            (4100, 1152, 640, 1), (8192, 384, 1152, 3)])
def test_tn_wgrad(dev, Mtok, N, K, split, force_tile):
    """dw[N,K] = dy[Mtok,N]^T @ x[Mtok,K]; dbias = colsum(dy)"""
    from reed_amd import ops
    if force_tile == 256:   # the 256^2 TN kernel has no fused bias gradient: check dw there, dbias via colsum
        g = torch.Generator().manual_seed(11)
        dy = _bf(torch.randn(Mtok, N, generator=g)).to(dev)
        x = _bf(torch.randn(Mtok, K, generator=g)).to(dev)
        dw = torch.full((N, K), float("nan"), device=dev)
        ops.linear_wgrad(dy, x, dw, split_k=split)
        torch.testing.assert_close(dw, dy.float().t() @ x.float(), atol=1e-2, rtol=1e-3)
        ws = torch.empty(ops.colsum_ws_floats(Mtok, N), device=dev)
        db = torch.zeros(N, device=dev)
        ops.colsum_bf16(dy, N, ws, db, Mtok, N)
        torch.testing.assert_close(db, dy.float().sum(0), atol=1e-2, rtol=1e-3)
        return
    g = torch.Generator().manual_seed(11)
    dy = _bf(torch.randn(Mtok, N, generator=g)).to(dev)
    x = _bf(torch.randn(Mtok, K, generator=g)).to(dev)
    dw = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)
    ops.linear_wgrad(dy, x, dw, dbias=db, split_k=split)
    ref = dy.float().t() @ x.float()
    torch.testing.assert_close(dw, ref, atol=1e-2, rtol=1e-3)
    torch.testing.assert_close(db, dy.float().sum(0), atol=1e-2, rtol=1e-3)
    if split == 1:
        ops.linear_wgrad(dy, x, dw, dbias=db, accumulate=True)
        torch.testing.assert_close(dw, 2 * ref, atol=2e-2, rtol=1e-3)
        torch.testing.assert_close(db, 2 * dy.float().sum(0), atol=2e-2, rtol=1e-3)


@pytest.mark.parametrize("lay", ["tall", "wide"])
@pytest.mark.parametrize("Mtok,N,K,split", [(256, 128, 256, 1), (1000, 384, 256, 1), (40, 128, 512, 1), (2048, 256, 256, 4),
                                            (4100, 1152, 768, 1), (8192, 384, 1280, 3), (8192, 3456, 1152, 4),
                                            (4096, 1152, 4608, 3)])
def test_tn_wgrad_tall_wide_tiles(dev, lay, Mtok, N, K, split, force_tile):
    """gemm_tn.hip: the 256x128 / 128x256 weight-gradient tiles (128x64 / 64x128 per wave, BK 32) against fp32 torch
    and against the 128^2 kernel: ragged last tile row (N % 256 != 0), ragged token count (Mtok % 32 != 0), split-K
    slabs with the fused bias gradient in the slab, accumulate."""
    from reed_amd import ops
    if force_tile != 0:
        pytest.skip("tile forcing does not apply to the explicit TN tile layouts")
    L = ops.TN_TALL if lay == "tall" else ops.TN_WIDE
    if lay == "wide" and K % 256:
        pytest.skip("128x256 tile needs K % 256 == 0")
    g = torch.Generator().manual_seed(13)
    dy = _bf(torch.randn(Mtok, N, generator=g)).to(dev)
    x = _bf(torch.randn(Mtok, K, generator=g)).to(dev)
    out = torch.full((N * K + N,), float("nan"), device=dev)
    dw, db = out[:N * K].view(N, K), out[N * K:]
    ops.linear_wgrad(dy, x, dw, dbias=db, split_k=split, lay=L)       # bias right behind the weight: one slab reduce
    ref = dy.float().t() @ x.float()
    torch.testing.assert_close(dw, ref, atol=1e-2, rtol=1e-3)
    torch.testing.assert_close(db, dy.float().sum(0), atol=1e-2, rtol=1e-3)
    dw0, db0 = torch.zeros(N, K, device=dev), torch.zeros(N, device=dev)
    ops.linear_wgrad(dy, x, dw0, dbias=db0, split_k=split, lay=ops.TN)
    torch.testing.assert_close(dw, dw0, atol=2e-3, rtol=1e-4)        # same products, different summation split
    if split == 1:
        ops.linear_wgrad(dy, x, dw, dbias=db, accumulate=True, lay=L)
        torch.testing.assert_close(dw, 2 * ref, atol=2e-2, rtol=1e-3)
        torch.testing.assert_close(db, 2 * dy.float().sum(0), atol=2e-2, rtol=1e-3)
    # identity operand: catches a transposed / shifted fragment map exactly
    if Mtok == 256:
        eye = torch.eye(Mtok, N, device=dev).to(torch.bfloat16)
        xx = (torch.arange(Mtok * K, device=dev).reshape(Mtok, K) % 251).float().to(torch.bfloat16)
        d2 = torch.zeros(N, K, device=dev)
        ops.linear_wgrad(eye, xx, d2, lay=L)
        assert torch.equal(d2, xx.float()[:N])


@pytest.mark.parametrize("tokens,shapes", [
    (512, [(1152, 4608), (4608, 1152), (1152, 1152), (3456, 1152)]),     # the SiT-XL/2 block: 162 + 162 + 45 + 126 tiles
    (1000, [(1152, 4608), (4608, 1152), (1152, 1152), (3456, 1152)]),    # ragged token count
    (2048, [(1152, 4608), (4608, 1152), (1152, 1152), (3456, 1152)]),    # (force_tile 0: gemm256w.hip's one-item-per-CU form at every
                                                                         # token count since round 6)
    (256, [(384, 256), (640, 128)]),                                     # ragged last tile rows (1.5 and 2.5 tiles of 256)
    (300, [(128, 128)]),
])
def test_wgrad_group(dev, tokens, shapes, force_tile):
    """reed_wgrad_group (csrc/gemm_tn.hip): the weight + bias gradients of up to four linears in one launch without
    split-K, against fp32 torch and against the per-GEMM path; accumulate; run-to-run bit-identical; identity operand."""
    from reed_amd import ops
    if force_tile not in (0, 128):   # 0: gemm256w.hip's one-item-per-CU form for the XL/2 block; 128: always gemm_tn.hip's grouped kernel
        pytest.skip("two kernels: force_tile 0 / 128")
    g = torch.Generator().manual_seed(17)
    probs, refs = [], []
    for n_out, k_in in shapes:
        dy = _bf(torch.randn(tokens, n_out, generator=g)).to(dev)
        x = _bf(torch.randn(tokens, k_in, generator=g)).to(dev)
        out = torch.full((n_out * k_in + n_out,), float("nan"), device=dev)
        probs.append((dy, x, out[:n_out * k_in].view(n_out, k_in), out[n_out * k_in:], n_out, k_in))
        refs.append((dy.float().t() @ x.float(), dy.float().sum(0)))
    assert ops.wgrad_group(probs, tokens)
    first = [(q[2].clone(), q[3].clone()) for q in probs]
    for (dw, db), (rw, rb), q in zip(first, refs, probs):
        torch.testing.assert_close(dw, rw, atol=1e-2, rtol=1e-3)
        torch.testing.assert_close(db, rb, atol=1e-2, rtol=1e-3)
        dw0, db0 = torch.zeros_like(dw), torch.zeros_like(db)
        ops.gemm_force_tile(128)
        ops.linear_wgrad(q[0], q[1], dw0, dbias=db0, lay=ops.TN)
        ops.gemm_force_tile(force_tile)
        torch.testing.assert_close(dw, dw0, atol=2e-3, rtol=1e-4)        # same products, different summation order
    for q in probs:
        q[2].fill_(float("nan"))
    assert ops.wgrad_group(probs, tokens)
    assert all(torch.equal(q[2], f[0]) and torch.equal(q[3], f[1]) for q, f in zip(probs, first))
    assert ops.wgrad_group(probs, tokens, accumulate=True)
    for q, (rw, rb) in zip(probs, refs):
        torch.testing.assert_close(q[2], 2 * rw, atol=2e-2, rtol=1e-3)
        torch.testing.assert_close(q[3], 2 * rb, atol=2e-2, rtol=1e-3)
    # without bias gradients, identity dy: dw = the first n_out rows of x exactly
    n_out, k_in = shapes[-1]
    eye = torch.eye(tokens, n_out, device=dev).to(torch.bfloat16)
    xx = (torch.arange(tokens * k_in, device=dev).reshape(tokens, k_in) % 251).float().to(torch.bfloat16)
    d2 = torch.full((n_out, k_in), float("nan"), device=dev)
    assert ops.wgrad_group([(eye, xx, d2, None, n_out, k_in)], tokens)
    assert torch.equal(d2[:min(n_out, tokens)], xx.float()[:min(n_out, tokens)])


def test_wgrad_group_beside_a_collective_is_the_two_workgroup_kernel(dev):
    """While gradient buckets are in flight (ops.set_concurrent_comm: the data-parallel backward) the grouped launch keeps to
    gemm_tn.hip's half-size tiles in dynamic order — the one-workgroup-per-CU form of gemm256w.hip takes twice as long when
    RCCL's channels hold CUs (profiles/r4_wgrad_under_cu_hog.txt).  Since round 6 both forms walk every tile's tokens as ONE
    sequence (no K-cut pieces): the N = 1 plan and the plan beside collectives give the same bits."""
    from reed_amd import ops
    ops.set_comm_forms(True)   # (a tuner or REED_COMM_FORMS=0 may have switched them off)
    tokens, shapes = 4096, [(1152, 4608), (4608, 1152), (1152, 1152), (3456, 1152)]
    g = torch.Generator().manual_seed(23)
    probs = []
    for n_out, k_in in shapes:
        dy = _bf(torch.randn(tokens, n_out, generator=g)).to(dev)
        x = _bf(torch.randn(tokens, k_in, generator=g)).to(dev)
        out = torch.full((n_out * k_in + n_out,), float("nan"), device=dev)
        probs.append((dy, x, out[:n_out * k_in].view(n_out, k_in), out[n_out * k_in:], n_out, k_in))

    def run(tile, comm):
        ops.gemm_force_tile(tile)
        ops.set_concurrent_comm(comm)
        try:
            for q in probs:
                q[2].fill_(float("nan"))
                q[3].fill_(float("nan"))
            assert ops.wgrad_group(probs, tokens)
            return [(q[2].clone(), q[3].clone()) for q in probs]
        finally:
            ops.set_concurrent_comm(False)
            ops.gemm_force_tile(0)

    two = run(128, False)
    beside = run(0, True)
    alone = run(0, False)
    assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(two, beside))
    assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(two, alone))


@pytest.mark.parametrize("tokens", [192, 1024, 8192])
@pytest.mark.parametrize("bias", ["all", "none", "mixed"])
def test_wgrad_group_items_bit_identical_to_the_two_workgroup_kernel(dev, tokens, bias):
    """Round 6's deal of the XL/2 block (tests/test_host_cpu.py checks its cover on the host): 212 full tiles + 24 items of
    384 x 128 + 18 of 128 x 384 + 2 bias-only items, every one a whole-K sequence — the same bits as gemm_tn.hip's grouped kernel
    for every weight and bias gradient, with and without bias gradients (the items' fourth waves), for an odd K-tile count
    (192 tokens = 3 K-tiles), and accumulating."""
    from reed_amd import ops
    shapes = [(1152, 4608), (4608, 1152), (1152, 1152), (3456, 1152)]
    g = torch.Generator().manual_seed(29)
    probs = []
    for i, (n_out, k_in) in enumerate(shapes):
        dy = _bf(torch.randn(tokens, n_out, generator=g)).to(dev)
        x = _bf(torch.randn(tokens, k_in, generator=g)).to(dev)
        out = torch.full((n_out * k_in + n_out,), float("nan"), device=dev)
        has_b = bias == "all" or (bias == "mixed" and i % 2 == 0)
        probs.append((dy, x, out[:n_out * k_in].view(n_out, k_in), out[n_out * k_in:] if has_b else None, n_out, k_in))

    def run(tile, acc):
        ops.gemm_force_tile(tile)
        try:
            for q in probs:
                q[2].fill_(0.25 if acc else float("nan"))
                if q[3] is not None:
                    q[3].fill_(-0.5 if acc else float("nan"))
            assert ops.wgrad_group(probs, tokens, accumulate=acc)
            return [(q[2].clone(), None if q[3] is None else q[3].clone()) for q in probs]
        finally:
            ops.gemm_force_tile(0)

    for acc in (False, True):
        ref, got = run(128, acc), run(0, acc)
        for (rw, rb), (gw, gb), q in zip(ref, got, probs):
            assert torch.isfinite(gw).all()
            assert torch.equal(rw, gw)
            if rb is not None:
                assert torch.equal(rb, gb)
    # against fp32 torch
    for (gw, gb), q in zip(got, probs):
        torch.testing.assert_close(gw - 0.25, q[0].float().t() @ q[1].float(), atol=2e-2, rtol=1e-3)
        if gb is not None:
            torch.testing.assert_close(gb + 0.5, q[0].float().sum(0), atol=2e-2, rtol=1e-3)


def test_wgrad_group_planning(dev):
    """The SiT-XL/2 block fills the 512 workgroup slots exactly and is grouped; small models are left to split-K; a group
    that does not fit one round is refused without launching."""
    from reed_amd import ops
    D, Hm = 1152, 4608
    assert ops.wgrad_group_blocks([(D, Hm), (Hm, D), (D, D), (3 * D, D)]) == 512
    assert ops.wgrad_group_fits([(D, Hm), (Hm, D), (D, D), (3 * D, D)])
    assert not ops.wgrad_group_fits([(384, 1536), (1536, 384), (384, 384), (1152, 384)])      # SiT-S/2: 72 of 512
    assert not ops.wgrad_group_fits([(Hm, Hm)] * 4)
    z = torch.zeros(64, Hm, dtype=torch.bfloat16, device=dev)
    dw = torch.zeros(Hm, Hm, device=dev)
    assert ops.wgrad_group([(z, z, dw, None, Hm, Hm)] * 4, 64) is False


def test_plan_wgrad_choices():
    """ops.plan_wgrad on the SiT-XL/2 block shapes: the tile the wave-quantisation model picks (tools/wgrad_sweep.py)."""
    from reed_amd import ops
    M = 256 * 256
    assert ops.plan_wgrad(M, 4608, 1152) == (ops.TN_TALL, 3)
    assert ops.plan_wgrad(M, 1152, 4608) == (ops.TN_WIDE, 3)
    assert ops.plan_wgrad(M, 3456, 1152) == (ops.TN_TALL, 4)
    assert ops.plan_wgrad(M, 1152, 1152) == (ops.TN, 6)


def test_tn_asymmetric(dev):
    from reed_amd import ops
    Mtok, N, K = 128, 128, 128
    dy = torch.eye(Mtok, N, device=dev).to(torch.bfloat16)
    x = (torch.arange(Mtok * K, device=dev).reshape(Mtok, K) % 251).float().to(torch.bfloat16)
    dw = torch.zeros(N, K, device=dev)
    ops.linear_wgrad(dy, x, dw)
    assert torch.equal(dw, x.float())
    # and the other way: x = I
    x2 = torch.eye(Mtok, K, device=dev).to(torch.bfloat16)
    dy2 = (torch.arange(Mtok * N, device=dev).reshape(Mtok, N) % 241).float().to(torch.bfloat16)
    ops.linear_wgrad(dy2, x2, dw)
    assert torch.equal(dw, dy2.float().t().contiguous())


def test_gemm_arg_errors(dev):
    from reed_amd import ops
    x = torch.zeros(128, 64, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(100, 64, dtype=torch.bfloat16, device=dev)
    out = torch.zeros(128, 100, dtype=torch.bfloat16, device=dev)
    with pytest.raises(RuntimeError, match="multiple of 128"):
        ops.linear_fwd(x, w, None, out)


@pytest.mark.parametrize("lay", ["NT", "NN"])
@pytest.mark.parametrize("M,N,K", [(8192, 4608, 128), (9000, 2432, 320), (65536, 1152, 64), (16640, 1152, 256), (9000, 2432, 512), (16640, 1152, 384), (9000, 2432, 768)])
def test_many_tiles(dev, lay, M, N, K):
    """Several rounds of tiles per CU, ragged M edge and a half-empty last 256-column tile: every tile written exactly
    once through the LDS-staged epilogue, and the 128^2 and 256^2 kernels agree bit for bit."""
    from reed_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    b = _bf(torch.randn(N, generator=g)).to(dev)
    wq = w if lay == "NT" else w.t().contiguous()   # NN reads the k-strided operand [K, N]

    def run(tile):
        ops.gemm_force_tile(tile)
        out = torch.full((M + 1, N), float("nan"), dtype=torch.bfloat16, device=dev)   # +1 guard row
        if lay == "NT":
            ops.gemm(ops.NT, ops.EPI_BF16, x, wq, M, N, K, out, K, K, N, bias=b)
        else:
            ops.gemm(ops.NN, ops.EPI_BF16, x, wq, M, N, K, out, K, N, N, bias=b)
        return out

    o128, o256, o257, o258, o64 = run(128), run(256), run(257), run(258), run(64)
    ops.gemm_force_tile(0)
    assert torch.isnan(o64[M]).all() and torch.equal(o64[:M], o256[:M])   # the one-wave 16 x 64 tiles (NT): the same bits
    assert torch.isnan(o128[M]).all() and torch.isnan(o256[M]).all()       # nothing written past row M-1
    assert torch.isnan(o257[M]).all() and torch.isnan(o258[M]).all()
    # four 128x128 waves, one tile per workgroup / the persistent walk over the tile list / 128x256 tiles, two workgroups per
    # CU: the same bits
    assert torch.equal(o257[:M], o256[:M]) and torch.equal(o258[:M], o256[:M])
    ref = x.float() @ w.float().t() + b.float()
    err = (o256[:M].float() - ref).abs().max().item()
    assert err <= ref.abs().max().item() * 2 ** -7, err
    assert torch.equal(o128[:M], o256[:M])


@pytest.mark.parametrize("N", [2304, 1152])   # 1152 = 4.5 x 256: the last column tile takes the ragged path
def test_fused_epilogues_many_tiles(dev, N):
    """gate+residual and gelu epilogues over many tiles (gate rows change inside and across tiles)."""
    from reed_amd import ops
    M, K, T = 16384, 128, 64
    g = torch.Generator().manual_seed(5)
    x = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.1).to(dev)
    b = _bf(torch.randn(N, generator=g)).to(dev)
    pre_ref = _bf(x.float() @ w.float().t() + b.float())
    pre = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    act = torch.zeros_like(pre)
    ops.linear_fwd(x, w, b, pre, epi=ops.EPI_GELU, act_out=act)
    torch.testing.assert_close(pre.float(), pre_ref.float(), atol=2e-2, rtol=2e-2)
    torch.testing.assert_close(act.float(), _bf(_gelu(pre.float())).float(), atol=1e-2, rtol=1e-2)
    gate = _bf(torch.randn(M // T, N, generator=g)).to(dev)
    xin = torch.randn(M, N, generator=g).to(dev)
    xout = torch.zeros(M, N, device=dev)
    y = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    ops.linear_fwd(x, w, b, xout, epi=ops.EPI_GATE_RES, R=xin, gate=gate, ldgate=gate.stride(0), rows_per_gate=T,
                   y_out=y)
    ref = xin + _bf(gate.float().repeat_interleave(T, 0) * y.float()).float()
    torch.testing.assert_close(y.float(), pre_ref.float(), atol=2e-2, rtol=2e-2)
    torch.testing.assert_close(xout, ref, atol=1e-5, rtol=1e-5)
    # rows_per_gate not a multiple of 16: per-lane gate row (slow path)
    T2 = 40
    Mr = (M // T2) * T2
    gate2 = _bf(torch.randn(Mr // T2, N, generator=g)).to(dev)
    xout2 = torch.zeros(Mr, N, device=dev)
    ops.linear_fwd(x[:Mr], w, b, xout2, epi=ops.EPI_GATE_RES, R=xin[:Mr], gate=gate2, ldgate=gate2.stride(0),
                   rows_per_gate=T2, M=Mr, N=N, K=K, ldx=K, ldw=K, ldo=N)
    ref2 = xin[:Mr] + _bf(gate2.float().repeat_interleave(T2, 0) * pre_ref[:Mr].float()).float()
    torch.testing.assert_close(xout2, ref2, atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("lay", ["NT", "NN"])
@pytest.mark.parametrize("M,N,K", [(256, 144, 64), (300, 288, 192), (8, 144, 128), (8192, 1152, 1152), (2050, 3456, 128),
                                   (512, 4608, 64), (1000, 1152, 4608), (256, 1152, 192)])
def test_tile144(dev, lay, M, N, K, force_tile):
    """gemm144.hip (256x144 tile, 4 x 2 waves of 64x80 / 64x64, 3-slot LDS-DMA ring): 1, 2, 3 and many K-tiles, ragged M,
    N not a multiple of 128, the 16-column strip and the B piece (columns 128..143 of a tile); bit-identical to the 128^2
    kernel where that one applies (same products, same k order inside a lane)."""
    from reed_amd import ops
    if force_tile != 144:
        pytest.skip("runs once, forcing the tiles itself")
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    b = _bf(torch.randn(N, generator=g)).to(dev)
    wq = w if lay == "NT" else w.t().contiguous()

    def run(tile):
        ops.gemm_force_tile(tile)
        out = torch.full((M + 1, N), float("nan"), dtype=torch.bfloat16, device=dev)
        if lay == "NT":
            ops.gemm(ops.NT, ops.EPI_BF16, x, wq, M, N, K, out, K, K, N, bias=b)
        else:
            ops.gemm(ops.NN, ops.EPI_BF16, x, wq, M, N, K, out, K, N, N, bias=b)
        return out

    o = run(144)
    assert torch.isnan(o[M]).all() and not torch.isnan(o[:M]).any()
    ref = x.float() @ w.float().t() + b.float()
    err = (o[:M].float() - ref).abs().max().item()
    assert err <= ref.abs().max().item() * 2 ** -7, err
    if N % 128 == 0:
        assert torch.equal(o[:M], run(128)[:M])
    ops.gemm_force_tile(144)


def test_tile144_identity(dev, force_tile):
    """A = I against an asymmetric B on the 256x144 tile, both layouts: exact fragment / piece maps."""
    from reed_amd import ops
    if force_tile != 144:
        pytest.skip("runs once")
    M, N, K = 256, 288, 256
    x = torch.eye(M, K, device=dev).to(torch.bfloat16)
    w = (torch.arange(N * K, device=dev).reshape(N, K) % 251).float().to(torch.bfloat16)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(ops.NT, ops.EPI_BF16, x, w, M, N, K, out, K, K, N)
    assert torch.equal(out.float(), w.float().t().contiguous())
    out.zero_()
    ops.gemm(ops.NN, ops.EPI_BF16, x, w.t().contiguous(), M, N, K, out, K, N, N)
    assert torch.equal(out.float(), w.float().t().contiguous())


def test_concurrent_comm_switch_keeps_results(dev, force_tile):
    """reed_set_concurrent_comm(1) (a data-parallel step: collectives beside the GEMMs) only changes WHICH kernel runs — the
    persistent form of the four-wave kernel is not selected — never the result."""
    from reed_amd import ops
    if force_tile != 0:
        pytest.skip("heuristic path only")
    g = torch.Generator(device="cpu").manual_seed(11)
    M, N, K = 32768, 4608, 256
    x = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    outs = []
    try:
        for on in (0, 1):
            ops.set_concurrent_comm(on)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            ops.linear_fwd(x, w, None, out)
            outs.append(out)
    finally:
        ops.set_concurrent_comm(0)
    assert torch.equal(outs[0], outs[1])
    torch.testing.assert_close(outs[0][:512].float(), x[:512].float() @ w.float().t(), atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("reserve", [16, 40])
def test_persistent_form_with_cu_reserve(dev, force_tile, reserve):
    """With CUs held back for RCCL's channels the persistent form runs fewer workgroups per XCD (positions s + 30 k, s + 27 k of
    the run): every tile still computed exactly once, same bits as the eight-wave kernel."""
    from reed_amd import ops
    if force_tile != 258:
        pytest.skip("persistent form only")
    g = torch.Generator(device="cpu").manual_seed(reserve)
    M, N, K = 20000, 1152, 384   # nt = 6; ragged last row tile and ragged last column tile
    x = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    outs = {}
    try:
        ops.set_cu_reserve(reserve)
        for tile in (258, 256):
            ops.gemm_force_tile(tile)
            out = torch.full((M + 1, N), float("nan"), dtype=torch.bfloat16, device=dev)
            ops.gemm(ops.NT, ops.EPI_BF16, x, w, M, N, K, out, K, K, N)
            outs[tile] = out
    finally:
        ops.set_cu_reserve(0)
        ops.gemm_force_tile(258)
    assert torch.isnan(outs[258][M]).all() and not torch.isnan(outs[258][:M]).any()
    assert torch.equal(outs[258][:M], outs[256][:M])


@pytest.mark.parametrize("N,K", [(1024, 1024), (4096, 1024), (1024, 4096)])
def test_ragged_m_split_is_bit_invisible(dev, N, K, force_tile):
    """M = 64 x 257 (a ViT tower at batch 64): the dispatcher sends the 64 full tile rows and the 64 tail rows out as two
    launches where the ragged 65th row tile would cost a whole round of the chip (csrc/gemm.hip: ragged-M split) — outputs
    bit-identical to the one-launch form of a forced tile, for the plain, QuickGELU, GELU(erf), bf16-residual and LayerScale
    epilogues, and nothing written past row M - 1."""
    if force_tile != 0:
        pytest.skip("runs once, forcing the tiles itself")
    from reed_amd import ops
    M = 64 * 257
    g = torch.Generator().manual_seed(N + K)
    x = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    b = _bf(torch.randn(N, generator=g)).to(dev)
    r16 = _bf(torch.randn(M, N, generator=g)).to(dev)
    r32 = torch.randn(M, N, generator=g).to(dev)
    gamma = torch.randn(N, generator=g).to(dev)

    def run(tile):
        ops.gemm_force_tile(tile)
        try:
            o_plain = torch.full((M + 1, N), float("nan"), dtype=torch.bfloat16, device=dev)
            ops.gemm(ops.NT, ops.EPI_BF16, x, w, M, N, K, o_plain, K, K, N, bias=b)
            o_q = torch.full((M + 1, N), float("nan"), dtype=torch.bfloat16, device=dev)
            ops.gemm(ops.NT, ops.EPI_QGELU, x, w, M, N, K, None, K, K, N, C2=o_q, ldc2=N, bias=b)
            o_e = torch.full((M + 1, N), float("nan"), dtype=torch.bfloat16, device=dev)
            ops.gemm(ops.NT, ops.EPI_GELU_ERF, x, w, M, N, K, None, K, K, N, C2=o_e, ldc2=N, bias=b)
            o_r = torch.full((M + 1, N), float("nan"), dtype=torch.bfloat16, device=dev)
            ops.gemm(ops.NT, ops.EPI_RES_BF16, x, w, M, N, K, o_r, K, K, N, R=r16, ldr=N, bias=b)
            o_l = torch.full((M + 1, N), float("nan"), device=dev)
            ops.gemm(ops.NT, ops.EPI_LS_RES, x, w, M, N, K, o_l, K, K, N, R=r32, ldr=N, bias=b, gate=gamma)
            torch.cuda.synchronize()
            return o_plain, o_q, o_e, o_r, o_l
        finally:
            ops.gemm_force_tile(0)

    split, whole = run(0), run(256)
    for name, a, c in zip(("plain", "quickgelu", "gelu_erf", "res_bf16", "ls_res"), split, whole):
        assert torch.isnan(a[M].float()).all() and torch.isfinite(a[:M].float()).all(), name
        ne = (a[:M] != c[:M])
        assert not ne.any(), (name, int(ne.sum()), ne.nonzero()[:4].tolist(), a[:M][ne][:4].tolist(), c[:M][ne][:4].tolist())
    ref = _bf(x.float() @ w.float().t() + b.float()).float()
    torch.testing.assert_close(split[0][:M].float(), ref, atol=2e-2, rtol=2e-2)
    torch.testing.assert_close(split[2][:M].float(), _bf(torch.nn.functional.gelu(ref)).float(), atol=4e-2, rtol=2e-2)   # one bf16 ulp


@pytest.mark.parametrize("M,n_out,k_in", [(4096, 1152, 4608), (4096, 4608, 1152), (1024, 1152, 1152), (300 * 16, 3456, 1152)])
@pytest.mark.parametrize("epi", ["plain", "mul", "dgelu", "dot"])
def test_dgrad_as_nt_on_transposed_weights_is_bit_identical(dev, M, n_out, k_in, epi, force_tile):
    """Round 6: the blocks' input gradients dx = dy W run as NT GEMMs on a transposed copy W^T [k_in, n_out] (both operands
    k-contiguous) instead of NN GEMMs on W (k-strided operand, transposing LDS reads): the same products in the same order — the
    same bits, for the plain store, the activation-backward epilogues (16: multiply by the saved derivative; 4: dGELU) and the
    head-dot epilogue 13 (the attention backward's delta where dO is produced: NT too since round 6), on whatever tile runs."""
    from reed_amd import ops
    g = torch.Generator().manual_seed(M + n_out)
    dy = _bf(torch.randn(M, n_out, generator=g)).to(dev)
    w = _bf(torch.randn(n_out, k_in, generator=g) * 0.05).to(dev)
    wt = w.t().contiguous()
    r = _bf(torch.randn(M, k_in, generator=g)).to(dev)
    outs = []
    for use_t in (False, True):
        dx = torch.full((M, k_in), float("nan"), dtype=torch.bfloat16, device=dev)
        if epi == "dot":
            if k_in % 72:
                pytest.skip("head_dim 72 rows only")
            dpart = torch.full((k_in // 72, 2, M), float("nan"), device=dev)
            ok = ops.dgrad_with_head_dots(dy, w, dx, r, dpart, M, n_out, k_in, 72, wt=wt if use_t else None)
            if not ok:
                pytest.skip("this shape's kernel has no head-dot epilogue")
            outs.append((dx, dpart))
        else:
            e = {"plain": ops.EPI_BF16, "mul": ops.EPI_MUL, "dgelu": ops.EPI_DGELU}[epi]
            kw = {} if epi == "plain" else dict(R=r, ldr=k_in)
            if use_t:
                ops.gemm(ops.NT, e, dy, wt, M, k_in, n_out, dx, n_out, n_out, k_in, **kw)
            else:
                ops.gemm(ops.NN, e, dy, w, M, k_in, n_out, dx, n_out, k_in, k_in, **kw)
            outs.append((dx,))
    torch.cuda.synchronize()
    for a, b in zip(outs[0], outs[1]):
        assert torch.isfinite(a.float()).all() and torch.equal(a, b)
    ref = dy.float() @ w.float()
    if epi == "plain" or epi == "dot":
        torch.testing.assert_close(outs[1][0].float(), ref, atol=2e-2 * float(ref.abs().max()), rtol=2e-2)


@pytest.mark.parametrize("case", ["fc1_fwd_gelu", "fc1_fwd_gelu_g", "fc2_dgrad_mul_nn", "fc2_dgrad_dgelu_nt", "gate_res", "plain"])
def test_column_split_is_bit_identical_to_one_launch(dev, case):
    """Round 6 (a switch, off by default: force_tile 259 / REED_GEMM_COLSPLIT=1 — measured equal in the b = 32 step): where a leading
    block of tile columns fills whole rounds of the chip exactly (8192 tokens x 4608 columns = 2.25 rounds of 256^2 tiles: 16 of the
    18 tile columns are two rounds) reed_gemm sends that block to the four-wave 256^2 kernel and the remaining columns out as a
    second launch on offset pointers (csrc/gemm.hip).  Every epilogue operand with a column index moves
    with it — Q, C, C2, R, bias, gate — and every element is formed by the same products in the same order: the same bits as the
    single launch on 256x144 tiles (force_tile 144) and on 256^2 tiles (257)."""
    from reed_amd import ops
    if ops.wgrad_slots() != 512:
        pytest.skip("the shape is chosen for 256 CUs")
    M, D, Hm = 8192, 1152, 4608
    g = torch.Generator().manual_seed(len(case))
    K = D
    x = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(Hm, K, generator=g) * 0.05).to(dev)          # NT operand [N, K]
    bias = _bf(torch.randn(Hm, generator=g)).to(dev)
    r16 = _bf(torch.randn(M, Hm, generator=g)).to(dev)
    gate = _bf(torch.randn(M // 256, Hm, generator=g)).to(dev)
    xin = torch.randn(M, Hm, generator=g).to(dev)
    outs = []
    for tile in (259, 144, 257):
        ops.gemm_force_tile(tile)
        try:
            c = torch.full((M, Hm), float("nan"), dtype=torch.bfloat16, device=dev)
            c2 = torch.full((M, Hm), float("nan"), dtype=torch.bfloat16, device=dev)
            if case in ("fc1_fwd_gelu", "fc1_fwd_gelu_g"):
                e = ops.EPI_GELU if case == "fc1_fwd_gelu" else ops.EPI_GELU_G
                ops.gemm(ops.NT, e, x, w, M, Hm, K, c, K, K, Hm, C2=c2, ldc2=Hm, bias=bias)
                outs.append((c, c2))
            elif case == "fc2_dgrad_mul_nn":
                wn = w.t().contiguous()                                   # NN operand [K, N]
                ops.gemm(ops.NN, ops.EPI_MUL, x, wn, M, Hm, K, c, K, Hm, Hm, R=r16, ldr=Hm)
                outs.append((c,))
            elif case == "fc2_dgrad_dgelu_nt":
                ops.gemm(ops.NT, ops.EPI_DGELU, x, w, M, Hm, K, c, K, K, Hm, R=r16, ldr=Hm)
                outs.append((c,))
            elif case == "gate_res":
                xo = torch.full((M, Hm), float("nan"), device=dev)
                ops.gemm(ops.NT, ops.EPI_GATE_RES, x, w, M, Hm, K, xo, K, K, Hm, C2=c2, ldc2=Hm, R=xin, ldr=Hm, bias=bias, gate=gate,
                         ldgate=Hm, rows_per_gate=256)
                outs.append((xo, c2))
            else:
                ops.gemm(ops.NT, ops.EPI_BF16, x, w, M, Hm, K, c, K, K, Hm, bias=bias)
                outs.append((c,))
        finally:
            ops.gemm_force_tile(0)
    torch.cuda.synchronize()
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.isfinite(a.float()).all() and torch.equal(a, b)
    if case == "plain":
        ref = x.float() @ w.float().t() + bias.float()
        torch.testing.assert_close(outs[0][0].float(), ref, atol=2e-2 * float(ref.abs().max()), rtol=2e-2)


@pytest.mark.parametrize("epi", ["plain", "gelu", "gelu_g", "dgelu", "mul"])
@pytest.mark.parametrize("M,N,K", [(256, 288, 128), (300, 576, 192), (8, 288, 256), (1000, 1152, 1152), (8192, 4608, 1152),
                                   (512, 4608, 64 * 5)])
def test_tile288(dev, M, N, K, epi, force_tile):
    """gemm288.hip (round 6; NT, 256x288 tile, 4 x 2 waves of 64x144, two 68 KiB LDS stages, the product pipelined in groups of three
    column tiles): 2, 3, 5 and 18 K-tiles, ragged M, the 16-column strip of each wave column, every epilogue it builds — the bits of
    the 256x144 kernel (same products, same k order inside a lane, same epilogue arithmetic) and close to an fp32 reference."""
    from reed_amd import ops
    if force_tile != 0:
        pytest.skip("runs once, forcing the tiles itself")
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    b = _bf(torch.randn(N, generator=g)).to(dev)
    r = _bf(torch.randn(M, N, generator=g)).to(dev)

    def run(tile):
        ops.gemm_force_tile(tile)
        try:
            c = torch.full((M + 1, N), float("nan"), dtype=torch.bfloat16, device=dev)
            c2 = torch.full((M + 1, N), float("nan"), dtype=torch.bfloat16, device=dev)
            if epi == "plain":
                ops.gemm(ops.NT, ops.EPI_BF16, x, w, M, N, K, c, K, K, N, bias=b)
                return (c,)
            if epi in ("gelu", "gelu_g"):
                ops.gemm(ops.NT, ops.EPI_GELU if epi == "gelu" else ops.EPI_GELU_G, x, w, M, N, K, c, K, K, N, C2=c2, ldc2=N, bias=b)
                return (c, c2)
            ops.gemm(ops.NT, ops.EPI_DGELU if epi == "dgelu" else ops.EPI_MUL, x, w, M, N, K, c, K, K, N, R=r, ldr=N)
            return (c,)
        finally:
            ops.gemm_force_tile(0)

    o288, o144 = run(288), run(144)
    torch.cuda.synchronize()
    for a, bb in zip(o288, o144):
        assert torch.isnan(a[M]).all() and not torch.isnan(a[:M].float()).any()
        assert torch.equal(a[:M], bb[:M])
    if epi == "plain":
        ref = x.float() @ w.float().t() + b.float()
        err = (o288[0][:M].float() - ref).abs().max().item()
        assert err <= ref.abs().max().item() * 2 ** -7, err


def test_tile288_identity(dev, force_tile):
    """A = I against an asymmetric B on the 256x288 tile: exact fragment / piece maps, both wave columns, both tile columns, the
    fifth DMA piece (the last 32 rows of the B stage) and the chunk swizzle of the 64-byte rows."""
    from reed_amd import ops
    if force_tile != 0:
        pytest.skip("runs once")
    M, N, K = 256, 576, 256
    x = torch.eye(M, K, device=dev).to(torch.bfloat16)
    w = (torch.arange(N * K, device=dev).reshape(N, K) % 251).float().to(torch.bfloat16)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm_force_tile(288)
    try:
        ops.gemm(ops.NT, ops.EPI_BF16, x, w, M, N, K, out, K, K, N)
    finally:
        ops.gemm_force_tile(0)
    assert torch.equal(out.float(), w.float().t().contiguous())
