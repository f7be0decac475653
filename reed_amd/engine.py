"""Forward / backward of the SiT model as a sequence of HIP kernel launches (reference: image/models/sit.py:271-311
and its autograd). Python here is orchestration only: shapes, pointers into the arenas, launch order, the
activation tape. Numerics follow the reference under accelerate bf16 mixed precision (SURVEY.md §3.2 dtype map):
fp32 residual stream / LayerNorm / loss, bf16 GEMM operands with fp32 MFMA accumulation, bf16 linear outputs.

Per block the launch sequence is
  fwd : LN+modulate -> qkv GEMM(+bias) -> attention -> proj GEMM(+bias, gate, residual) -> LN+modulate
        -> fc1 GEMM(+bias, GELU) -> fc2 GEMM(+bias, gate, residual)                       (7 launches)
  bwd : gate-bwd, 2x{wgrad, dgrad} for the MLP, LN-bwd, gate-bwd, proj {wgrad, dgrad}, attention-bwd,
        qkv {wgrad, dgrad}, LN-bwd, modulation-grad reduce                                (14 launches)
and the modulation of ALL blocks is one GEMM forward (silu(c) @ W_ada_all^T) and two backward.
"""
import contextlib
import os
import types

import torch

from . import ops
from .ops import (NT, NN, TN, EPI_BF16, EPI_GELU, EPI_SILU, EPI_GATE_RES, EPI_DGELU, EPI_DSILU, EPI_F32, EPI_ADDF32_RB,
                  EPI_GELU_G, EPI_SILU_G, EPI_MUL)


class Engine:
    # Kernel forms picked by the token count per GPU (both measured on MI355X with SiT-XL/2; bench.py prints them under
    # "kernel_forms", train.py logs them): the activation backward as one multiply by a derivative the forward saved — above this
    # many tokens (b > 48), see save_act_grad below — and the block's weight gradients on a second stream — up to this many.
    # NOTE for comparisons across GPU counts: the two activation-backward forms differ by one bf16 rounding of the derivative, so
    # 1 GPU x 256 and 8 GPUs x 32 of the same global batch do not give the same gradient bits unless the form is pinned
    # (train.py --save-act-grad {auto,0,1}; bench.py --save-act-grad).
    # (Round 6) the blocks' input gradients as NT GEMMs on a transposed copy of the weights from this many tokens per GPU on
    # (None: never): + 0.9 GB, kept fresh by the fused optimiser on its side stream; 3.6-5 % per dgrad GEMM at b = 256
    DGRAD_NT_MIN_TOKENS = 12289
    SAVE_ACT_GRAD_MIN_TOKENS = 12288
    # (Round 6: 0 = never by default.  With the one-item-per-CU weight-gradient launch the second stream measures equal at b = 32 —
    # 957.1 / 965.8 images/s with it, 963.7 / 958.4 without (profiles/r6_b32_side_stream_choices.txt) — and on the timeline it only
    # parks the main queue's row kernels behind a launch that holds every CU (ln_mod_bwd2: 131 us per launch for 30 us of work).
    # REED_WGRAD_STREAM=1 / engine.wgrad_stream = True switch it on; rounds 2-5 used it up to 12288 tokens: + 3.7 % at b = 32 with
    # the kernels of that time.)
    WGRAD_STREAM_MAX_TOKENS = 0

    def __init__(self, model):
        self.m = model
        self.L = model._layout
        self.A = model._arena
        if self.A.master.device.type != "cuda":
            raise RuntimeError("reed_amd.SiT: parameters are on %s; move the model to an AMD GPU (model.cuda()). "
                               "The hot path has no CPU fallback." % self.A.master.device)
        self.D = model.hidden_size
        self.H = model.num_heads
        self.hd = self.D // self.H
        self.Hm = int(self.D * model.mlp_ratio)
        self.T = model.num_patches
        self.P = model.patch_size
        self.C = model.in_channels
        self.depth = model.depth
        self.NO = self.P * self.P * self.C
        self._dims_ok = set()    # precisions whose shape restrictions this model has passed (_check_dims)
        split = model.encoder_depth_text is not None and model.encoder_depth_text != model.encoder_depth
        self.split = split
        self.tap_depth = [model.encoder_depth if (zt == "i" or not split) else model.encoder_depth_text
                          for zt in model.z_types]
        self.reducer = None      # set by reed_amd.parallel.GradReducer
        self._dot_delta = {}     # (tokens, operand type, reducing, forced tile, CU reserve, comm forms) -> the dO GEMM has the head-dot epilogue
        self._ws = None
        self._named = None       # [(name, parameter)] and the first trainable parameter, cached for backward
        self._sentinel = None
        self._ws_side = None
        self._shadow_t = None
        self.dgrad_nt = None     # True / False / None = by DGRAD_NT_MIN_TOKENS: the blocks' dgrads as NT GEMMs on transposed weights
        self.split_ada_wgrad = None   # None: per-block adaLN weight gradients iff a reducer is attached (see backward)
        self._side = None        # second HIP stream: the blocks' weight-gradient GEMMs run beside the dgrad chain
        # True / False / None = by WGRAD_STREAM_MAX_TOKENS (above)
        self.wgrad_stream = {"0": False, "1": True}.get(os.environ.get("REED_WGRAD_STREAM", "auto"))
        self.wgrad_stream_max_tokens = self.WGRAD_STREAM_MAX_TOKENS
        self.grad_live = False   # True: param grads hold a previous micro-step -> accumulate
        self.table_rows = model.num_classes + (1 if model.class_dropout_prob > 0 else 0)
        self._hb = 2             # bytes per element of the current build's operand arrays (set per forward / backward)
        # Round 5: the forward epilogues of fc1 / the t-MLP / the projector layers save the activation's DERIVATIVE at the
        # pre-activation (in the array that used to hold the pre-activation: nothing but the backward's dGELU / dSiLU epilogue
        # read it), so that backward epilogue is one multiply per element.  False: the pre-activation and the recomputing
        # epilogues (the form the round-1..4 records were measured with; tests compare the two).
        # (same-box A/B at b = 256: dgrad fc2 0.729 -> 0.659 ms, fc1 forward 0.681 -> 0.708, step 199.1 -> 197.5 ms: profiles/r5_actgrad.txt)
        # None = by the token count: on above 12288 tokens (b > 48 per GPU), where the block's GEMMs run on the four-wave 256^2
        # kernels — one wave per SIMD, the dGELU epilogue's vector work exposed.  Below, the 256x144 kernel's two waves per SIMD
        # hide that work already and the derivative only costs the forward: b = 32, in-step, fc1 forward 123.8 -> 136.5 us per launch
        # against 118.1 -> 116.4 for the fc2 dgrad (profiles/r4_b32_timeline.txt, r5_b32_timeline_first.txt).
        self.save_act_grad = None
        # the attention backward's delta = rowsum(dO * O) from the epilogue of the GEMM that produces dO (False: the row kernel;
        # tests/test_model_gpu.py::test_bench_plan_matches_the_golden_pinned_plan_at_b256 compares the two)
        self.fused_delta = True
        self._err = None         # sticky device flag: a label outside the embedding table was seen (see check_errors)

    def _check_dims(self, prec):
        """Shape restrictions of the build that evaluates the model: the 16-bit MFMA tiles want every GEMM width in multiples
        of 128; the fp32-operand build (csrc/gemm_f32.hip) guards every bound and only needs 4-element alignment."""
        if prec in self._dims_ok:
            return
        q = 4 if prec == "fp32" else 128
        what = "4 (fp32 operands)" if prec == "fp32" else "128 for the MFMA GEMM tiles"
        for n, v in (("hidden_size", self.D), ("mlp hidden", self.Hm), ("projector_dim", self.m.projector_dim)):
            if v % q:
                raise ValueError(f"{n}={v} must be a multiple of {what}")
        for z in self.m.z_dims:
            if z % q:
                raise ValueError(f"z_dim={z} must be a multiple of {what}")
        self._dims_ok.add(prec)

    def check_errors(self):
        """Raise if any forward since the last check saw a class label outside the embedding table (the reference's
        nn.Embedding raises IndexError at that forward, sit.py:98; here the kernel substitutes row 0, sets a device flag
        and the error surfaces at the caller's next synchronisation point: the trainer's logging sync, the samplers'
        return, state_dict()). Costs one 4-byte D2H copy — call it where the host synchronises anyway."""
        if self._err is not None and int(self._err.item()) != 0:
            self._err.zero_()
            raise IndexError(f"reed_amd.SiT: a class label outside [0, {self.table_rows}) was passed to the model "
                             f"(num_classes={self.m.num_classes}, class_dropout_prob={self.m.class_dropout_prob}); "
                             "the affected forwards used row 0 instead")

    # ---- pointers into the arenas -------------------------------------------------
    def W(self, name):
        return self._shadow.data_ptr() + self._hb * self.L.off(name)

    def Wf(self, name):
        return self.A.master.data_ptr() + 4 * self.L.off(name)

    def G(self, name):
        return self.A.grad.data_ptr() + 4 * self.L.off(name)

    def ws(self, nfloats, dev):
        if self._ws is None or self._ws.numel() < nfloats:
            self._ws = torch.empty(int(nfloats), dtype=torch.float32, device=dev)
        return self._ws

    def ws_side(self, nfloats, dev):
        """Split-K slab workspace of the weight-gradient stream (allocated while that stream is current)."""
        if self._ws_side is None or self._ws_side.numel() < nfloats:
            self._ws_side = torch.empty(int(nfloats), dtype=torch.float32, device=dev)
        return self._ws_side

    # ---- forward -------------------------------------------------------------------
    def forward(self, x, t, y, inference, need_grad, drop):
        prec = getattr(self.m, "precision", "bf16")
        prev = ops.use(prec)
        try:
            return self._forward(x, t, y, inference, need_grad, drop, prec)
        finally:
            ops.use(prev)

    def _forward(self, x, t, y, inference, need_grad, drop, prec):
        m, L = self.m, self.L
        ops.require_cuda(x, "x")
        dev = x.device
        B, C, HW = x.shape[0], x.shape[1], x.shape[-1]
        D, H, hd, Hm, T, P = self.D, self.H, self.hd, self.Hm, self.T, self.P
        if C != self.C or HW != m.input_size or x.shape[-2] != HW:
            raise ValueError(f"input {tuple(x.shape)} does not match (N,{self.C},{m.input_size},{m.input_size})")
        M = B * T
        self._check_dims(prec)
        self._shadow = self.A.ensure_shadow(prec)
        hdt = ops.half_dtype(prec)
        hb = self._hb = hdt.itemsize    # bytes per element of the operand / activation arrays (2; 4 in the fp32-operand build)
        pend = self.A.pending   # parameter buckets an overlapped optimiser step is still rewriting (optim.py)
        if "all" in pend:
            self.A.wait_all()
        elif pend:
            self.A.wait("embed")
        x = x.contiguous().float()
        t = t.contiguous().float()
        y = y.contiguous().long()

        def bf(*s):
            return torch.empty(s, dtype=hdt, device=dev)

        def f32(*s):
            return torch.empty(s, dtype=torch.float32, device=dev)

        sag = self.save_act_grad if self.save_act_grad is not None else M > self.SAVE_ACT_GRAD_MIN_TOKENS
        self._sag = sag   # (the projector forward of this call follows it)
        tp = types.SimpleNamespace(B=B, blocks=[], proj={}, x=x, prec=prec, act_grad=sag) if need_grad else None
        # epilogues of a layer whose saved array feeds the backward (gemm.h): derivative-saving forms when there is a backward
        epi_silu = EPI_SILU_G if need_grad and sag else EPI_SILU
        epi_gelu = EPI_GELU_G if need_grad and sag else EPI_GELU
        # -- embedders
        tok = f32(M, D)
        ops.patch_embed_fwd(x, self.W("x_embedder.proj.weight"), self.W("x_embedder.proj.bias"), self.Wf("pos_embed"),
                            tok, B, C, HW, P, D)
        sin = bf(B, 256)
        ops.timestep_sinusoid(t, sin, B)
        t1p = bf(B, D) if need_grad else None
        t1 = bf(B, D)
        ops.gemm(NT, epi_silu, sin, self.W("t_embedder.mlp.0.weight"), B, D, 256, t1p, 256, 256, D, C2=t1, ldc2=D,
                 bias=self.W("t_embedder.mlp.0.bias"))
        temb = bf(B, D)
        ops.gemm(NT, EPI_BF16, t1, self.W("t_embedder.mlp.2.weight"), B, D, D, temb, D, D, D,
                 bias=self.W("t_embedder.mlp.2.bias"))
        drop_u8 = None
        if drop is not None:
            drop_u8 = drop.to(device=dev, dtype=torch.uint8).contiguous()
        labels_eff = torch.empty(B, dtype=torch.int64, device=dev)
        c = f32(B, D)
        silu_c = bf(B, D)
        if self._err is None:
            self._err = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.label_cond(y, drop_u8, m.num_classes, self.Wf("y_embedder.embedding_table.weight"), temb, labels_eff, c,
                       silu_c, B, D, table_rows=self.table_rows, err=self._err)
        Nall = L.ada_rows
        mod = bf(B, Nall)
        sp = self._shadow.data_ptr()
        mp = mod.data_ptr()
        # modulation of every block: one GEMM on silu(c) — or two when an overlapped optimiser step is still rewriting the
        # tail of the adaLN matrix (a third of all parameters): the head blocks' rows now, the rest when block kh starts
        kh = min(L.ADA_HEAD_BLOCKS, self.depth) if "ada_tail" in pend else 0
        n_head = kh * 6 * D if kh else Nall
        if pend:
            self.A.wait("ada_head")
        ops.gemm(NT, EPI_BF16, silu_c, sp + hb * L.ada_w_off, B, n_head, D, mod, D, D, Nall, bias=sp + hb * L.ada_b_off)
        if need_grad:
            tp.sin, tp.t1p, tp.t1, tp.labels_eff, tp.c, tp.silu_c, tp.mod = sin, t1p, t1, labels_eff, c, silu_c, mod

        # -- blocks
        if not need_grad:  # inference: ping-pong buffers, nothing saved
            xa, xb = tok, f32(M, D)
            h, qkv, o, u = bf(M, D), bf(M, 3 * D), bf(M, D), bf(M, Hm)
        xcur = tok
        zs_by_proj = {}
        for i in range(self.depth):
            b = f"blocks.{i}."
            mb = mp + hb * (i * 6 * D)
            if kh and i == kh:
                self.A.wait("ada_tail")
                ops.gemm(NT, EPI_BF16, silu_c, sp + hb * (L.ada_w_off + n_head * D), B, Nall - n_head, D, mp + hb * n_head,
                         D, D, Nall, bias=sp + hb * (L.ada_b_off + n_head))
                kh = 0
            if pend:
                self.A.wait(f"block{i}")
            if need_grad:
                h, qkv, o, u = bf(M, D), bf(M, 3 * D), bf(M, D), bf(M, Hm)
                h2, a1, y1, y2 = bf(M, D), bf(M, Hm), bf(M, D), bf(M, D)
                mean1, rstd1, mean2, rstd2 = f32(M), f32(M), f32(M), f32(M)
                lse = f32(B, H, T)
                xmid, xout = f32(M, D), f32(M, D)
            else:
                h2, a1, y1, y2 = h, None, None, None
                mean1 = rstd1 = mean2 = rstd2 = lse = None
                xmid = xb if xcur is xa else xa
                xout = xcur  # safe: fc2's epilogue reads R=xmid, writes xout; xcur is dead after proj
            ops.ln_modulate_fwd(xcur, mb, mb + hb * D, Nall, h, mean1, rstd1, M, D, T)
            ops.gemm(NT, EPI_BF16, h, self.W(b + "attn.qkv.weight"), M, 3 * D, D, qkv, D, D, 3 * D,
                     bias=self.W(b + "attn.qkv.bias"))
            qkv_a, qstats = qkv, None
            if m.qk_norm:
                qkv_a = bf(M, 3 * D)
                qstats = f32(M, 2, H, 2) if need_grad else None
                ops.qk_norm_fwd(qkv, self.Wf(b + "attn.q_norm.weight"), self.Wf(b + "attn.q_norm.bias"),
                                self.Wf(b + "attn.k_norm.weight"), self.Wf(b + "attn.k_norm.bias"), qkv_a, qstats, M, H, hd)
            ops.attention_fwd(qkv_a, o, lse, B, T, H, hd)
            ops.gemm(NT, EPI_GATE_RES, o, self.W(b + "attn.proj.weight"), M, D, D, xmid, D, D, D, C2=y1, ldc2=D,
                     R=xcur, ldr=D, bias=self.W(b + "attn.proj.bias"), gate=mb + 2 * hb * D, ldgate=Nall, rows_per_gate=T)
            ops.ln_modulate_fwd(xmid, mb + 3 * hb * D, mb + 4 * hb * D, Nall, h2, mean2, rstd2, M, D, T)
            ops.gemm(NT, epi_gelu, h2, self.W(b + "mlp.fc1.weight"), M, Hm, D, a1, D, D, Hm, C2=u, ldc2=Hm,
                     bias=self.W(b + "mlp.fc1.bias"))
            ops.gemm(NT, EPI_GATE_RES, u, self.W(b + "mlp.fc2.weight"), M, D, Hm, xout, Hm, Hm, D, C2=y2, ldc2=D,
                     R=xmid, ldr=D, bias=self.W(b + "mlp.fc2.bias"), gate=mb + 5 * hb * D, ldgate=Nall, rows_per_gate=T)
            if need_grad:
                tp.blocks.append(types.SimpleNamespace(x=xcur, mean1=mean1, rstd1=rstd1, h=h, qkv=qkv, qkv_a=qkv_a,
                                                       qstats=qstats, o=o, lse=lse,
                                                       y1=y1, xmid=xmid, mean2=mean2, rstd2=rstd2, h2=h2, a1=a1, u=u,
                                                       y2=y2))
            xcur = xout
            if not inference:
                for j, dj in enumerate(self.tap_depth):
                    if dj == i + 1:
                        zs_by_proj[j] = self._projector_fwd(j, xcur, B, need_grad, tp)
        # -- final layer
        if kh:   # depth <= ADA_HEAD_BLOCKS: the tail (final layer's rows) was never launched
            self.A.wait("ada_tail")
            ops.gemm(NT, EPI_BF16, silu_c, sp + hb * (L.ada_w_off + n_head * D), B, Nall - n_head, D, mp + hb * n_head,
                     D, D, Nall, bias=sp + hb * (L.ada_b_off + n_head))
        if pend:
            self.A.wait_all()
        out = f32(B, C, HW, HW)
        meanF = f32(M) if need_grad else None
        rstdF = f32(M) if need_grad else None
        mf = mp + hb * (self.depth * 6 * D)
        ops.final_layer_fwd(xcur, mf, mf + hb * D, Nall, self.W("final_layer.linear.weight"),
                            self.W("final_layer.linear.bias"), out, meanF, rstdF, B, T, D, C, P)
        if need_grad:
            tp.x_last, tp.meanF, tp.rstdF = xcur, meanF, rstdF
        zs = None
        if not inference:
            if self.split:
                ji, jt = m.z_types.index("i"), m.z_types.index("t")
                zs = [zs_by_proj[ji], zs_by_proj[jt]]
                if need_grad:
                    tp.zs_order = [ji, jt]
            else:
                zs = [zs_by_proj[j] for j in range(len(m.z_dims))]
                if need_grad:
                    tp.zs_order = list(range(len(m.z_dims)))
        return out, zs, tp

    def _projector_fwd(self, j, x, B, need_grad, tp):
        m, D, T = self.m, self.D, self.T
        Pd, Z = m.projector_dim, m.z_dims[j]
        dev = x.device
        img = m.z_types[j] == "i"
        self.A.wait("projectors")
        R = B * T if img else B
        hdt = ops.half_dtype()   # the build Engine.forward selected
        bf = lambda *s: torch.empty(s, dtype=hdt, device=dev)  # noqa: E731
        xin = bf(R, D)
        if img:
            ops.ln_modulate_fwd(x, None, None, 0, xin, None, None, R, D, T)  # f32 -> bf16 cast (autocast input cast)
        else:
            ops.token_mean_fwd(x, xin, B, T, D)
        pre = f"projectors.{j}."
        epi_silu = EPI_SILU_G if need_grad and self._sag else EPI_SILU
        p1p, p1 = (bf(R, Pd) if need_grad else None), bf(R, Pd)
        ops.gemm(NT, epi_silu, xin, self.W(pre + "0.weight"), R, Pd, D, p1p, D, D, Pd, C2=p1, ldc2=Pd,
                 bias=self.W(pre + "0.bias"))
        p2p, p2 = (bf(R, Pd) if need_grad else None), bf(R, Pd)
        ops.gemm(NT, epi_silu, p1, self.W(pre + "2.weight"), R, Pd, Pd, p2p, Pd, Pd, Pd, C2=p2, ldc2=Pd,
                 bias=self.W(pre + "2.bias"))
        zt = bf(R, Z)
        ops.gemm(NT, EPI_BF16, p2, self.W(pre + "4.weight"), R, Z, Pd, zt, Pd, Pd, Z, bias=self.W(pre + "4.bias"))
        if need_grad:
            tp.proj[j] = types.SimpleNamespace(xin=xin, p1p=p1p, p1=p1, p2p=p2p, p2=p2, R=R)
        return zt.view(B, T, Z) if img else zt

    # ---- backward ------------------------------------------------------------------
    def _wgrad_block(self, probs, Mtok, acc, dev, side):
        """The four weight gradients of a block as one launch without split-K (ops.wgrad_group; csrc/gemm_tn.hip): probs =
        [(dy, x, weight name, n_out, k_in), ...].  Falls back to the per-GEMM path when the device is too small."""
        args = [(dy, x, self.G(w), self.G(w.replace("weight", "bias")), N, K) for dy, x, w, N, K in probs]
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            done = ops.wgrad_group(args, Mtok, accumulate=acc)
        if not done:
            for dy, x, w, N, K in probs:
                self._wgrad(dy, x, w, Mtok, N, K, acc, dev, side=side)
            return
        if side is not None:
            for dy, x, _, _, _ in probs:
                dy.record_stream(side)
                x.record_stream(side)

    def _wgrad(self, dy, x, wname, Mtok, N, K, acc, dev, bias_done=False, side=None):
        """dW (+)= dy^T x and db (+)= colsum(dy) into the gradient arena: the TN kernel / tile ops.plan_wgrad picks
        (256x128, 128x256 or 128x128) with wave-quantised split-K through slabs (deterministic reduce); the bias gradient rides along as an extra
        ones-MFMA in the blocks of the first column tile unless the caller already has it (bias_done).
        side = the weight-gradient stream: the launch is ordered after everything queued on the current stream so far
        (dy is the newest tensor it reads) and runs beside the dgrad chain, which nothing downstream of it in
        backward depends on — it fills the CUs the chain's ragged last tile rounds leave idle (at b = 32/GPU a
        256-row-tile dgrad grid covers 56-84 % of the 256 CUs)."""
        bname = wname.replace("weight", "bias")
        lay, split = ops.plan_wgrad(Mtok, N, K)
        if side is None:
            ws = self.ws(split * (N * K + N), dev) if split > 1 else None
            ops.linear_wgrad(dy, x, self.G(wname), dbias=None if bias_done else self.G(bname), accumulate=acc,
                             split_k=split, Mtok=Mtok, N=N, K=K, ws=ws, lay=lay)
            return
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ws = self.ws_side(split * (N * K + N), dev) if split > 1 else None
            ops.linear_wgrad(dy, x, self.G(wname), dbias=None if bias_done else self.G(bname), accumulate=acc,
                             split_k=split, Mtok=Mtok, N=N, K=K, ws=ws, lay=lay)
        dy.record_stream(side)   # allocated on the main stream: the caching allocator must not hand these
        x.record_stream(side)    # blocks out again before the side stream's reads have finished

    def WT(self, name):
        """Device address of the transposed 16-bit copy W^T [k_in, n_out] of a block linear's weight (arena.py), or None."""
        if self._shadow_t is None:
            return None
        seg = self.A.t_seg.get(name)
        return None if seg is None else self._shadow_t.data_ptr() + self._hb * seg[0]

    def _dgrad(self, epi, dy, wname, Mtok, N, K, out, **kw):
        """dx[Mtok,K] = dy[Mtok,N] W[N,K].  With the transposed copy of the weight (round 6: the blocks' linears above
        DGRAD_NT_MIN_TOKENS) an NT GEMM on W^T — both operands k-contiguous; otherwise NN straight on the weight shadow (W is the
        k-strided operand, read with transposing LDS reads).  The same products in the same order: the same bits."""
        wt = self.WT(wname)
        if wt is not None:
            ops.gemm(NT, epi, dy, wt, Mtok, K, N, out, N, N, K, **kw)
        else:
            ops.gemm(NN, epi, dy, self.W(wname), Mtok, K, N, out, N, K, K, **kw)

    def backward(self, tp, dout, dzs):
        prec = getattr(self.m, "precision", "bf16")
        if getattr(tp, "prec", prec) != prec:
            raise RuntimeError("reed_amd.SiT: model.precision changed between forward and backward")
        prev = ops.use(prec)
        # gradient buckets are reduced beside this backward's GEMMs (RCCL channels hold CUs): only then does the library keep to
        # kernels that degrade gracefully without every CU (csrc/gemm256.hip:reed_set_concurrent_comm); the forward, the
        # optimiser and the sampler run with no collective in flight (the step waits for the last bucket before the update)
        comm = self.reducer is not None and self.reducer.active()
        if comm:
            ops.set_concurrent_comm(True)
        try:
            return self._backward(tp, dout, dzs, ops.half_dtype(prec))
        finally:
            if comm:
                ops.set_concurrent_comm(False)
            ops.use(prev)

    def _backward(self, tp, dout, dzs, hdt):
        m, L = self.m, self.L
        D, H, hd, Hm, T, P, C = self.D, self.H, self.hd, self.Hm, self.T, self.P, self.C
        B = tp.B
        M = B * T
        dev = dout.device
        Nall = L.ada_rows
        hb = self._hb = hdt.itemsize
        self.A.ensure_grad()
        # the transposed copies of the blocks' weights (the dgrads' NT operand): built on first use, then kept fresh by the optimiser
        use_t = self.dgrad_nt if self.dgrad_nt is not None else (self.DGRAD_NT_MIN_TOKENS is not None and M >= self.DGRAD_NT_MIN_TOKENS)
        self._shadow_t = self.A.ensure_shadow_t(tp.prec) if (use_t and hb == 2) else None
        acc = self.grad_live
        mp = tp.mod.data_ptr()
        ch = T // 16  # 16-row chunks per sample

        def bf(*s):
            return torch.empty(s, dtype=hdt, device=dev)

        def f32(*s):
            return torch.empty(s, dtype=torch.float32, device=dev)

        group_wgrad = ops.wgrad_group_fits([(D, Hm), (Hm, D), (D, D), (3 * D, D)])
        nws = ops.attention_bwd_ws_floats(B, T, H)       # delta = rowsum(dO * O) of the persistent attention backward
        attn_ws = f32(nws) if nws else None
        # the answer of the library ("this shape's dO GEMM has the head-dot epilogue") depends on which kernel the GEMM runs on:
        # the token count, the operand type, whether gradient buckets are being reduced beside this backward (the library then
        # keeps off the persistent kernels) and a forced tile — all in the key (a 1002 seen once under one of
        # them was remembered for the rest of the run under the token count alone).
        # (and the CU reserve and the forms-beside-collectives switch, which the tuner cycles under one token count)
        dkey = (M, str(hdt), self.reducer is not None and self.reducer.active(), ops.gemm_forced_tile(), ops.cu_reserve(),
                ops.comm_forms())
        dot_delta = self._dot_delta if self.fused_delta and hdt != torch.float32 else {dkey: False}
        side = None
        if self.wgrad_stream or (self.wgrad_stream is None and M <= self.wgrad_stream_max_tokens):
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
            side = self._side
        dout = dout.contiguous().float()
        # -- final layer
        mf = mp + hb * (self.depth * 6 * D)
        hbuf, dlin, dh = bf(M, D), bf(M, self.NO), bf(M, D)
        ops.final_layer_bwd_rows(dout, tp.x_last, tp.meanF, tp.rstdF, mf, mf + hb * D, Nall,
                                 self.W("final_layer.linear.weight"), hbuf, dlin, dh, B, T, D, C, P)
        dx = torch.zeros(M, D, dtype=torch.float32, device=dev)
        partF = f32(M // 16, 2, D)
        # which projectors fire after block i's output?
        pending = {j: d for j, d in enumerate(self.tap_depth)}

        def tap_fires(depth_idx):   # a projector backward adds into dx at this depth (between block depth_idx-1 and depth_idx)
            return any(dj == depth_idx and j in tp.proj for j, dj in pending.items())

        def ln_bwd(dh_, x_, mean_, rstd_, scale_ptr, part_, nxt):
            """LayerNorm+modulate backward into dx. nxt = index of the block whose MLP-branch gate consumes the finished
            dx next (or None): its gate backward rides along in the same pass unless a projector tap adds into dx in
            between. Returns (dy2, pg2) of that block when fused."""
            if nxt is None or nxt < 0 or tap_fires(nxt + 1):
                ops.ln_modulate_bwd(dh_, x_, mean_, rstd_, scale_ptr, Nall, dx, part_, M, D, T)
                return None
            nb_ = tp.blocks[nxt]
            dy_, pg_ = bf(M, D), f32(M // 16, D)
            ops.ln_modulate_bwd_gate(dh_, x_, mean_, rstd_, scale_ptr, Nall, dx, part_, nb_.y2,
                                     mp + hb * (nxt * 6 * D + 5 * D), Nall, dy_, pg_, None, M, D, T)
            return dy_, pg_

        pre = ln_bwd(dh, tp.x_last, tp.meanF, tp.rstdF, mf + hb * D, partF, self.depth - 1)
        wsf = self.ws(ops.smallk_ws_floats(D, max(self.NO, C * P * P)), dev)
        ops.smallk_wgrad(hbuf, False, dlin, wsf, self.G("final_layer.linear.weight"), None,
                         self.G("final_layer.linear.bias"), M, D, self.NO, 1, acc)
        dmod = bf(B, Nall)
        offF = self.depth * 6 * D
        ops.reduce_mod_parts([(partF.data_ptr(), 2 * D, offF), (partF.data_ptr() + 4 * D, 2 * D, offF + D)], dmod, Nall,
                             B, D, ch)
        del hbuf, dlin, dh
        # With a reducer attached the adaLN weight gradient is computed block by block (rows of block i right after
        # block i's backward: dW_ada[i] = dmod[:, i]^T silu(c), a K = b GEMM) instead of one GEMM at the end, so each
        # slice's all-reduce overlaps the rest of backward; the arithmetic per output element is the same.
        split_ada = self.reducer is not None if self.split_ada_wgrad is None else self.split_ada_wgrad
        gp = self.A.grad.data_ptr()
        dmp = dmod.data_ptr()
        # Factor path (a reducer that is reducing THIS backward, no gradient accumulation in flight): the adaLN weight
        # gradient contracts over the local batch only, so instead of all-reducing the [Nall, D] fp32 matrix (a third of
        # the payload) the ranks all-gather the two bf16 factors — dmod rows of block i (pre-scaled by 1 / world: exact,
        # power-of-two worlds only) right after block i's backward, silu(c) once — and every rank forms the global-batch
        # product itself at the end of backward (K = world * b). At world = 1 this is the per-block GEMM on a copy.
        red = self.reducer
        fg = (red is not None and red.ada_gather and red.active() and not acc and self.split_ada_wgrad is None)
        if fg:
            split_ada = False
            W = red.world
            g_send, g_recv = red.gather_buffers("dmod", B * Nall, hdt, dev)
            s_send, s_all = red.gather_buffers("silu", B * D, hdt, dev)
            s_send.view(B, D).copy_(tp.silu_c)
            red.gather(s_send, s_all)

        def ada_rows(i):
            return i * 6 * D, (6 * D if i < self.depth else 2 * D)

        def ada_wgrad(i):
            c0, rows = ada_rows(i)
            ops.gemm(TN, EPI_F32, dmp + hb * c0, tp.silu_c, rows, D, B, gp + 4 * (L.ada_w_off + c0 * D), Nall, D, D,
                     dbias=gp + 4 * (L.ada_b_off + c0), accumulate=acc)

        def ada_gather(i):   # block i's dmod rows are final: scale, pack [b, rows] contiguously, all-gather
            c0, rows = ada_rows(i)
            snd = g_send[B * c0:B * (c0 + rows)]
            torch.mul(dmod[:, c0:c0 + rows], 1.0 / W, out=snd.view(B, rows))
            red.gather(snd, g_recv[W * B * c0:W * B * (c0 + rows)])

        def ada_after(i):
            if fg:
                ada_gather(i)
            elif split_ada:
                ada_wgrad(i)

        ada_after(self.depth)
        if self.reducer is not None:
            self.reducer.ready("final")
            if split_ada:
                self.reducer.ready(f"ada{self.depth}")
        proj_left = len([j for j in pending if j in tp.proj])
        dz_by_proj = {}
        if dzs is not None:
            for k, j in enumerate(tp.zs_order):
                dz_by_proj[j] = dzs[k]
        for i in reversed(range(self.depth)):
            for j, dj in pending.items():
                if dj == i + 1 and j in tp.proj:
                    self._projector_bwd(j, tp, dz_by_proj.get(j), dx, B, acc, dev)
                    proj_left -= 1
                    if proj_left == 0 and self.reducer is not None:
                        self.reducer.ready("projectors")
            bk = tp.blocks[i]
            b = f"blocks.{i}."
            mb = mp + hb * (i * 6 * D)
            # MLP branch (its gate backward usually came with the previous LayerNorm backward)
            # (bias gradients of fc2 / proj = column sums of dy2 / dy1: fused into their weight-gradient GEMMs)
            if pre is not None:
                dy2, pg2 = pre
            else:
                pg2, dy2 = f32(M // 16, D), bf(M, D)
                ops.gate_bwd(dx, bk.y2, mb + 5 * hb * D, Nall, dy2, pg2, M, D, T)
            wg = [] if group_wgrad else None   # the block's four weight gradients: one launch at the end of the block
            if wg is not None:
                wg.append((dy2, bk.u, b + "mlp.fc2.weight", D, Hm))
            else:
                self._wgrad(dy2, bk.u, b + "mlp.fc2.weight", M, D, Hm, acc, dev, side=side)
            da1 = bf(M, Hm)
            self._dgrad(EPI_MUL if tp.act_grad else EPI_DGELU, dy2, b + "mlp.fc2.weight", M, D, Hm, da1, R=bk.a1, ldr=Hm)
            if wg is not None:
                wg.append((da1, bk.h2, b + "mlp.fc1.weight", Hm, D))
            else:
                self._wgrad(da1, bk.h2, b + "mlp.fc1.weight", M, Hm, D, acc, dev, side=side)
            # reuse dy2 unless its weight gradient (side stream / grouped launch) may still have to read it
            dh2 = dy2 if side is None and wg is None else bf(M, D)
            self._dgrad(EPI_BF16, da1, b + "mlp.fc1.weight", M, Hm, D, dh2)
            # LN2 backward + the attention branch's gate backward in one pass over dx
            pl2 = f32(M // 16, 2, D)
            pg1, dy1 = f32(M // 16, D), bf(M, D)
            ops.ln_modulate_bwd_gate(dh2, bk.xmid, bk.mean2, bk.rstd2, mb + 4 * hb * D, Nall, dx, pl2, bk.y1, mb + 2 * hb * D, Nall,
                                     dy1, pg1, None, M, D, T)
            if wg is not None:
                wg.append((dy1, bk.o, b + "attn.proj.weight", D, D))
            else:
                self._wgrad(dy1, bk.o, b + "attn.proj.weight", M, D, D, acc, dev, side=side)
            do = bf(M, D)
            dqkv = bf(M, 3 * D)
            # delta = rowsum(dO * O) of the attention backward where dO is produced: the GEMM's epilogue 13 leaves per-row,
            # per-head partial dot products with O (12 MB instead of a 302 MB row pass over dO and O at b = 256); where this
            # shape's GEMM kernel has no such epilogue (False, decided once per token count) the plain store + the row kernel
            fused = False
            if dot_delta.get(dkey, True) and attn_ws is not None and T <= 256:
                dpart = f32(M * H * (1 if hd == 64 else 2))
                fused = dot_delta[dkey] = ops.dgrad_with_head_dots(dy1, self.W(b + "attn.proj.weight"), do, bk.o, dpart, M, D, D, hd,
                                                                   wt=self.WT(b + "attn.proj.weight"))
            if fused:
                ops.attention_bwd_dp(bk.qkv_a, do, bk.lse, dpart, dqkv, attn_ws, B, T, H, hd)
            else:
                self._dgrad(EPI_BF16, dy1, b + "attn.proj.weight", M, D, D, do)
                ops.attention_bwd(bk.qkv_a, bk.o, do, bk.lse, dqkv, B, T, H, hd, ws=attn_ws)
            if m.qk_norm:  # back through the per-head LayerNorm of q and k; parameter grads via per-block partials
                nb = (M * 3 * H + 255) // 256
                part = f32(nb, 4 * hd)
                dpre = bf(M, 3 * D)
                ops.qk_norm_bwd(dqkv, bk.qkv, bk.qstats, self.Wf(b + "attn.q_norm.weight"),
                                self.Wf(b + "attn.k_norm.weight"), dpre, part, M, H, hd)
                ops.rowsum_f32(part, nb, self.G(b + "attn.q_norm.weight"), 4 * hd, acc,
                               ws=self.ws((nb + 63) // 64 * 4 * hd, dev))
                dqkv = dpre
            if wg is not None:
                wg.append((dqkv, bk.h, b + "attn.qkv.weight", 3 * D, D))
                self._wgrad_block(wg, M, acc, dev, side)
            else:
                self._wgrad(dqkv, bk.h, b + "attn.qkv.weight", M, 3 * D, D, acc, dev, side=side)
            dh1 = do  # reuse
            self._dgrad(EPI_BF16, dqkv, b + "attn.qkv.weight", M, 3 * D, D, dh1)
            pl1 = f32(M // 16, 2, D)
            pre = ln_bwd(dh1, bk.x, bk.mean1, bk.rstd1, mb + hb * D, pl1, i - 1)
            o6 = i * 6 * D
            p1, p2 = pl1.data_ptr(), pl2.data_ptr()
            ops.reduce_mod_parts([(p1, 2 * D, o6), (p1 + 4 * D, 2 * D, o6 + D), (pg1.data_ptr(), D, o6 + 2 * D),
                                  (p2, 2 * D, o6 + 3 * D), (p2 + 4 * D, 2 * D, o6 + 4 * D),
                                  (pg2.data_ptr(), D, o6 + 5 * D)], dmod, Nall, B, D, ch)
            tp.blocks[i] = None  # free this block's activations
            ada_after(i)
            if self.reducer is not None:
                if side is None:
                    self.reducer.ready(f"block{i}")
                    if split_ada:
                        self.reducer.ready(f"ada{i}")
                else:  # the bucket holds gradients written on both streams: fire it from the side stream, after main
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        self.reducer.ready(f"block{i}")
                        if split_ada:
                            self.reducer.ready(f"ada{i}")
        if proj_left > 0 and m.z_dims and self.reducer is not None:   # projector outputs unused by the loss
            self.reducer.ready("projectors")
        # -- adaLN (all blocks + final): dW = dmod^T silu(c); d silu(c) = dmod @ W   (one GEMM each)
        sp = self._shadow.data_ptr()
        if fg:
            red.gather_sync()
            for i in reversed(range(self.depth + 1)):   # rank-major [world * b, rows] factors: K = the global batch
                c0, rows = ada_rows(i)
                ops.gemm(TN, EPI_F32, g_recv.data_ptr() + hb * W * B * c0, s_all, rows, D, W * B,
                         gp + 4 * (L.ada_w_off + c0 * D), rows, D, D, dbias=gp + 4 * (L.ada_b_off + c0), accumulate=False)
        elif not split_ada:
            ops.gemm(TN, EPI_F32, dmod, tp.silu_c, Nall, D, B, gp + 4 * L.ada_w_off, Nall, D, D,
                     dbias=gp + 4 * L.ada_b_off, accumulate=acc)
        ksteps = Nall // 64
        split = min(64, max(1, ksteps // 8))
        slab = B * D
        wsd = self.ws(split * slab, dev)
        ops.gemm(NN, EPI_F32, dmod, sp + hb * L.ada_w_off, B, D, Nall, wsd, Nall, D, D, split_k=split, slab_stride=slab)
        per = (ksteps + split - 1) // split
        eff = (ksteps + per - 1) // per
        dsilu = f32(B, D)
        ops.reduce_slabs(wsd, slab, eff, dsilu, slab, False)
        # -- conditioning: label table + timestep MLP
        dtemb = bf(B, D)
        gt = self.A.view(self.A.grad, "y_embedder.embedding_table.weight")
        if not acc:
            gt.zero_()
        ops.label_cond_bwd(dsilu, tp.c, tp.labels_eff, dtemb, gt, B, D)
        ops.linear_wgrad(dtemb, tp.t1, self.G("t_embedder.mlp.2.weight"), dbias=self.G("t_embedder.mlp.2.bias"),
                         accumulate=acc, Mtok=B, N=D, K=D)
        dt1 = bf(B, D)
        ops.gemm(NN, EPI_MUL if tp.act_grad else EPI_DSILU, dtemb, self.W("t_embedder.mlp.2.weight"), B, D, D, dt1, D, D, D, R=tp.t1p, ldr=D)
        ops.linear_wgrad(dt1, tp.sin, self.G("t_embedder.mlp.0.weight"), dbias=self.G("t_embedder.mlp.0.bias"),
                         accumulate=acc, Mtok=B, N=D, K=256)
        # -- patch embed: dW[d,k] = sum_tokens bf16(dx)[token,d] * patch[token,k]
        K = C * P * P
        xb = bf(M, K)
        ops.patchify_bf16(tp.x, xb, B, C, tp.x.shape[-1], P, 0)
        wsf = self.ws(ops.smallk_ws_floats(D, max(self.NO, K)), dev)
        ops.smallk_wgrad(dx, True, xb, wsf, self.G("x_embedder.proj.weight"), self.G("x_embedder.proj.bias"), None, M, D,
                         K, 0, acc)
        if self.reducer is not None:
            if fg:   # adaLN weights and biases are global-batch averages already: reduce only the embedders' part
                eb, ee = self.reducer.buckets["embed"]
                self.reducer.ready_range(L.ada_b_off + Nall, ee)
            else:
                if not split_ada:
                    for i in reversed(range(self.depth + 1)):
                        self.reducer.ready(f"ada{i}")
                self.reducer.ready("embed")
        if side is not None:  # the optimiser / next micro-step (same stream as this backward) sees every weight gradient
            torch.cuda.current_stream().wait_stream(side)
        self.grad_live = True
        self._attach_grads()

    def _projector_bwd(self, j, tp, dz, dx, B, acc, dev):
        m, D, T = self.m, self.D, self.T
        Pd, Z = m.projector_dim, m.z_dims[j]
        pj = tp.proj[j]
        R = pj.R
        pre = f"projectors.{j}."
        if dz is None:  # projector output unused by the loss: zero grads (torch would leave them None)
            if not acc:
                b, e = self.L.range_of(pre)
                self.A.grad[b:e].zero_()
            return
        dz = dz.reshape(R, Z)
        hdt = ops.half_dtype()   # the build Engine.backward selected
        if dz.dtype != hdt:
            dz = dz.to(hdt)
        dz = dz.contiguous()
        bf = lambda *s: torch.empty(s, dtype=hdt, device=dev)  # noqa: E731
        self._wgrad(dz, pj.p2, pre + "4.weight", R, Z, Pd, acc, dev)
        d2 = bf(R, Pd)
        epi_dsilu = EPI_MUL if tp.act_grad else EPI_DSILU
        ops.gemm(NN, epi_dsilu, dz, self.W(pre + "4.weight"), R, Pd, Z, d2, Z, Pd, Pd, R=pj.p2p, ldr=Pd)
        self._wgrad(d2, pj.p1, pre + "2.weight", R, Pd, Pd, acc, dev)
        d1 = bf(R, Pd)
        ops.gemm(NN, epi_dsilu, d2, self.W(pre + "2.weight"), R, Pd, Pd, d1, Pd, Pd, Pd, R=pj.p1p, ldr=Pd)
        self._wgrad(d1, pj.xin, pre + "0.weight", R, Pd, D, acc, dev)
        if m.z_types[j] == "i":
            ops.gemm(NN, EPI_ADDF32_RB, d1, self.W(pre + "0.weight"), R, D, Pd, dx, Pd, D, D)
        else:
            dmean = bf(R, D)
            ops.gemm(NN, EPI_BF16, d1, self.W(pre + "0.weight"), R, D, Pd, dmean, Pd, D, D)
            ops.token_mean_bwd(dmean, dx, B, T, D)
        tp.proj[j] = None

    def _attach_grads(self):
        """Expose the grad arena through param.grad (views), for torch.optim / clip_grad_norm_ compatibility."""
        g = self.A.grad
        if self._named is None:   # named_parameters() walks the module tree (1 ms per call on SiT-XL/2): once
            self._named = [(name, p) for name, p in self.m.named_parameters()]
        for name, p in self._named:
            if p.requires_grad and p.grad is None:
                p.grad = self.A.view(g, name)

    def zero_grad(self):
        """Equivalent of optimizer.zero_grad(set_to_none=True): the next backward overwrites."""
        self.grad_live = False


class _SiTFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, x, t, y, inference, drop):
        eng = model.engine()
        out, zs, tape = eng.forward(x, t, y, inference, True, drop)
        ctx.eng, ctx.tape = eng, tape
        ctx.nz = 0 if zs is None else len(zs)
        ctx.set_materialize_grads(False)
        return (out,) + (tuple(zs) if zs else ())

    @staticmethod
    def backward(ctx, dout, *dzs):
        if ctx.tape is None:
            raise RuntimeError("reed_amd.SiT: backward called twice on the same forward")
        eng, tape = ctx.eng, ctx.tape
        ctx.tape = None
        if dout is None:
            dout = torch.zeros_like(tape.x)
        # the user may have dropped p.grad (zero_grad(set_to_none=True)) -> overwrite semantics
        if eng._sentinel is None:
            eng._sentinel = next(p for p in eng.m.parameters() if p.requires_grad)
        if eng._sentinel.grad is None:
            eng.grad_live = False
        eng.backward(tape, dout, list(dzs) if ctx.nz else None)
        return (None,) * 7


def sit_apply(model, x, t, y, inference=True):
    ops.require_cuda(x, "x")
    eng = model.engine()
    drop = None
    if model.class_dropout_prob > 0:
        if model.force_drop_mask is not None:
            drop = model.force_drop_mask
        elif model.training:  # LabelEmbedder.token_drop (sit.py:84-93): device RNG draw
            drop = torch.rand(y.shape[0], device=y.device) < model.class_dropout_prob
    need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in (model.x_embedder.proj.weight,))
    if not need_grad:
        out, zs, _ = eng.forward(x, t, y, inference, False, drop)
        return out, zs
    if not hasattr(model, "_anchor") or model._anchor.device != x.device:
        model._anchor = torch.zeros(1, device=x.device, requires_grad=True)
    res = _SiTFunction.apply(model._anchor, model, x, t, y, inference, drop)
    out, zs = res[0], (list(res[1:]) if len(res) > 1 else None)
    if inference:
        zs = None
    elif zs is None:
        zs = []
    return out, zs
