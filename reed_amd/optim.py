"""Fused clip_grad_norm_ + AdamW + EMA (+ bf16 weight re-cast) over the flat parameter arena — replaces
image/train.py:402-412 (accelerator.clip_grad_norm_, torch.optim.AdamW.step, zero_grad, update_ema) with three
HIP launches and no host synchronisation (the gradient norm stays on the device until someone asks for it).
"""
import os

import torch

from . import ops

_NB = 2048  # partial-sum blocks of the squared-norm reduction


class FusedAdamWEMA:
    def __init__(self, model, ema=None, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.0, eps=1e-8,
                 max_grad_norm=1.0, ema_decay=0.9999, overlap=False, init_scale=65536.0, growth_factor=2.0,
                 backoff_factor=0.5, growth_interval=2000):
        """overlap=True: the update runs on the optimiser's own HIP stream, one launch per parameter bucket in the
        order the next forward touches them (ArenaLayout.update_chunks), each followed by an event that the forward
        waits for just before the first kernel that reads the bucket. The pass is HBM-bound (38 B/param), the
        forward MFMA-bound, so the two overlap; at b = 32/GPU it also fills the CUs that one-round GEMM grids leave
        idle. Readers other than the model forward (state_dict, EMA sampling, torch ops on p.data) are ordered by
        model.state_dict() / Engine.forward / flush()."""
        self.model, self.ema = model, ema
        self.overlap = bool(overlap) and os.environ.get("REED_OPT_OVERLAP", "1") != "0"
        self._stream = None
        self._chunks = None
        self._params = None
        # REED_OPT_SHARD=1 (data-parallel runs, opt-in; "auto": TrainStep measures it against the replicated pass): every rank
        # updates 1 / world of every update chunk and the 16-bit shadows are all-gathered in place (see _shard_plan /
        # sync_replicas / set_sharded); decided at the first step
        self._want_shard = os.environ.get("REED_OPT_SHARD", "0") == "1"
        self._shard = None       # [(chunk name, [(begin, end, owner rank | -1 = every rank), ...]), ...]; False = replicated
        self._rank, self._world = 0, 1
        self.lr, self.betas, self.weight_decay, self.eps = lr, tuple(betas), weight_decay, eps
        self.max_grad_norm, self.ema_decay = max_grad_norm, ema_decay
        self.step_count = 0
        A = model._arena
        if A.master.device.type != "cuda":
            raise RuntimeError("FusedAdamWEMA: move the model to the GPU first")
        dev = A.master.device
        n = model._layout.n_train
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.partial = torch.empty(_NB, dtype=torch.float32, device=dev)
        # [||g||, clip coefficient] of the last step — two buffers used in turn, so that the norm a caller took from step k
        # (grad_norm below: a view, no copy kernel) stays valid until step k + 2 writes it again
        self._norm_clips = [torch.zeros(2, dtype=torch.float32, device=dev) for _ in range(2)]
        self.norm_clip = self._norm_clips[0]
        # fp16 training (model.precision == "fp16"; the reference's README recipe): dynamic loss scaling with
        # torch.amp.GradScaler's defaults, as accelerate builds it — [scale, growth_tracker, found_inf, steps_taken] on the
        # device. TrainStep multiplies the loss by scaler_state[0]; step() unscales inside the clip coefficient, skips the
        # update on overflow and updates the scale, all without a host synchronisation (csrc/optim.hip).
        self.scaler_state = None
        self.scaler_cfg = (float(growth_factor), float(backoff_factor), int(growth_interval))
        if getattr(model, "precision", "bf16") == "fp16":
            self.scaler_state = torch.tensor([float(init_scale), 0.0, 0.0, 0.0], dtype=torch.float32, device=dev)
        if ema is not None:
            if ema._layout.n_total != model._layout.n_total or ema._arena.master.device != dev:
                raise ValueError("EMA model must be a deepcopy of the model on the same device")

    @property
    def grad_norm(self):
        """0-d device tensor: the pre-clip global gradient norm of the last step (no host sync until .item()); a view that the
        step after next overwrites."""
        return self.norm_clip[0]

    @torch.no_grad()
    def step(self):
        m = self.model
        A, L = m._arena, m._layout
        if A.grad is None:
            raise RuntimeError("FusedAdamWEMA.step() called before any backward")
        prec = getattr(m, "precision", "bf16")
        prev = ops.use(prec)      # the update writes the 16-bit shadow in the model's operand type
        try:
            self._step(m, A, L, prec)
        finally:
            ops.use(prev)

    def _step(self, m, A, L, prec):
        A.ensure_shadow(prec)
        self.norm_clip = self._norm_clips[(self.step_count + 1) & 1]
        nc = None
        st = self.scaler_state
        clip = self.max_grad_norm is not None and self.max_grad_norm > 0
        if st is not None:        # scaled gradients: the norm pass also finds overflows, the coefficient also unscales
            ops.grad_sqnorm(A.grad, L.n_train, self.partial, _NB)
            ops.clip_finalize_scaled(self.partial, _NB, float(self.max_grad_norm) if clip else 0.0, self.norm_clip, st,
                                     *self.scaler_cfg)
            nc = self.norm_clip
        elif clip:
            ops.grad_sqnorm(A.grad, L.n_train, self.partial, _NB)
            ops.clip_finalize(self.partial, _NB, float(self.max_grad_norm), self.norm_clip)
            nc = self.norm_clip
        self.step_count += 1
        b1, b2 = self.betas
        bc1 = 1.0 - b1 ** self.step_count
        bc2 = 1.0 - b2 ** self.step_count
        ema_buf = self.ema._arena.master if self.ema is not None else None
        keep_t = False
        if not self.overlap and not self._want_shard:
            ops.adamw_ema(A.master, A.grad, self.exp_avg, self.exp_avg_sq, ema_buf, A.shadow, L.n_train, L.n_total, nc,
                          self.lr, b1, b2, self.eps, self.weight_decay, bc1, bc2, self.ema_decay, scaler_state=st)
        else:
            A.wait_all()
            main = torch.cuda.current_stream()
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=A.master.device)
                eng = m.engine()
                self._chunks = L.update_chunks(list(eng.tap_depth) if m.z_dims else ())
            if self._shard is None:
                self._shard = self._shard_plan(L) if self._want_shard else False
            side = self._stream
            side.wait_stream(main)  # grads, clip coefficient
            # the transposed copies of the blocks' weights (arena.py: the input gradients' NT operand) follow each block's update on
            # this stream where the engine has asked for them and they were fresh before this step
            keep_t = A.shadow_t is not None and A.shadow_t_gen == A.shadow_gen and A.shadow_t.dtype == A.shadow.dtype
            pp, gp, sp = A.master.data_ptr(), A.grad.data_ptr(), A.shadow.data_ptr()
            hb = A.shadow.element_size()
            mp, vp = self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
            ep = ema_buf.data_ptr() if ema_buf is not None else None
            ev = None

            def update(b, e):
                nt = max(0, min(e, L.n_train) - b)
                ops.adamw_ema(pp + 4 * b, gp + 4 * b if nt else None, mp + 4 * b if nt else None,
                              vp + 4 * b if nt else None, ep + 4 * b if ep is not None else None, sp + hb * b, nt,
                              e - b, nc, self.lr, b1, b2, self.eps, self.weight_decay, bc1, bc2, self.ema_decay,
                              scaler_state=st)

            with torch.cuda.stream(side):
                if self._shard:
                    import torch.distributed as dist
                    sh32 = A.shadow.view(torch.int32)   # bit patterns (every backend carries int32; pieces are 4-element aligned)
                    per = 4 // hb                        # shadow elements per int32
                    for name, subs in self._shard:
                        for b, e, owner in subs:
                            if owner in (self._rank, -1):
                                update(b, e)
                        own = [s_ for s_ in subs if s_[2] >= 0]
                        if own:     # equal pieces, rank order: one in-place all-gather of the operand copies of this chunk
                            self._gather(sh32, own[0][0] // per, (own[0][1] - own[0][0]) // per)
                            for fb, fe in self._f32_read:    # ... and the few parameters the forward reads from the fp32 master
                                for b, e, owner in own:
                                    lo, hi = max(b, fb), min(e, fe)
                                    if hi > lo:
                                        dist.broadcast(A.master[lo:hi], src=owner)
                        ev = torch.cuda.Event()
                        ev.record(side)
                        A.pending[name] = ev
                else:
                    for name, b, e in self._chunks:
                        update(b, e)
                        ev = torch.cuda.Event()
                        ev.record(side)
                        A.pending[name] = ev
            if keep_t:   # behind every chunk (the forward waits for none of this); first read by the next backward
                with torch.cuda.stream(side):
                    for i in range(L.depth):
                        A.transpose_block(i)
                    evt = torch.cuda.Event()
                    evt.record(side)
                A.pending_t = evt
            if self.ema is not None:
                self.ema._arena.pending["all"] = ev
        A.mark_shadow_fresh()
        if keep_t:
            A.shadow_t_gen = A.shadow_gen
        if self.ema is not None:
            self.ema._arena.shadow_version = -1  # EMA master changed behind torch's back: re-cast on next use

    def _shard_plan(self, L, world=None, rank=None):
        """Sharded update (REED_OPT_SHARD=1 | auto).  The fused pass moves 38 bytes per parameter whatever
        the batch: at b = 32 per GPU it is 13 % of the step and every rank of a data-parallel run repeats it identically.
        Here every update chunk (ArenaLayout.update_chunks: next-forward order) is cut into `world` equal 4-aligned pieces, rank
        r owning piece r, plus a tail of fewer than 4 world elements that every rank updates; a rank runs the fused kernel on
        ITS pieces only (master, Adam moments, EMA, 16-bit shadow) and the chunk's shadow is completed by ONE in-place
        all-gather on the optimiser's stream (every rank sends on all its links at once: 7/8 of 2 bytes per parameter received
        per rank, where owner-by-owner broadcasts — the first form of this — would have used one rank's links at a time), the
        forward waiting per chunk as before.  2 bytes per parameter on the wire instead of 38 through HBM on world - 1 of
        world ranks.  The gradients are still all-reduced (every rank holds the averaged gradient and computes the same norm
        and clip coefficient), the arithmetic of a piece is the replicated step's, so the shadows — what the forward computes
        with — are bit-identical to the replicated run's on every rank.  What a non-owner does NOT have is the fp32 master /
        moments / EMA of the pieces it does not own: sync_replicas() (collective) brings them up to date before anything
        reads them (checkpoints, EMA sampling, state_dict)."""
        if world is None:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
                return False
            world, rank = dist.get_world_size(), dist.get_rank()
        self._rank, self._world = rank, world
        # parameters the forward reads in fp32 straight from the master arena (the label table, the q / k LayerNorm affines):
        # their owners broadcast them behind the shadows
        self._f32_read = []
        for name, (off, shp) in L.seg.items():
            if name == "y_embedder.embedding_table.weight" or ".q_norm." in name or ".k_norm." in name:
                n = 1
                for d in shp:
                    n *= d
                self._f32_read.append((off, off + n))
        plan = []
        for name, b, e in self._chunks:
            piece = ((e - b) // world) // 4 * 4
            subs = [(b + r * piece, b + (r + 1) * piece, r) for r in range(world)] if piece else []
            if b + world * piece < e:
                subs.append((b + world * piece, e, -1))
            plan.append((name, subs))
        return plan

    def _gather(self, buf, start, n):
        """buf[start : start + world n] <- every rank's buf[start + r n : start + (r + 1) n], in place."""
        import torch.distributed as dist
        out = buf[start:start + self._world * n]
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(out, buf[start + self._rank * n:start + (self._rank + 1) * n])
        else:       # gloo (CPU tests, rehearsals) has no all-gather on device tensors: the same bytes, owner by owner
            for r in range(self._world):
                dist.broadcast(buf[start + r * n:start + (r + 1) * n], src=r)

    def shard_selftest(self):
        """Collective.  One in-place all-gather of a scratch buffer through the path the sharded update uses, checked, and the
        verdict agreed over the ranks (MIN): TrainStep tries the sharded pass only where this returned True everywhere."""
        import torch.distributed as dist
        ok = 1
        try:
            if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
                return False
            self._rank, self._world = dist.get_rank(), dist.get_world_size()
            n = 256
            buf = torch.full((self._world * n,), -1, dtype=torch.int32, device=self.exp_avg.device)
            buf[self._rank * n:(self._rank + 1) * n] = self._rank
            self._gather(buf, 0, n)
            want = torch.arange(self._world, dtype=torch.int32, device=buf.device).repeat_interleave(n)
            ok = int(torch.equal(buf, want))
        except Exception:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=self.exp_avg.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    def set_sharded(self, on):
        """Collective.  Switch between the replicated and the sharded update between two steps (TrainStep's run-time
        measurement): turning it off first makes every rank a full replica again."""
        if on:
            if not self._shard:
                self._want_shard, self._shard = True, None
        elif self._shard or self._want_shard:
            self.sync_replicas()
            self._want_shard, self._shard = False, False

    def sync_replicas(self):
        """Collective (every rank must call it).  After a sharded step only a piece's owner holds its fp32 master weights,
        Adam moments and EMA: broadcast them from the owners so that every rank is a full replica again — before a checkpoint,
        an EMA forward, state_dict() or any torch read of the parameters.  No-op for the replicated update."""
        if not self._shard:
            return
        import torch.distributed as dist
        self.flush()
        A, L = self.model._arena, self.model._layout
        bufs = [A.master, self.exp_avg, self.exp_avg_sq] + ([self.ema._arena.master] if self.ema is not None else [])
        for _, subs in self._shard:
            for b, e, owner in subs:
                if owner < 0:
                    continue
                for t in bufs:
                    hi = min(e, t.numel())
                    if hi > b:
                        dist.broadcast(t[b:hi], src=owner)
        if self.ema is not None:
            self.ema._arena.shadow_version = -1

    def flush(self):
        """Order the current stream after an overlapped update (before reading parameters / EMA with torch ops)."""
        self.model._arena.wait_all()
        if self.ema is not None:
            self.ema._arena.wait_all()

    def zero_grad(self, set_to_none=False):
        """The next backward OVERWRITES the gradient arena (no memset).  param.grad stays attached as a view of the arena
        (stale until that backward) unless set_to_none=True: dropping and re-creating ~300 views costs 2 ms of host time per
        step, which at b = 32 per GPU is the difference between a GPU-bound and an enqueue-bound backward."""
        eng = self.model._engine
        if eng is not None:
            eng.zero_grad()
        if set_to_none:
            if self._params is None:
                self._params = list(self.model.parameters())
            for p in self._params:
                p.grad = None

    # ---- checkpoint compatibility with torch.optim.AdamW.state_dict() (train.py:423) ----
    def state_dict(self):
        self.flush()
        L = self.model._layout
        state, idx = {}, []
        # with a loss scaler torch's per-parameter "step" counts the updates actually applied (skipped steps do not count)
        steps = self.step_count if self.scaler_state is None else int(self.scaler_state[3].item())
        for i, (name, p) in enumerate(self.model.named_parameters()):
            idx.append(i)
            if not p.requires_grad or steps == 0:
                continue
            off, shp = L.seg[name]
            n = p.numel()
            state[i] = {"step": torch.tensor(float(steps)),
                        "exp_avg": self.exp_avg[off:off + n].view(shp).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + n].view(shp).clone()}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "params": idx}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        L = self.model._layout
        names = [n for n, _ in self.model.named_parameters()]
        steps = 0
        for i, st in sd["state"].items():
            name = names[int(i)]
            off, shp = L.seg[name]
            n = int(torch.tensor(shp).prod())
            self.exp_avg[off:off + n].copy_(st["exp_avg"].flatten())
            self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].flatten())
            steps = max(steps, int(float(st["step"])))
        self.step_count = steps
        if self.scaler_state is not None:   # the scale itself restarts at init_scale, as accelerate's does on resume
            self.scaler_state[3] = float(steps)
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.weight_decay = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]


@torch.no_grad()
def update_ema(ema_model, model, decay=0.9999):
    """train.py:94-105 as a single fused pass (used for the decay=0 initial sync; the per-step update is fused
    into FusedAdamWEMA.step)."""
    A, E = model._arena, ema_model._arena
    ops.adamw_ema(A.master, None, None, None, E.master, None, 0, model._layout.n_total, None, 0.0, 0.9, 0.999, 1e-8,
                  0.0, 1.0, 1.0, decay)
    E.shadow_version = -1
