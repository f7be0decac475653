"""One optimisation step of image/train.py (:349 sample_posterior, :363-385 schedules, :387-412 loss combine,
backward, clip, AdamW, EMA) on the HIP path, without per-step host synchronisation: every returned scalar is a
device tensor; call .item() only when logging.
"""
import os

import numpy as np
import torch

from . import ops


def repa_weight_decay(kind, global_step, repa_steps):
    """train.py:364-371"""
    if kind == "constant":
        return 1.0
    if kind == "linear":
        return max(1.0 - global_step / repa_steps, 0.0)
    if kind == "cosine":
        return max((1.0 + np.cos(np.pi * global_step / repa_steps)) / 2, 0.0)
    raise NotImplementedError(kind)


def diffusion_loss_decay(kind, global_step, start_steps, warm_up_steps, max_train_steps):
    """train.py:373-385 (the 'cosine' branch keeps the reference's operator precedence)."""
    top = warm_up_steps + start_steps
    if global_step < start_steps:
        return 0.0
    if start_steps <= global_step < top:
        return (global_step - start_steps) / warm_up_steps
    if kind == "constant":
        return 1.0
    if kind == "linear":
        return 1.0 - (global_step - top) / (max_train_steps - top)
    if kind == "cosine":
        return (1.0 + np.cos(np.pi * (global_step - top) / max_train_steps - top)) / 2
    raise NotImplementedError(kind)


@torch.no_grad()
def sample_posterior(moments, latents_scale=0.18215, latents_bias=0.0, noise=None):
    """train.py:84-91 with scalar scale/bias (the reference uses the same value for all 4 channels)."""
    ops.require_cuda(moments, "moments")
    B, C2 = moments.shape[0], moments.shape[1]
    moments = moments.contiguous().float()
    half = moments.numel() // B // 2
    if noise is None:
        noise = torch.randn((B, C2 // 2) + tuple(moments.shape[2:]), device=moments.device)
    out = torch.empty_like(noise)
    ops.sample_posterior(moments, noise.contiguous(), out, B, half, latents_scale, latents_bias)
    return out


class TrainStep:
    CU_CANDIDATES = (0, 16, 32)   # CU reserves measured by REED_COMM_CUS=auto
    TUNE_STEPS = 3                # event-timed steps per candidate (after one settling step)

    def __init__(self, model, loss_fn, optimizer, reducer=None, proj_coeff=0.5, repa_decay="constant",
                 repa_steps=400000, start_diffusion_steps=0, diffusion_warm_up_steps=50000,
                 diffusion_decay="constant", max_train_steps=400000, latents_scale=0.18215, latents_bias=0.0,
                 grad_accum=1):
        self.model, self.loss_fn, self.opt, self.reducer = model, loss_fn, optimizer, reducer
        if hasattr(optimizer, "overlap") and os.environ.get("REED_OPT_OVERLAP", "1") != "0":
            optimizer.overlap = True   # the step is always followed by a forward (or by flush()/state_dict())
        self.proj_coeff = proj_coeff
        self.sched = (repa_decay, repa_steps, start_diffusion_steps, diffusion_warm_up_steps, diffusion_decay,
                      max_train_steps)
        self.latents_scale, self.latents_bias = latents_scale, latents_bias
        self.grad_accum = max(1, int(grad_accum))
        self.global_step = 0
        self._micro = 0
        # CU reserve for the GEMM grids under data parallelism (csrc/gemm256.hip:reed_num_cus): RCCL's channels hold CUs while
        # a bucket is in flight, and grids planned as exactly one round of all CUs then take two.  How many CUs RCCL takes
        # depends on its version, topology and message size, so it can be MEASURED — opt-in (no 8-GPU run has
        # validated it yet, and a measured choice makes runs differ in their kernel plans: any reserve > 0 leaves fewer than
        # 512 workgroup slots, so the grouped weight-gradient launch gives way to the split-K path):
        #   REED_COMM_CUS unset / off   reserve 0, nothing measured (the default; bit-reproducible run to run)
        #   REED_COMM_CUS=<n>           reserve n
        #   REED_COMM_CUS=auto          the first optimiser steps run with each candidate of CU_CANDIDATES (0, 16, 32):
        #                               one settling step + TUNE_STEPS (3) event-timed steps; per candidate the MEDIAN,
        #                               MAX over ranks; the fastest is kept and logged on rank 0.  bench.py asks for this.
        #                               Then, at that reserve, one more candidate ("static"): the backward WITHOUT the kernel forms
        #                               ops.set_concurrent_comm selects beside a collective (ops.set_comm_forms) — they cost ~ 3 % of
        #                               a step while no RCCL channel holds a CU and save ~ 6 % while 16 CUs are held (one-GPU
        #                               rehearsal, profiles/r4_step_under_cu_hog.txt); how long buckets are in flight is the node's.
        #   REED_COMM_ALGO=auto         afterwards the same measurement for the bucket form (ncclAllReduce vs ncclReduceScatter +
        #                               ncclAllGather, parallel.py); otherwise the form stays what REED_COMM_ALGO names
        #                               (default allreduce): the fp32 summation order is then fixed from run to run.
        #   REED_OPT_SHARD=auto         last, the sharded optimiser pass (optim.py:_shard_plan: 1 / world of every chunk per rank +
        #                               an in-place all-gather of the 16-bit shadows) against the best replicated time, after a
        #                               collective self-test of its gather path; kept only where it measured faster (the
        #                               arithmetic and the operand copies are identical either way).  bench.py asks for this.
        # Any failure inside the measurement's bookkeeping ends it with reserve 0 / allreduce on every rank (tune_error).
        self.cu_reserve = 0
        self.comm_forms = os.environ.get("REED_COMM_FORMS", "1") != "0"   # ops.set_comm_forms: the kernel forms the backward
        ops.set_comm_forms(self.comm_forms)                                # selects beside gradient buckets
        self.cu_tuning = None
        self.tune_error = None
        self._tune = []
        self._tune_algo = False
        self._tune_shard = False
        self.shard_tuning = None
        self.plan_tuning(os.environ.get("REED_COMM_CUS", "off"), os.environ.get("REED_OPT_SHARD"), os.environ.get("REED_COMM_ALGO"))

    def plan_tuning(self, mode="auto", shard="auto", algo=None):
        """Schedule the run-time measurements described in __init__ for the next optimiser steps (what the constructor does from
        the environment).  bench.py calls it AFTER it has timed the plain plan — reserve 0, all-reduce buckets, replicated
        optimiser pass — so that a measurement that fails on first contact with the real interconnect cannot lose that number."""
        reducer, optimizer = self.reducer, self.opt
        mode = mode or "off"
        if self.cu_tuning is not None or self.shard_tuning is not None or self._tune:
            #  a second pass would compare against the first one's records: the keep-the-fastest block is keyed on
            # cu_tuning being unset, the "static" and "rsag" verdicts on their absence from it — refuse instead of mis-measuring
            raise RuntimeError("TrainStep.plan_tuning: this step has already measured (or is measuring) its plan; build a new "
                               "TrainStep to measure again")
        self._tune_shard = (shard == "auto" and reducer is not None and reducer.active()
                            and hasattr(optimizer, "set_sharded") and getattr(optimizer, "overlap", False))
        if reducer is not None and reducer.active() and (mode != "off" or self._tune_shard):
            if mode not in ("off", "auto"):
                self.cu_reserve = int(mode)
                ops.set_cu_reserve(self.cu_reserve)
            if mode == "auto" or self._tune_shard:
                cands = (list(self.CU_CANDIDATES) if mode == "auto"
                         else [self.cu_reserve])          # (only the sharded pass is measured: its replicated baseline)
                nt = max(1, int(self.TUNE_STEPS))
                self._tune = [(c, k) for c in cands for k in range(nt + 1)]   # k = 0: settling step, k >= 1: timed
                self._tune_times = {}
                if mode == "auto" and self.comm_forms:   # (REED_COMM_FORMS=0: already off, nothing to compare)
                    self._tune += [("static", k) for k in range(nt + 1)]
                self._tune_algo = algo == "auto" and hasattr(reducer, "algo")
                if self._tune_algo:
                    self._tune += [("rsag", k) for k in range(nt + 1)]
                if self._tune_shard:
                    self._tune += [("shard", k) for k in range(nt + 1)]

    def __call__(self, x, labels, zs, moments=None, **inject):
        """x: latents [b,4,32,32] (or pass moments=[b,8,32,32] to run sample_posterior). Returns device scalars.
        res["grad_norm"] is a VIEW of one of the optimiser's two alternating norm buffers: read (or clone) it before the step
        after next — a caller that keeps device scalars over a logging window must clone it."""
        rd, rs, sd, wu, dd, mx = self.sched
        w_repa = repa_weight_decay(rd, self.global_step, rs)
        w_diff = diffusion_loss_decay(dd, self.global_step, sd, wu, mx)
        if moments is not None:
            x = sample_posterior(moments, self.latents_scale, self.latents_bias)
        self._micro += 1
        syncing = self._micro % self.grad_accum == 0
        tune = self._tune[0] if self._tune and self._micro % self.grad_accum == 1 % self.grad_accum else None
        if tune is not None:
            if tune[0] == "rsag":
                self.reducer.algo = "rsag"
            elif tune[0] == "static":
                ops.set_comm_forms(False)
            elif tune[0] == "shard":
                if tune[1] == 0 and not self._shard_start():      # self-test failed somewhere: the candidate is dropped
                    tune = None
            else:
                ops.set_cu_reserve(tune[0])
        if tune is not None:
            if tune[1]:   # a timed step of this candidate (tune[1] == 0 is its settling step)
                self._tune_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                self._tune_ev[0].record()
        if self.reducer is not None:
            self.reducer.enabled = syncing  # DDP no_sync on non-final micro-steps (accelerate accumulate())
        out = self.loss_fn(self.model, x, dict(y=labels), zs=zs, **inject)
        den = out["denoising_loss"].mean()
        proj = out["proj_loss"]
        proj_mean = proj.mean() if torch.is_tensor(proj) else torch.zeros((), device=x.device)
        loss = den * w_diff + proj_mean * self.proj_coeff * w_repa
        st = getattr(self.opt, "scaler_state", None)
        if st is not None:   # fp16: scaler.scale(loss).backward() (accelerate, train.py:401); the scale is a device scalar
            (loss * (st[0].detach() / self.grad_accum)).backward()
        else:
            (loss / self.grad_accum).backward()
        res = {"loss": loss.detach(), "denoising_loss": den.detach(), "proj_loss": proj_mean.detach(),
               "img_proj_loss": out["img_proj_loss"], "text_proj_loss": out["text_proj_loss"]}
        if syncing:
            if self.reducer is not None:
                self.reducer.sync()
            self.opt.step()          # clip + AdamW + EMA + bf16 re-cast, fused
            self.opt.zero_grad()
            self.global_step += 1
            # a view of the optimiser's norm buffer of this step (two buffers in turn: valid until the step after next).  A
            # .clone() here was a 4-byte copy kernel at the very point where the overlapped optimiser's chunks start: in the
            # b = 32 kernel trace it sat 1.2 ms in its queue and ran 0.6 ms (tools/step_dump.py), with the next step behind it
            res["grad_norm"] = self.opt.grad_norm
            if self._tune:
                self._tune_step()
        return res

    def tune_steps_left(self):
        """Optimiser steps the CU-reserve measurement still needs (bench.py keeps them out of its timed region)."""
        return len(self._tune)

    def _agree(self, times):
        import torch.distributed as dist
        t = torch.tensor(times, dtype=torch.float64, device=self.opt.grad_norm.device)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.tolist()

    def _shard_start(self):
        """First step of the sharded-pass candidate: collective self-test of its gather path, then switch the optimiser over.
        False (on every rank alike) drops the candidate."""
        ok = False
        try:
            ok = self.opt.shard_selftest()
            if ok:
                self.opt.set_sharded(True)
        except Exception as e:
            ok = False
            self.tune_error = repr(e)
        if not ok:
            self._tune = [t for t in self._tune if t[0] != "shard"]
            self.shard_tuning = {"kept": False, "reason": "self-test of the in-place all-gather failed"}
            self._log("sharded optimiser pass not measured: the self-test of its all-gather failed on some rank")
        return ok

    def _tune_abort(self, err):
        """Leave the measurement with the safe plan (reserve 0, all-reduce buckets, replicated optimiser pass).  Every rank runs
        the same bookkeeping on the same schedule, so a deterministic failure ends it on all of them at the same step."""
        self._tune = []
        self.tune_error = repr(err)
        self.cu_reserve = 0
        ops.set_cu_reserve(0)
        self.comm_forms = os.environ.get("REED_COMM_FORMS", "1") != "0"
        ops.set_comm_forms(self.comm_forms)
        if self._tune_algo:
            self.reducer.algo = "allreduce"
        if self._tune_shard:
            try:
                self.opt.set_sharded(False)
            except Exception:
                pass

    def _median_ms(self, cand):
        ms = []
        for e0, e1 in self._tune_times[cand]:
            e1.synchronize()
            ms.append(e0.elapsed_time(e1))
        ms.sort()
        return ms[len(ms) // 2]

    def _tune_step(self):
        try:
            cand, k = self._tune.pop(0)
            if k:
                self._tune_ev[1].record()
                self._tune_times.setdefault(cand, []).append(self._tune_ev)
            nxt = self._tune[0][0] if self._tune else None
            if self.cu_tuning is None and (nxt is None or nxt in ("static", "rsag", "shard")):   # every reserve candidate is timed: keep the fastest
                cands = sorted(c for c in self._tune_times if c not in ("static", "rsag", "shard"))
                t = self._agree([self._median_ms(c) for c in cands])
                best = min(range(len(cands)), key=lambda i: t[i])
                self.cu_reserve = cands[best]
                self.cu_tuning = {str(c): round(v, 3) for c, v in zip(cands, t)}
                self._best_ms = t[best]
                ops.set_cu_reserve(self.cu_reserve)
                self._log(f"CU reserve {self.cu_reserve} kept (median ms per step, MAX over ranks: {self.cu_tuning})")
            if "static" in self._tune_times and nxt != "static" and "static" not in self.cu_tuning:   # the kernel forms beside buckets
                t = self._agree([self._median_ms("static")])[0]
                self.cu_tuning["static"] = round(t, 3)
                self.comm_forms = not (t < self._best_ms)
                ops.set_comm_forms(self.comm_forms)
                self._log(f"kernel forms beside collectives {'kept' if self.comm_forms else 'dropped'} "
                          f"(without them {t:.3f} ms vs {self._best_ms:.3f} ms per step)")
                self._best_ms = min(t, self._best_ms)
            if self._tune_algo and "rsag" in self._tune_times and nxt != "rsag" and "rsag" not in self.cu_tuning:   # the bucket form
                t = self._agree([self._median_ms("rsag")])[0]
                self.cu_tuning["rsag"] = round(t, 3)
                self.reducer.algo = "rsag" if t < self._best_ms else "allreduce"
                self._log(f"bucket form {self.reducer.algo} kept (rsag {t:.3f} ms vs allreduce {self._best_ms:.3f} ms)")
                self._best_ms = min(t, self._best_ms)
            if not self._tune and "shard" in self._tune_times:                # the sharded optimiser pass at that plan
                t = self._agree([self._median_ms("shard")])[0]
                keep = t < self._best_ms
                self.shard_tuning = {"kept": keep, "sharded_ms": round(t, 3), "replicated_ms": round(self._best_ms, 3)}
                if not keep:
                    self.opt.set_sharded(False)
                self._log(f"optimiser pass {'sharded' if keep else 'replicated'} kept (sharded {t:.3f} ms vs replicated "
                          f"{self._best_ms:.3f} ms per step)")
        except Exception as e:   # never fatal: the safe plan on every rank
            self._tune_abort(e)
            self._log(f"measurement abandoned ({e!r}): CU reserve 0, all-reduce buckets")

    def _log(self, msg):
        if self.reducer is None or getattr(self.reducer, "rank", 0) == 0:
            import sys
            print(f"[reed_amd.TrainStep] {msg}", file=sys.stderr, flush=True)
