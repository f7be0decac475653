"""Data-parallel gradient reduction: one process per GPU, RCCL over xGMI (replaces accelerate ->
DistributedDataParallel, image/train.py:151,293-295,401).

The gradient arena is cut into buckets that are contiguous ranges in backward-completion order
(final layer, blocks L-1..0, projectors, embedders+adaLN; reed_amd/arena.py:ArenaLayout.buckets). The engine calls
`ready(name)` as soon as a bucket's last weight gradient has been launched; the reduction runs on the
communicator's own high-priority stream (ordered after the compute stream by an event) and therefore overlaps the
rest of backward. `sync()` makes the compute stream wait for the last bucket before grad-norm / AdamW.
DDP semantics kept: average (not sum), fp32, parameters broadcast from rank 0 at start.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import _lib


def shard_batch(global_batch, world):
    """Local batch per rank (train.py:263: int(batch_size // num_processes))."""
    return int(global_batch // world)


def rank_seed(seed, rank):
    """train.py:175-176: set_seed(args.seed + accelerator.process_index)."""
    return seed + rank


def sample_seed(global_seed, world, rank):
    """generate.py:49: seed = global_seed * world_size + rank."""
    return global_seed * world + rank


def sample_index(i, world, rank, total):
    """generate.py:164: index = i * world_size + rank + total."""
    return i * world + rank + total


def check_buckets(layout):
    """Every trainable parameter segment lies in exactly one bucket; buckets do not overlap."""
    bk = layout.buckets()
    spans = sorted(r for _, r in bk)
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        if a1 > b0:
            raise AssertionError(f"overlapping buckets {(a0, a1)} {(b0, b1)}")
    for name, (off, shp) in layout.seg.items():
        if name == "pos_embed":
            continue
        n = 1
        for s in shp:
            n *= s
        hits = [k for k, (b, e) in bk if b <= off and off + n <= e]
        if len(hits) != 1:
            raise AssertionError(f"{name} covered by buckets {hits}")
    return bk


class GradReducer:
    """RCCL-backed reducer (torch.distributed must be initialised: it carries the 128-byte RCCL id of the native
    communicator, the ranks' agreement on the binding, and the torch binding's collectives).

    Two bindings of the same collective (RCCL all-reduce(avg), fp32, in place, on a stream beside backward):
      torch    torch.distributed's NCCL(=RCCL) process group on a torch side stream — the default: it is the one RCCL
               instance every torch job on this image already runs, and no N > 1 run of the native binding exists yet;
      native   libreed_hip.so's own communicator + high-priority stream (csrc/comm.cpp), `REED_COMM=native`. The ranks
               AGREE on it: after the attempt every rank contributes an ok flag to a MIN all-reduce over the torch
               process group, and if any rank failed, every rank destroys its native communicator and uses the torch
               binding (a per-rank decision would leave the ranks issuing mismatched collectives: a hang, not an error).
    Two forms of the reduction (`REED_COMM_ALGO`): `allreduce` (default) = one ncclAllReduce(avg) per bucket;
    `rsag` = ncclReduceScatter(avg) + ncclAllGather on the bucket, the direct form SURVEY.md §5 derives for the fully
    connected xGMI node. Both average in fp32, but in RCCL's own summation order per form: the low bits of the averaged
    gradients may differ between the two, so the form is FIXED for a run (and from run to run) unless the caller asks for the
    run-time measurement with REED_COMM_ALGO=auto (trainer.TrainStep)."""

    def __init__(self, model, rank=None, world=None):
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.model = model
        self.buckets = dict(check_buckets(model._layout))
        self.enabled = True
        self.force = os.environ.get("REED_FORCE_REDUCER", "0") == "1"  # run the RCCL calls even at world == 1 (tests)
        self._lib = _lib.load()
        self.comm = None
        self._tstream = None
        # adaLN weight gradients by factor all-gather instead of all-reduce (see gather()): power-of-two worlds only
        # (the 1 / world average is folded into the bf16 factor, exact for powers of two); REED_ADA_GATHER=0 turns it off
        self.ada_gather = os.environ.get("REED_ADA_GATHER", "1") != "0" and (self.world & (self.world - 1)) == 0
        self._gbuf = {}
        mode = os.environ.get("REED_COMM", "torch")
        self.algo = os.environ.get("REED_COMM_ALGO", "allreduce")
        if self.algo == "auto":     # trainer.TrainStep measures the two forms in its first steps; until then the default
            self.algo = "allreduce"
        if self.algo not in ("allreduce", "rsag"):
            raise ValueError(f"REED_COMM_ALGO={self.algo!r}: expected 'allreduce', 'rsag' or 'auto'")
        self.timing = None       # per-bucket event pairs when bench.py asks for the exposed-communication diagnosis
        if mode == "native":
            err = None
            try:
                self._init_native()
            except RuntimeError as e:   # e.g. a second RCCL instance that cannot bootstrap next to torch's
                err = e
            if not self._agree(err is None):
                if self.comm is not None:
                    self._lib.reed_comm_destroy(self.comm)
                    self.comm = None
                import warnings
                warnings.warn(f"reed_amd: native RCCL communicator unavailable on at least one rank ({err}); every "
                              "rank reduces through torch.distributed's RCCL process group instead")
        elif mode != "torch":
            raise ValueError(f"REED_COMM={mode!r}: expected 'torch' or 'native'")
        if self.comm is None:
            if not dist.is_initialized():
                raise RuntimeError("GradReducer: torch.distributed must be initialised for the torch RCCL binding")
            self._tstream = torch.cuda.Stream(device=model._arena.master.device, priority=-1)
            # AVG and all_gather_into_tensor on device tensors are RCCL features; on any other backend (gloo: the
            # two-ranks-on-one-GPU test of tests/test_cli_gpu.py) the same results come from SUM + scale / a SUM over a
            # zero-padded buffer
            self._rccl = dist.get_backend() == "nccl"
        model.engine().reducer = self

    def _agree(self, ok):
        """True iff EVERY rank's `ok` is true (MIN all-reduce over the torch process group)."""
        if self.world == 1 or not dist.is_initialized():
            return ok
        dev = self.model._arena.master.device if dist.get_backend() == "nccl" else "cpu"
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    def _init_native(self):
        L = self._lib
        idbuf = ctypes.create_string_buffer(128)
        obj = [b""]
        id_err = None
        if self.rank == 0:
            # a failure here must not leave the other ranks blocked in the broadcast below while rank 0 goes on to the
            # agreement all-reduce: rank 0 still broadcasts, an EMPTY id, and every rank raises after it
            try:
                _lib.check(L.reed_comm_unique_id(idbuf), "comm_unique_id")
                obj = [bytes(idbuf.raw)]
            except RuntimeError as e:
                id_err = e
        if self.world > 1:
            dist.broadcast_object_list(obj, src=0)
        if len(obj[0]) != 128:
            raise RuntimeError(f"rank 0 could not create an RCCL unique id ({id_err})")
        comm = ctypes.c_void_p()
        _lib.check(L.reed_comm_init(obj[0], self.rank, self.world, ctypes.byref(comm)), "comm_init")
        self.comm = comm

    @property
    def binding(self):
        return "native" if self.comm is not None else "torch"

    def _stream(self):
        return torch.cuda.current_stream().cuda_stream

    def broadcast_params(self, root=0):
        A = self.model._arena
        A.wait_all()
        if self.comm is not None:
            _lib.check(self._lib.reed_comm_broadcast(self.comm, A.master.data_ptr(), A.master.numel(), root,
                                                     self._stream()), "comm_broadcast")
        else:
            dist.broadcast(A.master, src=root)
        A.shadow_version = -1

    def ready(self, name):
        if not self.enabled or (self.world == 1 and not self.force):
            return
        b, e = self.buckets[name]
        self._reduce(b, e, name)

    def _reduce(self, b, e, name):
        """All-reduce(avg) of grad[b:e) on the communicator's stream, ordered after the current stream."""
        g = self.model._arena.grad
        if self.comm is not None:
            fn = self._lib.reed_comm_allreduce_avg_rsag if self.algo == "rsag" else self._lib.reed_comm_allreduce_avg
            _lib.check(fn(self.comm, g.data_ptr() + 4 * b, e - b, self._stream()), "comm_allreduce_avg")
            return
        self._tstream.wait_stream(torch.cuda.current_stream())   # the bucket's gradients are written
        with torch.cuda.stream(self._tstream):
            if self.timing is not None:
                ev0 = torch.cuda.Event(enable_timing=True)
                ev0.record()
            self._t_allreduce_avg(g[b:e])                          # stream-ordered: no host wait
            if self.timing is not None:
                ev1 = torch.cuda.Event(enable_timing=True)
                ev1.record()
                self.timing.append((name, 4 * (e - b), ev0, ev1))

    def _t_allreduce_avg(self, t):
        n = t.numel()
        chunk = n // self.world
        if self.algo == "rsag" and self._rccl and chunk > 0:
            body = t[:chunk * self.world]
            mine = body[self.rank * chunk:(self.rank + 1) * chunk]
            dist.reduce_scatter_tensor(mine, body, op=dist.ReduceOp.AVG)
            dist.all_gather_into_tensor(body, mine)
            if n > chunk * self.world:
                dist.all_reduce(t[chunk * self.world:], op=dist.ReduceOp.AVG)
        elif self._rccl:
            dist.all_reduce(t, op=dist.ReduceOp.AVG)
        elif self.algo == "rsag" and chunk > 0:   # gloo (tests): the same two-phase structure with SUM + scale
            body = t[:chunk * self.world]
            parts = list(body.view(self.world, chunk).unbind(0))
            mine = torch.empty_like(parts[self.rank])
            dist.reduce_scatter(mine, [p.contiguous() for p in parts])
            mine.mul_(1.0 / self.world)
            dist.all_gather(parts, mine)
            if n > chunk * self.world:
                tail = t[chunk * self.world:]
                dist.all_reduce(tail)
                tail.mul_(1.0 / self.world)
        else:
            dist.all_reduce(t)
            t.mul_(1.0 / self.world)

    def active(self):
        """True when this backward's gradients are being reduced (not a no_sync micro-step, more than one rank)."""
        return self.enabled and (self.world > 1 or self.force)

    def ready_range(self, b, e):
        """All-reduce(avg) of the gradient range [b, e) (a bucket cut short: see Engine.backward's factor path)."""
        if not self.active() or e <= b:
            return
        self._reduce(b, e, f"range[{b}:{e}]")

    def gather_buffers(self, name, numel, dtype, device):
        """Persistent (send [numel], recv [world * numel]) pair: the collectives run on the communicator's stream, so
        the buffers must not go back to the caching allocator while a gather may still be reading them."""
        key = (name, numel, dtype)
        if key not in self._gbuf:
            self._gbuf[key] = (torch.empty(numel, dtype=dtype, device=device),
                               torch.empty(self.world * numel, dtype=dtype, device=device))
        return self._gbuf[key]

    def gather(self, send, recv):
        """recv[r * n : (r + 1) * n] = rank r's send (contiguous tensors; n = send.numel()), asynchronously on the
        communicator's stream, ordered after the current stream. The adaLN matrix is a third of all parameters and its
        gradient dW = dmod^T silu(c) contracts over the LOCAL batch only: exchanging the two factors (world x b x 6D
        bf16 per block) and forming the global-batch product on every rank moves 0.1 GB per step instead of the
        0.9 GB the all-reduce of the matrix does (XL/2, 8 ranks)."""
        assert send.is_contiguous() and recv.is_contiguous() and recv.numel() == self.world * send.numel()
        nbytes = send.numel() * send.element_size()
        if self.comm is not None:
            _lib.check(self._lib.reed_comm_allgather(self.comm, send.data_ptr(), recv.data_ptr(), nbytes, self._stream()),
                       "comm_allgather")
            return
        self._tstream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._tstream):
            if self._rccl:
                dist.all_gather_into_tensor(recv, send)
            else:   # exact: every element is one rank's bit pattern plus zeros
                ri, si = recv.view(torch.int32), send.view(torch.int32)   # (16-bit payloads: an even element count)
                n = si.numel()
                ri.zero_()
                ri[self.rank * n:(self.rank + 1) * n].copy_(si)
                dist.all_reduce(ri)
            self._gev = torch.cuda.Event()
            self._gev.record()

    def gather_sync(self):
        """The current stream waits for every gather issued so far (not for bucket reductions queued behind them)."""
        if self.comm is not None:
            _lib.check(self._lib.reed_comm_sync_gather(self.comm, self._stream()), "comm_sync_gather")
        elif getattr(self, "_gev", None) is not None:
            torch.cuda.current_stream().wait_event(self._gev)

    def sync(self):
        if not self.enabled or (self.world == 1 and not self.force):
            return
        if self.comm is not None:
            _lib.check(self._lib.reed_comm_sync(self.comm, self._stream()), "comm_sync")
        else:
            torch.cuda.current_stream().wait_stream(self._tstream)

    def close(self):
        if self.comm:
            self._lib.reed_comm_destroy(self.comm)
            self.comm = None


class TorchDistGradReducer:
    """Same bucket plan through torch.distributed.all_reduce (gloo on CPU in the tests; checks the plan, the
    averaging and the ordering contract without a GPU). Not used on the product path."""

    def __init__(self, layout, grad, world):
        self.buckets = dict(check_buckets(layout))
        self.grad, self.world = grad, world
        self.fired = []

    def ready(self, name):
        b, e = self.buckets[name]
        dist.all_reduce(self.grad[b:e])
        self.grad[b:e].div_(self.world)
        self.fired.append(name)

    def sync(self):
        pass
