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
    """RCCL-backed reducer. Needs torch.distributed initialised (any backend) only to ship the 128-byte RCCL id.

    Two bindings of the same collective (RCCL all-reduce(avg), fp32, in place, on a stream beside backward):
      native   libreed_hip.so's own communicator + high-priority stream (csrc/comm.cpp) — the default;
      torch    torch.distributed's NCCL(=RCCL) process group on a torch side stream — `REED_COMM=torch`, and the
               automatic fall-back (with a warning) when the native communicator cannot be created, so that a
               multi-GPU job still runs on RCCL rather than not at all."""

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
        mode = os.environ.get("REED_COMM", "native")
        if mode != "torch":
            try:
                self._init_native()
            except RuntimeError as e:   # e.g. a second RCCL instance that cannot bootstrap next to torch's
                import warnings
                warnings.warn(f"reed_amd: native RCCL communicator unavailable ({e}); gradient all-reduce goes "
                              "through torch.distributed's RCCL process group instead")
                self.comm = None
        if self.comm is None:
            if not dist.is_initialized():
                raise RuntimeError("GradReducer: torch.distributed must be initialised for the torch RCCL binding")
            self._tstream = torch.cuda.Stream(device=model._arena.master.device, priority=-1)
        model.engine().reducer = self

    def _init_native(self):
        L = self._lib
        idbuf = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _lib.check(L.reed_comm_unique_id(idbuf), "comm_unique_id")
        obj = [bytes(idbuf.raw)]
        if self.world > 1:
            dist.broadcast_object_list(obj, src=0)
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
        g = self.model._arena.grad
        if self.comm is not None:
            _lib.check(self._lib.reed_comm_allreduce_avg(self.comm, g.data_ptr() + 4 * b, e - b, self._stream()),
                       "comm_allreduce_avg")
            return
        self._tstream.wait_stream(torch.cuda.current_stream())   # the bucket's gradients are written
        with torch.cuda.stream(self._tstream):
            dist.all_reduce(g[b:e], op=dist.ReduceOp.AVG)           # stream-ordered: no host wait

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
