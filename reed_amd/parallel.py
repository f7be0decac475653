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
        # adaLN weight gradients by factor all-gather instead of all-reduce (see gather()): power-of-two worlds only
        # (the 1 / world average is folded into the bf16 factor, exact for powers of two); REED_ADA_GATHER=0 turns it off
        self.ada_gather = os.environ.get("REED_ADA_GATHER", "1") != "0" and (self.world & (self.world - 1)) == 0
        self._gbuf = {}
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
            # AVG and all_gather_into_tensor on device tensors are RCCL features; on any other backend (gloo: the
            # two-ranks-on-one-GPU test of tests/test_cli_gpu.py) the same results come from SUM + scale / a SUM over a
            # zero-padded buffer
            self._rccl = dist.get_backend() == "nccl"
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
            self._t_allreduce_avg(g[b:e])                          # stream-ordered: no host wait

    def _t_allreduce_avg(self, t):
        if self._rccl:
            dist.all_reduce(t, op=dist.ReduceOp.AVG)
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
        g = self.model._arena.grad
        if self.comm is not None:
            _lib.check(self._lib.reed_comm_allreduce_avg(self.comm, g.data_ptr() + 4 * b, e - b, self._stream()),
                       "comm_allreduce_avg")
            return
        self._tstream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._tstream):
            self._t_allreduce_avg(g[b:e])

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
