"""Flat parameter arena for the SiT model.

MI355X-first memory layout: every parameter lives in ONE contiguous fp32 master buffer; gradients, both Adam
moments, the EMA copy and the bf16 compute shadow are separate arenas with the SAME offsets. Consequences:
  * the optimiser side of the step (grad-norm, clip, AdamW, EMA, bf16 re-cast) is one pass over flat memory
    instead of the reference's 298-tensor foreach loops (image/train.py:94-105,402-412);
  * gradient all-reduce buckets are plain [begin, end) ranges, in backward completion order;
  * all 28 blocks' adaLN linears + the final layer's are adjacent rows of one [N_all, D] matrix, so the whole
    model's modulation is ONE GEMM on silu(c) (the reference runs 29 skinny GEMMs, sit.py:126-129,146-150).
nn.Parameters of reed_amd.models.sit.SiT are views into the master arena, so state_dict()/load_state_dict()
keep the reference's key names and shapes.
"""
from collections import OrderedDict

import numpy as np
import torch

ALIGN = 8  # elements; keeps every segment 16-byte aligned in the bf16 shadow and 32-byte in fp32


def _align(n, a=ALIGN):
    return (n + a - 1) // a * a


class ArenaLayout:
    """Segment table: name -> (offset, shape). Order = storage order (see module docstring)."""

    def __init__(self, shapes, depth, n_proj):
        """shapes: OrderedDict name->shape in state_dict order."""
        order = []
        ada_w = [f"blocks.{i}.adaLN_modulation.1.weight" for i in range(depth)] + ["final_layer.adaLN_modulation.1.weight"]
        ada_b = [n.replace("weight", "bias") for n in ada_w]
        order += ada_w + ada_b
        rest = [n for n in shapes if n not in set(order) and n != "pos_embed"]
        emb = [n for n in rest if n.split(".")[0] in ("x_embedder", "t_embedder", "y_embedder")]
        blocks = [n for n in rest if n.startswith("blocks.")]
        proj = [n for n in rest if n.startswith("projectors.")]
        fin = [n for n in rest if n.startswith("final_layer.")]
        order += emb + blocks + proj + fin
        assert len(order) == len(shapes) - 1, "unclassified parameter"
        self.seg = OrderedDict()
        off = 0
        for n in order:
            shp = tuple(shapes[n])
            numel = int(np.prod(shp))
            if n in ada_w or n in ada_b:
                assert numel % ALIGN == 0, "hidden size must be a multiple of 8"
            self.seg[n] = (off, shp)
            off += _align(numel)
        self.n_train = _align(off, 64)
        self.seg["pos_embed"] = (self.n_train, tuple(shapes["pos_embed"]))
        self.n_total = _align(self.n_train + int(np.prod(shapes["pos_embed"])), 64)
        self.ada_w_off = self.seg[ada_w[0]][0]
        self.ada_b_off = self.seg[ada_b[0]][0]
        self.ada_rows = sum(shapes[n][0] for n in ada_w)
        self.depth = depth

    def off(self, name):
        return self.seg[name][0]

    def numel(self, name):
        return int(np.prod(self.seg[name][1]))

    def range_of(self, prefix):
        """[begin, end) element range covering every segment whose name starts with prefix (must be adjacent)."""
        names = [n for n in self.seg if n.startswith(prefix)]
        b = min(self.seg[n][0] for n in names)
        e = max(self.seg[n][0] + _align(self.numel(n)) for n in names)
        return b, e

    ADA_HEAD_BLOCKS = 4   # adaLN rows of this many leading blocks are updated first (see update_chunks)

    def update_chunks(self, tap_blocks=()):
        """[(name, begin, end)] covering [0, n_total) exactly once, in the order the NEXT forward first touches the
        weights. The fused optimiser walks these on its own stream and the forward waits per chunk, so the HBM-bound
        update of later chunks runs under the MFMA-bound forward of earlier blocks. Order:
          embed        adaLN biases + embedders (small; the conditioning path reads them first)
          ada_head     adaLN weight rows of blocks 0..ADA_HEAD_BLOCKS-1  (the forward computes their modulation first)
          block0..     blocks in forward order, with `ada_tail` (the other 2/3 of the adaLN matrix — a third of all
                       parameters) after the head blocks and the projectors right after the earliest tap block
          final        final layer + pos_embed (frozen; only the EMA pass touches it)."""
        bk = dict(self.buckets())
        starts = sorted((r[0], n) for n, r in bk.items())
        ends = {}
        for (b0, n), nxt in zip(starts, starts[1:] + [(self.n_total, None)]):
            ends[n] = (b0, nxt[0])
        _, e0 = ends["embed"]
        kh = min(self.ADA_HEAD_BLOCKS, self.depth)
        per_block = (self.seg["blocks.0.adaLN_modulation.1.weight"][1][0] *
                     self.seg["blocks.0.adaLN_modulation.1.weight"][1][1])
        split = self.ada_w_off + kh * per_block
        assert self.ada_w_off == 0 and split <= self.ada_b_off
        order = [("embed", self.ada_b_off, e0), ("ada_head", 0, split)]
        first_tap = min(tap_blocks) if tap_blocks else None   # 1-based depth: projectors read the output of block first_tap-1
        for i in range(self.depth):
            if i == kh:
                order.append(("ada_tail", split, self.ada_b_off))
            if first_tap is not None and i == first_tap and "projectors" in ends:
                order.append(("projectors",) + ends["projectors"])
            order.append((f"block{i}",) + ends[f"block{i}"])
        if kh >= self.depth:
            order.append(("ada_tail", split, self.ada_b_off))
        if "projectors" in ends and not any(n == "projectors" for n, _, _ in order):
            order.append(("projectors",) + ends["projectors"])
        order.append(("final",) + ends["final"])
        order = [c for c in order if c[2] > c[1]]
        assert sum(e - b for _, b, e in order) == self.n_total and all(b % 4 == 0 and e % 4 == 0 for _, b, e in order)
        return order

    def ada_row_range(self, i):
        """[begin, end) of the adaLN weight rows of block i (i == depth: the final layer's 2D rows) in the arena."""
        name = f"blocks.{i}.adaLN_modulation.1.weight" if i < self.depth else "final_layer.adaLN_modulation.1.weight"
        off, shp = self.seg[name]
        return off, off + _align(int(np.prod(shp)))

    def buckets(self):
        """Gradient all-reduce buckets in the order backward finishes them: final layer, blocks L-1..0, each followed
        by ITS rows of the adaLN matrix (`ada{i}`; the engine computes those rows' gradient right after the block when
        a reducer is attached, so that the adaLN third of the payload — 0.9 GB for XL/2 — rides under backward instead
        of being reduced after it), the projectors (fired when the last tap's backward is done), and at the very end
        the small rest: adaLN biases + embedders."""
        out = [("final", self.range_of("final_layer.linear")), (f"ada{self.depth}", self.ada_row_range(self.depth))]
        for i in reversed(range(self.depth)):
            out.append((f"block{i}", self.range_of(f"blocks.{i}.attn.qkv")[0:1] + (self.range_of(f"blocks.{i}.mlp.fc2")[1],)))
            out.append((f"ada{i}", self.ada_row_range(i)))
        names = [n for n in self.seg if n.startswith("projectors.")]
        if names:
            out.append(("projectors", self.range_of("projectors.")))
        out.append(("embed", (self.ada_b_off, self.range_of("y_embedder")[1])))
        return out


class ParamArena:
    def __init__(self, layout, device, dtype=torch.float32):
        self.layout = layout
        self.device = torch.device(device)
        self.master = torch.zeros(layout.n_total, dtype=dtype, device=device)
        self.grad = None
        self.shadow = None
        self.shadow_version = -1
        # name -> torch.cuda.Event: parameter ranges a fused optimiser step is still rewriting on its own stream
        # (reed_amd/optim.py, overlap=True). Whoever reads weights waits for the range it needs (Engine.forward,
        # per block) or for everything (state_dict, EMA forward): wait() / wait_all().
        self.pending = OrderedDict()
        # Round 6: a TRANSPOSED 16-bit copy of the blocks' linear weights (W^T [k_in, n_out] per weight), so that the input
        # gradients run as NT GEMMs — both operands k-contiguous: one ds_read_b128 per fragment where the k-strided weight takes
        # two transposing reads — 3.6-5 % faster at b = 256, the same bits (profiles/r6_dgrad_nt_on_transposed_weights.txt).
        # Built on first use (Engine.backward), kept fresh by the fused optimiser on its side stream (block by block behind the
        # block's update chunk: it is first read a whole forward later), rebuilt here whenever it is older than the shadow.
        self.shadow_t = None
        self.t_seg = None          # name -> (offset in shadow_t, n_out, k_in)
        self.shadow_gen = 0        # bumped whenever the shadow changes (a cast of the master; an optimiser step)
        self.shadow_t_gen = -1
        self.pending_t = None      # event behind the optimiser's transposes

    def wait(self, name):
        """Order the current stream after the optimiser's update of bucket `name` (no-op when none is in flight)."""
        ev = self.pending.pop(name, None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def wait_all(self):
        while self.pending:
            _, ev = self.pending.popitem(last=False)
            torch.cuda.current_stream().wait_event(ev)

    def view(self, buf, name):
        off, shp = self.layout.seg[name]
        return buf[off:off + int(np.prod(shp))].view(shp)

    def ensure_grad(self):
        if self.grad is None:
            self.grad = torch.zeros(self.layout.n_train, dtype=torch.float32, device=self.device)
        return self.grad

    def ensure_shadow(self, precision="bf16"):
        """16-bit copy of the master weights (GEMM operands; bf16, or IEEE half for a model sampling at
        precision="fp16"). Refreshed when the master changed through torch (version counter) — the fused optimiser
        rewrites the bf16 one itself and calls mark_shadow_fresh()."""
        from . import ops
        dt = ops.half_dtype(precision)
        if self.shadow is None or self.shadow.dtype != dt:
            self.shadow = torch.empty(self.layout.n_total, dtype=dt, device=self.device)
            self.shadow_version = -1
        if self.shadow_version != self.master._version:
            self.wait_all()   # an overlapped optimiser step may still be rewriting the master
            prev = ops.use(precision)
            try:
                ops.cast_bf16(self.master, self.shadow, self.layout.n_total)
            finally:
                ops.use(prev)
            self.shadow_version = self.master._version
            self.shadow_gen += 1
        return self.shadow

    def mark_shadow_fresh(self):
        self.shadow_version = self.master._version
        self.shadow_gen += 1

    def t_names(self):
        """The weights that have a transposed copy: the four linears of every block."""
        return [f"blocks.{i}.{w}.weight" for i in range(self.layout.depth) for w in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2")]

    def transpose_block(self, i):
        """W^T of block i's four linears from the shadow, on the current stream (the caller orders it behind the shadow's update)."""
        from . import ops
        es = self.shadow.element_size()
        for w in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2"):
            name = f"blocks.{i}.{w}.weight"
            toff, n_out, k_in = self.t_seg[name]
            ops.transpose_bf16(self.shadow.data_ptr() + es * self.layout.off(name), self.shadow_t.data_ptr() + es * toff, n_out, k_in)

    def ensure_shadow_t(self, precision="bf16"):
        """The transposed copies, as fresh as the shadow (which the caller has ensured): allocate on first use; rebuild everything
        on the current stream if an update happened that did not refresh them (a cast of the master, a non-overlapped optimiser
        step); otherwise just order the current stream behind the optimiser's own transposes."""
        from . import ops
        if self.shadow.element_size() != 2:
            return None
        if self.shadow_t is None or self.shadow_t.dtype != self.shadow.dtype:
            self.t_seg, off = {}, 0
            for name in self.t_names():
                n_out, k_in = self.layout.seg[name][1]
                self.t_seg[name] = (off, int(n_out), int(k_in))
                off += int(n_out) * int(k_in)
            self.shadow_t = torch.empty(off, dtype=self.shadow.dtype, device=self.device)
            self.shadow_t_gen = -1
        if self.shadow_t_gen != self.shadow_gen:
            self.wait_all()
            self.pending_t = None
            prev = ops.use(precision)
            try:
                for i in range(self.layout.depth):
                    self.transpose_block(i)
            finally:
                ops.use(prev)
            self.shadow_t_gen = self.shadow_gen
        elif self.pending_t is not None:
            torch.cuda.current_stream().wait_event(self.pending_t)
            self.pending_t = None
        return self.shadow_t
