"""SiT (DiT-style transformer with adaLN-Zero) — drop-in for the reference's image/models/sit.py.

Same constructor signature (sit.py:161-181), same `forward(x, t, y, inference=True) -> (x, zs)` (:271-311),
same state_dict key names and shapes, same `SiT_models` registry (:373-415) — but the module
holds no torch layers: parameters are views into one flat arena (reed_amd/arena.py) and forward/backward run
hand-written HIP kernels through reed_amd/engine.py. There is no CPU path: forward raises off-GPU.

Deliberate deviations from reference defects: S presets set decoder_hidden_size=hidden_size
(§9-1: the reference's SiT-S/* cannot run a forward); z_dims=[] is allowed (§9-4, "alignment off").
"""
import math

import numpy as np
import torch
import torch.nn as nn

from ..arena import ArenaLayout, ParamArena


class _Holder(nn.Module):
    """Parameter container that reproduces the reference's module tree (for state_dict key names only)."""


def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    """sit.py:349-366 — float64 numpy: [sin(pos*w) | cos(pos*w)], w_k = 10000^(-k/(dim/2))."""
    assert embed_dim % 2 == 0
    omega = np.arange(embed_dim // 2, dtype=np.float64)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False, extra_tokens=0):
    """sit.py:319-346 — note meshgrid(w, h): the first half of the channels encodes the COLUMN index."""
    grid_h = np.arange(grid_size, dtype=np.float32)
    grid_w = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(grid_w, grid_h), axis=0).reshape([2, 1, grid_size, grid_size])
    emb = np.concatenate([get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0]),
                          get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])], axis=1)
    if cls_token and extra_tokens > 0:
        emb = np.concatenate([np.zeros([extra_tokens, embed_dim]), emb], axis=0)
    return emb


class SiT(nn.Module):
    def __init__(self, path_type="edm", input_size=32, patch_size=2, in_channels=4, hidden_size=1152,
                 decoder_hidden_size=768, encoder_depth=8, encoder_depth_text=None, depth=28, num_heads=16,
                 mlp_ratio=4.0, class_dropout_prob=0.1, num_classes=1000, use_cfg=False, z_dims=[768],
                 z_types=["i"], projector_dim=2048, **block_kwargs):
        super().__init__()
        if decoder_hidden_size != hidden_size:
            raise ValueError(f"decoder_hidden_size ({decoder_hidden_size}) must equal hidden_size ({hidden_size}): "
                             "FinalLayer consumes the last block's tokens (reference sit.py:215,308)")
        unknown = set(block_kwargs) - {"fused_attn", "qk_norm"}
        if unknown:
            raise TypeError(f"unexpected block kwargs {sorted(unknown)}")
        self.path_type = path_type
        self.in_channels = in_channels
        self.out_channels = in_channels
        self.patch_size = patch_size
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.depth = depth
        self.num_heads = num_heads
        self.mlp_ratio = mlp_ratio
        self.use_cfg = use_cfg
        self.num_classes = num_classes
        self.class_dropout_prob = class_dropout_prob
        self.z_dims = list(z_dims)
        self.z_types = list(z_types)
        self.encoder_depth = encoder_depth
        self.encoder_depth_text = encoder_depth_text
        self.projector_dim = projector_dim
        self.qk_norm = bool(block_kwargs.get("qk_norm", False))
        self.fused_attn = bool(block_kwargs.get("fused_attn", True))  # both settings run the same fused HIP kernel
        if len(self.z_types) < len(self.z_dims):
            raise ValueError("z_types must name the kind ('i' | 't') of every entry of z_dims")
        self.z_types = self.z_types[:len(self.z_dims)]  # the reference zips projectors with z_types (sit.py:292)
        split = encoder_depth_text is not None and encoder_depth_text != encoder_depth
        if split and (self.z_types.count("i") != 1 or self.z_types.count("t") != 1):
            raise ValueError("encoder_depth_text != encoder_depth needs exactly one image ('i') and one text ('t') "
                             "projector (the reference keeps only the last of each, sit.py:291-304)")
        if hidden_size % num_heads or hidden_size // num_heads not in (64, 72):
            raise ValueError(f"head_dim {hidden_size / num_heads} unsupported: the HIP attention kernels cover 64 and 72")
        self.num_patches = (input_size // patch_size) ** 2

        shapes = self._param_shapes()
        self._layout = ArenaLayout(shapes, depth, len(self.z_dims))
        self._arena = ParamArena(self._layout, "cpu")
        self._engine = None
        self._build_tree(shapes)
        self.initialize_weights()
        self.force_drop_mask = None  # tests: bool [N] replacing LabelEmbedder's torch.rand draw
        # operand type of the kernels = which build of the library evaluates this model (reed_amd/_lib.py): "bf16" (the reference
        # under accelerate bf16), "fp16" (IEEE half: sampling at the mantissa of the reference's TF32, and --mixed-precision fp16
        # training with the on-device GradScaler; csrc/common.hpp REED_FP16) or "fp32" (fp32 operands and activations on the
        # fp32 MFMA: --mixed-precision no / generate.py --no-tf32; REED_FP32)
        self.precision = "bf16"

    # ------------------------------------------------------------------ structure
    def _param_shapes(self):
        from collections import OrderedDict
        D, p, C = self.hidden_size, self.patch_size, self.in_channels
        Hm = int(D * self.mlp_ratio)
        sh = OrderedDict()
        sh["x_embedder.proj.weight"] = (D, C, p, p)
        sh["x_embedder.proj.bias"] = (D,)
        sh["t_embedder.mlp.0.weight"] = (D, 256)
        sh["t_embedder.mlp.0.bias"] = (D,)
        sh["t_embedder.mlp.2.weight"] = (D, D)
        sh["t_embedder.mlp.2.bias"] = (D,)
        sh["y_embedder.embedding_table.weight"] = (self.num_classes + (1 if self.class_dropout_prob > 0 else 0), D)
        sh["pos_embed"] = (1, self.num_patches, D)
        for i in range(self.depth):
            b = f"blocks.{i}."
            sh[b + "attn.qkv.weight"] = (3 * D, D)
            sh[b + "attn.qkv.bias"] = (3 * D,)
            if self.qk_norm:  # timm Attention: q_norm / k_norm = LayerNorm(head_dim) between qkv and proj
                hd = D // self.num_heads
                for n in ("q_norm", "k_norm"):
                    sh[b + f"attn.{n}.weight"] = (hd,)
                    sh[b + f"attn.{n}.bias"] = (hd,)
            sh[b + "attn.proj.weight"] = (D, D)
            sh[b + "attn.proj.bias"] = (D,)
            sh[b + "mlp.fc1.weight"] = (Hm, D)
            sh[b + "mlp.fc1.bias"] = (Hm,)
            sh[b + "mlp.fc2.weight"] = (D, Hm)
            sh[b + "mlp.fc2.bias"] = (D,)
            sh[b + "adaLN_modulation.1.weight"] = (6 * D, D)
            sh[b + "adaLN_modulation.1.bias"] = (6 * D,)
        P = self.projector_dim
        for j, z in enumerate(self.z_dims):
            b = f"projectors.{j}."
            sh[b + "0.weight"], sh[b + "0.bias"] = (P, D), (P,)
            sh[b + "2.weight"], sh[b + "2.bias"] = (P, P), (P,)
            sh[b + "4.weight"], sh[b + "4.bias"] = (z, P), (z,)
        sh["final_layer.linear.weight"] = (p * p * C, D)
        sh["final_layer.linear.bias"] = (p * p * C,)
        sh["final_layer.adaLN_modulation.1.weight"] = (2 * D, D)
        sh["final_layer.adaLN_modulation.1.bias"] = (2 * D,)
        return sh

    def _build_tree(self, shapes):
        """Register parameters under the reference's dotted names (module tree of empty holders)."""
        self._pnames = list(shapes)
        for name in shapes:
            parts = name.split(".")
            mod = self
            for part in parts[:-1]:
                if isinstance(mod, nn.ModuleList):
                    while len(mod) <= int(part):
                        mod.append(_Holder())
                    mod = mod[int(part)]
                else:
                    if part not in mod._modules:
                        is_list = mod is self and part in ("blocks", "projectors")
                        mod.add_module(part, nn.ModuleList() if is_list else _Holder())
                    mod = mod._modules[part]
            mod.register_parameter(parts[-1], nn.Parameter(self._arena.view(self._arena.master, name),
                                                           requires_grad=(name != "pos_embed")))
        if "projectors" not in self._modules:
            self.add_module("projectors", nn.ModuleList())

    def _rebind(self):
        """Point every Parameter at its arena view (after the arena moved device/dtype)."""
        sd = dict(self.named_parameters())
        for name in self._pnames:
            sd[name].data = self._arena.view(self._arena.master, name)
            sd[name].grad = None
        self._engine = None

    def _apply(self, fn, recurse=True):
        self._arena.wait_all()
        new_master = fn(self._arena.master)
        if new_master.dtype != torch.float32:
            raise TypeError("reed_amd.SiT keeps fp32 master weights (bf16 compute copies are internal), as the "
                            "reference does under accelerate mixed precision")
        if new_master is not self._arena.master:
            arena = ParamArena(self._layout, new_master.device)
            arena.master = new_master
            self._arena = arena
            self._rebind()
        return self

    def __deepcopy__(self, memo):
        import copy
        self._arena.wait_all()
        eng, self._engine = self._engine, None
        try:
            new = self.__class__.__new__(self.__class__)
            memo[id(self)] = new
            for k, v in self.__dict__.items():
                new.__dict__[k] = copy.deepcopy(v, memo)
        finally:
            self._engine = eng
        req = {n: p.requires_grad for n, p in self.named_parameters()}
        new._rebind()
        for n, p in new.named_parameters():
            p.requires_grad_(req[n])
        return new

    # ------------------------------------------------------------------ init
    @torch.no_grad()
    def initialize_weights(self):
        """Reference init scheme (sit.py:217-254). On the CPU generator the draws are consumed in the reference's
        order — including the default nn.Linear/Conv2d/Embedding inits that it immediately overwrites — so
        `torch.manual_seed(s); SiT(...)` yields the reference's initial weights for the same seed."""
        D, p, C = self.hidden_size, self.patch_size, self.in_channels
        sd = dict(self.named_parameters())
        shapes = self._param_shapes()

        def burn_linear(name):  # nn.Linear.reset_parameters: kaiming_uniform_(w), uniform_(b)
            torch.empty(int(np.prod(shapes[name + ".weight"]))).uniform_()
            torch.empty(int(np.prod(shapes[name + ".bias"]))).uniform_()

        burn_linear("x_embedder.proj")
        burn_linear("t_embedder.mlp.0")
        burn_linear("t_embedder.mlp.2")
        torch.empty(int(np.prod(shapes["y_embedder.embedding_table.weight"]))).normal_()
        lin_order = ["t_embedder.mlp.0", "t_embedder.mlp.2"]
        for i in range(self.depth):
            b = f"blocks.{i}."
            for n in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2", "adaLN_modulation.1"):
                burn_linear(b + n)
                lin_order.append(b + n)
        for j in range(len(self.z_dims)):
            for n in ("0", "2", "4"):
                burn_linear(f"projectors.{j}.{n}")
                lin_order.append(f"projectors.{j}.{n}")
        burn_linear("final_layer.linear")
        burn_linear("final_layer.adaLN_modulation.1")
        lin_order += ["final_layer.linear", "final_layer.adaLN_modulation.1"]
        # self.apply(_basic_init): xavier_uniform_ on every nn.Linear weight, zero bias
        for n in lin_order:
            w = torch.empty(shapes[n + ".weight"])
            nn.init.xavier_uniform_(w)
            sd[n + ".weight"].copy_(w)
            sd[n + ".bias"].zero_()
        pe = get_2d_sincos_pos_embed(D, int(self.num_patches ** 0.5))
        sd["pos_embed"].copy_(torch.from_numpy(pe).float().unsqueeze(0))
        w = torch.empty(D, C * p * p)
        nn.init.xavier_uniform_(w)
        sd["x_embedder.proj.weight"].copy_(w.view(D, C, p, p))
        sd["x_embedder.proj.bias"].zero_()
        for n in ("y_embedder.embedding_table.weight", "t_embedder.mlp.0.weight", "t_embedder.mlp.2.weight"):
            w = torch.empty(shapes[n])
            nn.init.normal_(w, std=0.02)
            sd[n].copy_(w)
        for i in range(self.depth):
            sd[f"blocks.{i}.adaLN_modulation.1.weight"].zero_()
            sd[f"blocks.{i}.adaLN_modulation.1.bias"].zero_()
            if self.qk_norm:  # nn.LayerNorm default init (no RNG): weight 1, bias 0
                for n in ("q_norm", "k_norm"):
                    sd[f"blocks.{i}.attn.{n}.weight"].fill_(1.0)
                    sd[f"blocks.{i}.attn.{n}.bias"].zero_()
        for n in ("final_layer.adaLN_modulation.1.weight", "final_layer.adaLN_modulation.1.bias",
                  "final_layer.linear.weight", "final_layer.linear.bias"):
            sd[n].zero_()

    # ------------------------------------------------------------------ compute
    def unpatchify(self, x, patch_size=None):
        """(N, T, p*p*C) -> (N, C, H, W), channel order (p_row, p_col, c) (sit.py:256-269). Pure index permutation;
        the HIP final-layer kernel writes this layout directly, this method exists for API parity."""
        c = self.out_channels
        p = self.patch_size if patch_size is None else patch_size
        h = w = int(x.shape[1] ** 0.5)
        assert h * w == x.shape[1]
        x = x.reshape(x.shape[0], h, w, p, p, c).permute(0, 5, 1, 3, 2, 4)
        return x.reshape(x.shape[0], c, h * p, w * p)

    def state_dict(self, *args, **kwargs):
        self._arena.wait_all()   # an overlapped optimiser step (reed_amd/optim.py) may still be writing the arena
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._arena.wait_all()
        return super().load_state_dict(*args, **kwargs)

    def engine(self):
        if self._engine is None:
            from ..engine import Engine
            self._engine = Engine(self)
        return self._engine

    def forward(self, x, t, y, inference=True):
        """x (N,C,H,W) f32, t (N,) f32 in [0,1], y (N,) int64 -> (velocity (N,C,H,W) f32, zs list | None).
        zs entries are bf16 ([N,T,z] for 'i', [N,z] for 't'), as the reference's raw autocast outputs."""
        from ..engine import sit_apply
        return sit_apply(self, x, t, y, inference)


# ---------------------------------------------------------------------------------------------
# registry (sit.py:373-415)
def _mk(depth, hidden, heads, patch):
    def f(**kw):
        kw.setdefault("decoder_hidden_size", hidden)
        return SiT(depth=depth, hidden_size=hidden, patch_size=patch, num_heads=heads, **kw)
    return f


SiT_models = {}
for _n, (_d, _h, _nh) in {"XL": (28, 1152, 16), "L": (24, 1024, 16), "B": (12, 768, 12), "S": (12, 384, 6)}.items():
    for _p in (2, 4, 8):
        SiT_models[f"SiT-{_n}/{_p}"] = _mk(_d, _h, _nh, _p)
