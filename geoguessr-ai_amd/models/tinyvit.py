"""Drop-in for the reference's ``models/tinyvit.py`` (``TinyViTAdapter``, :17-150): same constructor, same
``.config`` / ``.vision_model.encoder.layers`` shims, same ``freeze_*`` / ``train`` semantics, same forward return
(``pooler_output`` (B,C), ``last_hidden_state`` (B,1,C)), same ``backbone.*`` state-dict keys (timm names,
SURVEY.md App. A.5) -- but the encoder itself is the HIP runtime in ``csrc/tinyvit.hip`` (one C call for the
whole forward, one for the whole backward) instead of ``timm.create_model``.

Parameters are views into ONE flat fp32 device buffer (gradients likewise), which is what the RCCL gradient
all-reduce and the fused AdamW kernel operate on.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import warnings
from types import SimpleNamespace
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import _lib as L
from .flat import FlatStore, _Tree

# timm variants; drop_path_rate is timm's per-variant default
VARIANTS = {
    "tiny_vit_5m_224": dict(embed_dims=(64, 128, 160, 320), num_heads=(2, 4, 5, 10), window_sizes=(7, 7, 14, 7),
                            img_size=224, drop_path_rate=0.0),
    "tiny_vit_11m_224": dict(embed_dims=(64, 128, 256, 448), num_heads=(2, 4, 8, 14), window_sizes=(7, 7, 14, 7),
                             img_size=224, drop_path_rate=0.1),
    "tiny_vit_21m_224": dict(embed_dims=(96, 192, 384, 576), num_heads=(3, 6, 12, 18), window_sizes=(7, 7, 14, 7),
                             img_size=224, drop_path_rate=0.2),
    "tiny_vit_21m_384": dict(embed_dims=(96, 192, 384, 576), num_heads=(3, 6, 12, 18), window_sizes=(12, 12, 24, 12),
                             img_size=384, drop_path_rate=0.1),
    "tiny_vit_21m_512": dict(embed_dims=(96, 192, 384, 576), num_heads=(3, 6, 12, 18), window_sizes=(16, 16, 32, 16),
                             img_size=512, drop_path_rate=0.1),
}
DEPTHS = (2, 2, 6, 2)


PRECISIONS = {"bf16": 0, "bfloat16": 0, "fp32": 1, "f32": 1, "float32": 1, "fp32_split": 3}      # fp32_split: f32 storage, the transformer blocks' Linears as f32-accurate split-bf16 products


def default_precision() -> str:
    """``GG_PRECISION`` = "fp32" (default: the reference's own arithmetic -- f32 activations, f32 MFMA, fp32-accurate GELU)
    or "bf16" (bf16 activations / MFMA operands, fp32 accumulation and master weights: 3.4x the throughput, bf16-level parity)."""
    p = os.environ.get("GG_PRECISION", "fp32").lower()
    if p not in PRECISIONS:
        raise ValueError(f"GG_PRECISION='{p}' (known: bf16, fp32)")
    return p


def make_cfg(model_name: str, **overrides) -> tuple:
    base = model_name.split(".")[0]
    if base not in VARIANTS:
        raise ValueError(f"unknown TinyViT variant '{model_name}' (known: {sorted(VARIANTS)})")
    v = dict(VARIANTS[base])
    v.update(overrides)
    depths = tuple(v.get("depths", DEPTHS))
    c = L.TinyVitCfg()
    c.img_size, c.in_chans = v["img_size"], 3
    c.embed_dims = (C.c_int * 4)(*v["embed_dims"])
    c.depths = (C.c_int * 4)(*depths)
    c.num_heads = (C.c_int * 4)(*v["num_heads"])
    c.window_sizes = (C.c_int * 4)(*v["window_sizes"])
    c.mlp_ratio, c.mbconv_expand_ratio = 4.0, 4.0
    c.bn_eps, c.ln_eps, c.bn_momentum = 1e-5, 1e-5, 0.1
    prec = v.get("precision") or default_precision()
    if prec not in PRECISIONS:
        raise ValueError(f"precision='{prec}' (known: bf16, fp32)")
    c.act_dtype = PRECISIONS[prec]
    c.features_only = int(bool(v.get("features_only", False)))
    return c, v, depths


def _tensor_table(cfg: L.TinyVitCfg):
    lib = L.lib()
    n = lib.gg_tinyvit_num_tensors(C.byref(cfg))
    if n < 0:
        raise L.GgError(lib.gg_last_error().decode())
    out = []
    name = C.create_string_buffer(256)
    off, numel, ndim, kind = C.c_int64(), C.c_int64(), C.c_int(), C.c_int()
    shape = (C.c_int64 * 4)()
    for i in range(n):
        L.check(lib.gg_tinyvit_tensor_info(C.byref(cfg), i, name, 256, C.byref(off), C.byref(numel), C.byref(ndim), shape,
                                           C.byref(kind)), "gg_tinyvit_tensor_info")
        out.append(dict(name=name.value.decode(), offset=off.value, numel=numel.value,
                        shape=tuple(shape[j] for j in range(ndim.value)), kind=kind.value, index=i))
    return out


def _init_tensor(name: str, shape, g: torch.Generator) -> torch.Tensor:
    """timm TinyVit init: Linear trunc_normal(.02)/0, LayerNorm 1/0, Conv2d torch default, BN 1/0 except the last
    BN of an MBConv (0), attention biases 0."""
    if name.endswith("conv.weight"):
        fan_in = shape[1] * shape[2] * shape[3]
        return (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
    if name.endswith("bn.weight"):
        return torch.zeros(shape) if (name.startswith("stages.0.") and ".conv3." in name) else torch.ones(shape)
    if name.endswith("norm.weight"):
        return torch.ones(shape)
    if name.endswith("attention_biases") or name.endswith(".bias"):
        return torch.zeros(shape)
    if name.endswith(".weight"):
        t = torch.empty(shape)
        nn.init.trunc_normal_(t, std=0.02, generator=g)
        return t
    raise AssertionError(name)


class TinyVitBackbone(FlatStore):
    """Owner of the flat parameter / buffer storage and of the HIP workspace (``self.backbone`` of the adapter)."""

    def __init__(self, model_name: str, seed: Optional[int] = None, **overrides):
        super().__init__()
        self.cfg, self.variant, self.depths = make_cfg(model_name, **overrides)
        self.model_name = model_name.split(".")[0]
        self.precision = "fp32" if self.cfg.act_dtype in (1, 3) else "bf16"          # storage / arithmetic class ("fp32_split" stores and checks like fp32)
        self.split = self.cfg.act_dtype == 3
        self.table = _tensor_table(self.cfg)
        lib = L.lib()
        self.num_features = int(self.variant["embed_dims"][-1])
        self.param_floats = lib.gg_tinyvit_param_floats(C.byref(self.cfg))
        self.buffer_floats = lib.gg_tinyvit_buffer_floats(C.byref(self.cfg))
        self.num_counters = lib.gg_tinyvit_num_counters(C.byref(self.cfg))
        self.num_drop_slots = lib.gg_tinyvit_num_drop_slots(C.byref(self.cfg))
        n_blocks = sum(self.depths)
        rates = torch.linspace(0, float(self.variant["drop_path_rate"]), n_blocks).tolist()
        slot_rates: List[float] = []
        for b in range(n_blocks):
            slot_rates += [rates[b]] if b < self.depths[0] else [rates[b], rates[b]]
        self.drop_rates = slot_rates
        g = torch.Generator().manual_seed(torch.initial_seed() if seed is None else seed)
        self._register_table(lambda name, shape: _init_tensor(name, shape, g))
        self._wcache = None
        self._wcache_version = -1
        self._ws: Dict[bool, torch.Tensor] = {}
        self._last = None
        self._gen = 0                    # generation of the training workspace contents (one per training forward)
        self._grad_ready_hook = None     # set by optim.AdamW.overlap_allreduce: fn(lo, hi) over flat gradient floats

    # -- HIP calls ------------------------------------------------------------------------------------------
    def _ensure_weights(self):
        lib = L.lib()
        if self._wcache is None:
            nbytes = lib.gg_tinyvit_wcache_bytes(C.byref(self.cfg))
            self._wcache = torch.zeros(nbytes, dtype=torch.uint8, device=self._flat.device)
            self._wcache_version = -1
            self._dirty_all = True
        ver = self._param_version()
        if self._wcache_version != ver:
            # full rebuild unless the only writers since the last sync were masked raw-pointer writers (the fused optimizer) AND torch's own
            # version counters did not move (no load_state_dict / copy_ / torch.optim step in between)
            only = None if (getattr(self, "_dirty_all", True) or getattr(self, "_synced_ver", None) != ver) else getattr(self, "_dirty_only", None)
            if only is None:
                L.check(lib.gg_tinyvit_refresh_weights(C.byref(self.cfg), L.ptr(self._flat, torch.float32, "params"),
                                                       L.ptr(self._wcache), L.stream()), "gg_tinyvit_refresh_weights")
            else:
                L.check(lib.gg_tinyvit_refresh_weights_masked(C.byref(self.cfg), L.ptr(self._flat, torch.float32, "params"),
                                                              L.ptr(self._wcache), only, L.stream()), "gg_tinyvit_refresh_weights_masked")
            self._wcache_version = self._synced_ver = ver
            self._dirty_all, self._dirty_only = False, None

    def _workspace(self, batch: int, training: bool, mask=None) -> torch.Tensor:
        need = L.lib().gg_tinyvit_workspace_bytes_masked(C.byref(self.cfg), batch, int(training), mask)
        if need < 0:
            raise L.GgError(L.lib().gg_last_error().decode())
        ws = self._ws.get(training)
        if ws is None or ws.numel() < need or ws.device != self._flat.device:
            self._ws[training] = None
            ws = torch.empty(need, dtype=torch.uint8, device=self._flat.device)
            self._ws[training] = ws
        return ws

    def make_drop_scales(self, batch: int, generator: Optional[torch.Generator] = None) -> Optional[torch.Tensor]:
        """timm DropPath (scale_by_keep): per-sample Bernoulli(1-p)/(1-p), one row per slot, drawn by ``gg_drop_path_scales`` (a counter-based
        generator on the device).  The seed comes from torch's RNG once per backbone (so ``torch.manual_seed`` makes runs repeatable), the counter
        advances with every call.  With ``generator`` the rows are a function of that generator's seed and of how many times that generator object has
        been passed here (its own state is neither read on the device nor advanced); these per-generator counts are not part of ``drop_path_state()``."""
        if max(self.drop_rates) <= 0:
            return None
        dev = self._flat.device
        rates = getattr(self, "_rates_dev", None)
        if rates is None or rates.device != dev:          # cached on the device: no host-to-device copy per step (graph-capturable)
            rates = self._rates_dev = torch.tensor(self.drop_rates, dtype=torch.float32, device=dev)
        if generator is not None:
            # host-side (seed, counter) from the generator's seed + a per-generator call count: no device round trip (a CUDA generator's randint +
            # .item() was a blocking sync per step), deterministic for a given generator seed and call order.  The count lives with the generator OBJECT
            # (torch generators can be neither weakly referenced nor given attributes: the entry holds a strong reference, so the object's id cannot be
            # reused by another generator while the entry lives; the table keeps the 16 most recently used generators) and restarts when the generator is
            # given another seed; re-seeding it with the SAME seed continues the count (initial_seed() cannot tell) -- use a fresh generator for that
            counts = self.__dict__.setdefault("_drop_gen_counts", {})
            seed = int(generator.initial_seed()) & (2 ** 62 - 1)
            ent = counts.pop(id(generator), None)
            if ent is None or ent[0] is not generator or ent[1] != seed:
                ent = [generator, seed, 0]
            counter = ent[2]
            ent[2] += 1
            counts[id(generator)] = ent                 # (re-inserted last: dicts keep insertion order, the oldest entry leaves first)
            while len(counts) > 16:
                counts.pop(next(iter(counts)))
        else:
            if getattr(self, "_drop_seed", None) is None:
                self._drop_seed, self._drop_counter = int(torch.randint(0, 2 ** 62, (1,)).item()), 0      # (CPU RNG, once per backbone)
            seed, counter = self._drop_seed, self._drop_counter
            self._drop_counter += 1
        out = torch.empty((self.num_drop_slots, batch), dtype=torch.float32, device=dev)
        L.check(L.lib().gg_drop_path_scales(L.ptr(rates), self.num_drop_slots, batch, seed, counter, L.ptr(out), L.stream()), "gg_drop_path_scales")
        return out

    # DropPath stream position: kept OUT of state_dict (its keys are the timm contract) and saved by checkpoint.save_checkpoint under its own key, so
    # a resumed run continues the mask sequence instead of replaying it from counter 0
    def drop_path_state(self):
        return dict(drop_seed=getattr(self, "_drop_seed", None), drop_counter=getattr(self, "_drop_counter", 0))

    def load_drop_path_state(self, state):
        if state and state.get("drop_seed") is not None:
            self._drop_seed, self._drop_counter = int(state["drop_seed"]), int(state.get("drop_counter", 0))

    def forward_hip(self, x: torch.Tensor, training: bool, drop_scales: Optional[torch.Tensor] = None) -> torch.Tensor:
        L.require_gpu()
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != self.cfg.img_size or x.shape[3] != self.cfg.img_size:
            raise L.GgError(f"TinyViT expects (B,3,{self.cfg.img_size},{self.cfg.img_size}) pixel_values, got {tuple(x.shape)}")
        if not self._flat.is_cuda:
            raise L.GgError("TinyViTAdapter parameters are on the CPU; call .to('cuda') -- there is no CPU fallback")
        x = x.to(device=self._flat.device, dtype=torch.float32).contiguous()
        B = x.shape[0]
        self._ensure_weights()
        mask = self.trainable_mask() if training else None      # the workspace keeps no activation that only a frozen weight's gradient would read
        ws = self._workspace(B, training, mask)
        out = torch.empty((B, self.num_features), dtype=torch.float32, device=x.device)
        if drop_scales is not None:
            assert drop_scales.shape == (self.num_drop_slots, B) and drop_scales.dtype == torch.float32
        self._fwd_mask = mask
        if training:
            self._train_mask = mask          # (an eval forward in between does not change what the training workspace was laid out for)
        L.check(L.lib().gg_tinyvit_forward(C.byref(self.cfg), B, int(training), L.ptr(self._flat), L.ptr(self._flat_buf),
                                           L.ptr(self._counters), L.ptr(self._wcache), L.ptr(x), L.ptr(drop_scales), L.ptr(ws),
                                           L.ptr(out), mask, L.stream()),
                "gg_tinyvit_forward")
        if training:
            self._counters += 1          # num_batches_tracked (int64 bookkeeping)
            self._flat_buf_dirty = True
            self._gen += 1
            self._last = (B, drop_scales, self._gen)
        return out

    def _stage_ranges(self):
        """Flat float range of each backward stage id: 3, 2, 1 (TinyVitStages), 0 (MBConv stage), -1 (patch_embed); the range of
        stage 3 also holds head.norm (final before the stage loop starts)."""
        if getattr(self, "_stage_rng", None) is None:
            first = {}
            for t in self.table:
                if t["kind"] != 0:
                    continue
                key = -1 if t["name"].startswith("patch_embed.") else (int(t["name"].split(".")[1]) if t["name"].startswith("stages.") else 4)
                first.setdefault(key, t["offset"])
            order = [-1, 0, 1, 2, 3]
            ends = {k: (first[order[i + 1]] if i + 1 < len(order) else self.param_floats) for i, k in enumerate(order)}
            self._stage_rng = {k: (first[k], ends[k]) for k in order}
        return self._stage_rng

    def backward_hip(self, d_out: torch.Tensor, gen: Optional[int] = None):
        if self._last is None:
            raise L.GgError("TinyViT backward without a training forward")
        B, drop, last_gen = self._last
        if gen is not None and gen != last_gen:
            raise L.GgError(f"TinyViT backward for training forward #{gen}, but the workspace now holds the activations of forward "
                            f"#{last_gen}: saved activations live in ONE workspace per backbone, so every training forward must be "
                            "followed by its backward before the next training forward")
        if d_out.shape[0] != B:
            raise L.GgError(f"TinyViT backward: gradient batch {d_out.shape[0]} != forward batch {B}")
        fg = self.attach_grads()
        ws = self._ws[True]
        mask = self.trainable_mask()
        if getattr(self, "_train_mask", None) is not None and mask != self._train_mask:
            # the workspace was laid out (and activations were dropped) for the forward's mask: backward must see the same one
            changed = [t["name"] for t, a, b in zip([t for t in self.table], mask, self._train_mask) if bool(a) != bool(b)]
            raise L.GgError("requires_grad changed between forward and backward for " + ", ".join(changed[:4]) +
                            " ...: the training forward laid out its workspace for the mask it saw (activations only a frozen weight's gradient "
                            "needs are not kept); run the forward again")
        hook = self._grad_ready_hook
        cb = L.STAGE_DONE_FN(0)
        failed = []
        if hook is not None:
            rng = self._stage_ranges()

            def _stage_done(stage, _user):
                # ctypes swallows an exception raised inside a callback (it only prints "Exception ignored ..."), and a rank that skips one
                # bucket would then launch its collectives in another order than its peers: keep the first failure, send nothing further
                # from inside this backward (allreduce_grads() sends whatever is left in one fixed order) and re-raise below
                if failed:
                    return
                try:
                    hook(*rng[stage])
                except BaseException as exc:      # noqa: BLE001 -- must not escape into the C caller
                    failed.append(exc)
            cb = L.STAGE_DONE_FN(_stage_done)
        L.check(L.lib().gg_tinyvit_backward(C.byref(self.cfg), B, L.ptr(self._flat), L.ptr(self._wcache), L.ptr(drop), L.ptr(ws),
                                            L.ptr(d_out.contiguous(), torch.float32, "d_out"), L.ptr(fg), mask, L.stream(), cb, None),
                "gg_tinyvit_backward")
        if failed:
            raise L.GgError(f"TinyViT backward: the gradient-ready hook failed ({type(failed[0]).__name__}: {failed[0]}); the gradients are "
                            "complete, but no further bucket was sent from inside this backward pass") from failed[0]

    def activation(self, name: str, batch: int) -> torch.Tensor:
        """Raw bytes of a saved activation of the last training forward (parity tests)."""
        off, nbytes = C.c_int64(), C.c_int64()
        L.check(L.lib().gg_tinyvit_activation_info_masked(C.byref(self.cfg), batch, name.encode(), getattr(self, "_train_mask", None), C.byref(off),
                                                          C.byref(nbytes)), "gg_tinyvit_activation_info")
        return self._ws[True][off.value:off.value + nbytes.value]

    def forward(self, x):
        need = torch.is_grad_enabled() and any(p.requires_grad for p in self._params.values())
        return _EncoderFn.apply(self, x, _anchor(self) if need else _anchor(self).detach())


def _anchor(bb: TinyVitBackbone) -> torch.Tensor:
    a = getattr(bb, "_anchor_t", None)
    if a is None or a.device != bb._flat.device:
        a = torch.zeros((), device=bb._flat.device, requires_grad=True)
        bb._anchor_t = a
    return a


class _EncoderFn(torch.autograd.Function):
    """Whole-encoder autograd node.  Parameter gradients are accumulated straight into the flat gradient buffer
    (``p.grad`` views), not returned through autograd; the zero-dim ``anchor`` input only keeps the node alive."""

    @staticmethod
    def forward(ctx, bb: TinyVitBackbone, x: torch.Tensor, anchor: torch.Tensor):
        need_grad = ctx.needs_input_grad[2]      # (grad mode is off inside Function.forward)
        training = bb.training
        drop = bb.make_drop_scales(x.shape[0]) if training else None
        out = bb.forward_hip(x, training, drop)
        ctx.bb = bb
        ctx.valid = training and need_grad
        ctx.gen = bb._gen
        return out

    @staticmethod
    def backward(ctx, d_out):
        if not ctx.valid:
            raise L.GgError("backward through a TinyViT forward that ran in eval mode (running-stat BatchNorm keeps no "
                            "activations); call .train() before the forward pass")
        ctx.bb.backward_hip(d_out, ctx.gen)
        return None, None, torch.zeros((), device=d_out.device)


class TinyViTAdapter(nn.Module):
    """See module docstring.  ``pretrained=True`` cannot download (no network): weights are looked up in
    ``$GG_PRETRAINED_DIR/<model_name>.pt`` (a timm state dict) and otherwise a warning is issued and the timm
    initialisation is used."""

    def __init__(self, model_name: str = "tiny_vit_21m_512.dist_in22k_ft_in1k", pretrained: bool = True,
                 global_pool: str = "avg", features_only: bool = False, **overrides):
        """``overrides`` (not in the reference): ``precision="bf16"|"fp32"`` (default ``$GG_PRECISION`` or fp32), ``seed``,
        ``drop_path_rate``, ``img_size`` ... (timm ``create_model`` kwargs)."""
        super().__init__()
        if global_pool != "avg":
            raise NotImplementedError("only global_pool='avg' (the reference's setting) is built")
        # features_only=True (models/tinyvit.py:38-46): timm returns the stage feature maps and the adapter's forward pools the last
        # one (:139-143) -- i.e. the encoder output WITHOUT head.norm; the runtime writes exactly that vector
        self.features_only = features_only
        self.backbone = TinyVitBackbone(model_name, features_only=features_only, **overrides)
        hidden = self.backbone.num_features
        self.config = SimpleNamespace(hidden_size=hidden, hidden_sizes=[hidden], _name_or_path=model_name)
        stages = self.backbone._modules["stages"]
        self.vision_model = SimpleNamespace(encoder=SimpleNamespace(layers=list(stages)))
        self._fully_frozen = False
        if pretrained:
            d = os.environ.get("GG_PRETRAINED_DIR")
            path = os.path.join(d, model_name + ".pt") if d else None
            if path and os.path.exists(path):
                self.backbone.load_state_dict(torch.load(path, map_location="cpu"), strict=False)
            else:
                warnings.warn(f"pretrained weights for {model_name} not available offline (set GG_PRETRAINED_DIR); "
                              "using the timm initialisation")

    def build_transform(self):
        raise RuntimeError("timm data transforms not available in this environment.")   # models/tinyvit.py:83-86

    def freeze_all(self, eval_mode: bool = True):
        for p in self.parameters():
            p.requires_grad = False
        self._fully_frozen = True
        if eval_mode:
            super().train(False)
        return self

    def unfreeze_all(self):
        for p in self.parameters():
            p.requires_grad = True
        self._fully_frozen = False
        return self

    def freeze_all_but_last_stage(self):
        layers = list(self.vision_model.encoder.layers)
        for m in layers[:-1]:
            for p in m.parameters():
                p.requires_grad = False
        return self

    def train(self, mode: bool = True):
        if self._fully_frozen:
            return super().train(False)
        return super().train(mode)

    def forward(self, pixel_values: torch.Tensor = None, x: Optional[torch.Tensor] = None):
        if pixel_values is None and x is not None:
            pixel_values = x
        out = self.backbone(pixel_values)
        return SimpleNamespace(pooler_output=out, last_hidden_state=out.unsqueeze(1))
