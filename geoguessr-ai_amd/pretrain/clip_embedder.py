"""Drop-in for the reference's ``pretrain/clip_embedder.py`` (``CLIPEmbedding``, :10-101) plus the HIP vision tower it wraps
(``CLIPVisionTower``: transformers ``CLIPVisionModel`` semantics and state-dict keys, ``csrc/clip.hip``).

* ``CLIPEmbedding``: frozen, inference only -- ``forward`` returns ``last_hidden_state.mean(dim=1)`` (:63-65); panorama kwargs
  ``image, image_2..4`` stack on dim 1 (:94-101).  Float tensors are pixel_values; PIL images / uint8 arrays or tensors go through the
  processor's tensor side on the device (``clip_preprocess``).
* ``CLIPVisionTower``: the base model of ``SuperGuessr`` for CLIP runs (models/super_guessr.py:134-150: ``config._name_or_path`` contains
  "clip-vit", ``config.hidden_size``, ``vision_model.encoder.layers``).  It trains: the whole forward and backward are one C call each
  (``gg_clip_forward`` / ``gg_clip_backward``), parameters are views into one flat fp32 buffer like the TinyViT backbone's, so
  ``optim.AdamW`` and the RCCL gradient exchange treat both encoders alike.

Arithmetic: ``precision="fp32"`` (default; the reference runs the tower in fp32), ``"bf16"``, or ``"fp16"`` (inference only: the precision BASELINE
config c4 names).  No hub download: weights come from a
state dict (HF names, with or without the leading ``vision_model.``)."""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace
from typing import Dict, Optional

import torch
from torch import Tensor, nn

from .. import _lib as L
from ..models.flat import FlatStore

CLIP_CONFIGS = {
    "openai/clip-vit-base-patch32": dict(hidden_size=768, intermediate_size=3072, num_layers=12, num_heads=12, image_size=224, patch_size=32),
    "openai/clip-vit-large-patch14-336": dict(hidden_size=1024, intermediate_size=4096, num_layers=24, num_heads=16, image_size=336, patch_size=14),
}
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # CLIPProcessor's image_mean / image_std (openai/clip-vit-*)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _precision_code(precision: Optional[str]) -> int:
    from ..models.tinyvit import PRECISIONS, default_precision
    p = precision or default_precision()
    if p in ("fp16", "f16", "float16", "half"):
        return 2                      # CLIP only, inference only (BASELINE config c4: "MFMA fp16")
    if p not in PRECISIONS:
        raise ValueError(f"precision='{p}' (known: bf16, fp16, fp32)")
    return PRECISIONS[p]


class _VisionModel(FlatStore):
    """``vision_model`` of the HF module tree: owner of the flat storage.  Its children (``embeddings``, ``pre_layrnorm``,
    ``encoder.layers.N....``, ``post_layernorm``) are rebuilt from the tensor names, so ``encoder.layers[i].parameters()`` is what
    ``SuperGuessr._freeze_params`` expects."""

    def __init__(self, cfg: L.ClipCfg, seed: int):
        super().__init__()
        self.cfg = cfg
        self.precision = {0: "bf16", 1: "fp32", 2: "fp16"}[cfg.act_dtype]
        lib = L.lib()
        n = lib.gg_clip_num_tensors(C.byref(cfg))
        if n < 0:
            raise L.GgError(lib.gg_last_error().decode())
        self.table = []
        name = C.create_string_buffer(256)
        off, numel, ndim = C.c_int64(), C.c_int64(), C.c_int()
        shape = (C.c_int64 * 4)()
        for i in range(n):
            L.check(lib.gg_clip_tensor_info(C.byref(cfg), i, name, 256, C.byref(off), C.byref(numel), C.byref(ndim), shape), "gg_clip_tensor_info")
            self.table.append(dict(name=name.value.decode(), offset=off.value, numel=numel.value, shape=tuple(shape[j] for j in range(ndim.value)),
                                   kind=0, index=i))
        self.param_floats = lib.gg_clip_param_floats(C.byref(cfg))
        self.buffer_floats, self.num_counters = 0, 0
        g = torch.Generator().manual_seed(seed)

        def init(name, shape):
            if name.endswith(("norm.weight", "norm1.weight", "norm2.weight")):
                return torch.ones(shape)
            if name.endswith(".bias"):
                return torch.zeros(shape)
            return torch.randn(shape, generator=g) * 0.02
        self._register_table(init)
        self._wcache, self._wcache_version, self._ws = None, -1, {}
        self._last = None
        self._gen = 0
        self._grad_ready_hook = None      # (optim.AdamW.overlap_allreduce sets it; the CLIP backward has no stage callback: buckets leave after it)


class CLIPVisionTower(nn.Module):
    def __init__(self, model_name: str = "openai/clip-vit-base-patch32", seed: int = 0, precision: Optional[str] = None, **cfg_overrides):
        super().__init__()
        kw = dict(CLIP_CONFIGS.get(model_name, CLIP_CONFIGS["openai/clip-vit-base-patch32"]))
        kw.update(cfg_overrides)
        c = L.ClipCfg()
        c.hidden_size, c.intermediate_size, c.num_layers, c.num_heads = kw["hidden_size"], kw["intermediate_size"], kw["num_layers"], kw["num_heads"]
        c.image_size, c.patch_size, c.ln_eps = kw["image_size"], kw["patch_size"], 1e-5
        c.act_dtype = _precision_code(precision)
        self.cfg = c
        self.precision = {0: "bf16", 1: "fp32", 2: "fp16"}[c.act_dtype]
        self.config = SimpleNamespace(hidden_size=kw["hidden_size"], _name_or_path=model_name, **{k: v for k, v in kw.items() if k != "hidden_size"})
        self.vision_model = _VisionModel(c, seed)
        self.num_tokens = (c.image_size // c.patch_size) ** 2 + 1

    # ---- weights ----------------------------------------------------------------------------------------------------------------------
    @property
    def backbone(self):          # same attribute name as TinyViTAdapter's flat-storage owner (SuperGuessr reads .backbone.precision)
        return self.vision_model

    def named_views(self) -> Dict[str, Tensor]:
        return {n: p.data for n, p in self.vision_model._params.items()}

    def load_hf_state_dict(self, sd: Dict[str, Tensor]):
        """HF ``CLIPVisionModel`` keys; a leading ``vision_model.`` (transformers 4.x nesting) is stripped.  Unknown keys (``position_ids``) are ignored."""
        views = self.named_views()
        with torch.no_grad():
            for k, v in sd.items():
                k = k[len("vision_model."):] if k.startswith("vision_model.") else k
                if k in views:
                    views[k].copy_(torch.as_tensor(v).to(views[k].device, torch.float32))
        self.vision_model.mark_params_dirty()

    # ---- HIP calls --------------------------------------------------------------------------------------------------------------------
    def _ensure_weights(self):
        vm, lib = self.vision_model, L.lib()
        if vm._wcache is None or vm._wcache.device != vm._flat.device:
            vm._wcache = torch.zeros(lib.gg_clip_wcache_bytes(C.byref(self.cfg)), dtype=torch.uint8, device=vm._flat.device)
            vm._wcache_version = -1
        ver = vm._param_version()
        if vm._wcache_version != ver:
            L.check(lib.gg_clip_refresh_weights(C.byref(self.cfg), L.ptr(vm._flat), L.ptr(vm._wcache), L.stream()), "gg_clip_refresh_weights")
            vm._wcache_version = ver

    def _workspace(self, batch: int, training: bool, mask) -> Tensor:
        vm = self.vision_model
        need = L.lib().gg_clip_workspace_bytes(C.byref(self.cfg), batch, int(training), mask)
        if need < 0:
            raise L.GgError(L.lib().gg_last_error().decode())
        ws = vm._ws.get(training)
        if ws is None or ws.numel() < need or ws.device != vm._flat.device:
            vm._ws[training] = None
            ws = vm._ws[training] = torch.empty(need, dtype=torch.uint8, device=vm._flat.device)
        return ws

    def forward_hip(self, x: Tensor, training: bool, return_last_hidden: bool):
        L.require_gpu()
        vm = self.vision_model
        if not vm._flat.is_cuda:
            raise L.GgError("CLIPVisionTower parameters are on the CPU; call .to('cuda') -- there is no CPU fallback")
        S = self.cfg.image_size
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != S or x.shape[3] != S:
            raise L.GgError(f"CLIPVisionTower expects (B,3,{S},{S}) pixel_values, got {tuple(x.shape)}")
        x = x.to(device=vm._flat.device, dtype=torch.float32).contiguous()
        B = x.shape[0]
        self._ensure_weights()
        mask = vm.trainable_mask() if training else None
        ws = self._workspace(B, training, mask)
        D, T = self.cfg.hidden_size, self.num_tokens
        out = torch.empty((B, D), dtype=torch.float32, device=x.device)
        last = torch.empty((B, T, D), dtype=torch.float32, device=x.device) if return_last_hidden else None
        L.check(L.lib().gg_clip_forward(C.byref(self.cfg), B, int(training), L.ptr(vm._flat), L.ptr(vm._wcache), L.ptr(x), L.ptr(ws), L.ptr(out),
                                        L.ptr(last), mask, L.stream()), "gg_clip_forward")
        if training:
            vm._gen += 1
            vm._last = (B, mask, vm._gen)
        return out, last

    def backward_hip(self, d_out: Optional[Tensor], d_last: Optional[Tensor], gen: int):
        vm = self.vision_model
        if vm._last is None:
            raise L.GgError("CLIP backward without a training forward")
        B, mask, last_gen = vm._last
        if gen != last_gen:
            raise L.GgError(f"CLIP backward for training forward #{gen}, but the workspace now holds the activations of forward #{last_gen}: every "
                            "training forward must be followed by its backward before the next training forward")
        if vm.trainable_mask() != mask:
            raise L.GgError("requires_grad changed between the CLIP forward and its backward; run the forward again")
        fg = vm.attach_grads()
        f = lambda t: None if t is None else t.to(torch.float32).contiguous()
        d_out, d_last = f(d_out), f(d_last)
        L.check(L.lib().gg_clip_backward(C.byref(self.cfg), B, L.ptr(vm._flat), L.ptr(vm._wcache), L.ptr(vm._ws[True]), L.ptr(d_out), L.ptr(d_last),
                                         L.ptr(fg), mask, L.stream()), "gg_clip_backward")
        hook = vm._grad_ready_hook
        if hook is not None:
            hook(0, vm.param_floats)

    def forward(self, pixel_values: Tensor = None, return_last_hidden: bool = True):
        vm = self.vision_model
        need = torch.is_grad_enabled() and any(p.requires_grad for p in vm._params.values())      # (no dropout / BatchNorm: train and eval compute the same)
        if need and self.precision == "fp16":
            raise L.GgError("CLIPVisionTower(precision='fp16') is inference-only: run under torch.no_grad() / freeze it, or train in fp32 / bf16")
        if not need:
            out, last = self.forward_hip(pixel_values, False, return_last_hidden)
        else:
            out, last = _ClipFn.apply(self, pixel_values, _anchor(vm), return_last_hidden)
        return SimpleNamespace(last_hidden_state=last, pooled_mean=out, pooler_output=out)


def _anchor(vm: _VisionModel) -> Tensor:
    a = getattr(vm, "_anchor_t", None)
    if a is None or a.device != vm._flat.device:
        a = vm._anchor_t = torch.zeros((), device=vm._flat.device, requires_grad=True)
    return a


class _ClipFn(torch.autograd.Function):
    """Whole-tower autograd node (the counterpart of the TinyViT backbone's): parameter gradients are accumulated straight into the flat
    gradient buffer the parameters' ``.grad`` views point into; the zero-dim ``anchor`` input only keeps the node alive."""

    @staticmethod
    def forward(ctx, tower: CLIPVisionTower, x: Tensor, anchor: Tensor, want_last: bool):
        out, last = tower.forward_hip(x, True, want_last)
        ctx.tower, ctx.gen = tower, tower.vision_model._gen
        ctx.set_materialize_grads(False)          # an unused last_hidden_state must not cost a (B,T,D) zero gradient
        return out, last

    @staticmethod
    def backward(ctx, d_out, d_last):
        ctx.tower.backward_hip(d_out, d_last, ctx.gen)
        dev = (d_out if d_out is not None else d_last).device
        return None, None, torch.zeros((), device=dev), None


def clip_preprocess(images, size: int, device) -> Tensor:
    """``CLIPProcessor(images=image, return_tensors="pt")["pixel_values"]`` (pretrain/clip_embedder.py:55-57) on the device: see
    ``training.preprocess.images_to_pixel_values`` (Pillow bicubic resize of the shortest edge to ``size``, centre crop, * 1/255, CLIP mean / std; the uint8
    image is bit-identical to the processor's, pinned by transformers' own output in tests/golden/preprocess_pil.npz)."""
    from ..training.preprocess import images_to_pixel_values
    return images_to_pixel_values(images, size, CLIP_MEAN, CLIP_STD, device, pipeline="clip")


class CLIPEmbedding(nn.Module):
    def __init__(self, model_name: str = "openai/clip-vit-base-patch32", device: str = "cuda", load_checkpoint: bool = False,
                 panorama: bool = False, state_dict: Optional[Dict[str, Tensor]] = None, precision: Optional[str] = None, **cfg_overrides):
        super().__init__()
        self.device = device
        self.panorama = panorama
        self.clip_model = CLIPVisionTower(model_name if not load_checkpoint else "openai/clip-vit-base-patch32", precision=precision, **cfg_overrides)
        if load_checkpoint:
            state_dict = torch.load(model_name, map_location="cpu")
            print("Loaded embedder from checkpoint:", model_name)
        if state_dict is not None:
            self.clip_model.load_hf_state_dict({(k.partition(".")[2] if "base_model" in k else k): v for k, v in state_dict.items()})
        self.clip_model = self.clip_model.to(device if isinstance(device, str) else f"cuda:{device}")
        for p in self.clip_model.parameters():          # the embedder is frozen (pretrain/clip_embedder.py:51: torch.no_grad())
            p.requires_grad = False
        self.eval()

    def _get_embedding(self, image) -> Tensor:
        """A float tensor is taken as ``pixel_values`` (pretrain/clip_embedder.py:58-59); anything else -- PIL image, uint8 array / tensor, list of
        images -- goes through the processor's tensor side on the device (:55-57)."""
        dev = next(self.clip_model.parameters()).device
        if torch.is_tensor(image) and image.is_floating_point():
            pixel_values = image
        else:
            pixel_values = clip_preprocess(image, self.clip_model.cfg.image_size, dev)
        with torch.no_grad():
            return self.clip_model(pixel_values=pixel_values, return_last_hidden=False).pooled_mean

    def forward(self, image, **kwargs) -> Tensor:
        if "image_2" not in kwargs:
            return self._get_embedding(image)
        embs = [self._get_embedding(image)] + [self._get_embedding(kwargs[c]) for c in ("image_2", "image_3", "image_4")]
        return torch.stack(embs, dim=1)
