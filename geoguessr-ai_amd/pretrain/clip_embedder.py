"""Drop-in for the reference's ``pretrain/clip_embedder.py`` (``CLIPEmbedding``, :10-101) plus the HIP vision tower it
wraps (``CLIPVisionTower``: transformers ``CLIPVisionModel`` semantics, HF state-dict keys).  Frozen, inference only:
``forward`` returns ``last_hidden_state.mean(dim=1)`` (:63-65); panorama kwargs ``image, image_2..4`` stack on dim 1.
No ``CLIPProcessor`` / hub download: inputs are preprocessed pixel tensors and weights are loaded from a state dict."""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace
from typing import Dict, Optional

import torch
from torch import Tensor, nn

from .. import _lib as L

CLIP_CONFIGS = {
    "openai/clip-vit-base-patch32": dict(hidden_size=768, intermediate_size=3072, num_layers=12, num_heads=12, image_size=224, patch_size=32),
    "openai/clip-vit-large-patch14-336": dict(hidden_size=1024, intermediate_size=4096, num_layers=24, num_heads=16, image_size=336, patch_size=14),
}


class CLIPVisionTower(nn.Module):
    def __init__(self, model_name: str = "openai/clip-vit-base-patch32", seed: int = 0, **cfg_overrides):
        super().__init__()
        kw = dict(CLIP_CONFIGS.get(model_name, CLIP_CONFIGS["openai/clip-vit-base-patch32"]))
        kw.update(cfg_overrides)
        c = L.ClipCfg()
        c.hidden_size, c.intermediate_size, c.num_layers, c.num_heads = kw["hidden_size"], kw["intermediate_size"], kw["num_layers"], kw["num_heads"]
        c.image_size, c.patch_size, c.ln_eps = kw["image_size"], kw["patch_size"], 1e-5
        self.cfg = c
        self.config = SimpleNamespace(hidden_size=kw["hidden_size"], _name_or_path=model_name, **{k: v for k, v in kw.items() if k != "hidden_size"})
        lib = L.lib()
        n = lib.gg_clip_num_tensors(C.byref(c))
        if n < 0:
            raise L.GgError(lib.gg_last_error().decode())
        self.table = []
        name = C.create_string_buffer(256)
        off, numel, ndim = C.c_int64(), C.c_int64(), C.c_int()
        shape = (C.c_int64 * 4)()
        for i in range(n):
            L.check(lib.gg_clip_tensor_info(C.byref(c), i, name, 256, C.byref(off), C.byref(numel), C.byref(ndim), shape), "gg_clip_tensor_info")
            self.table.append(dict(name=name.value.decode(), offset=off.value, numel=numel.value, shape=tuple(shape[j] for j in range(ndim.value))))
        self.param_floats = lib.gg_clip_param_floats(C.byref(c))
        g = torch.Generator().manual_seed(seed)
        self.flat = nn.Parameter(torch.zeros(self.param_floats), requires_grad=False)
        for t in self.table:
            v = self.flat.data[t["offset"]:t["offset"] + t["numel"]].view(t["shape"])
            if t["name"].endswith("norm.weight") or t["name"].endswith("norm1.weight") or t["name"].endswith("norm2.weight") or t["name"].endswith("layrnorm.weight"):
                v.fill_(1.0)
            elif t["name"].endswith(".bias"):
                v.zero_()
            else:
                v.copy_(torch.randn(t["shape"], generator=g) * 0.02)
        self._wcache = None
        self._ver = -1
        self._ws = None

    def named_views(self) -> Dict[str, Tensor]:
        return {t["name"]: self.flat.data[t["offset"]:t["offset"] + t["numel"]].view(t["shape"]) for t in self.table}

    def load_hf_state_dict(self, sd: Dict[str, Tensor]):
        """HF ``CLIPVisionModel`` keys; a leading ``vision_model.`` (transformers 4.x nesting) is stripped."""
        views = self.named_views()
        for k, v in sd.items():
            k = k[len("vision_model."):] if k.startswith("vision_model.") else k
            if k in views:
                views[k].copy_(torch.as_tensor(v).to(views[k].device, torch.float32))
        self.flat.data.add_(0)      # bump the version counter -> weight cache refresh

    @torch.no_grad()
    def forward(self, pixel_values: Tensor = None, return_last_hidden: bool = True):
        L.require_gpu()
        if not self.flat.is_cuda:
            raise L.GgError("CLIPVisionTower parameters are on the CPU; call .to('cuda') -- there is no CPU fallback")
        x = pixel_values.to(device=self.flat.device, dtype=torch.float32).contiguous()
        B = x.shape[0]
        lib = L.lib()
        if self._wcache is None or self._wcache.device != self.flat.device:
            self._wcache = torch.zeros(lib.gg_clip_wcache_bytes(C.byref(self.cfg)), dtype=torch.uint8, device=self.flat.device)
            self._ver = -1
        if self._ver != self.flat._version:
            L.check(lib.gg_clip_refresh_weights(C.byref(self.cfg), L.ptr(self.flat.data), L.ptr(self._wcache), L.stream()), "gg_clip_refresh_weights")
            self._ver = self.flat._version
        need = lib.gg_clip_workspace_bytes(C.byref(self.cfg), B)
        if self._ws is None or self._ws.numel() < need or self._ws.device != self.flat.device:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.flat.device)
        D = self.cfg.hidden_size
        T = (self.cfg.image_size // self.cfg.patch_size) ** 2 + 1
        out = torch.empty((B, D), dtype=torch.float32, device=x.device)
        last = torch.empty((B, T, D), dtype=torch.float32, device=x.device) if return_last_hidden else None
        L.check(lib.gg_clip_forward(C.byref(self.cfg), B, L.ptr(self.flat.data), L.ptr(self._wcache), L.ptr(x), L.ptr(self._ws), L.ptr(out),
                                    L.ptr(last), L.stream()), "gg_clip_forward")
        return SimpleNamespace(last_hidden_state=last, pooled_mean=out, pooler_output=out)


class CLIPEmbedding(nn.Module):
    def __init__(self, model_name: str = "openai/clip-vit-base-patch32", device: str = "cuda", load_checkpoint: bool = False,
                 panorama: bool = False, state_dict: Optional[Dict[str, Tensor]] = None, **cfg_overrides):
        super().__init__()
        self.device = device
        self.panorama = panorama
        self.clip_model = CLIPVisionTower(model_name if not load_checkpoint else "openai/clip-vit-base-patch32", **cfg_overrides)
        if load_checkpoint:
            state_dict = torch.load(model_name, map_location="cpu")
            print("Loaded embedder from checkpoint:", model_name)
        if state_dict is not None:
            self.clip_model.load_hf_state_dict({(".".join(k.split(".")[1:]) if "base_model" in k else k): v for k, v in state_dict.items()})
        self.clip_model = self.clip_model.to(device if isinstance(device, str) else f"cuda:{device}")
        self.eval()

    def _get_embedding(self, image: Tensor) -> Tensor:
        if not isinstance(image, Tensor):
            raise L.GgError("CLIPEmbedding expects preprocessed pixel tensors (no CLIPProcessor in this build)")
        return self.clip_model(pixel_values=image, return_last_hidden=False).pooled_mean

    def forward(self, image, **kwargs) -> Tensor:
        if isinstance(image, Tensor) or "image_2" not in kwargs:
            return self._get_embedding(image)
        embs = [self._get_embedding(image)] + [self._get_embedding(kwargs[c]) for c in ("image_2", "image_3", "image_4")]
        return torch.stack(embs, dim=1)
