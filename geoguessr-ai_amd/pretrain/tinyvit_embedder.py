"""Drop-in for the reference's ``pretrain/tinyvit_embedder.py`` (``TinyViTEmbedding``, :8-124): frozen TinyViT with
``num_classes=0`` run under ``no_grad``; panorama kwargs ``image_2..4`` stack on dim 1.  PIL inputs need timm's eval
transform (absent here): tensors only."""
from __future__ import annotations

import torch
from torch import Tensor

from .. import _lib as L
from ..models.tinyvit import TinyViTAdapter


class TinyViTEmbedding(torch.nn.Module):
    def __init__(self, model_name: str = "tiny_vit_21m_512.dist_in22k_ft_in1k", device: str = "cuda", load_checkpoint: bool = False,
                 panorama: bool = False):
        super().__init__()
        self.device, self.panorama, self.model_name = device, panorama, model_name
        arch = "tiny_vit_21m_224" if load_checkpoint else model_name
        self.tinyvit_model = TinyViTAdapter(arch, pretrained=not load_checkpoint)
        if load_checkpoint:
            self.tinyvit_model.backbone.load_state_dict(torch.load(model_name, map_location="cpu"))
            print("Loaded embedder from checkpoint:", model_name)
        self.tinyvit_model = self.tinyvit_model.to(device if isinstance(device, str) else f"cuda:{device}")
        self.eval()

    def _get_embedding(self, image) -> Tensor:
        if not isinstance(image, Tensor):
            raise L.GgError("TinyViTEmbedding expects preprocessed pixel tensors (timm transforms are not available)")
        with torch.no_grad():
            return self.tinyvit_model(pixel_values=image).pooler_output

    def forward(self, image, **kwargs) -> Tensor:
        if isinstance(image, Tensor) or "image_2" not in kwargs:
            return self._get_embedding(image)
        embs = [self._get_embedding(image)] + [self._get_embedding(kwargs[c]) for c in ("image_2", "image_3", "image_4")]
        return torch.stack(embs, dim=1)
