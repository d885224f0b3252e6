"""Drop-in for the reference's ``pretrain/tinyvit_embedder.py`` (``TinyViTEmbedding``, :8-124): frozen TinyViT with
``num_classes=0`` run under ``no_grad``; panorama kwargs ``image_2..4`` stack on dim 1.  Float tensors are pixel_values (:70-72); PIL images /
uint8 arrays or tensors go through timm's eval transform on the device (:51-53,67-69: Pillow bicubic resize of the shortest edge to floor(size / crop_pct) --
of both edges in "squash" mode --, centre crop, /255, ImageNet mean / std: ``training.preprocess.images_to_pixel_values``, bit-identical to Pillow on the
uint8 image)."""
from __future__ import annotations

import torch
from torch import Tensor

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
        if isinstance(image, Tensor) and image.is_floating_point():
            pixel_values = image
        else:
            from ..training.preprocess import TINYVIT_MEAN, TINYVIT_STD, images_to_pixel_values
            bb = self.tinyvit_model.backbone
            # timm's published default_cfgs (timm/models/tiny_vit.py; timm itself is not in the image): crop_pct 0.95 for the 224 variants, 1.0 for the 384 one,
            # 1.0 with crop_mode "squash" for the 512 one; bicubic everywhere
            crop = 0.95 if bb.cfg.img_size == 224 else 1.0
            pixel_values = images_to_pixel_values(image, bb.cfg.img_size, TINYVIT_MEAN, TINYVIT_STD, bb.flat_params.device, crop_pct=crop, pipeline="timm",
                                                  crop_mode="squash" if bb.cfg.img_size == 512 else "center")
        with torch.no_grad():
            return self.tinyvit_model(pixel_values=pixel_values).pooler_output

    def forward(self, image, **kwargs) -> Tensor:
        if "image_2" not in kwargs:
            return self._get_embedding(image)
        embs = [self._get_embedding(image)] + [self._get_embedding(kwargs[c]) for c in ("image_2", "image_3", "image_4")]
        return torch.stack(embs, dim=1)
