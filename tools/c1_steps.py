#!/usr/bin/env python3
"""A few c1 training steps (TinyViT-5M, batch 8) for a rocprofv3 kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
from geoguessr_ai_amd.models.super_guessr import SuperGuessr
from geoguessr_ai_amd.optim import AdamW
dev = "cuda"
base = TinyViTAdapter("tiny_vit_5m_224", pretrained=False, precision="fp32")
model = SuperGuessr(base, panorama=False, should_smooth_labels=False, serving=False).to(dev).train()
opt = AdamW(model, lr=5e-5, betas=(0.9, 0.999), weight_decay=0.01)
g = torch.Generator(device=dev).manual_seed(330)
x = torch.randn(8, 3, 224, 224, device=dev, generator=g)
lab = torch.stack([torch.rand(8, device=dev, generator=g) * 360 - 180, torch.rand(8, device=dev, generator=g) * 180 - 90], 1)
clf = torch.randint(0, model.num_cells, (8,), device=dev, generator=g)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    o = model(pixel_values=x, labels=lab, labels_clf=clf)
    o.loss.backward(); opt.step(); opt.zero_grad()
torch.cuda.synchronize()
