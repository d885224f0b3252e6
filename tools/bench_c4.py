#!/usr/bin/env python3
"""dev: CLIP ViT-B/32 tower inference at batch 1024 in one precision (argv[1]: fp32|bf16|fp16), a few calls -- for rocprofv3 --kernel-trace --stats."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd.pretrain.clip_embedder import CLIPVisionTower
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
tower = CLIPVisionTower("openai/clip-vit-base-patch32", precision=prec).cuda().eval()
for p_ in tower.parameters():
    p_.requires_grad = False
x = torch.randn(1024, 3, 224, 224, device="cuda")
with torch.no_grad():
    for _ in range(2):
        tower(pixel_values=x, return_last_hidden=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        tower(pixel_values=x, return_last_hidden=False)
    torch.cuda.synchronize()
print(prec, round((time.perf_counter() - t0) / 5 * 1e3, 3), "ms")
