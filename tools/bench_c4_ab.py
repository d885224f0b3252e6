#!/usr/bin/env python3
"""dev: CLIP ViT-B/32 fp16 tower, batch 1024, under dev switch settings given as NAME=VALUE,... groups on the command line, interleaved rounds in one process."""
import sys, os, time
os.environ["GG_DEV_SWITCHES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd.pretrain.clip_embedder import CLIPVisionTower
groups = [dict(kv.split("=") for kv in g.split(",") if kv) for g in sys.argv[1:]] or [{}]
keys = sorted({k for g in groups for k in g})
tower = CLIPVisionTower("openai/clip-vit-base-patch32", precision="fp16").cuda().eval()
x = torch.randn(1024, 3, 224, 224, device="cuda")
res = {i: [] for i in range(len(groups))}
with torch.no_grad():
    for r in range(4):
        for i, g in enumerate(groups):
            for k in keys: os.environ.pop(k, None)
            os.environ.update(g)
            tower(pixel_values=x, return_last_hidden=False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(4): tower(pixel_values=x, return_last_hidden=False)
            torch.cuda.synchronize()
            res[i].append((time.perf_counter() - t0) / 4 * 1e3)
for i, g in enumerate(groups):
    print(g, "ms per forward:", " ".join(f"{v:.2f}" for v in res[i]), " median", f"{sorted(res[i])[len(res[i]) // 2]:.2f}")
