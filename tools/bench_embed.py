#!/usr/bin/env python3
"""Bulk-embedding forward (the "next" row f2's inner loop): TinyViT-21M-224 inference at 1024 images per call, fp32 vs fp32_split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
x = torch.randn(1024, 3, 224, 224, device="cuda")
for prec in ("fp32", "fp32_split", "bf16"):
    m = TinyViTAdapter("tiny_vit_21m_224", pretrained=False, precision=prec).cuda().eval()
    with torch.no_grad():
        for _ in range(2): m(pixel_values=x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): m(pixel_values=x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{prec:11s} {dt * 1e3:8.2f} ms per 1024 images = {1024 / dt:9.0f} images/s", flush=True)
    del m; torch.cuda.empty_cache()
