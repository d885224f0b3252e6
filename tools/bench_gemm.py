#!/usr/bin/env python3
"""Per-shape timing of the GEMM kernel on the shapes of the TinyViT-21M-224 step at 1024 images (dev tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import ops

B = 1024
M1, M0, Ms1, Ms2, Ms3 = B * 112 * 112, B * 56 * 56, B * 28 * 28, B * 14 * 14, B * 7 * 7
shapes = [
    ("pe.conv1", M1, 48, 32, {}), ("pe.conv2", M0, 96, 432, {}),
    ("mb.conv1", M0, 384, 96, {}), ("mb.conv3", M0, 96, 384, {}),
    ("mb.conv3.dgrad", M0, 384, 96, {}), ("mb.conv1.dgrad", M0, 96, 384, {"residual": True}),
    ("s1.merge1", M0, 192, 96, {}), ("s1.qkv", Ms1, 576, 192, {"bias": True}), ("s1.proj", Ms1, 192, 192, {"bias": True, "residual": True}),
    ("s1.fc1", Ms1, 768, 192, {"bias": True, "act": "gelu", "preact": True}), ("s1.fc2", Ms1, 192, 768, {"bias": True, "residual": True}),
    ("s1.fc2.dgrad", Ms1, 768, 192, {"dact": True}),
    ("s2.qkv", Ms2, 1152, 384, {"bias": True}), ("s2.fc1", Ms2, 1536, 384, {"bias": True, "act": "gelu", "preact": True}),
    ("s2.fc1.nopre", Ms2, 1536, 384, {"bias": True, "act": "gelu"}), ("s2.fc1.bias", Ms2, 1536, 384, {"bias": True}), ("s2.fc1.plain", Ms2, 1536, 384, {}),
    ("s2.fc2", Ms2, 384, 1536, {"bias": True, "residual": True}), ("s2.fc2.dgrad", Ms2, 1536, 384, {"dact": True}),
    ("s3.fc1", Ms3, 2304, 576, {"bias": True, "act": "gelu", "preact": True}), ("s3.fc2", Ms3, 576, 2304, {"bias": True, "residual": True}),
    ("head", 256, 12647, 576, {"bias": True, "out_f32": True}),
    ("square4k", 4096, 4096, 4096, {}),
]
for name, M, N, K, o in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    kw = {}
    if o.get("bias"): kw["bias"] = torch.randn(N, device="cuda")
    if o.get("act"): kw["act"] = o["act"]
    if o.get("preact"): kw["preact"] = True
    if o.get("residual"): kw["residual"] = torch.randn(M, N, device="cuda").bfloat16()
    if o.get("dact"): kw["dact_preact"], kw["dact"] = torch.randn(M, N, device="cuda").bfloat16(), "gelu"
    if o.get("out_f32"): kw["out_f32"] = True
    out = torch.empty((M, N), dtype=torch.float32 if o.get("out_f32") else torch.bfloat16, device="cuda")
    for _ in range(2): ops.gemm_nt(A, W, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n): ops.gemm_nt(A, W, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    byt = 2 * (M * K + N * K + M * N * (1 + bool(o.get("preact")) + bool(o.get("residual")) + bool(o.get("dact"))))
    if o.get("out_f32"): byt += 2 * M * N
    print(f"{name:16s} M={M:9d} N={N:5d} K={K:5d}  {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TF/s  {byt/ms/1e6:8.1f} GB/s")
    del A, W, out, kw
