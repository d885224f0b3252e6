#!/usr/bin/env python3
"""dev: GG_GEMM_DEPHASE sweep (cycles the second workgroup of a CU waits before its first tile)."""
import os, sys
os.environ.setdefault("GG_DEV_SWITCHES", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import ops
T = 1024 * 50
shapes = [("c4.qkv", T, 2304, 768, {"bias": 1}), ("c4.fc1", T, 3072, 768, {"bias": 1, "act": "quick_gelu"}), ("c4.fc2", T, 768, 3072, {"bias": 1}), ("c4.proj", T, 768, 768, {"bias": 1}),
          ("s2.fc1", 200704, 1536, 384, {"bias": 1, "act": "gelu"}), ("s2.fc2", 200704, 384, 1536, {"bias": 1}), ("sq8k", 8192, 8192, 8192, {})]
vals = ["0", "5000", "10000", "15000", "20000", "30000", "50000"]
for name, M, N, K, o in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    kw = {}
    if o.get("bias"): kw["bias"] = torch.randn(N, device="cuda")
    if o.get("act"): kw["act"] = o["act"]
    t = {v: [] for v in vals}
    for _ in range(5):
        for v in vals:
            os.environ["GG_GEMM_DEPHASE"] = v
            ops.gemm_nt(A, W, out=out, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): ops.gemm_nt(A, W, out=out, **kw)
            e1.record(); torch.cuda.synchronize()
            t[v].append(e0.elapsed_time(e1) / 3)
    print(f"{name:8s} " + " | ".join(f"{v:>6s}: {2.0*M*N*K / sorted(t[v])[2] / 1e9:6.0f}" for v in vals), flush=True)
