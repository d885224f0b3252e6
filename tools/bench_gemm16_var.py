#!/usr/bin/env python3
"""dev: schedule variants of the LDS-DMA bf16 GEMM (GG_GEMM_DMA = 0 old | 1 default | 2 setprio around MFMA groups | 3 no operand DMA | 4 no MFMAs | 5 neither) x tile-order group (GG_GEMM_GM),
plain epilogue, interleaved rounds in one process."""
import os, sys
os.environ.setdefault("GG_DEV_SWITCHES", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import ops
T = 1024 * 50
shapes = [("sq8k", 8192, 8192, 8192), ("sq4k", 4096, 4096, 4096), ("c4.qkv", T, 2304, 768), ("c4.fc1", T, 3072, 768), ("c4.fc2", T, 768, 3072), ("c4.proj", T, 768, 768),
          ("s2.fc1", 200704, 1536, 384), ("s2.fc2", 200704, 384, 1536)]
variants = [("0", "0"), ("1", "0"), ("1", "8"), ("2", "8"), ("1", "4"), ("1", "16"), ("3", "8"), ("4", "8"), ("5", "8")]
for name, M, N, K in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    t = {v: [] for v in variants}
    for _ in range(4):
        for v in variants:
            os.environ["GG_GEMM_DMA"], os.environ["GG_GEMM_GM"] = v
            ops.gemm_nt(A, W, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): ops.gemm_nt(A, W, out=out)
            e1.record(); torch.cuda.synchronize()
            t[v].append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * M * N * K
    print(f"{name:8s} " + " | ".join(f"v{v[0]}/g{v[1]} {fl / sorted(t[v])[1] / 1e9:6.0f}" for v in variants), flush=True)
