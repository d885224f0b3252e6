#!/usr/bin/env python3
"""dev: repeatability of the 16-bit NT GEMM (a race shows as run-to-run differences): every epilogue class on the bf16 TinyViT / CLIP shapes, 30 launches each into a poisoned output."""
import os, sys
os.environ.setdefault("GG_DEV_SWITCHES", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import ops
T = 1024 * 50
Ms1, Ms2, Ms3 = 1024 * 28 * 28, 1024 * 14 * 14, 1024 * 7 * 7
shapes = [("s2.fc1", Ms2, 1536, 384), ("s2.qkv", Ms2, 1152, 384), ("s2.fc2", Ms2, 384, 1536), ("s2.proj", Ms2, 384, 384), ("s1.fc2", Ms1, 192, 768), ("s1.fc1", Ms1, 768, 192),
          ("s3.fc1", Ms3, 2304, 576), ("c4.qkv", T, 2304, 768), ("c4.fc2", T, 768, 3072)]
for name, M, N, K in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda").bfloat16(); pre = torch.randn(M, N, device="cuda").bfloat16()
    rps = 196 if M % 196 == 0 else 50
    rs = torch.rand(M // rps, device="cuda")
    cases = {"plain": {}, "bias": {"bias": bias}, "bias+rowscale+res": {"bias": bias, "residual": res, "rowscale": rs, "rows_per_scale": rps},
             "gelu+preact": {"bias": bias, "act": "gelu", "preact": True}, "dgelu": {"dact_preact": pre, "dact": "gelu"}, "dgelu+rowscale": {"dact_preact": pre, "dact": "gelu", "rowscale": rs, "rows_per_scale": rps}, "res": {"residual": res}}
    for cn, kw in cases.items():
        ref = None; bad = 0
        for it in range(30):
            out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            r = ops.gemm_nt(A, W, out=out, **kw)
            outs = r if isinstance(r, tuple) else (r,)
            cur = [o.clone() for o in outs]
            if ref is None: ref = cur
            else: bad += int(any(not torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(ref, cur)))
        nan = int(torch.isnan(ref[0].float()).sum())
        print(f"{name:8s} {cn:18s} differing runs {bad}/29  nan {nan}", flush=True)
    del A, W, res, pre
