#!/usr/bin/env python3
"""dev: fp32 depthwise 3x3 stride-1 kernels at the TinyViT-21M-224 / 1024-image shapes (GG_DW_F32_NO_MULTI=1: one-column kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import ops

B = 1024
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for name, H, Cc in [("mb", 56, 384), ("s1.local", 28, 192), ("s2.local", 14, 384), ("s3.local", 7, 576)]:
    x = torch.randn(B, H, H, Cc, device="cuda"); x2 = torch.randn(B, H, H, Cc, device="cuda"); x3 = torch.randn(B, H, H, Cc, device="cuda")
    taps = torch.randn(9, Cc, device="cuda")
    stat = torch.stack([torch.zeros(Cc), torch.ones(Cc)]).cuda(); g = torch.ones(Cc, device="cuda"); b = torch.zeros(Cc, device="cuda")
    coef = torch.ones(3, Cc, device="cuda")
    byt = 4 * B * H * H * Cc
    t = {}
    t["fwd+stats"] = (timed(lambda: ops.dwconv3x3_fwd(x, taps, 1, colstats=True)), 2)
    t["fwd bn+gelu on load"] = (timed(lambda: ops.dwconv3x3_fwd_fused(x, stat, g, b, taps, act="gelu", stride=1)), 2)
    t["bwd_data"] = (timed(lambda: ops.dwconv3x3_bwd_data(x, taps, B, H, H, Cc, 1)), 2)
    t["bwd_data in"] = (timed(lambda: ops.dwconv3x3_bwd_data_fused(x, x2, coef, taps)), 3)
    t["bwd_data in+epi"] = (timed(lambda: ops.dwconv3x3_bwd_data_fused(x, x2, coef, taps, ep_y=x3, ep_stat=stat, ep_gamma=g, ep_beta=b, ep_act="gelu")), 4)
    print(f"{name:9s} {H}x{H}x{Cc}: " + "  ".join(f"{k} {v*1e3:7.1f} us ({n*byt/v/1e6:6.0f} GB/s)" for k, (v, n) in t.items()))
