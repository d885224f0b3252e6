#!/usr/bin/env python3
"""Depthwise 3x3 kernel timing at the TinyViT-21M-224 / 1024-image shapes (dev tool).  GG_DW_TILED=1 selects the LDS-tiled kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import _lib as L

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
    x = torch.randn(B, H, H, Cc, device="cuda").bfloat16(); y = torch.empty_like(x)
    taps = torch.randn(9, Cc, device="cuda")
    rows = L.lib().gg_dwconv_stat_rows(B, H, H, Cc, 1)
    part = torch.zeros(L.lib().gg_stat_rows_capacity(rows), 2, Cc, device="cuda")
    tf = timed(lambda: L.check(L.lib().gg_dwconv3x3_fwd(x.data_ptr(), taps.data_ptr(), y.data_ptr(), B, H, H, Cc, 1, part.data_ptr(), L.stream())))
    tb = timed(lambda: L.check(L.lib().gg_dwconv3x3_bwd_data(x.data_ptr(), taps.data_ptr(), y.data_ptr(), B, H, H, Cc, 1, L.stream())))
    byt = 4 * B * H * H * Cc
    print(f"{name:10s} {H}x{H}x{Cc}  fwd+stats {tf*1e3:8.1f} us ({byt/tf/1e6:7.1f} GB/s)   bwd_data {tb*1e3:8.1f} us ({byt/tb/1e6:7.1f} GB/s)   stat rows {rows}")

# ---- backward fusions (MBConv shape) ----
for name, H, Cc in [("mb", 56, 384), ("s2.local", 14, 384)]:
    dz = torch.randn(B, H, H, Cc, device="cuda").bfloat16(); yin = torch.randn_like(dz); out = torch.empty_like(dz); epy = torch.randn_like(dz)
    taps = torch.randn(9, Cc, device="cuda"); coef = torch.randn(3, Cc, device="cuda") * 0.1
    stat = torch.stack([torch.zeros(Cc), torch.ones(Cc)]).cuda(); g = torch.ones(Cc, device="cuda"); bt = torch.zeros(Cc, device="cuda")
    rows = L.lib().gg_dwconv_fused_stat_rows(B, H, H, Cc, 1)
    part = torch.zeros(L.lib().gg_stat_rows_capacity(max(rows, L.lib().gg_dwconv_fused_stat_rows(B, H, H, Cc, 0))), 2, Cc, device="cuda")
    N = None
    t_in = timed(lambda: L.check(L.lib().gg_dwconv3x3_bwd_data_fused(dz.data_ptr(), yin.data_ptr(), coef.data_ptr(), taps.data_ptr(), out.data_ptr(), B, H, H, Cc, N, N, N, N, 0, N, L.stream())))
    t_ep = timed(lambda: L.check(L.lib().gg_dwconv3x3_bwd_data_fused(dz.data_ptr(), N, N, taps.data_ptr(), out.data_ptr(), B, H, H, Cc, epy.data_ptr(), stat.data_ptr(), g.data_ptr(), bt.data_ptr(), 1, part.data_ptr(), L.stream())))
    t_both = timed(lambda: L.check(L.lib().gg_dwconv3x3_bwd_data_fused(dz.data_ptr(), yin.data_ptr(), coef.data_ptr(), taps.data_ptr(), out.data_ptr(), B, H, H, Cc, epy.data_ptr(), stat.data_ptr(), g.data_ptr(), bt.data_ptr(), 1, part.data_ptr(), L.stream())))
    L1 = 2 * B * H * H * Cc
    print(f"{name:10s} fused bwd: in2 {t_in*1e3:8.1f} us ({3*L1/t_in/1e6:7.1f} GB/s)  epi {t_ep*1e3:8.1f} us ({3*L1/t_ep/1e6:7.1f} GB/s)  both {t_both*1e3:8.1f} us ({4*L1/t_both/1e6:7.1f} GB/s)")
