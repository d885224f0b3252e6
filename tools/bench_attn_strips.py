#!/usr/bin/env python3
"""dev: does the 13-strip (196-token) resident attention lose to SIMD imbalance?  Plain sequences (no window geometry, no bias) of N tokens,
head dim 32, 12 heads, 1024 sequences, f32: time per (16 x 16) score block for N = 176 (11 strips) ... 256 (16 strips)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from geoguessr_ai_amd import _lib as L
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B, nh, Cc = 1024, 12, 384
for N in (128, 176, 192, 196, 208, 224, 256):
    M = B * N
    qkv = torch.randn(M, 3 * Cc, device="cuda"); out = torch.empty(M, Cc, device="cuda"); dout = torch.randn(M, Cc, device="cuda")
    dqkv = torch.empty_like(qkv); lse = torch.empty(M, nh, device="cuda")
    a = L.AttnArgs()
    a.qkv, a.ld, a.q_off, a.k_off, a.v_off, a.head_stride, a.head_dim = qkv.data_ptr(), 3 * Cc, 0, 32, 64, 96, 32
    a.num_heads, a.num_windows, a.tokens_per_window = nh, B, N
    a.scale = 32 ** -0.5
    a.out, a.ldo, a.lse = out.data_ptr(), Cc, lse.data_ptr()
    tf = timed(lambda: L.check(L.lib().gg_attention_flash_fwd(C.byref(a), 1, L.stream())))
    a.dout, a.lddo, a.dqkv = dout.data_ptr(), Cc, dqkv.data_ptr()
    ds = torch.empty(L.lib().gg_attention_flash_ds_scratch_floats(B, nh, N), device="cuda")
    a.ds_scratch = ds.data_ptr()
    td = timed(lambda: L.check(L.lib().gg_attention_flash_bwd(C.byref(a), 1, L.stream())))
    s = (N + 15) // 16
    blocks = B * nh * s * s
    print(f"N={N:4d} strips={s:2d}  fwd {tf*1e3:8.1f} us = {tf*1e6/blocks*1e3*1024:7.1f} SIMD-ns/block   bwd(dS) {td*1e3:8.1f} us = {td*1e6/blocks*1e3*1024:7.1f} SIMD-ns/block"
          f"   (MFMA floor at 2.0 GHz: fwd {16*32/2.0:.0f}, bwd {40*32/2.0:.0f} ns)")
