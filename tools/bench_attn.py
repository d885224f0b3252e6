#!/usr/bin/env python3
"""Window-attention kernel timing at the TinyViT-21M-224 / 1024-image shapes (dev tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
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


for name, res, ws, Cc, nh in [("s1", 28, 7, 192, 6), ("s2", 14, 14, 384, 12), ("s3", 7, 7, 576, 18)]:
    M = B * res * res
    N = ws * ws
    qkv = torch.randn(M, 3 * Cc, device="cuda").bfloat16()
    out = torch.empty(M, Cc, device="cuda").bfloat16(); dout = torch.randn(M, Cc, device="cuda").bfloat16()
    dqkv = torch.empty_like(qkv); lse = torch.empty(M, nh, device="cuda")
    table = torch.randn(nh, N, device="cuda") * 0.1
    Np = L.lib().gg_attention_padded_tokens(N)
    full = torch.empty(nh, Np, Np, device="cuda").bfloat16()
    L.check(L.lib().gg_attention_expand_bias(table.data_ptr(), nh, ws, 32 ** -0.5, full.data_ptr(), L.stream()))
    dbias = torch.zeros_like(table)
    for with_bias in (True, False):
        a = L.AttnArgs()
        a.qkv, a.ld, a.q_off, a.k_off, a.v_off, a.head_stride, a.head_dim = qkv.data_ptr(), 3 * Cc, 0, 32, 64, 96, 32
        a.num_heads, a.num_windows, a.tokens_per_window = nh, B * (res // ws) ** 2, N
        a.window_size, a.map_h, a.map_w = ws, res, res
        a.bias = full.data_ptr() if with_bias else None
        a.scale = 32 ** -0.5
        a.out, a.ldo, a.lse = out.data_ptr(), Cc, lse.data_ptr()
        tf = timed(lambda: L.check(L.lib().gg_attention_fwd(C.byref(a), L.stream())))
        a.dout, a.lddo, a.dqkv = dout.data_ptr(), Cc, dqkv.data_ptr()
        a.dbias = None
        tb = timed(lambda: L.check(L.lib().gg_attention_bwd(C.byref(a), L.stream())))
        a.dbias = dbias.data_ptr() if with_bias else None
        tbb = timed(lambda: L.check(L.lib().gg_attention_bwd(C.byref(a), L.stream())))
        fb, bb = 2 * M * 4 * Cc, 2 * M * (3 * Cc + 2 * Cc + 3 * Cc)
        print(f"{name} ws={ws} bias={int(with_bias)}  fwd {tf*1e3:8.1f} us ({fb/tf/1e6:7.1f} GB/s)   bwd {tb*1e3:8.1f} us ({bb/tb/1e6:7.1f} GB/s)   bwd+dbias {tbb*1e3:8.1f} us")
