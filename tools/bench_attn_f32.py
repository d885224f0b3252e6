#!/usr/bin/env python3
"""dev: fp32 flash attention (gg_attention_flash_fwd / _bwd, dtype 1) at the TinyViT-21M-224 / 1024-image shapes."""
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
    M, N = B * res * res, ws * ws
    qkv = torch.randn(M, 3 * Cc, device="cuda"); out = torch.empty(M, Cc, device="cuda"); dout = torch.randn(M, Cc, device="cuda")
    dqkv = torch.empty_like(qkv); lse = torch.empty(M, nh, device="cuda"); table = torch.randn(nh, N, device="cuda") * 0.1
    a = L.AttnArgs()
    a.qkv, a.ld, a.q_off, a.k_off, a.v_off, a.head_stride, a.head_dim = qkv.data_ptr(), 3 * Cc, 0, 32, 64, 96, 32
    a.num_heads, a.num_windows, a.tokens_per_window = nh, B * (res // ws) ** 2, N
    a.window_size, a.map_h, a.map_w = ws, res, res
    a.bias_table = table.data_ptr(); a.scale = 32 ** -0.5
    a.out, a.ldo, a.lse = out.data_ptr(), Cc, lse.data_ptr()
    tf = timed(lambda: L.check(L.lib().gg_attention_flash_fwd(C.byref(a), 1, L.stream())))
    a.dout, a.lddo, a.dqkv = dout.data_ptr(), Cc, dqkv.data_ptr()
    tb = timed(lambda: L.check(L.lib().gg_attention_flash_bwd(C.byref(a), 1, L.stream())))
    ds = torch.empty(L.lib().gg_attention_flash_ds_scratch_floats(a.num_windows, nh, N), device="cuda")
    a.ds_scratch = ds.data_ptr()
    td = timed(lambda: L.check(L.lib().gg_attention_flash_bwd(C.byref(a), 1, L.stream())))
    fl = 4.0 * M * N * Cc
    print(f"{name} ws={ws}  fwd {tf*1e3:8.1f} us ({fl/tf/1e9:6.1f} TF/s)   bwd {tb*1e3:8.1f} us ({2.5*fl/tb/1e9:6.1f} TF/s on 5 products)   bwd with dS hand-off {td*1e3:8.1f} us")
