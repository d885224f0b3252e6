#!/usr/bin/env python3
"""One window-attention shape, a few launches (dev tool for rocprofv3 --pmc): stage (s1|s2|s3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from geoguessr_ai_amd import _lib as L
B = 1024
res, ws, Cc, nh = {"s1": (28, 7, 192, 6), "s2": (14, 14, 384, 12), "s3": (7, 7, 576, 18)}[sys.argv[1] if len(sys.argv) > 1 else "s2"]
M, N = B * res * res, ws * ws
qkv = torch.randn(M, 3 * Cc, device="cuda").bfloat16(); out = torch.empty(M, Cc, device="cuda").bfloat16()
dout = torch.randn(M, Cc, device="cuda").bfloat16(); dqkv = torch.empty_like(qkv); lse = torch.empty(M, nh, device="cuda")
table = torch.randn(nh, N, device="cuda") * 0.1
Np = L.lib().gg_attention_padded_tokens(N)
full = torch.empty(nh, Np, Np, device="cuda").bfloat16()
L.check(L.lib().gg_attention_expand_bias(table.data_ptr(), nh, ws, 32 ** -0.5, full.data_ptr(), L.stream()))
a = L.AttnArgs()
a.qkv, a.ld, a.q_off, a.k_off, a.v_off, a.head_stride, a.head_dim = qkv.data_ptr(), 3 * Cc, 0, 32, 64, 96, 32
a.num_heads, a.num_windows, a.tokens_per_window = nh, B * (res // ws) ** 2, N
a.window_size, a.map_h, a.map_w = ws, res, res
a.bias, a.scale = full.data_ptr(), 32 ** -0.5
a.out, a.ldo, a.lse = out.data_ptr(), Cc, lse.data_ptr()
a.dout, a.lddo, a.dqkv = dout.data_ptr(), Cc, dqkv.data_ptr()
for _ in range(3):
    L.check(L.lib().gg_attention_fwd(C.byref(a), L.stream()))
    L.check(L.lib().gg_attention_bwd(C.byref(a), L.stream()))
torch.cuda.synchronize()
