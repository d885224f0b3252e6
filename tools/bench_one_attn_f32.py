#!/usr/bin/env python3
"""dev: ONE fp32 flash-attention shape, a few launches (for rocprofv3 --pmc / --kernel-trace): stage s1|s2|s3, which = fwd|bwd|both."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from geoguessr_ai_amd import _lib as L
stage = sys.argv[1] if len(sys.argv) > 1 else "s2"
which = sys.argv[2] if len(sys.argv) > 2 else "both"
B = int(os.environ.get("GG_B", "1024"))
res, ws, Cc, nh = {"s1": (28, 7, 192, 6), "s2": (14, 14, 384, 12), "s3": (7, 7, 576, 18)}[stage]
M, N = B * res * res, ws * ws
qkv = torch.randn(M, 3 * Cc, device="cuda"); out = torch.empty(M, Cc, device="cuda"); dout = torch.randn(M, Cc, device="cuda")
dqkv = torch.empty_like(qkv); lse = torch.empty(M, nh, device="cuda"); table = torch.randn(nh, N, device="cuda") * 0.1
a = L.AttnArgs()
a.qkv, a.ld, a.q_off, a.k_off, a.v_off, a.head_stride, a.head_dim = qkv.data_ptr(), 3 * Cc, 0, 32, 64, 96, 32
a.num_heads, a.num_windows, a.tokens_per_window = nh, B * (res // ws) ** 2, N
a.window_size, a.map_h, a.map_w = ws, res, res
a.bias_table = table.data_ptr(); a.scale = 32 ** -0.5
a.out, a.ldo, a.lse = out.data_ptr(), Cc, lse.data_ptr()
a.dout, a.lddo, a.dqkv = dout.data_ptr(), Cc, dqkv.data_ptr()
if os.environ.get("GG_DS"):
    ds = torch.empty(L.lib().gg_attention_flash_ds_scratch_floats(a.num_windows, nh, N), device="cuda")
    a.ds_scratch = ds.data_ptr()
L.check(L.lib().gg_attention_flash_fwd(C.byref(a), 1, L.stream()))
for _ in range(3):
    if which in ("fwd", "both"): L.check(L.lib().gg_attention_flash_fwd(C.byref(a), 1, L.stream()))
    if which in ("bwd", "both"): L.check(L.lib().gg_attention_flash_bwd(C.byref(a), 1, L.stream()))
torch.cuda.synchronize()
