#!/usr/bin/env python3
"""dev: two launches each of the split-product kernels that carry the fp32_split step, at the 1024-image TinyViT-21M shapes -- the program tools/pmc_split.sh profiles
(SQ counters per kernel).  NT: stage-2 fc2 (split3a<0,4>: 256 x 128 tiles), stage-1 fc1 (split3b<4>: 128 x 128, two workgroups per CU), stage-1 fc2 dgrad-like
N = 192 (96-column tiles); TN: stage-3 fc1 weight gradient; window attention 14 x 14 (stage 2) and 7 x 7 (stage 1) forward + backward."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from geoguessr_ai_amd import _lib as L

lib = L.lib()
REP = int(os.environ.get("REP", "2"))
only = sys.argv[1:]


def want(name):
    return not only or any(name.startswith(o) for o in only)


def planes(x):
    out = torch.empty((3,) + tuple(x.shape), dtype=torch.bfloat16, device=x.device)
    L.check(lib.gg_split3_bf16(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), out.data_ptr(), L.stream()), "gg_split3_bf16")
    return out


for name, M, N, K in [("nt.s2fc2", 200704, 384, 1536), ("nt.s1fc1", 802816, 768, 192), ("nt.s1fc2", 802816, 192, 768), ("nt.s3qkv", 50176, 1728, 576), ("nt.s2fc1", 200704, 1536, 384), ("nt.s3fc1", 50176, 2304, 576)]:
    if not want(name): continue
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda") * K ** -0.5; bias = torch.randn(N, device="cuda")
    Bp = planes(B); out = torch.empty(M, N, device="cuda")
    a = L.Split3Args()
    a.b_planes, a.ldb, a.M, a.N, a.K, a.C, a.ldc, a.bias = Bp.data_ptr(), K, M, N, K, out.data_ptr(), N, bias.data_ptr()
    for _ in range(REP):
        L.check(lib.gg_gemm_nt_split3_af32(C.byref(a), A.data_ptr(), K, 0, L.stream()), name)
    torch.cuda.synchronize()
    del A, B, Bp, out
if want("tn"):
    M, N, K = 50176, 2304, 576
    dY = torch.randn(M, N, device="cuda"); X = torch.randn(M, K, device="cuda"); scratch = torch.empty(32 << 20, device="cuda")
    s = lib.gg_gemm_tn_split3_splits(M, N, K)
    for _ in range(REP):
        L.check(lib.gg_gemm_tn_split3(dY.data_ptr(), N, X.data_ptr(), K, M, N, K, None, 0, scratch.data_ptr(), s, L.stream()), "tn")
    torch.cuda.synchronize()
    del dY, X
Bt = 1024
for name, res, ws, Cc, nh in [("attn.s2", 14, 14, 384, 12), ("attn.s1", 28, 7, 192, 6)]:
    if not want(name): continue
    M, N = Bt * res * res, ws * ws
    qkv = torch.randn(M, 3 * Cc, device="cuda"); out = torch.empty(M, Cc, device="cuda"); dout = torch.randn(M, Cc, device="cuda")
    dqkv = torch.empty_like(qkv); lse = torch.empty(M, nh, device="cuda"); table = torch.randn(nh, N, device="cuda") * 0.1
    a = L.AttnArgs()
    a.qkv, a.ld, a.q_off, a.k_off, a.v_off, a.head_stride, a.head_dim = qkv.data_ptr(), 3 * Cc, 0, 32, 64, 96, 32
    a.num_heads, a.num_windows, a.tokens_per_window = nh, Bt * (res // ws) ** 2, N
    a.window_size, a.map_h, a.map_w = ws, res, res
    a.bias_table = table.data_ptr(); a.scale = 32 ** -0.5
    a.out, a.ldo, a.lse = out.data_ptr(), Cc, lse.data_ptr()
    a.dout, a.lddo, a.dqkv = dout.data_ptr(), Cc, dqkv.data_ptr()
    for _ in range(REP):
        L.check(lib.gg_attention_flash_fwd(C.byref(a), 1, L.stream()), name)
        L.check(lib.gg_attention_flash_bwd(C.byref(a), 1, L.stream()), name)
    torch.cuda.synchronize()
print("done")
