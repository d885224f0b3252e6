#!/usr/bin/env python3
"""Per-shape timing of the GEMM kernel on the shapes of the TinyViT-21M-224 step at 1024 images (dev tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import ops

B = 1024
M1, M0, Ms1, Ms2, Ms3 = B * 112 * 112, B * 56 * 56, B * 28 * 28, B * 14 * 14, B * 7 * 7
shapes = [
    ("pe.conv1", M1, 48, 32, {}), ("pe.conv2", M0, 96, 432, {}),
    ("mb.conv1", M0, 384, 96, {}), ("mb.conv3", M0, 96, 384, {}),
    ("mb.conv3.dgrad", M0, 384, 96, {}), ("mb.conv1.dgrad", M0, 96, 384, {"residual": True}),
    ("s1.merge1", M0, 192, 96, {}), ("s1.qkv", Ms1, 576, 192, {"bias": True}), ("s1.proj", Ms1, 192, 192, {"bias": True, "residual": True}),
    ("s1.fc1", Ms1, 768, 192, {"bias": True, "act": "gelu", "preact": True}), ("s1.fc2", Ms1, 192, 768, {"bias": True, "residual": True}),
    ("s1.fc2.dgrad", Ms1, 768, 192, {"dact": True}),
    ("s2.qkv", Ms2, 1152, 384, {"bias": True}), ("s2.fc1", Ms2, 1536, 384, {"bias": True, "act": "gelu", "preact": True}),
    ("s2.fc1.nopre", Ms2, 1536, 384, {"bias": True, "act": "gelu"}), ("s2.fc1.bias", Ms2, 1536, 384, {"bias": True}), ("s2.fc1.plain", Ms2, 1536, 384, {}),
    ("s2.fc2", Ms2, 384, 1536, {"bias": True, "residual": True}), ("s2.fc2.dgrad", Ms2, 1536, 384, {"dact": True}),
    ("s3.fc1", Ms3, 2304, 576, {"bias": True, "act": "gelu", "preact": True}), ("s3.fc2", Ms3, 576, 2304, {"bias": True, "residual": True}),
    ("head", 256, 12647, 576, {"bias": True, "out_f32": True}),
    ("square4k", 4096, 4096, 4096, {}),
]
for name, M, N, K, o in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    kw = {}
    if o.get("bias"): kw["bias"] = torch.randn(N, device="cuda")
    if o.get("act"): kw["act"] = o["act"]
    if o.get("preact"): kw["preact"] = True
    if o.get("residual"): kw["residual"] = torch.randn(M, N, device="cuda").bfloat16()
    if o.get("dact"): kw["dact_preact"], kw["dact"] = torch.randn(M, N, device="cuda").bfloat16(), "gelu"
    if o.get("out_f32"): kw["out_f32"] = True
    out = torch.empty((M, N), dtype=torch.float32 if o.get("out_f32") else torch.bfloat16, device="cuda")
    for _ in range(2): ops.gemm_nt(A, W, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n): ops.gemm_nt(A, W, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    byt = 2 * (M * K + N * K + M * N * (1 + bool(o.get("preact")) + bool(o.get("residual")) + bool(o.get("dact"))))
    if o.get("out_f32"): byt += 2 * M * N
    print(f"{name:16s} M={M:9d} N={N:5d} K={K:5d}  {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TF/s  {byt/ms/1e6:8.1f} GB/s")
    del A, W, out, kw

# ---- BatchNorm-backward-fused dgrads (raw GemmArgs so that only the kernel is timed) ----
import ctypes as C
from geoguessr_ai_amd import _lib as L


def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, M, N, K in [("mb.c3.dgrad+bnbwd", M0, 384, 96), ("mg1.c3.dgrad+bnbwd", Ms1, 192, 192), ("mg2.c3.dgrad+bnbwd", Ms2, 384, 384)]:
    dY = torch.randn(M, K, device="cuda").bfloat16(); Wt = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    y = torch.randn(M, N, device="cuda").bfloat16(); dz = torch.empty_like(y)
    stat = torch.stack([torch.zeros(N), torch.ones(N)]).cuda(); g = torch.ones(N, device="cuda"); b = torch.zeros(N, device="cuda")
    rows = L.lib().gg_gemm_colstats_rows(M)
    part = torch.zeros((L.lib().gg_stat_rows_capacity(rows), 2, N), device="cuda")
    for act in (1, 0):
        a = L.GemmArgs()
        a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.M, a.N, a.K = dY.data_ptr(), K, Wt.data_ptr(), K, dz.data_ptr(), N, M, N, K
        a.bn_y, a.bn_stat, a.bn_gamma, a.bn_beta, a.bn_act, a.colstats, a.split_k = y.data_ptr(), stat.data_ptr(), g.data_ptr(), b.data_ptr(), act, part.data_ptr(), 1
        ms = timed(lambda: L.check(L.lib().gg_gemm_nt(C.byref(a), L.stream())))
        byt = 2 * (M * K + 2 * M * N)
        print(f"{name:20s} act={act} M={M:9d} N={N:5d} K={K:5d}  {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TF/s  {byt/ms/1e6:8.1f} GB/s")
    a = L.GemmArgs()
    a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.M, a.N, a.K = dY.data_ptr(), K, Wt.data_ptr(), K, dz.data_ptr(), N, M, N, K
    a.colstats, a.split_k = part.data_ptr(), 1
    ms = timed(lambda: L.check(L.lib().gg_gemm_nt(C.byref(a), L.stream())))
    print(f"{name:20s} plain+stats            {ms*1e3:9.1f} us")
    del dY, Wt, y, dz, part
for name, M, Cin, Cout in [("mb.c1.dgrad.fold", M0, 96, 384), ("mg1.c1.dgrad.fold", M0, 96, 192), ("mg2.c1.dgrad.fold", Ms1, 192, 384)]:
    dz = torch.randn(M, Cout, device="cuda").bfloat16(); y = torch.randn(M, Cout, device="cuda").bfloat16()
    Bf = (torch.randn(Cin, 2 * Cout, device="cuda") * 0.05).bfloat16(); bias = torch.zeros(Cin, device="cuda")
    dx = torch.empty(M, Cin, device="cuda").bfloat16(); res = torch.randn(M, Cin, device="cuda").bfloat16()
    a = L.GemmArgs()
    a.A, a.lda, a.A2, a.k_split, a.B, a.ldb, a.C, a.ldc = dz.data_ptr(), Cout, y.data_ptr(), Cout, Bf.data_ptr(), 2 * Cout, dx.data_ptr(), Cin
    a.M, a.N, a.K, a.bias, a.residual, a.ldr, a.split_k = M, Cin, 2 * Cout, bias.data_ptr(), res.data_ptr(), Cin, 1
    ms = timed(lambda: L.check(L.lib().gg_gemm_nt(C.byref(a), L.stream())))
    byt = 2 * (2 * M * Cout + 2 * M * Cin)
    print(f"{name:20s} M={M:9d} N={Cin:5d} K={2*Cout:5d}  {ms*1e3:9.1f} us  {2*M*Cin*2*Cout/ms/1e9:8.1f} TF/s  {byt/ms/1e6:8.1f} GB/s")
    del dz, y, Bf, dx, res

# ---- TN (weight-gradient) GEMMs: kernel + slab reduction ----
for name, M, N, K, rs in [("s3.fc1.wgrad", Ms3, 2304, 576, True), ("s3.fc2.wgrad", Ms3, 576, 2304, True), ("s3.qkv.wgrad", Ms3, 1728, 576, True),
                          ("s3.proj.wgrad", Ms3, 576, 576, True), ("pe.conv2.wgrad", M0, 96, 432, False), ("pe.conv1.wgrad", M1, 48, 32, False)]:
    dY = torch.randn(M, N, device="cuda").bfloat16(); X = torch.randn(M, K, device="cuda").bfloat16()
    scale = torch.rand(B, device="cuda") if rs else None
    ms = timed(lambda: ops.gemm_tn(dY, X, rowscale=scale, rows_per_scale=M // B if rs else 0))
    print(f"{name:16s} M={M:9d} N={N:5d} K={K:5d}  {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TF/s  {2*M*(N+K)/ms/1e6:8.1f} GB/s")
    del dY, X
