#!/usr/bin/env python3
"""dev: MBConv.conv3 forward at the 1024-image size (M = 3 211 264, N = 96, K = 384) with BatchNorm2 + GELU of the depthwise output formed while the GEMM stages A:
gg_gemm_nt_f32's prologue kernel against gg_gemm_nt_split3_af32_pro, both with the BatchNorm partials of the result."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from geoguessr_ai_amd import _lib as L
lib = L.lib()


def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M, N, K in [(3211264, 96, 384), (802816, 192, 768)]:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * K ** -0.5
    stat = torch.stack([torch.randn(K) * 0.3, torch.rand(K) + 0.5]).contiguous().cuda(); gm = torch.randn(K, device="cuda"); bt = torch.randn(K, device="cuda") * 0.2
    Wp = torch.empty(3, N, K, dtype=torch.bfloat16, device="cuda")
    L.check(lib.gg_split3_bf16(W.data_ptr(), N, K, K, Wp.data_ptr(), L.stream()))
    parts = lib.gg_gemm_colstats_rows(M)
    stats = torch.zeros(lib.gg_stat_rows_capacity(parts), 2, N, device="cuda")
    o1, o2 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ga = L.GemmArgs()
    ga.A, ga.lda, ga.B, ga.ldb, ga.C, ga.ldc, ga.M, ga.N, ga.K = A.data_ptr(), K, W.data_ptr(), K, o1.data_ptr(), N, M, N, K
    ga.a_bn_stat, ga.a_bn_gamma, ga.a_bn_beta, ga.a_bn_act, ga.colstats = stat.data_ptr(), gm.data_ptr(), bt.data_ptr(), 1, stats.data_ptr()
    a = L.Split3Args()
    a.b_planes, a.ldb, a.M, a.N, a.K, a.C, a.ldc = Wp.data_ptr(), K, M, N, K, o2.data_ptr(), N
    t32 = timed(lambda: L.check(lib.gg_gemm_nt_f32(C.byref(ga), L.stream())))
    t3 = timed(lambda: L.check(lib.gg_gemm_nt_split3_af32_pro(C.byref(a), A.data_ptr(), K, 0, stat.data_ptr(), gm.data_ptr(), bt.data_ptr(), 1, stats.data_ptr(), L.stream())))
    rel = float((o1[:4096].double() - o2[:4096].double()).norm() / o1[:4096].double().norm())
    gb = 4.0 * (M * K + M * N) / 1e9
    print(f"{M} x {N} x {K}: f32-MFMA prologue GEMM {t32*1e3:8.1f} us ({gb/t32:6.0f} GB/s)   split prologue GEMM {t3*1e3:8.1f} us ({gb/t3:6.0f} GB/s)   x{t32/t3:.2f}   rel diff {rel:.1e}", flush=True)
    del A, o1, o2
