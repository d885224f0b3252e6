#!/usr/bin/env python3
"""One shape of the BatchNorm-backward-fused dgrad GEMM, a few launches (dev tool for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from geoguessr_ai_amd import _lib as L

M, N, K = 1024 * 56 * 56, 384, 96
act = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dY = torch.randn(M, K, device="cuda").bfloat16(); Wt = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
y = torch.randn(M, N, device="cuda").bfloat16(); dz = torch.empty_like(y)
stat = torch.stack([torch.zeros(N), torch.ones(N)]).cuda(); g = torch.ones(N, device="cuda"); b = torch.zeros(N, device="cuda")
rows = L.lib().gg_gemm_colstats_rows(M)
part = torch.zeros((L.lib().gg_stat_rows_capacity(rows), 2, N), device="cuda")
a = L.GemmArgs()
a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.M, a.N, a.K = dY.data_ptr(), K, Wt.data_ptr(), K, dz.data_ptr(), N, M, N, K
if act >= 0:
    a.bn_y, a.bn_stat, a.bn_gamma, a.bn_beta, a.bn_act = y.data_ptr(), stat.data_ptr(), g.data_ptr(), b.data_ptr(), act
a.colstats, a.split_k = part.data_ptr(), 1
for _ in range(3):
    L.check(L.lib().gg_gemm_nt(C.byref(a), L.stream()))
torch.cuda.synchronize()
