#!/usr/bin/env python3
"""dev: per-workgroup cycle trace of the LDS-DMA bf16 GEMM (GG_GEMM_DMA=6: wave 0 of every workgroup stamps its phases): python tools/trace_gemm16.py M N K"""
import os, sys, ctypes as C
os.environ["GG_DEV_SWITCHES"] = "1"; os.environ["GG_GEMM_DMA"] = "6"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from geoguessr_ai_amd import _lib as L
M, N, K = (int(x) for x in sys.argv[1:4])
A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
tiles = ((M + 191) // 192) * ((N + 127) // 128)
tr = torch.zeros((tiles, 8), dtype=torch.int64, device="cuda")
a = L.GemmArgs()
a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.M, a.N, a.K = A.data_ptr(), K, W.data_ptr(), K, out.data_ptr(), N, M, N, K
a.colstats, a.split_k = tr.data_ptr(), 1
for _ in range(3): L.check(L.lib().gg_gemm_nt(C.byref(a), L.stream()))
torch.cuda.synchronize()
t = tr.cpu().numpy().astype(np.float64)
nk = (K + 63) // 64
names = ["first operands (start -> stage 0 landed)", "k-loop", "  of it: waiting at the sync point (vmcnt + lgkm + barrier)", "  of it: issuing the next stage's DMA", "epilogue (issue + store drain)"]
print(f"M={M} N={N} K={K}: {tiles} tiles, {nk} stages; MFMA time of a k-loop at one wave per SIMD: {nk * 48 * 16} cycles")
for i, n in enumerate(names):
    print(f"  {n:62s} median {np.median(t[:, i]):9.0f}  p10 {np.percentile(t[:, i], 10):9.0f}  p90 {np.percentile(t[:, i], 90):9.0f}   per stage {np.median(t[:, i]) / nk:7.0f}")
life = t[:, 0] + t[:, 1] + t[:, 4]
ti = tr.cpu().numpy()
hw = ti[:, 6]
cu = ((hw >> 32) & 0xF) * 4096 + ((hw >> 13) & 7) * 512 + ((hw >> 12) & 1) * 256 + ((hw >> 8) & 0xF) * 16      # (xcc, se, sh, cu)
slot = cu + (hw & 0xF)
start, end = ti[:, 5].astype(np.float64), ti[:, 7].astype(np.float64)        # 100 MHz ticks
span = (end.max() - start.min()) * 10e-9
clk = np.median(life / np.maximum(end - start, 1)) * 100e6
print(f"  workgroup lifetime median {np.median(life):.0f} cycles = {np.median(end - start) / 100:.1f} us -> shader clock {clk / 1e9:.2f} GHz; launch span {span * 1e6:.1f} us; {len(np.unique(cu))} CUs, {len(np.unique(slot))} (CU, wave slot) pairs")
gaps, busy = [], []
for sl in np.unique(slot):
    m = slot == sl
    o = np.argsort(start[m]); s0, e0 = start[m][o], end[m][o]
    gaps += list((s0[1:] - e0[:-1]) / 100)
    busy.append((e0 - s0).sum() / (e0.max() - s0.min()))
print(f"  per wave slot: gap between a workgroup's end and its successor's start median {np.median(gaps):.2f} us (p90 {np.percentile(gaps, 90):.2f}); slot occupied {np.mean(busy):.2f} of its span; tiles per slot {tiles / len(np.unique(slot)):.1f}")
first_start = np.array([start[slot == sl].min() for sl in np.unique(slot)]); last_end = np.array([end[slot == sl].max() for sl in np.unique(slot)])
print(f"  first start spread {(first_start.max() - first_start.min()) / 100:.1f} us; last end spread {(last_end.max() - last_end.min()) / 100:.1f} us")
