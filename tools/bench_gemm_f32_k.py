"""dev: gg_gemm_nt_f32 time vs K at fixed M, N (per-tile overhead = intercept, per-k cost = slope)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import ops
for M, N in [(3211264, 384), (3211264, 96), (802816, 768)]:
    out = torch.empty(M, N, device="cuda")
    for K in (16, 32, 48, 64, 96, 128, 192, 256, 384):
        A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda") * 0.05
        for _ in range(2): ops.gemm_nt(A, B, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): ops.gemm_nt(A, B, out=out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"M={M} N={N} K={K:4d} {dt*1e6:8.1f} us  {2.0*M*N*K/dt/1e12:6.1f} TF/s  {4.0*(M*K+M*N)/dt/1e9:7.1f} GB/s", flush=True)
        del A, B
    del out
