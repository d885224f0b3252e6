"""dev: gg_gemm_nt_f32 time for a grid of exactly T tiles (N = 128, M = 128 T): one / two / three workgroups per CU, several K --
separates a workgroup's exposed overhead (launch + first loads + epilogue) from its k-loop time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import ops
for K in (16, 96, 384, 1536):
    for tiles in (256, 512, 768, 1536, 2304, 7680):
        M, N = 128 * tiles, 128
        A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda") * 0.05; out = torch.empty(M, N, device="cuda")
        for _ in range(3): ops.gemm_nt(A, B, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n): ops.gemm_nt(A, B, out=out)
        e1.record(); torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) / n * 1e3
        print(f"K={K:5d} tiles={tiles:5d} ({tiles/256:4.1f}/CU) {dt:8.1f} us   {dt/ (tiles/256):7.2f} us per tile-per-CU   {2.0*M*N*K/dt/1e6:6.1f} TF/s", flush=True)
        del A, B, out
