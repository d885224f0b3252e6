"""dev: does an HBM-bound kernel stream overlap with the split GEMM (144 KB of LDS, 8 waves per CU) when the two are issued on two HIP streams?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from geoguessr_ai_amd import ops, _lib as L
lib = L.lib()
torch.manual_seed(0)
M0, C0 = 3211264, 96
y = torch.randn(M0, C0, device="cuda"); res = torch.randn(M0, C0, device="cuda")
stat = torch.stack([torch.zeros(C0), torch.ones(C0)]).cuda(); g = torch.ones(C0, device="cuda"); b = torch.zeros(C0, device="cuda")
M, N, K = 200704, 384, 1536
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * K ** -0.5; out = torch.empty(M, N, device="cuda")
Wp = torch.empty(3, N, K, dtype=torch.bfloat16, device="cuda")
L.check(lib.gg_split3_bf16(W.data_ptr(), N, K, K, Wp.data_ptr(), L.stream()))
a = L.Split3Args()
a.b_planes, a.ldb, a.M, a.N, a.K, a.C, a.ldc = Wp.data_ptr(), K, M, N, K, out.data_ptr(), N
xl = torch.randn(M, 384, device="cuda"); gl = torch.ones(384, device="cuda"); bl = torch.zeros(384, device="cuda")
def hbm(n=10):
    for _ in range(n): ops.bn_apply(y, stat, g, b, residual=res)
def mfma(n=8):
    for _ in range(n): L.check(lib.gg_gemm_nt_split3_af32(C.byref(a), A.data_ptr(), K, 0, L.stream()))
s2 = torch.cuda.Stream()
def timed(f):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
t_h, t_m = timed(hbm), timed(mfma)
def both():
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(s2):
        s2.wait_event(ev)
        mfma()
    hbm()
    torch.cuda.current_stream().wait_stream(s2)
t_b = timed(both)
print(f"BatchNorm-apply stream alone {t_h:.2f} ms, split GEMMs alone {t_m:.2f} ms, serial sum {t_h + t_m:.2f} ms, on two streams {t_b:.2f} ms (overlap {(t_h + t_m - t_b) / min(t_h, t_m):.0%} of the shorter)")
