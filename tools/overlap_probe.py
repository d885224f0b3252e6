"""dev: do an HBM-bound kernel stream and an MFMA-bound (fp32 TN weight-gradient) kernel stream overlap when issued on two HIP streams?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import ops
torch.manual_seed(0)
M0, C0 = 3211264, 96
y = torch.randn(M0, C0, device="cuda"); res = torch.randn(M0, C0, device="cuda")
stat = torch.stack([torch.zeros(C0), torch.ones(C0)]).cuda(); g = torch.ones(C0, device="cuda"); b = torch.zeros(C0, device="cuda")
M3 = 50176
dY = torch.randn(M3, 2304, device="cuda"); X = torch.randn(M3, 576, device="cuda")
def hbm(n=12):
    for _ in range(n): ops.bn_apply(y, stat, g, b, residual=res)
def mfma(n=8):
    for _ in range(n): ops.gemm_tn(dY, X)
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
print(f"HBM-bound stream alone {t_h:.2f} ms, TN GEMMs alone {t_m:.2f} ms, serial sum {t_h + t_m:.2f} ms, on two streams {t_b:.2f} ms")
def hbm2(n=8):
    for _ in range(n): ops.gemm_nt(dY[:, :576].contiguous() if False else X, X[:2304])   # MFMA-bound NT GEMM instead of the HBM stream
