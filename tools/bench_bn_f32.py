"""dev: fp32 BatchNorm apply / backward passes at the stage-0 and stage-2 sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import ops
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M, C in [(3211264, 384), (3211264, 96), (802816, 192), (200704, 384)]:
    y = torch.randn(M, C, device="cuda"); res = torch.randn(M, C, device="cuda"); d = torch.randn(M, C, device="cuda")
    stat = torch.stack([torch.zeros(C), torch.ones(C)]).cuda(); g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    by = 4.0 * M * C
    t1 = timed(lambda: ops.bn_apply(y, stat, g, b, act="gelu"))
    t2 = timed(lambda: ops.bn_apply(y, stat, g, b, act="gelu", residual=res))
    t3 = timed(lambda: ops.bn_bwd(d, y, stat, g, b, act="gelu", want_param_grads=False))
    print(f"M={M} C={C}: apply {t1*1e3:7.1f} us ({2*by/t1/1e6:5.0f} GB/s)  apply+res {t2*1e3:7.1f} us ({3*by/t2/1e6:5.0f} GB/s)  bwd(reduce+apply) {t3*1e3:7.1f} us ({6*by/t3/1e6:5.0f} GB/s)")
    del y, res, d
