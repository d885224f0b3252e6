#!/usr/bin/env python3
"""dev: cost of the epilogue classes of the 16-bit NT GEMM on one shape (old / LDS-DMA form)."""
import os, sys
os.environ.setdefault("GG_DEV_SWITCHES", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import ops
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (51200, 2304, 768)))
A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda").bfloat16(); pre = torch.randn(M, N, device="cuda").bfloat16()
rs = torch.rand(M // 50, device="cuda")
cases = {"plain": {}, "bias": {"bias": bias}, "bias+res": {"bias": bias, "residual": res}, "bias+rowscale+res": {"bias": bias, "residual": res, "rowscale": rs, "rows_per_scale": 50},
         "qgelu": {"bias": bias, "act": "quick_gelu"}, "gelu": {"bias": bias, "act": "gelu"}, "gelu+preact": {"bias": bias, "act": "gelu", "preact": True},
         "dgelu": {"dact_preact": pre, "dact": "gelu"}, "f32out": {"bias": bias, "out_f32": True}}
for name, kw in cases.items():
    o = torch.empty((M, N), dtype=torch.float32, device="cuda") if kw.get("out_f32") else out
    r = {}
    for f in "01":
        os.environ["GG_GEMM_DMA"] = f
        ts = []
        for _ in range(3):
            ops.gemm_nt(A, W, out=o, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): ops.gemm_nt(A, W, out=o, **kw)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 3)
        r[f] = sorted(ts)[1]
    print(f"{name:18s} old {r['0']*1e3:7.1f} us {2.0*M*N*K/r['0']/1e9:6.0f} TF | dma {r['1']*1e3:7.1f} us {2.0*M*N*K/r['1']/1e9:6.0f} TF", flush=True)
