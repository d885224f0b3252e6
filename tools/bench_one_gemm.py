#!/usr/bin/env python3
"""One GEMM shape, a few launches (dev tool for rocprofv3 --pmc runs): M N K [bias] [gelu] [preact]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4])
flags = sys.argv[4:]
A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
kw = {}
if "bias" in flags: kw["bias"] = torch.randn(N, device="cuda")
if "gelu" in flags: kw["act"] = "gelu"
if "preact" in flags: kw["preact"] = True
out = torch.empty(M, N, device="cuda").bfloat16()
for _ in range(3): ops.gemm_nt(A, W, out=out, **kw)
torch.cuda.synchronize()
