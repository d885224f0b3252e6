"""dev: a few launches of one bf16 gg_gemm_nt shape (for rocprofv3 runs): python tools/one_gemm16.py M N K [bias] [residual] [gelu]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import ops
M, N, K = (int(a) for a in sys.argv[1:4])
fl = sys.argv[4:]
A = torch.randn(M, K, device="cuda").bfloat16(); B = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
kw = {}
if "bias" in fl: kw["bias"] = torch.randn(N, device="cuda")
if "residual" in fl: kw["residual"] = torch.randn(M, N, device="cuda").bfloat16()
if "gelu" in fl: kw["act"] = "gelu"
if "qgelu" in fl: kw["act"] = "quick_gelu"
for _ in range(6): ops.gemm_nt(A, B, out=out, **kw)
torch.cuda.synchronize()
