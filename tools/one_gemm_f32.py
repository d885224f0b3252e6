"""dev: a few launches of one gg_gemm_nt_f32 shape (for rocprofv3 --pmc runs): python tools/one_gemm_f32.py M N K"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import ops
M, N, K = (int(a) for a in sys.argv[1:4])
A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda") * 0.05; out = torch.empty(M, N, device="cuda")
for _ in range(4): ops.gemm_nt(A, B, out=out)
torch.cuda.synchronize()
