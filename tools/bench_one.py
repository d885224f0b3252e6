import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import ops
M, N, K = [int(v) for v in sys.argv[1:4]]
A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
for _ in range(3): ops.gemm_nt(A, W, out=out)
torch.cuda.synchronize()
