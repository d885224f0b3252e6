# dev: memory-side / matrix-side ablations of the LDS-DMA GEMM (GG_GEMM_DEBUG: 1 = no operand DMA, 2 = no stores, 32 = no MFMAs)
cd $GRAFT_REPO_ROOT
for d in 0 1 32 2 34 35; do
  echo "== GG_GEMM_DEBUG=$d"
  GG_DEV_SWITCHES=1 GG_GEMM_DEBUG=$d timeout -k 10 120 python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from geoguessr_ai_amd import ops
T = 1024 * 50
for name, M, N, K in [("sq8k", 8192, 8192, 8192), ("c4.qkv", T, 2304, 768), ("c4.fc2", T, 768, 3072), ("s2.fc1", 200704, 1536, 384)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    for v in ("1", "9"):
        os.environ["GG_GEMM_DMA"] = v
        ops.gemm_nt(A, W, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.gemm_nt(A, W, out=out)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"  {name:8s} v{v} {ms*1e3:8.1f} us  = {2.0*M*N*K/ms/1e9:7.0f} TF-equivalent, operand bytes into LDS {((M/256)*(N/128)*(K/32)*24576)/ms/1e9:7.2f} TB/s", flush=True)
PY
done
