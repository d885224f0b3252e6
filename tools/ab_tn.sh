# dev: A/B of two builds on the fp32 weight-gradient (TN) GEMM shapes (see tools/ab_lib.sh)
cd $GRAFT_REPO_ROOT
cat > /tmp/tnb.py <<'P'
import sys, os, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from geoguessr_ai_amd import ops
for name, M, N, K in [("pe1 wgrad", 12845056, 48, 32), ("pe2 wgrad", 3211264, 96, 432), ("s3.fc1 wgrad", 50176, 2304, 576), ("s3.fc2 wgrad", 50176, 576, 2304), ("s3.qkv wgrad", 50176, 1728, 576), ("s3.proj wgrad", 50176, 576, 576),
                      ("merge2.c3 wgrad", 50176, 576, 2304), ("s2.fc1 wgrad", 200704, 1536, 384)]:
    dY = torch.randn(M, N, device="cuda"); X = torch.randn(M, K, device="cuda")
    for _ in range(2): ops.gemm_tn(dY, X)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ops.gemm_tn(dY, X)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name:18s} M={M:8d} N={N:5d} K={K:5d}  {dt*1e6:9.1f} us  {2.0*M*N*K/dt/1e12:7.1f} TF/s")
    del dY, X
P
for i in 1 2; do
echo prev; GG_LIB=$PWD/tools/bin/libgg_prev.so python /tmp/tnb.py 2>/dev/null | grep " us "
echo new; python /tmp/tnb.py 2>/dev/null | grep " us "
done
