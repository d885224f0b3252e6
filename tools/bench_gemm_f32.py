"""dev: time gg_gemm_nt_f32 / gg_gemm_tn_f32 on the model's shapes (1024 images)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import ops
F32 = torch.float32
if "--quick" in sys.argv:
    shapes = None
shapes = [("4096^3", 4096, 4096, 4096, {}), ("s2.fc1 gelu+pre", 200704, 1536, 384, dict(act="gelu", preact=True, bias=True)),
          ("s2.fc2 res", 200704, 384, 1536, dict(bias=True, residual=True)), ("s2.qkv", 200704, 1152, 384, dict(bias=True)),
          ("s0.conv1 stats", 3211264, 384, 96, dict(colstats=True)), ("s0.conv3 stats", 3211264, 96, 384, dict(colstats=True)),
          ("s0.conv1 plain", 3211264, 384, 96, {}), ("s1.fc1", 802816, 768, 192, dict(act="gelu", preact=True, bias=True)),
          ("pe2 stats", 3211264, 96, 432, dict(colstats=True)), ("s3.fc1", 50176, 2304, 576, dict(act="gelu", preact=True, bias=True)),
          ("s3.fc2 res", 50176, 576, 2304, dict(bias=True, residual=True)), ("s1.fc2 res", 802816, 192, 768, dict(bias=True, residual=True)),
          ("s1.qkv", 802816, 576, 192, dict(bias=True)), ("s0.mb.c1 dgrad", 3211264, 64, 256, {}), ("head", 1024, 12647, 576, dict(bias=True)),
          ("s2.fc2 dgrad dgelu", 200704, 1536, 384, dict(dgelu=True)), ("s1.fc2 dgrad dgelu", 802816, 768, 192, dict(dgelu=True)),
          ("s0.conv3 dgrad bnbwd", 3211264, 384, 96, dict(bnbwd=True)), ("s0.conv1 dgrad 2src", 3211264, 96, 384, dict(twosrc=True)),
          ("s0.conv3 fwd pro", 3211264, 96, 384, dict(pro=True))]
if "--quick" in sys.argv:
    shapes = [sh for sh in shapes if sh[0] in ("4096^3", "s2.qkv", "s0.conv1 plain", "s2.fc1 gelu+pre")] + [("8192x8192x4096", 8192, 8192, 4096, {})]
for name, M, N, K, kw in shapes:
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda") * 0.05
    kw = dict(kw)
    if kw.pop("bias", False): kw["bias"] = torch.randn(N, device="cuda")
    if kw.pop("residual", False): kw["residual"] = torch.randn(M, N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    fn = lambda: ops.gemm_nt(A, B, out=out, **kw)
    if kw.pop("dgelu", False):
        kw["dact_preact"] = torch.randn(M, N, device="cuda"); kw["dact"] = "gelu"
    if kw.pop("bnbwd", False):
        y = torch.randn(M, N, device="cuda"); stat = torch.stack([torch.zeros(N), torch.ones(N)]).cuda(); g_ = torch.ones(N, device="cuda"); b_ = torch.zeros(N, device="cuda")
        fn = lambda: ops.conv_dgrad_bn_bwd(A, B, y, stat, g_, b_, act="gelu", want_param_grads=False)
    if kw.pop("twosrc", False):
        y = torch.randn(M, K, device="cuda"); coef = torch.ones(3, K, device="cuda"); W = torch.randn(K, N, device="cuda")
        fn = lambda: ops.folded_dgrad(A, y, W, coef, None, residual=out)
    if kw.pop("pro", False):
        stat = torch.stack([torch.zeros(K), torch.ones(K)]).cuda(); g_ = torch.ones(K, device="cuda"); b_ = torch.zeros(K, device="cuda")
        fn = lambda: ops.conv_bn_prologue(A, stat, g_, b_, B, act="gelu", colstats=True)
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{name:18s} M={M:8d} N={N:5d} K={K:5d}  {dt*1e6:9.1f} us  {2.0*M*N*K/dt/1e12:7.1f} TF/s  {4.0*(M*K+N*K+M*N)/dt/1e9:8.1f} GB/s(A+B+C)")
    del A, B, out, kw, fn

for name, M, N, K in [("pe1 wgrad", 12845056, 48, 32), ("pe2 wgrad", 3211264, 96, 432), ("s3.fc1 wgrad", 50176, 2304, 576)]:
    dY = torch.randn(M, N, device="cuda"); X = torch.randn(M, K, device="cuda")
    for _ in range(2): ops.gemm_tn(dY, X)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ops.gemm_tn(dY, X)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name:18s} M={M:8d} N={N:5d} K={K:5d}  {dt*1e6:9.1f} us  {2.0*M*N*K/dt/1e12:7.1f} TF/s  {4.0*M*(N+K)/dt/1e9:8.1f} GB/s")
    del dY, X
