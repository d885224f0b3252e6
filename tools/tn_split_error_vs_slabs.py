"""dev: error of the split weight-gradient GEMM against fp64 as a function of the number of row slabs (shorter f32 accumulation chains), next to gg_gemm_tn_f32."""
import sys, torch
sys.path.insert(0, "/root/repo")
from geoguessr_ai_amd import _lib as L
lib = L.lib()
scratch = torch.empty(64 << 20, device="cuda")
for name, M, N, K in [("s3.fc1", 50176, 2304, 576), ("s3.fc2", 50176, 576, 2304), ("s3.qkv", 50176, 1728, 576), ("s3.proj", 50176, 576, 576)]:
    g = torch.Generator(device="cuda").manual_seed(2)
    dY = torch.randn(M, N, device="cuda", generator=g); X = torch.randn(M, K, device="cuda", generator=g)
    ref = dY.double().T @ X.double()
    out = torch.empty(N, K, device="cuda")
    s32 = lib.gg_gemm_tn_f32_splits(M, N, K)
    L.check(lib.gg_gemm_tn_f32(dY.data_ptr(), N, X.data_ptr(), K, M, N, K, None, 0, scratch.data_ptr(), s32, L.stream())); L.check(lib.gg_splitk_reduce(scratch.data_ptr(), out.data_ptr(), N * K, s32, 0, 1.0, L.stream()))
    e32 = float((out.double() - ref).norm() / ref.norm())
    res = []
    s0 = lib.gg_gemm_tn_split3_splits(M, N, K)
    for s in (s0, 2 * s0, 4 * s0):
        if s * N * K * 4 > scratch.numel() * 4 or ((M + s - 1) // s + 31) // 32 * 32 * (s - 1) >= M: continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(3):
            if it == 1: e0.record()
            L.check(lib.gg_gemm_tn_split3(dY.data_ptr(), N, X.data_ptr(), K, M, N, K, None, 0, scratch.data_ptr(), s, L.stream())); L.check(lib.gg_splitk_reduce(scratch.data_ptr(), out.data_ptr(), N * K, s, 0, 1.0, L.stream()))
        e1.record(); torch.cuda.synchronize()
        res.append((s, float((out.double() - ref).norm() / ref.norm()), e0.elapsed_time(e1) / 2 * 1e3))
    print(name, "f32", s32, f"{e32:.2e}", "| split", [(s, f"{e:.2e}", f"{t:.0f} us") for s, e, t in res])
