#!/usr/bin/env python3
"""dev: the split weight-gradient GEMM (gg_gemm_tn_split3) next to gg_gemm_tn_f32 on the stage-3 shapes of the 1024-image TinyViT-21M step and a few ragged ones:
time (kernel + slab reduce) and rel-L2 error against an fp64 product."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import _lib as L


def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(kind, dY, X, scale, rps, scratch, out):
    M, N = dY.shape; K = X.shape[1]
    lib = L.lib()
    if kind == "split":
        s = lib.gg_gemm_tn_split3_splits(M, N, K)
        L.check(lib.gg_gemm_tn_split3(dY.data_ptr(), dY.stride(0), X.data_ptr(), X.stride(0), M, N, K, scale.data_ptr() if scale is not None else None, rps, scratch.data_ptr(), s, L.stream()))
    else:
        s = lib.gg_gemm_tn_f32_splits(M, N, K)
        L.check(lib.gg_gemm_tn_f32(dY.data_ptr(), dY.stride(0), X.data_ptr(), X.stride(0), M, N, K, scale.data_ptr() if scale is not None else None, rps, scratch.data_ptr(), s, L.stream()))
    L.check(lib.gg_splitk_reduce(scratch.data_ptr(), out.data_ptr(), N * K, s, 0, 1.0, L.stream()))
    return s


shapes = [("s3.fc1", 50176, 2304, 576, 49), ("s3.fc2", 50176, 576, 2304, 49), ("s3.qkv", 50176, 1728, 576, 0), ("s3.proj", 50176, 576, 576, 49),
          ("ragged", 5003, 200, 136, 7), ("tiny", 40, 8, 12, 0)]
scratch = torch.empty(32 << 20, device="cuda")
for name, M, N, K, rps in shapes:
    g = torch.Generator(device="cuda").manual_seed(2)
    dY = torch.randn(M, N, device="cuda", generator=g); X = torch.randn(M, K, device="cuda", generator=g)
    scale = ((torch.rand((M + rps - 1) // rps, device="cuda", generator=g) > 0.2).float() / 0.8) if rps else None
    o3 = torch.empty(N, K, device="cuda"); o32 = torch.empty(N, K, device="cuda")
    s3 = run("split", dY, X, scale, rps, scratch, o3); s32 = run("f32", dY, X, scale, rps, scratch, o32)
    dYs = dY.double() * (scale.double().repeat_interleave(rps)[:M, None] if rps else 1.0)
    ref = dYs.T @ X.double()
    err = lambda o: float((o.double() - ref).norm() / ref.norm())
    t3 = timed(lambda: run("split", dY, X, scale, rps, scratch, o3)); t32 = timed(lambda: run("f32", dY, X, scale, rps, scratch, o32))
    fl = 2.0 * M * N * K
    print(f"{name:8s} {M:6d} x {N:5d} x {K:5d} | f32 {t32*1e3:8.1f} us {fl/t32/1e9:6.1f} TF ({s32} slabs) | split {t3*1e3:8.1f} us {fl/t3/1e9:6.1f} TF ({s3} slabs) x{t32/t3:.2f} | err f32 {err(o32):.2e} split {err(o3):.2e}", flush=True)
