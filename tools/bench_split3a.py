#!/usr/bin/env python3
"""dev: the split GEMM with the A operand as f32 (split in the loader: gg_gemm_nt_split3_af32) next to the f32-MFMA GEMM and the plane-fed split GEMM on the
transformer shapes of the 1024-image TinyViT-21M step: time, and rel-L2 error against an fp64 product."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from geoguessr_ai_amd import ops, _lib as L


def planes(x):
    out = torch.empty((3,) + tuple(x.shape), dtype=torch.bfloat16, device=x.device)
    L.check(L.lib().gg_split3_bf16(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), out.data_ptr(), L.stream()), "gg_split3_bf16")
    return out


def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


EPI = os.environ.get("EPI", "")      # "gelu": bias + GELU + pre-activation copy (fc1 forward); "dact": x GELU'(saved pre-activation) (fc2 data gradient); "res": bias + row scale + residual
_keep = {}


def af32(A, Bp, out, bias=None):
    a = L.Split3Args()
    M, K = A.shape
    a.b_planes, a.ldb, a.M, a.N, a.K = Bp.data_ptr(), K, M, Bp.shape[1], K
    a.C, a.ldc = out.data_ptr(), out.stride(0)
    a.bias = bias.data_ptr() if bias is not None else None
    if EPI:
        key = (M, Bp.shape[1])
        if key not in _keep:
            _keep.clear()
            _keep[key] = (torch.randn(M, Bp.shape[1], device="cuda"), torch.ones((M + 48) // 49, device="cuda"))
        aux, sc = _keep[key]
        if EPI == "gelu": a.act, a.preact = 1, aux.data_ptr()
        elif EPI == "dact": a.bias, a.dact_preact, a.dact = None, aux.data_ptr(), 1
        elif EPI == "res": a.rowscale, a.rows_per_scale, a.residual, a.ldr = sc.data_ptr(), 49, aux.data_ptr(), Bp.shape[1]
    L.check(L.lib().gg_gemm_nt_split3_af32(C.byref(a), A.data_ptr(), A.stride(0), 0, L.stream()), "gg_gemm_nt_split3_af32")
    return out


shapes = [("s1.qkv", 802816, 576, 192), ("s1.fc1", 802816, 768, 192), ("s1.fc2", 802816, 192, 768),
          ("s1.fc2b", 802816, 192, 768), ("s1.proj", 802816, 192, 192), ("s3.proj", 50176, 576, 576), ("s2.qkv", 200704, 1152, 384), ("s2.fc1", 200704, 1536, 384), ("s2.fc2", 200704, 384, 1536), ("s2.proj", 200704, 384, 384),
          ("s3.qkv", 50176, 1728, 576), ("s3.fc1", 50176, 2304, 576), ("s3.fc2", 50176, 576, 2304), ("edge", 5000, 1000, 712)]
only = sys.argv[1:]
for name, M, N, K in shapes:
    if only and not any(name.startswith(x) for x in only): continue
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(M, K, device="cuda", generator=g)
    B = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
    bias = torch.randn(N, device="cuda", generator=g)
    if os.environ.get("ZERO") == "1": A.zero_(); B.zero_()          # (clock experiment: all-zero operands toggle no matrix-pipe inputs)
    if os.environ.get("ZERO") == "A": A.zero_()
    if os.environ.get("ZERO") == "small": A.mul_(2.0 ** -8).round_().mul_(2.0 ** 8)      # (operands exactly representable in bf16: planes 2 and 3 are zero)
    Ap, Bp = planes(A), planes(B)
    o32, o3, o3a = (torch.empty(M, N, device="cuda") for _ in range(3))
    t32 = timed(lambda: ops.gemm_nt(A, B, bias=bias, out=o32))
    t3 = timed(lambda: L.check(L.lib().gg_gemm_nt_split3(Ap.data_ptr(), K, Bp.data_ptr(), K, o3.data_ptr(), N, M, N, K, bias.data_ptr(), L.stream())))
    t3a = timed(lambda: af32(A, Bp, o3a, bias))
    rows = torch.cat([torch.arange(0, min(2048, M)), torch.arange(max(0, M - 1024), M)]).cuda()
    ref = A[rows].double() @ B.double().T + bias.double()
    err = lambda o: float((o[rows].double() - ref).norm() / ref.norm())
    fl = 2.0 * M * N * K
    print(f"{name:8s} {M:7d} {N:5d} {K:5d} | f32 MFMA {t32*1e3:8.1f} us {fl/t32/1e9:6.1f} TF | planes {t3*1e3:8.1f} us {fl/t3/1e9:6.1f} | A f32 {t3a*1e3:8.1f} us {fl/t3a/1e9:6.1f} (x{t32/t3a:.2f}) | "
          f"err f32 {err(o32):.2e} planes {err(o3):.2e} A-f32 {err(o3a):.2e}  same-as-planes {bool(torch.equal(o3, o3a))}", flush=True)
    del A, B, Ap, Bp, o32, o3, o3a
    torch.cuda.empty_cache()
