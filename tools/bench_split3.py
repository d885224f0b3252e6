#!/usr/bin/env python3
"""Experiment: fp32-accurate GEMM on the bf16 matrix pipe.  x = x1 + x2 + x3 with three bf16 terms (24 significand bits); the six
products x1y1 + (x1y2 + x2y1) + (x1y3 + x2y2 + x3y1) accumulated in fp32 reproduce the fp32 product to ~2^-24.  Here the six products run as ONE bf16
GEMM of the product library over a 6 K contraction ([x1|x1|x2|x1|x2|x3] . [y1|y2|y1|y3|y2|y1]^T, f32 output), which needs no new kernel: it answers
(a) the parity question -- error against an fp64 product next to the true f32-MFMA GEMM's -- and (b) gives a LOWER bound on the speed (the operand
split is done with torch glue and is not timed; a production form would split in the producer's epilogue / the GEMM's loader).
Shapes: the stage-2 / stage-3 transformer GEMMs of the 1024-image TinyViT-21M step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from geoguessr_ai_amd import ops, _lib as L


def planes(x):
    """gg_split3_bf16: f32 [rows][cols] -> bf16 [3][rows][cols]"""
    out = torch.empty((3,) + tuple(x.shape), dtype=torch.bfloat16, device=x.device)
    L.check(L.lib().gg_split3_bf16(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), out.data_ptr(), L.stream()), "gg_split3_bf16")
    return out


def gemm_split3(Ap, Bp, out):
    M, K = Ap.shape[1], Ap.shape[2]
    N = Bp.shape[1]
    L.check(L.lib().gg_gemm_nt_split3(Ap.data_ptr(), K, Bp.data_ptr(), K, out.data_ptr(), out.stride(0), M, N, K, None, L.stream()), "gg_gemm_nt_split3")
    return out


def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def split3(x):
    x1 = x.bfloat16(); r = x - x1.float()
    x2 = r.bfloat16(); r = r - x2.float()
    return x1, x2, r.bfloat16()


def main():
    dev = "cuda"
    shapes = [("s2.qkv", 200704, 1152, 384), ("s2.fc1", 200704, 1536, 384), ("s2.fc2", 200704, 384, 1536), ("s2.proj", 200704, 384, 384),
              ("s3.fc1", 50176, 2304, 576), ("s3.fc2", 50176, 576, 2304)]
    print(f"{'shape':8s} {'M':>7s} {'N':>5s} {'K':>5s} | f32-MFMA us  TF/s | bf16x3 (6K) us  TF/s-eq | split3 kernel us  TF/s-eq  (+A split us) | plain bf16 us | rel-L2 err vs fp64: f32-MFMA  bf16x3(6K)  split3   bf16 | max|err|/max|ref|: f32-MFMA  split3")
    for name, M, N, K in shapes:
        g = torch.Generator(device=dev).manual_seed(1)
        A = torch.randn(M, K, device=dev, generator=g)
        B = torch.randn(N, K, device=dev, generator=g) * K ** -0.5
        a1, a2, a3 = split3(A); b1, b2, b3 = split3(B)
        A6 = torch.cat([a1, a1, a2, a1, a2, a3], 1).contiguous()
        B6 = torch.cat([b1, b2, b1, b3, b2, b1], 1).contiguous()
        out32 = torch.empty(M, N, device=dev); out6 = torch.empty(M, N, device=dev); out16 = torch.empty(M, N, device=dev)
        t32 = timed(lambda: ops.gemm_nt(A, B, out=out32))
        t6 = timed(lambda: ops.gemm_nt(A6, B6, out_f32=True, out=out6))
        t16 = timed(lambda: ops.gemm_nt(a1, b1, out_f32=True, out=out16))
        Ap, Bp = planes(A), planes(B)
        assert torch.equal(Ap[0], a1) and torch.equal(Ap[1], a2) and torch.equal(Ap[2], a3)            # the split kernel = the torch expression
        outs = torch.empty(M, N, device=dev)
        ts = timed(lambda: gemm_split3(Ap, Bp, outs))
        tsp = timed(lambda: planes(A))
        rows = 4096
        ref = A[:rows].double() @ B.double().T
        def err(o):
            d = o[:rows].double() - ref
            return float(d.norm() / ref.norm()), float(d.abs().max() / ref.abs().max())
        e32, e6, e16, es = err(out32), err(out6), err(out16), err(outs)
        fl = 2.0 * M * N * K
        print(f"{name:8s} {M:7d} {N:5d} {K:5d} | {t32*1e3:9.1f} {fl/t32/1e9:6.1f} | {t6*1e3:12.1f} {fl/t6/1e9:8.1f} | {ts*1e3:13.1f} {fl/ts/1e9:8.1f}  ({tsp*1e3:8.1f}) | {t16*1e3:10.1f} | "
              f"{e32[0]:.2e} {e6[0]:.2e} {es[0]:.2e} {e16[0]:.2e} | {e32[1]:.2e} {es[1]:.2e}", flush=True)
        del A, B, A6, B6, out32, out6, out16, outs, Ap, Bp, a1, a2, a3, b1, b2, b3
        torch.cuda.empty_cache()


def chain():
    """Producer-side split: LayerNorm -> Linear as the model runs it (f32 LayerNorm, f32-MFMA GEMM with bias) against LayerNorm writing the three planes ->
    gg_gemm_nt_split3 with bias (weight planes made once).  Stage-2 shapes of the 1024-image step: norm1 -> qkv and norm2 -> fc1."""
    dev = "cuda"
    print("\nLayerNorm -> Linear chains (us): f32 LN + f32-MFMA GEMM | LN writing bf16 planes + split3 GEMM | rel-L2 of the chain's result against fp64")
    for name, M, C_, N in [("s2 norm1->qkv", 200704, 384, 1152), ("s2 norm2->fc1", 200704, 384, 1536), ("s1 norm2->fc1", 802816, 192, 768), ("s3 norm1->qkv", 50176, 576, 1728)]:
        g = torch.Generator(device=dev).manual_seed(2)
        x = torch.randn(M, C_, device=dev, generator=g) * 1.3 + 0.2
        gamma = 1 + 0.2 * torch.randn(C_, device=dev, generator=g); beta = 0.1 * torch.randn(C_, device=dev, generator=g)
        W = torch.randn(N, C_, device=dev, generator=g) * C_ ** -0.5; bias = 0.1 * torch.randn(N, device=dev, generator=g)
        Wp = planes(W)
        a = torch.empty(M, C_, device=dev); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
        out1 = torch.empty(M, N, device=dev); out2 = torch.empty(M, N, device=dev)
        ap = torch.empty(3, M, C_, dtype=torch.bfloat16, device=dev)
        lib = L.lib()
        def ln_f32():
            L.check(lib.gg_layernorm_fwd(x.data_ptr(), 1, gamma.data_ptr(), beta.data_ptr(), M, C_, 1e-5, a.data_ptr(), 1, mean.data_ptr(), rstd.data_ptr(), L.stream()), "ln")
        def ln_pl():
            L.check(lib.gg_layernorm_fwd_split3(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), M, C_, 1e-5, ap.data_ptr(), mean.data_ptr(), rstd.data_ptr(), L.stream()), "ln3")
        def g_f32():
            ops.gemm_nt(a, W, bias=bias, out=out1)
        def g_s3():
            L.check(lib.gg_gemm_nt_split3(ap.data_ptr(), C_, Wp.data_ptr(), C_, out2.data_ptr(), N, M, N, C_, bias.data_ptr(), L.stream()), "s3")
        t_ln, t_lp, t_g, t_s = timed(ln_f32), timed(ln_pl), timed(g_f32), timed(g_s3)
        rows = 2048
        xd = x[:rows].double()
        ref = torch.nn.functional.layer_norm(xd, (C_,), gamma.double(), beta.double(), 1e-5) @ W.double().T + bias.double()
        e1 = float((out1[:rows].double() - ref).norm() / ref.norm()); e2 = float((out2[:rows].double() - ref).norm() / ref.norm())
        print(f"{name:14s} M={M:7d} C={C_:4d} N={N:5d} | LN {t_ln*1e3:7.1f} + GEMM {t_g*1e3:7.1f} = {(t_ln+t_g)*1e3:7.1f} | LN {t_lp*1e3:7.1f} + GEMM {t_s*1e3:7.1f} = {(t_lp+t_s)*1e3:7.1f} "
              f"({(t_ln+t_g)/(t_lp+t_s):.2f}x) | {e1:.2e} {e2:.2e}", flush=True)
        del x, a, ap, out1, out2, W, Wp
        torch.cuda.empty_cache()


def quick():
    """only the split3 kernel next to the f32-MFMA GEMM (us), plain epilogue"""
    dev = "cuda"
    for name, M, N, K in [("s2.qkv", 200704, 1152, 384), ("s2.fc1", 200704, 1536, 384), ("s2.fc2", 200704, 384, 1536), ("s2.proj", 200704, 384, 384),
                          ("s3.fc1", 50176, 2304, 576), ("s3.fc2", 50176, 576, 2304)]:
        g = torch.Generator(device=dev).manual_seed(1)
        A = torch.randn(M, K, device=dev, generator=g); B = torch.randn(N, K, device=dev, generator=g) * K ** -0.5
        Ap, Bp = planes(A), planes(B)
        o1 = torch.empty(M, N, device=dev); o2 = torch.empty(M, N, device=dev)
        t32 = timed(lambda: ops.gemm_nt(A, B, out=o1)); ts = timed(lambda: gemm_split3(Ap, Bp, o2))
        ref = A[:2048].double() @ B.double().T
        e1 = float((o1[:2048].double() - ref).norm() / ref.norm()); e2 = float((o2[:2048].double() - ref).norm() / ref.norm())
        fl = 2.0 * M * N * K
        print(f"{name:8s} f32 {t32*1e3:8.1f} us {fl/t32/1e9:6.1f} | split3 {ts*1e3:8.1f} us {fl/ts/1e9:6.1f} TF-eq ({t32/ts:.2f}x) | rel-L2 {e1:.2e} {e2:.2e}", flush=True)
        del A, B, Ap, Bp, o1, o2
        torch.cuda.empty_cache()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "quick":
        quick()
    else:
        main()
        chain()
