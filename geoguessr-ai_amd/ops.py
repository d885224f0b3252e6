"""Thin tensor-level wrappers over the primitive C-ABI entry points (used by the model shims and by the
per-kernel parity tests).  bf16 activations are ``torch.bfloat16`` tensors; everything is enqueued on the
current torch stream.  No arithmetic happens in Python."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L

ACT = {"none": 0, None: 0, "gelu": 1, "quick_gelu": 2}
BF16, F32, I64 = torch.bfloat16, torch.float32, torch.int64


def _p(t, dtype=None, name="tensor"):
    return L.ptr(t, dtype, name)


def _pr(t, dtype=None, name="matrix"):
    """Pointer of a 2-D row-major matrix whose rows may be strided (a column slice of a wider buffer)."""
    if t is None:
        return None
    if t.dim() != 2 or t.stride(1) != 1:
        raise L.GgError(f"{name} must be a 2-D matrix with unit column stride")
    if not t.is_cuda:
        raise L.GgError(f"{name} must live on the GPU (got device {t.device}); there is no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise L.GgError(f"{name} must be {dtype}, got {t.dtype}")
    return t.data_ptr()


def gemm_nt(A, B, *, bias=None, act=None, preact=False, rowscale=None, rows_per_scale=0, residual=None,
            dact_preact=None, dact=None, colstats=False, out_f32=False, out=None, M=None, N=None, K=None,
            lda=None, ldb=None, ldc=None):
    """C[M,N] = epilogue(A[M,K] @ B[N,K]^T).  Returns C (and preact / colstats when requested)."""
    L.require_gpu()
    M = A.shape[0] if M is None else M
    K = A.shape[1] if K is None else K
    N = B.shape[0] if N is None else N
    lda = A.stride(0) if lda is None else lda
    ldb = B.stride(0) if ldb is None else ldb
    dev = A.device
    DT = A.dtype                     # BF16: bf16 MFMA kernels; F32: the reference-precision twins (gg_gemm_nt_f32)
    if DT not in (BF16, F32):
        raise L.GgError(f"gemm_nt: operands must be bf16 or f32, got {DT}")
    if out is None:
        out = torch.empty((M, N), dtype=F32 if (out_f32 or DT == F32) else BF16, device=dev)
    ldc = out.stride(0) if ldc is None else ldc
    pre = torch.empty((M, N), dtype=DT, device=dev) if preact else None
    stats = None
    if colstats:
        rows = L.lib().gg_gemm_colstats_rows(M)
        stats_buf = torch.zeros((L.lib().gg_stat_rows_capacity(rows), 2, N), dtype=F32, device=dev)
        stats = stats_buf[:rows]
    a = L.GemmArgs()
    a.A, a.lda, a.B, a.ldb, a.C, a.ldc = _pr(A, DT, "A"), lda, _pr(B, DT, "B"), ldb, _pr(out), ldc
    a.M, a.N, a.K = M, N, K
    a.bias = _p(bias, F32, "bias")
    a.act = ACT[act]
    a.preact = _p(pre)
    a.rowscale, a.rows_per_scale = _p(rowscale, F32, "rowscale"), rows_per_scale
    a.residual, a.ldr = _p(residual, DT, "residual"), (residual.stride(0) if residual is not None else 0)
    a.dact_preact, a.dact = _p(dact_preact, DT, "dact_preact"), ACT[dact]
    a.colstats = _p(stats)
    a.out_f32, a.split_k = int(out_f32), 1
    if DT == F32:
        L.check(L.lib().gg_gemm_nt_f32(C.byref(a), L.stream()), "gg_gemm_nt_f32")
    else:
        L.check(L.lib().gg_gemm_nt(C.byref(a), L.stream()), "gg_gemm_nt")
    res = [out]
    if preact:
        res.append(pre)
    if colstats:
        res.append(stats)
    return res[0] if len(res) == 1 else tuple(res)


def gemm_splitk(A, B, split_k: int, accumulate_into: Optional[torch.Tensor] = None, scale: float = 1.0):
    """f32 C[M,N] = A[M,K] @ B[N,K]^T with the reduction split over blockIdx.y, then reduced."""
    M, K, N = A.shape[0], A.shape[1], B.shape[0]
    part = torch.empty((split_k, M, N), dtype=F32, device=A.device)
    a = L.GemmArgs()
    a.A, a.lda, a.B, a.ldb, a.C, a.ldc = _p(A, BF16), A.stride(0), _p(B, BF16), B.stride(0), _p(part), N
    a.M, a.N, a.K, a.out_f32, a.split_k = M, N, K, 1, split_k
    L.check(L.lib().gg_gemm_nt(C.byref(a), L.stream()), "gg_gemm_nt(split)")
    out = accumulate_into if accumulate_into is not None else torch.empty((M, N), dtype=F32, device=A.device)
    L.check(L.lib().gg_splitk_reduce(_p(part), _p(out, F32), M * N, split_k, int(accumulate_into is not None),
                                     scale, L.stream()), "gg_splitk_reduce")
    return out


def gemm_tn(dY, X, rowscale=None, rows_per_scale=0, accumulate_into=None):
    """f32 dW[N,K] = dY[M,N]^T @ X[M,K] (weight-gradient form, reduction over rows split across workgroups)."""
    M, N = dY.shape
    K = X.shape[1]
    if dY.dtype == F32:
        splits = L.lib().gg_gemm_tn_f32_splits(M, N, K)
        part = torch.empty((splits, N, K), dtype=F32, device=dY.device)
        L.check(L.lib().gg_gemm_tn_f32(_pr(dY, F32, "dY"), dY.stride(0), _pr(X, F32, "X"), X.stride(0), M, N, K, _p(rowscale, F32),
                                       rows_per_scale, _p(part), splits, L.stream()), "gg_gemm_tn_f32")
    else:
        splits = L.lib().gg_gemm_tn_splits(M, N, K)
        part = torch.empty((splits, N, K), dtype=F32, device=dY.device)
        L.check(L.lib().gg_gemm_tn(_pr(dY, BF16, "dY"), dY.stride(0), _pr(X, BF16, "X"), X.stride(0), M, N, K, _p(rowscale, F32), rows_per_scale,
                                   _p(part), splits, L.stream()), "gg_gemm_tn")
    out = accumulate_into if accumulate_into is not None else torch.empty((N, K), dtype=F32, device=dY.device)
    L.check(L.lib().gg_splitk_reduce(_p(part), _p(out, F32), N * K, splits, int(accumulate_into is not None), 1.0, L.stream()),
            "gg_splitk_reduce")
    return out


def col2im_nhwc_bnbwd(dcol, y, stat, gamma, beta, act="gelu", nparts=1024):
    """stride-2 col2im fused with BatchNorm backward's reduce: returns (dz [B,H,W,C] bf16, partial sums [nparts,2,C] f32)."""
    B, H, W, Cc = y.shape
    dz = torch.empty_like(y)
    part = torch.zeros((nparts + 64, 2, Cc), dtype=F32, device=y.device)
    fn = L.lib().gg_col2im_nhwc_bnbwd_f32 if y.dtype == F32 else L.lib().gg_col2im_nhwc_bnbwd_bf16
    L.check(fn(_p(dcol, y.dtype), _p(y), _p(stat, F32), _p(gamma, F32), _p(beta, F32), ACT[act], _p(dz),
               _p(part), nparts, B, H, W, Cc, L.stream()), "gg_col2im_nhwc_bnbwd")
    return dz, part[:nparts]


def gemm_tn_bn(dz, y, coef, X, accumulate_into=None):
    """f32 dW[N,K] = (coef0*dz + coef1*y + coef2)^T @ X: a ConvNorm's weight gradient straight from BatchNorm backward's (dz, y, coef)."""
    M, N = dz.shape
    K = X.shape[1]
    f32 = dz.dtype == F32
    splits = (L.lib().gg_gemm_tn_f32_splits if f32 else L.lib().gg_gemm_tn_splits)(M, N, K)
    part = torch.empty((splits, N, K), dtype=F32, device=dz.device)
    fn = L.lib().gg_gemm_tn_bn_f32 if f32 else L.lib().gg_gemm_tn_bn
    L.check(fn(_pr(dz, dz.dtype, "dz"), _pr(y, dz.dtype, "y"), dz.stride(0), _p(coef, F32), _pr(X, dz.dtype, "X"), X.stride(0), M, N, K,
               _p(part), splits, L.stream()), "gg_gemm_tn_bn")
    out = accumulate_into if accumulate_into is not None else torch.empty((N, K), dtype=F32, device=dz.device)
    L.check(L.lib().gg_splitk_reduce(_p(part), _p(out, F32), N * K, splits, int(accumulate_into is not None), 1.0, L.stream()),
            "gg_splitk_reduce")
    return out


def transpose_bf16(x, rowscale=None, rows_per_scale=0, pad_to: int = 8):
    R, Cc = x.shape
    ldo = (R + pad_to - 1) // pad_to * pad_to
    out = torch.zeros((Cc, ldo), dtype=BF16, device=x.device)
    L.check(L.lib().gg_transpose_bf16(_pr(x, BF16), x.stride(0), _p(out), ldo, R, Cc, _p(rowscale, F32), rows_per_scale,
                                      L.stream()), "gg_transpose_bf16")
    return out


def colsum_bf16(x, rowscale=None, rows_per_scale=0, out=None):
    M, Cc = x.shape
    scratch = torch.empty((L.lib().gg_colsum_scratch_floats(M, Cc),), dtype=F32, device=x.device)
    acc = out is not None
    if out is None:
        out = torch.empty((Cc,), dtype=F32, device=x.device)
    if x.dtype == F32:
        L.check(L.lib().gg_colsum_f32(_pr(x, F32), x.stride(0), M, Cc, _p(rowscale, F32), rows_per_scale, _p(scratch),
                                      _p(out, F32), int(acc), L.stream()), "gg_colsum_f32")
    else:
        L.check(L.lib().gg_colsum_bf16(_pr(x, BF16), x.stride(0), M, Cc, _p(rowscale, F32), rows_per_scale, _p(scratch),
                                       _p(out, F32), int(acc), L.stream()), "gg_colsum_bf16")
    return out


def im2col_nchw3(x, stride=2, out_dtype=BF16):
    B, Cc, H, W = x.shape
    assert Cc == 3
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    col = torch.empty((B * Ho * Wo, 32), dtype=out_dtype, device=x.device)
    if out_dtype == F32:
        L.check(L.lib().gg_im2col_nchw3_f32_f32(_p(x, F32), _p(col), B, H, W, stride, L.stream()), "gg_im2col_nchw3_f32_f32")
    else:
        L.check(L.lib().gg_im2col_nchw3_f32(_p(x, F32), _p(col), B, H, W, stride, L.stream()), "gg_im2col_nchw3_f32")
    return col


def im2col_nhwc(x, stride=2):
    B, H, W, Cc = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    col = torch.empty((B * Ho * Wo, 9 * Cc), dtype=x.dtype, device=x.device)
    if x.dtype == F32:
        L.check(L.lib().gg_im2col_nhwc_f32(_p(x, F32), None, None, None, 0, _p(col), B, H, W, Cc, stride, L.stream()), "gg_im2col_nhwc_f32")
    else:
        L.check(L.lib().gg_im2col_nhwc_bf16(_p(x, BF16), _p(col), B, H, W, Cc, stride, L.stream()), "gg_im2col_nhwc_bf16")
    return col


def im2col_nhwc_bn(y, stat, gamma, beta, act="gelu", stride=2):
    """im2col of act(BatchNorm(y)) for a saved pre-BatchNorm conv output y (B,H,W,C); the activation tensor is never stored."""
    B, H, W, Cc = y.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    col = torch.empty((B * Ho * Wo, 9 * Cc), dtype=y.dtype, device=y.device)
    if y.dtype == F32:
        L.check(L.lib().gg_im2col_nhwc_f32(_p(y, F32), _p(stat, F32), _p(gamma, F32), _p(beta, F32), ACT[act], _p(col), B, H, W, Cc,
                                           stride, L.stream()), "gg_im2col_nhwc_f32")
    else:
        L.check(L.lib().gg_im2col_nhwc_bn_bf16(_p(y, BF16), _p(stat, F32), _p(gamma, F32), _p(beta, F32), ACT[act], _p(col), B, H, W, Cc,
                                               stride, L.stream()), "gg_im2col_nhwc_bn_bf16")
    return col


def col2im_nhwc(dcol, B, H, W, Cc, stride=2):
    dx = torch.empty((B, H, W, Cc), dtype=dcol.dtype, device=dcol.device)
    if dcol.dtype == F32:
        L.check(L.lib().gg_col2im_nhwc_f32(_p(dcol, F32), _p(dx), B, H, W, Cc, stride, L.stream()), "gg_col2im_nhwc_f32")
    else:
        L.check(L.lib().gg_col2im_nhwc_bf16(_p(dcol, BF16), _p(dx), B, H, W, Cc, stride, L.stream()), "gg_col2im_nhwc_bf16")
    return dx


def dwconv3x3_fwd(x, taps, stride=1, colstats=False):
    B, H, W, Cc = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty((B, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    stats = None
    f32 = x.dtype == F32
    if colstats:
        rows = L.lib().gg_dwconv_f32_stat_rows(B, Ho, Wo, Cc, stride) if f32 else L.lib().gg_dwconv_stat_rows(B, Ho, Wo, Cc, stride)
        stats = torch.zeros((L.lib().gg_stat_rows_capacity(rows), 2, Cc), dtype=F32, device=x.device)[:rows]
    if f32:
        L.check(L.lib().gg_dwconv3x3_fwd_f32(_p(x, F32), _p(taps, F32), _p(y), B, H, W, Cc, stride, _p(stats), L.stream()),
                "gg_dwconv3x3_fwd_f32")
    else:
        L.check(L.lib().gg_dwconv3x3_fwd(_p(x, BF16), _p(taps, F32), _p(y), B, H, W, Cc, stride, _p(stats), L.stream()),
                "gg_dwconv3x3_fwd")
    return (y, stats) if colstats else y


def dwconv3x3_fwd_fused(y_in, stat, gamma, beta, taps, act="gelu", stride=2, colstats=True):
    """depthwise conv over act(BatchNorm(y_in)) formed while loading (y_in = saved pre-BatchNorm output of the ConvNorm in front)."""
    B, H, W, Cc = y_in.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    f32 = y_in.dtype == F32
    y = torch.empty((B, Ho, Wo, Cc), dtype=y_in.dtype, device=y_in.device)
    stats = None
    if colstats:
        rows = L.lib().gg_dwconv_f32_stat_rows(B, Ho, Wo, Cc, stride) if f32 else L.lib().gg_dwconv_fwd_fused_stat_rows(B, H, W, Cc, stride)
        stats = torch.zeros((L.lib().gg_stat_rows_capacity(rows), 2, Cc), dtype=F32, device=y_in.device)[:rows]
    fn = L.lib().gg_dwconv3x3_fwd_fused_f32 if f32 else L.lib().gg_dwconv3x3_fwd_fused
    L.check(fn(_p(y_in, y_in.dtype), _p(stat, F32), _p(gamma, F32), _p(beta, F32), ACT[act], _p(taps, F32), _p(y), B, H, W,
               Cc, stride, _p(stats), L.stream()), "gg_dwconv3x3_fwd_fused")
    return (y, stats) if colstats else y


def dwconv3x3_bwd_data_fused(dz_in, y_in, in_coef, taps, ep_y=None, ep_stat=None, ep_gamma=None, ep_beta=None, ep_act=None):
    """Stride-1 depthwise data gradient; the BatchNorm-backward apply of the ConvNorm behind it rides on the loads (dy = c0*dz + c1*y + c2
    when y_in is given) and act'(BN(ep_y)) * . plus BatchNorm backward's column sums of the ConvNorm in front ride on the stores.
    Returns (out, partial rows [rows, 2, C] or None)."""
    L.require_gpu()
    B, H, W, Cc = dz_in.shape
    dt = dz_in.dtype
    f32 = dt == F32
    out = torch.empty((B, H, W, Cc), dtype=dt, device=dz_in.device)
    part, rows = None, 0
    if ep_y is not None:
        rows = L.lib().gg_dwconv_f32_stat_rows(B, H, W, Cc, 1) if f32 else L.lib().gg_dwconv_fused_stat_rows(B, H, W, Cc, int(y_in is not None))
        part = torch.zeros((L.lib().gg_stat_rows_capacity(rows), 2, Cc), dtype=F32, device=dz_in.device)
    fn = L.lib().gg_dwconv3x3_bwd_data_fused_f32 if f32 else L.lib().gg_dwconv3x3_bwd_data_fused
    L.check(fn(_p(dz_in, dt), _p(y_in, dt), _p(in_coef, F32), _p(taps, F32), _p(out), B, H, W, Cc, _p(ep_y, dt), _p(ep_stat, F32),
               _p(ep_gamma, F32), _p(ep_beta, F32), ACT[ep_act], _p(part), L.stream()), "gg_dwconv3x3_bwd_data_fused")
    return out, (part[:rows] if part is not None else None)


def dwconv3x3_bwd_data(dy, taps, B, H, W, Cc, stride=1):
    dx = torch.empty((B, H, W, Cc), dtype=dy.dtype, device=dy.device)
    if dy.dtype == F32:
        L.check(L.lib().gg_dwconv3x3_bwd_data_f32(_p(dy, F32), _p(taps, F32), _p(dx), B, H, W, Cc, stride, L.stream()),
                "gg_dwconv3x3_bwd_data_f32")
    else:
        L.check(L.lib().gg_dwconv3x3_bwd_data(_p(dy, BF16), _p(taps, F32), _p(dx), B, H, W, Cc, stride, L.stream()),
                "gg_dwconv3x3_bwd_data")
    return dx


def dwconv3x3_bwd_weight(x, dy, stride=1, grad=None):
    B, H, W, Cc = x.shape
    f32 = x.dtype == F32
    nscr = L.lib().gg_dwconv_f32_wgrad_scratch_floats(B, H, W, Cc, stride) if f32 else L.lib().gg_dwconv_wgrad_scratch_floats(B, H, W, Cc, stride)
    scratch = torch.empty((nscr,), dtype=F32, device=x.device)
    acc = grad is not None
    if grad is None:
        grad = torch.empty((Cc, 1, 3, 3), dtype=F32, device=x.device)
    if f32:
        L.check(L.lib().gg_dwconv3x3_bwd_weight_f32(_p(x, F32), _p(dy, F32), B, H, W, Cc, stride, _p(scratch), _p(grad, F32),
                                                    int(acc), L.stream()), "gg_dwconv3x3_bwd_weight_f32")
    else:
        L.check(L.lib().gg_dwconv3x3_bwd_weight(_p(x, BF16), _p(dy, BF16), B, H, W, Cc, stride, _p(scratch), _p(grad, F32),
                                                int(acc), L.stream()), "gg_dwconv3x3_bwd_weight")
    return grad


def bn_finalize(partials, count, eps=1e-5, momentum=0.1, running_mean=None, running_var=None):
    nparts, _, Cc = partials.shape
    buf = torch.zeros((L.lib().gg_stat_rows_capacity(nparts), 2, Cc), dtype=F32, device=partials.device)
    buf[:nparts] = partials
    partials = buf
    stat = torch.empty((2, Cc), dtype=F32, device=partials.device)
    L.check(L.lib().gg_bn_finalize(_p(partials, F32), nparts, Cc, count, eps, momentum, _p(stat), _p(running_mean, F32),
                                   _p(running_var, F32), L.stream()), "gg_bn_finalize")
    return stat


def bn_eval_stat(running_mean, running_var, eps=1e-5):
    Cc = running_mean.numel()
    stat = torch.empty((2, Cc), dtype=F32, device=running_mean.device)
    L.check(L.lib().gg_bn_eval_stat(_p(running_mean, F32), _p(running_var, F32), Cc, eps, _p(stat), L.stream()),
            "gg_bn_eval_stat")
    return stat


def bn_apply(y, stat, gamma, beta, act=None, residual=None, rowscale=None, rows_per_scale=0):
    M, Cc = y.shape
    out = torch.empty_like(y)
    fn = L.lib().gg_bn_apply_f32 if y.dtype == F32 else L.lib().gg_bn_apply
    L.check(fn(_p(y, y.dtype), _p(stat, F32), _p(gamma, F32), _p(beta, F32), M, Cc, ACT[act],
               _p(residual, y.dtype), _p(rowscale, F32), rows_per_scale, _p(out), L.stream()), "gg_bn_apply")
    return out


def bn_bwd(dout, y, stat, gamma, beta, act=None, residual=None, rowscale=None, rows_per_scale=0, want_param_grads=True):
    M, Cc = y.shape
    dev = y.device
    dz, dy = torch.empty_like(y), torch.empty_like(y)
    scratch = torch.empty((L.lib().gg_bn_bwd_scratch_floats(M, Cc),), dtype=F32, device=dev)
    dg = torch.zeros((Cc,), dtype=F32, device=dev) if want_param_grads else None
    db = torch.zeros((Cc,), dtype=F32, device=dev) if want_param_grads else None
    fn = L.lib().gg_bn_bwd_f32 if y.dtype == F32 else L.lib().gg_bn_bwd
    L.check(fn(_p(dout, y.dtype), _p(y), _p(stat, F32), _p(gamma, F32), _p(beta, F32), M, Cc, ACT[act],
               _p(residual, y.dtype), _p(rowscale, F32), rows_per_scale, _p(dz), _p(dy), _p(scratch), _p(dg),
               _p(db), 1, L.stream()), "gg_bn_bwd")
    return dz, dy, dg, db


def _gemm_nt_any(a, dtype):
    fn = L.lib().gg_gemm_nt_f32 if dtype == F32 else L.lib().gg_gemm_nt
    L.check(fn(C.byref(a), L.stream()), "gg_gemm_nt")


def conv_bn_prologue(y_prev, stat, gamma, beta, W, act=None, colstats=False):
    """C = act(BN(y_prev)) @ W^T with the BatchNorm + activation applied while the GEMM stages its A tile.

    y_prev [M,K] (bf16 or f32) is the previous ConvNorm's saved pre-BatchNorm output, stat = [mean[K], rstd[K]]."""
    L.require_gpu()
    M, K = y_prev.shape
    N = W.shape[0]
    dev, dt = y_prev.device, y_prev.dtype
    out = torch.empty((M, N), dtype=dt, device=dev)
    stats = None
    if colstats:
        rows = L.lib().gg_gemm_colstats_rows(M)
        stats = torch.zeros((L.lib().gg_stat_rows_capacity(rows), 2, N), dtype=F32, device=dev)[:rows]
    a = L.GemmArgs()
    a.A, a.lda, a.B, a.ldb, a.C, a.ldc = _pr(y_prev, dt, "y_prev"), y_prev.stride(0), _pr(W, dt, "W"), W.stride(0), _p(out), N
    a.M, a.N, a.K = M, N, K
    a.a_bn_stat, a.a_bn_gamma, a.a_bn_beta, a.a_bn_act = _p(stat, F32), _p(gamma, F32), _p(beta, F32), ACT[act]
    a.colstats = _p(stats)
    a.split_k = 1
    _gemm_nt_any(a, dt)
    return (out, stats) if colstats else out


def conv_dgrad_bn_bwd(dY, Wt, y, stat, gamma, beta, act=None, want_param_grads=True):
    """dz = (dY @ Wt^T) * act'(BN(y)) with BatchNorm backward's column sums taken in the GEMM epilogue, then finalize.

    dY [M,K], Wt [N,K] (the conv weight transposed), y [M,N] (the ConvNorm's saved conv output); all bf16 or all f32.
    Returns (dz, coef [3,N], dgamma, dbeta) with dy = coef0*dz + coef1*y + coef2."""
    L.require_gpu()
    M, K = dY.shape
    N = Wt.shape[0]
    dev, dt = dY.device, dY.dtype
    dz = torch.empty((M, N), dtype=dt, device=dev)
    rows = L.lib().gg_gemm_colstats_rows(M)
    part = torch.zeros((L.lib().gg_stat_rows_capacity(rows), 2, N), dtype=F32, device=dev)
    a = L.GemmArgs()
    a.A, a.lda, a.B, a.ldb, a.C, a.ldc = _pr(dY, dt, "dY"), dY.stride(0), _pr(Wt, dt, "Wt"), Wt.stride(0), _pr(dz), N
    a.M, a.N, a.K = M, N, K
    a.bn_y, a.bn_stat, a.bn_gamma, a.bn_beta, a.bn_act = _p(y, dt, "y"), _p(stat, F32), _p(gamma, F32), _p(beta, F32), ACT[act]
    a.colstats = _p(part)
    a.split_k = 1
    _gemm_nt_any(a, dt)
    coef = torch.empty((3, N), dtype=F32, device=dev)
    dg = torch.zeros((N,), dtype=F32, device=dev) if want_param_grads else None
    db = torch.zeros((N,), dtype=F32, device=dev) if want_param_grads else None
    L.check(L.lib().gg_bn_bwd_finalize(_p(part), rows, N, M, _p(stat, F32), _p(gamma, F32), _p(coef), _p(dg), _p(db), 0,
                                       L.stream()), "gg_bn_bwd_finalize")
    return dz, coef, dg, db


def folded_dgrad(dz, y, W, coef, stat, residual=None):
    """dx = BNbwd_apply(dz, y, coef) @ W  for a 1x1 conv W [Cout, Cin] f32, without forming dy.
    bf16: [dz | y] @ Bf^T + bias with the apply folded into the weights; f32: dy = c0*dz + c1*y + c2 formed from the two sources while the
    GEMM stages its A tile (a doubled contraction would cost real f32 MFMA time)."""
    L.require_gpu()
    M, Cout = dz.shape
    Cin = W.shape[1]
    dev = dz.device
    if dz.dtype == F32:
        Wt = W.t().contiguous()                                    # [Cin, Cout]
        dx = torch.empty((M, Cin), dtype=F32, device=dev)
        a = L.GemmArgs()
        a.A, a.lda, a.A2, a.a_bn_stat = _pr(dz, F32, "dz"), dz.stride(0), _pr(y, F32, "y"), _p(coef, F32)
        a.B, a.ldb, a.C, a.ldc = _p(Wt), Cout, _p(dx), Cin
        a.M, a.N, a.K = M, Cin, Cout
        a.residual, a.ldr = _p(residual, F32, "residual"), Cin
        a.split_k = 1
        _gemm_nt_any(a, F32)
        return dx
    Bf = torch.empty((Cin, 2 * Cout), dtype=BF16, device=dev)
    bias = torch.empty((Cin,), dtype=F32, device=dev)
    L.check(L.lib().gg_bn_bwd_fold_weights(_p(W, F32, "W"), _p(coef, F32), _p(stat, F32), Cout, Cin, _p(Bf), _p(bias),
                                           L.stream()), "gg_bn_bwd_fold_weights")
    dx = torch.empty((M, Cin), dtype=BF16, device=dev)
    a = L.GemmArgs()
    a.A, a.lda, a.A2, a.k_split = _pr(dz, BF16, "dz"), dz.stride(0), _pr(y, BF16, "y"), Cout
    a.B, a.ldb, a.C, a.ldc = _p(Bf), 2 * Cout, _p(dx), Cin
    a.M, a.N, a.K = M, Cin, 2 * Cout
    a.bias = _p(bias)
    a.residual, a.ldr = _p(residual, BF16, "residual"), Cin
    a.split_k = 1
    L.check(L.lib().gg_gemm_nt(C.byref(a), L.stream()), "gg_gemm_nt")
    return dx


def layernorm_fwd(x, gamma, beta, eps=1e-5, out_f32=None, save_stats=True):
    M, Cc = x.shape
    x_f32 = x.dtype == F32
    out_f32 = x_f32 if out_f32 is None else out_f32
    out = torch.empty((M, Cc), dtype=F32 if out_f32 else BF16, device=x.device)
    mean = torch.empty((M,), dtype=F32, device=x.device) if save_stats else None
    rstd = torch.empty((M,), dtype=F32, device=x.device) if save_stats else None
    L.check(L.lib().gg_layernorm_fwd(_p(x), int(x_f32), _p(gamma, F32), _p(beta, F32), M, Cc, eps, _p(out), int(out_f32),
                                     _p(mean), _p(rstd), L.stream()), "gg_layernorm_fwd")
    return out, mean, rstd


def layernorm_fwd_bn(y, bn_stat, bn_gamma, bn_beta, gamma, beta, eps=1e-5):
    """(x, LN(x), mean, rstd) with x = BatchNorm(y) (in y's storage type) formed on the way in (TinyViT local_conv -> norm2)."""
    M, Cc = y.shape
    x = torch.empty_like(y)
    out = torch.empty_like(y)
    mean = torch.empty((M,), dtype=F32, device=y.device)
    rstd = torch.empty((M,), dtype=F32, device=y.device)
    fn = L.lib().gg_layernorm_fwd_bn_f32 if y.dtype == F32 else L.lib().gg_layernorm_fwd_bn
    L.check(fn(_p(y, y.dtype), _p(bn_stat, F32), _p(bn_gamma, F32), _p(bn_beta, F32), _p(x), _p(gamma, F32), _p(beta, F32),
               M, Cc, eps, _p(out), _p(mean), _p(rstd), L.stream()), "gg_layernorm_fwd_bn")
    return x, out, mean, rstd


def layernorm_bwd(dout, x, mean, rstd, gamma, dres=None, want_param_grads=True):
    M, Cc = x.shape
    f32 = x.dtype == F32
    dx = torch.empty_like(x)
    scratch = torch.empty((L.lib().gg_layernorm_bwd_scratch_floats(M, Cc),), dtype=F32, device=x.device)
    dg = torch.zeros((Cc,), dtype=F32, device=x.device) if want_param_grads else None
    db = torch.zeros((Cc,), dtype=F32, device=x.device) if want_param_grads else None
    L.check(L.lib().gg_layernorm_bwd(_p(dout), _p(x), int(f32), _p(mean, F32), _p(rstd, F32), _p(gamma, F32), M, Cc, _p(dres),
                                     _p(dx), _p(scratch), _p(dg), _p(db), 1, L.stream()), "gg_layernorm_bwd")
    return dx, dg, db


def layernorm_bwd_colsum(dout, x, mean, rstd, gamma, dres=None):
    """LayerNorm backward that also leaves the per-block column sums (sum dx*x, sum dx) of its result; returns (dx, part, rows)."""
    M, Cc = x.shape
    dx = torch.empty_like(x)
    rows = L.lib().gg_layernorm_bwd_colsum_rows(M)
    part = torch.empty(((rows + 64) * 2 * Cc,), dtype=F32, device=x.device)
    L.check(L.lib().gg_layernorm_bwd_colsum(_p(dout), _p(x), int(x.dtype == F32), _p(mean, F32), _p(rstd, F32), _p(gamma, F32), M, Cc, _p(dres),
                                            _p(dx), _p(part), L.stream()), "gg_layernorm_bwd_colsum")
    return dx, part, rows


def bn_bwd_coef_from_x(part, rows, count, stat, gamma, beta):
    """coef [3][C] of dy = c0*g + c1*y + c2 (BatchNorm backward, frozen parameters) from layernorm_bwd_colsum's rows."""
    Cc = gamma.numel()
    coef = torch.empty((3, Cc), dtype=F32, device=part.device)
    L.check(L.lib().gg_bn_bwd_coef_from_x(_p(part), rows, Cc, count, _p(stat, F32), _p(gamma, F32), _p(beta, F32), _p(coef), L.stream()),
            "gg_bn_bwd_coef_from_x")
    return coef


def token_mean_fwd(x, B, T):
    Cc = x.shape[-1]
    out = torch.empty((B, Cc), dtype=F32, device=x.device)
    fn = L.lib().gg_token_mean_fwd_f32 if x.dtype == F32 else L.lib().gg_token_mean_fwd
    L.check(fn(_p(x), _p(out), B, T, Cc, L.stream()), "gg_token_mean_fwd")
    return out


def token_mean_bwd(dout, T, out_dtype=BF16):
    B, Cc = dout.shape
    dx = torch.empty((B * T, Cc), dtype=out_dtype, device=dout.device)
    fn = L.lib().gg_token_mean_bwd_f32 if out_dtype == F32 else L.lib().gg_token_mean_bwd
    L.check(fn(_p(dout, F32), _p(dx), B, T, Cc, L.stream()), "gg_token_mean_bwd")
    return dx


def view_mean_f32(emb):
    """(N, V, C) f32 -> (N, C) f32 mean over the V views, summed in view order (models/super_guessr.py:347; prototype building)."""
    L.require_gpu()
    N, V, Cc = emb.shape
    out = torch.empty((N, Cc), dtype=F32, device=emb.device)
    L.check(L.lib().gg_view_mean_fwd_f32(_p(emb, F32, "emb"), _p(out), Cc, N, V, Cc, L.stream()), "gg_view_mean_fwd_f32")
    return out


def attention_flash(qkv, *, num_windows, tokens_per_window, num_heads, head_dim, q_off, k_off, v_off, head_stride,
                    window_size=0, map_h=0, map_w=0, bias_table=None, scale=None, dout=None, want_dbias=False, out=None, lse=None,
                    want_lse=False, deterministic_dbias=True, ds_handoff=False):
    """Online-softmax attention (any tokens_per_window; bf16 or f32 storage by ``qkv.dtype``).  Forward when ``dout`` is None
    (returns out, or (out, lse)); else backward given the forward's ``out`` and ``lse`` -> (dqkv, dbias).  ``ds_handoff``: give the
    backward a dS scratch (the dK/dV pass then hands dS to a one-product dQ pass)."""
    L.require_gpu()
    DT = qkv.dtype
    a = L.AttnArgs()
    a.qkv, a.ld = _p(qkv, DT, "qkv"), qkv.stride(0)
    a.q_off, a.k_off, a.v_off, a.head_stride, a.head_dim = q_off, k_off, v_off, head_stride, head_dim
    a.num_heads, a.num_windows, a.tokens_per_window = num_heads, num_windows, tokens_per_window
    a.window_size, a.map_h, a.map_w = window_size, map_h, map_w
    a.scale = head_dim ** -0.5 if scale is None else scale
    a.bias_table = _p(bias_table, F32, "bias_table")
    tokens = qkv.shape[0]
    dt = int(DT == F32)
    if dout is None:
        out = torch.empty((tokens, num_heads * head_dim), dtype=DT, device=qkv.device)
        a.out, a.ldo = _p(out), out.stride(0)
        lse_t = torch.empty((tokens, num_heads), dtype=F32, device=qkv.device) if want_lse else None
        a.lse = _p(lse_t)
        L.check(L.lib().gg_attention_flash_fwd(C.byref(a), dt, L.stream()), "gg_attention_flash_fwd")
        return (out, lse_t) if want_lse else out
    dqkv = torch.zeros_like(qkv)
    dbias = torch.zeros_like(bias_table) if (want_dbias and bias_table is not None) else None
    a.dout, a.lddo, a.dqkv, a.dbias = _p(dout, DT, "dout"), dout.stride(0), _p(dqkv), _p(dbias)
    scratch = None
    if dbias is not None and deterministic_dbias:
        rows = L.lib().gg_attention_flash_dbias_rows(num_windows, tokens_per_window)
        scratch = torch.empty((rows * num_heads * window_size * window_size,), dtype=F32, device=qkv.device)
        a.dbias_scratch = _p(scratch)
    a.out, a.ldo, a.lse = _p(out, DT, "out"), out.stride(0), _p(lse, F32, "lse")
    ds = None
    if ds_handoff:
        ds = torch.empty((L.lib().gg_attention_flash_ds_scratch_floats(num_windows, num_heads, tokens_per_window),), dtype=F32, device=qkv.device)
        a.ds_scratch = _p(ds)
    L.check(L.lib().gg_attention_flash_bwd(C.byref(a), dt, L.stream()), "gg_attention_flash_bwd")
    return dqkv, dbias


def attention(qkv, *, num_windows, tokens_per_window, num_heads, head_dim, q_off, k_off, v_off, head_stride,
              window_size=0, map_h=0, map_w=0, bias=None, scale=None, dout=None, want_dbias=False, out=None, lse=None,
              want_lse=False):
    """forward when ``dout`` is None (returns out, or (out, lse) with ``want_lse``); else backward given the forward's
    ``out`` and ``lse`` -> (dqkv, dbias)."""
    a = L.AttnArgs()
    f16 = qkv.dtype == torch.float16            # fp16 storage + fp16 MFMA: forward only, no bias (the CLIP tower's fp16 inference mode)
    if f16 and (bias is not None or dout is not None):
        raise L.GgError("attention: the fp16 form is forward-only and takes no bias")
    a.qkv, a.ld = _p(qkv, torch.float16 if f16 else BF16, "qkv"), qkv.stride(0)
    a.q_off, a.k_off, a.v_off, a.head_stride, a.head_dim = q_off, k_off, v_off, head_stride, head_dim
    a.num_heads, a.num_windows, a.tokens_per_window = num_heads, num_windows, tokens_per_window
    a.window_size, a.map_h, a.map_w = window_size, map_h, map_w
    full = None
    if bias is not None:      # relative-position table [heads][ws*ws] -> expanded [heads][Np][Np]
        Np = L.lib().gg_attention_padded_tokens(tokens_per_window)
        full = torch.empty((num_heads, Np, Np), dtype=BF16, device=qkv.device)
    a.scale = head_dim ** -0.5 if scale is None else scale
    if bias is not None:
        L.check(L.lib().gg_attention_expand_bias(_p(bias, F32, "bias"), num_heads, window_size, a.scale, _p(full),
                                                 L.stream()), "gg_attention_expand_bias")
    a.bias = _p(full)
    if bias is not None:
        a.bias_table = _p(bias, F32, "bias")          # compact table: the resident-window kernels of 12 x 12 / 14 x 14 windows gather from it in LDS
    tokens = qkv.shape[0]
    if dout is None:
        out = torch.empty((tokens, num_heads * head_dim), dtype=qkv.dtype, device=qkv.device)
        a.out, a.ldo = _p(out), out.stride(0)
        lse_t = torch.empty((tokens, num_heads), dtype=F32, device=qkv.device) if want_lse else None
        a.lse = _p(lse_t)
        if f16:
            L.check(L.lib().gg_attention_fwd_f16(C.byref(a), L.stream()), "gg_attention_fwd_f16")
        else:
            L.check(L.lib().gg_attention_fwd(C.byref(a), L.stream()), "gg_attention_fwd")
        return (out, lse_t) if want_lse else out
    dqkv = torch.zeros_like(qkv)
    dbias = torch.zeros_like(bias) if (want_dbias and bias is not None) else None
    a.dout, a.lddo, a.dqkv, a.dbias = _p(dout, BF16, "dout"), dout.stride(0), _p(dqkv), _p(dbias)
    scratch = None
    if dbias is not None:      # per-window partials, summed deterministically in a second stage
        scratch = torch.empty(((num_windows + 64) * num_heads * window_size * window_size,), dtype=F32, device=qkv.device)
        a.dbias_scratch = _p(scratch)
    a.out, a.ldo, a.lse = _p(out, BF16, "out"), out.stride(0), _p(lse, F32, "lse")
    L.check(L.lib().gg_attention_bwd(C.byref(a), L.stream()), "gg_attention_bwd")
    return dqkv, dbias


def geo_head(logits, centroids, *, labels=None, labels_clf=None, mode=0, smoothing_km=65.0, grad_scale=None,
             want_dlogits=False, num_candidates=5, want_nearest=False, K=None, dlogits_f32=False):
    """Fused SuperGuessr head epilogue.  Returns a dict of device tensors."""
    N = logits.shape[0]
    K = logits.shape[1] if K is None else K
    dev = logits.device
    r = dict(loss_rows=torch.empty((N,), dtype=F32, device=dev), loss=torch.zeros((1,), dtype=F32, device=dev),
             preds=torch.empty((N,), dtype=I64, device=dev), llh=torch.empty((N, 2), dtype=F32, device=dev),
             topk_vals=torch.empty((N, num_candidates), dtype=F32, device=dev),
             topk_idx=torch.empty((N, num_candidates), dtype=I64, device=dev))
    if want_dlogits:
        ldd = (K + 7) // 8 * 8
        r["dlogits"] = torch.empty((N, ldd), dtype=F32 if dlogits_f32 else BF16, device=dev)
    if want_nearest:
        r["nearest"] = torch.empty((N,), dtype=I64, device=dev)
    a = L.GeoHeadArgs()
    a.logits, a.ldl, a.N, a.K = _p(logits, F32, "logits"), logits.stride(0), N, K
    a.labels, a.centroids, a.labels_clf = _p(labels, F32, "labels"), _p(centroids, F32, "centroids"), _p(labels_clf, I64, "labels_clf")
    a.mode, a.smoothing_km = mode, smoothing_km
    a.grad_scale = (1.0 / N) if grad_scale is None else grad_scale
    a.loss_rows, a.loss = _p(r["loss_rows"]), _p(r["loss"])
    if want_dlogits:
        a.dlogits, a.ldd, a.dlogits_f32 = _p(r["dlogits"]), r["dlogits"].stride(0), int(dlogits_f32)
    a.preds, a.llh, a.topk_vals, a.topk_idx, a.num_candidates = _p(r["preds"]), _p(r["llh"]), _p(r["topk_vals"]), _p(r["topk_idx"]), num_candidates
    a.nearest = _p(r.get("nearest"))
    L.check(L.lib().gg_geo_head(C.byref(a), L.stream()), "gg_geo_head")
    return r


def haversine_matrix(x, centroids):
    """models/utils.py:39 with y given as (K,2) centroids (the reference passes centroids.t())."""
    N, K = x.shape[0], centroids.shape[0]
    out = torch.empty((N, K), dtype=F32, device=x.device)
    L.check(L.lib().gg_haversine_matrix(_p(x, F32), _p(centroids, F32), _p(out), N, K, L.stream()), "gg_haversine_matrix")
    return out


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, grad_scale=1.0):
    L.check(L.lib().gg_adamw_step(_p(p, F32), _p(g, F32), _p(m, F32), _p(v, F32), p.numel(), step, lr, beta1, beta2, eps,
                                  weight_decay, grad_scale, L.stream()), "gg_adamw_step")


def geoguessr_score(pred_llh, true_llh):
    """run_benchmark.py:25-65 for a batch: (distance_km float64 (N,), score int32 (N,)) -- haversine_np + geoguessr_score_from_distance.
    float64 coordinates (the reference's arrays) go through the double entry point un-narrowed; anything else is read as float32."""
    N = pred_llh.shape[0]
    d = torch.empty((N,), dtype=torch.float64, device=pred_llh.device)
    s = torch.empty((N,), dtype=torch.int32, device=pred_llh.device)
    if pred_llh.dtype == torch.float64 or true_llh.dtype == torch.float64:
        F64 = torch.float64
        L.check(L.lib().gg_geoguessr_score_f64(_p(pred_llh.to(F64).contiguous(), F64), _p(true_llh.to(F64).contiguous(), F64), N, _p(d), _p(s),
                                               L.stream()), "gg_geoguessr_score_f64")
    else:
        L.check(L.lib().gg_geoguessr_score(_p(pred_llh.float().contiguous(), F32), _p(true_llh.float().contiguous(), F32), N, _p(d), _p(s),
                                           L.stream()), "gg_geoguessr_score")
    return d, s


def preprocess_bilinear(images, size=None, mean=None, std=None):
    """Batch conditioning of the training loop (main_coordinator_idun_s3.py:337-381) as one kernel: bilinear resize
    (align_corners=False) -> /255 for uint8 input -> (x - mean)/std.  images: (N,3,H,W) or (B,V,3,H,W), f32 or u8, on the GPU."""
    L.require_gpu()
    assert images.is_cuda and images.dtype in (torch.float32, torch.uint8), "preprocess_bilinear: CUDA f32 / u8 images"
    lead = images.shape[:-3]
    c, h, w = images.shape[-3:]
    assert c == 3, "preprocess_bilinear: 3-channel images"
    x = images.contiguous().view(-1, 3, h, w)
    hd, wd = (h, w) if size is None else (int(size[0]), int(size[1]))
    out = torch.empty((x.shape[0], 3, hd, wd), dtype=F32, device=images.device)
    m3 = (C.c_float * 3)(*[float(v) for v in mean]) if mean is not None else None
    s3 = (C.c_float * 3)(*[float(v) for v in std]) if std is not None else None
    L.check(L.lib().gg_preprocess_bilinear(_p(x), int(images.dtype == torch.uint8), x.shape[0], h, w, _p(out), hd, wd, m3, s3,
                                           L.stream()), "gg_preprocess_bilinear")
    return out.view(*lead, 3, hd, wd)


def preprocess_pil(img_hwc, flt, resized, crop_top_left, crop_size, mean=None, std=None, mul_rescale=False, want_u8=False):
    """One RGB image (H, W, 3) uint8 on the GPU -> (3, Hc, Wc) float32: Pillow resize of the whole image to ``resized`` = (Hr, Wr) with filter ``flt``
    (2 bilinear, 3 bicubic), the crop window, 1/255, (x - mean) / std (``gg_preprocess_pil``; the uint8 stage is bit-identical to PIL.Image.resize)."""
    L.require_gpu()
    assert img_hwc.is_cuda and img_hwc.dtype == torch.uint8 and img_hwc.dim() == 3 and img_hwc.shape[2] == 3, "preprocess_pil: CUDA uint8 (H, W, 3) image"
    x = img_hwc.contiguous()
    hs, ws = x.shape[:2]
    (hr, wr), (top, left), (hc, wc) = resized, crop_top_left, crop_size
    need = L.lib().gg_preprocess_pil_workspace_bytes(hs, ws, flt, hr, wr, wc)
    if need < 0:
        raise L.GgError(f"preprocess_pil: bad geometry {hs}x{ws} -> {hr}x{wr}, filter {flt}")
    ws_buf = torch.empty(need, dtype=torch.uint8, device=x.device)
    out = torch.empty((3, hc, wc), dtype=F32, device=x.device)
    u8 = torch.empty((hc, wc, 3), dtype=torch.uint8, device=x.device) if want_u8 else None
    m3 = (C.c_float * 3)(*[float(v) for v in mean]) if mean is not None else None
    s3 = (C.c_float * 3)(*[float(v) for v in std]) if std is not None else None
    L.check(L.lib().gg_preprocess_pil(_p(x), hs, ws, int(flt), hr, wr, top, left, hc, wc, int(mul_rescale), m3, s3, _p(out), _p(u8), _p(ws_buf), L.stream()),
            "gg_preprocess_pil")
    return (out, u8) if want_u8 else out


def segment_mean(emb, ptr, member):
    """out[k] = mean of emb[member[ptr[k]:ptr[k+1]]] in list order (prototype building); zeros for empty segments."""
    L.require_gpu()
    K = ptr.numel() - 1
    out = torch.empty((K, emb.shape[1]), dtype=F32, device=emb.device)
    L.check(L.lib().gg_segment_mean(_p(emb, F32, "emb"), emb.stride(0), _p(ptr, I64, "ptr"), _p(member, I64, "member"), K, emb.shape[1],
                                    _p(out), L.stream()), "gg_segment_mean")
    return out


def dwconv3x3_s2_bwd_data_fused(dz_in, y_in, in_coef, taps, H, W, ep_y=None, ep_stat=None, ep_gamma=None, ep_beta=None, ep_act=None):
    """Stride-2 depthwise data gradient with the BatchNorm-backward apply (input side) and act'(BN) + reduce (output side) fused.
    dz_in / y_in: (B, Ho, Wo, C) bf16 or f32 (by ``dz_in.dtype``); returns (out (B,H,W,C), partial rows [rows,2,C] or None)."""
    L.require_gpu()
    B, Ho, Wo, Cc = dz_in.shape
    DT = dz_in.dtype
    f32 = DT == F32
    lib = L.lib()
    out = torch.empty((B, H, W, Cc), dtype=DT, device=dz_in.device)
    part = None
    rows_fn = lib.gg_dwconv_f32_s2_fused_stat_rows if f32 else lib.gg_dwconv_s2_fused_stat_rows
    if ep_y is not None:
        part = torch.zeros((lib.gg_stat_rows_capacity(rows_fn(B, H, W, Cc)), 2, Cc), dtype=F32, device=dz_in.device)
    fn = lib.gg_dwconv3x3_s2_bwd_data_fused_f32 if f32 else lib.gg_dwconv3x3_s2_bwd_data_fused
    L.check(fn(_p(dz_in, DT), _p(y_in, DT), _p(in_coef, F32), _p(taps, F32), _p(out), B, H, W, Cc, _p(ep_y, DT), _p(ep_stat, F32), _p(ep_gamma, F32),
               _p(ep_beta, F32), ACT[ep_act], _p(part), L.stream()), "gg_dwconv3x3_s2_bwd_data_fused")
    return out, (part[:rows_fn(B, H, W, Cc)] if part is not None else None)
