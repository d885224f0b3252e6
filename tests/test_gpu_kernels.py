"""Per-kernel parity: every HIP primitive behind the C-ABI against a plain fp32 CPU reference of the same op on the
same (bf16-representable) inputs.  Tolerances: bf16 outputs carry one rounding of 2^-9 relative; contractions are
accumulated in fp32 on both sides.  All tests call through ``libgg.so``."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from geoguessr_ai_amd import ops as o
    from geoguessr_ai_amd import _lib
    _lib.require_gpu()
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF).float()      # bf16-representable fp32


def dev(t, dtype=None):
    t = t.cuda()
    return t.to(dtype) if dtype is not None else t


def close(got, ref, rtol=1e-2, atol=1e-2, what=""):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = (err > tol)
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.4g} (ref max {float(ref.abs().max()):.4g})"


# ------------------------------------------------------------------------------------------- GEMM
def test_mfma_fragment_layout(ops):
    """A = permutation-like, B asymmetric: catches swapped row/col maps and transposed outputs."""
    M = N = K = 128
    A = torch.zeros(M, K)
    A[torch.arange(M), (torch.arange(M) * 7 + 3) % K] = 1.0
    B = (torch.arange(N)[:, None] * 0.5 + torch.arange(K)[None, :] * 0.03125).to(BF).float()
    got = ops.gemm_nt(dev(A, BF), dev(B, BF), out_f32=True)
    close(got, A @ B.t(), rtol=1e-6, atol=1e-6, what="mfma layout")


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 200, 96), (1000, 48, 32), (4099, 384, 96), (96, 576, 2304),
                                   (64, 12647, 576), (257, 96, 432)])
def test_gemm_plain(ops, M, N, K):
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.1)
    ref = A @ B.t()
    got = ops.gemm_nt(dev(A, BF), dev(B, BF), out_f32=True)
    close(got, ref, rtol=1e-4, atol=1e-3, what=f"gemm f32 {M}x{N}x{K}")
    got16 = ops.gemm_nt(dev(A, BF), dev(B, BF))
    close(got16, ref, rtol=1e-2, atol=1e-2, what="gemm bf16 out")


def test_gemm_epilogues(ops):
    M, N, K, T = 520, 200, 96, 130            # 4 "samples" of 130 rows
    A, B = rnd(M, K, seed=3), rnd(N, K, seed=4, scale=0.2)
    bias = rnd(N, seed=5)
    res = rnd(M, N, seed=6)
    rs = torch.tensor([0.0, 1.25, 1.25, 0.0])
    z = A @ B.t() + bias
    # fc1-style: bias + gelu, pre-activation saved
    out, pre = ops.gemm_nt(dev(A, BF), dev(B, BF), bias=dev(bias), act="gelu", preact=True)
    close(pre, z, what="preact")
    close(out, F.gelu(z), what="gelu")
    # proj/fc2-style: bias, DropPath row scale, residual
    out = ops.gemm_nt(dev(A, BF), dev(B, BF), bias=dev(bias), rowscale=dev(rs), rows_per_scale=T, residual=dev(res, BF))
    close(out, res + rs.repeat_interleave(T)[:, None] * z, what="rowscale+residual")
    # quick_gelu
    out = ops.gemm_nt(dev(A, BF), dev(B, BF), bias=dev(bias), act="quick_gelu")
    close(out, z * torch.sigmoid(1.702 * z), what="quick_gelu")
    # backward through GELU: out = (A B^T) * gelu'(pre)
    pre_in = rnd(M, N, seed=7)
    xg = pre_in.clone().requires_grad_(True)
    F.gelu(xg).sum().backward()
    out = ops.gemm_nt(dev(A, BF), dev(B, BF), dact_preact=dev(pre_in, BF), dact="gelu")
    close(out, (A @ B.t()) * xg.grad, what="dact gelu")
    # BatchNorm column statistics of the raw accumulators
    out, stats = ops.gemm_nt(dev(A, BF), dev(B, BF), colstats=True)
    raw = (A @ B.t()).to(BF).float()            # statistics are taken from the stored (bf16-rounded) conv output
    s = stats.cpu().sum(0)
    close(s[0], raw.sum(0), rtol=1e-3, atol=0.3, what="colsum")
    close(s[1], (raw * raw).sum(0), rtol=1e-3, atol=0.3, what="colsumsq")


@pytest.mark.parametrize("M,N,K", [(5003, 48, 32), (3000, 200, 96)])
def test_gemm_tn_with_batchnorm_apply_on_load(ops, M, N, K):
    """dW = (c0*dz + c1*y + c2)^T X with the BatchNorm-backward apply formed inside the TN GEMM's loader == apply pass (bf16 dy) + plain TN GEMM;
    ragged rows / columns must not pick up the affine's constant term."""
    dz, y, X = rnd(M, N, seed=90, scale=0.1), rnd(M, N, seed=91), rnd(M, K, seed=92)
    coef = torch.stack([1.0 + 0.1 * rnd(N, seed=93), 0.05 * rnd(N, seed=94), 0.02 * rnd(N, seed=95)])
    dzq, yq, Xq = dz.to(BF).float(), y.to(BF).float(), X.to(BF).float()
    dy = (coef[0] * dzq + (coef[1] * yq + coef[2])).to(BF).float()
    got = ops.gemm_tn_bn(dev(dz, BF), dev(y, BF), dev(coef), dev(X, BF))
    close(got, dy.t() @ Xq, rtol=2e-3, atol=2e-2, what="tn gemm with bn apply")
    plain = ops.gemm_tn(dev(dy, BF), dev(X, BF))
    close(got, plain.cpu(), rtol=1e-3, atol=1e-2, what="tn fused vs unfused")


@pytest.mark.parametrize("M,N,K", [(70001, 48, 32), (3000, 200, 96), (9000, 96, 432), (517, 52, 36)])
def test_f32_gemm_tn_with_batchnorm_apply_on_load(ops, M, N, K):
    """f32 twin (gg_gemm_tn_bn_f32; patch_embed.conv1 / conv2 weight gradients in the fp32 mode), both the tiled and the row-split small form."""
    dz, y, X = rnd(M, N, seed=90, scale=0.1), rnd(M, N, seed=91), rnd(M, K, seed=92)
    coef = torch.stack([1.0 + 0.1 * rnd(N, seed=93), 0.05 * rnd(N, seed=94), 0.02 * rnd(N, seed=95)])
    dy = coef[0] * dz + (coef[1] * y + coef[2])
    ref = dy.double().t() @ X.double()
    got = ops.gemm_tn_bn(dev(dz), dev(y), dev(coef), dev(X))
    scale = float(ref.abs().max())
    assert float((got.cpu().double() - ref).abs().max()) < 2e-5 * scale
    plain = ops.gemm_tn(dev(dy), dev(X))
    assert float((got - plain).abs().max()) < 1e-5 * scale


@pytest.mark.parametrize("M,N,K", [(784, 384, 1536), (196, 576, 2304), (8, 320, 12648), (256, 576, 12648), (1, 12648, 576), (784, 1152, 384), (200, 100, 52),
                                   (70, 68, 1028)])
def test_f32_gemm_small_m_form_and_split_k(ops, M, N, K):
    """Launches with fewer 128-row tiles than CUs run as 64 x 64 tiles; few tiles with a long contraction (the head's data gradient: K = 12648; stage-3 fc2 at one
    panorama) additionally split K into slabs reduced in slab order: every epilogue of the form against torch fp64, ragged edges included, and repeatable bit
    for bit (no atomics)."""
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g).cuda(); B = (torch.randn(N, K, generator=g) * 0.1).cuda()
    bias = torch.randn(N, generator=g).cuda(); res = torch.randn(M, N, generator=g).cuda()
    rps = max(1, M // 4)
    rsc = (torch.rand((M + rps - 1) // rps, generator=g) + 0.5).cuda()
    ref = A.double() @ B.double().t()
    rs_rows = rsc.double().repeat_interleave(rps)[:M, None]
    def rel(a, b): return float((a.double() - b).abs().max() / b.abs().max())
    tol = 2e-5 if K > 1000 else 4e-6
    a = ops.gemm_nt(A, B)
    assert rel(a, ref) < tol and torch.equal(a, ops.gemm_nt(A, B))
    assert rel(ops.gemm_nt(A, B, bias=bias), ref + bias.double()) < tol
    assert rel(ops.gemm_nt(A, B, bias=bias, residual=res), ref + bias.double() + res.double()) < tol
    assert rel(ops.gemm_nt(A, B, rowscale=rsc, rows_per_scale=rps), ref * rs_rows) < tol
    assert rel(ops.gemm_nt(A, B, bias=bias, residual=res, rowscale=rsc, rows_per_scale=rps), (ref + bias.double()) * rs_rows + res.double()) < tol
    y, pre = ops.gemm_nt(A, B, bias=bias, act="gelu", preact=True)
    assert rel(y, F.gelu(ref + bias.double())) < tol and rel(pre, ref + bias.double()) < tol
    h = torch.randn(M, N, generator=g).cuda()
    hh = h.double().clone().requires_grad_(True); F.gelu(hh).sum().backward()
    assert rel(ops.gemm_nt(A, B, dact_preact=h, dact="gelu"), ref * hh.grad) < tol


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (384, 576, 96), (12544, 576, 2304), (12544, 2304, 576), (12544, 1728, 576), (256, 64, 48), (512, 96, 64), (25088, 96, 384),
                                   (1024, 288, 128)])
def test_f32_gemm_row_layout_epilogue(ops, M, N, K):
    """Interior, aligned fp32 launches leave through the row-layout epilogue (gemm_f32_epilogue_rows: accumulators transposed through LDS, all
    element-wise work on whole rows): every epilogue kind against torch fp64, on both tile widths.  The 12 544-row shapes keep ~2000 workgroups
    in flight: that is where a 16-byte buffer store with a register soffset lost a data register to the next instruction (1 workgroup in ~2000)."""
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g).cuda(); B = (torch.randn(N, K, generator=g) * 0.1).cuda()
    bias = torch.randn(N, generator=g).cuda(); res = torch.randn(M, N, generator=g).cuda(); rsc = (torch.rand(M // 64, generator=g) + 0.5).cuda()
    ref = A.double() @ B.double().t()
    rs_rows = rsc.double().repeat_interleave(64)[:, None]
    def rel(a, b): return float((a.double() - b).abs().max() / b.abs().max())
    tol = 2e-5 if K > 1000 else 4e-6
    assert rel(ops.gemm_nt(A, B), ref) < tol
    c, st = ops.gemm_nt(A, B, colstats=True)
    assert rel(c, ref) < tol and rel(st.sum(0)[0], ref.sum(0)) < 1e-4 and rel(st.sum(0)[1], (ref * ref).sum(0)) < 1e-4
    assert rel(ops.gemm_nt(A, B, bias=bias), ref + bias.double()) < tol
    assert rel(ops.gemm_nt(A, B, bias=bias, residual=res), ref + bias.double() + res.double()) < tol
    assert rel(ops.gemm_nt(A, B, bias=bias, residual=res, rowscale=rsc, rows_per_scale=64), (ref + bias.double()) * rs_rows + res.double()) < tol
    y, pre = ops.gemm_nt(A, B, bias=bias, act="gelu", preact=True)
    assert rel(y, F.gelu(ref + bias.double())) < tol and rel(pre, ref + bias.double()) < tol
    assert rel(ops.gemm_nt(A, B, bias=bias, act="gelu"), F.gelu(ref + bias.double())) < tol
    h = torch.randn(M, N, generator=g).cuda()
    hh = h.double().clone().requires_grad_(True); F.gelu(hh).sum().backward()
    assert rel(ops.gemm_nt(A, B, dact_preact=h, dact="gelu", rowscale=rsc, rows_per_scale=64), ref * hh.grad * rs_rows) < tol
    # BatchNorm-backward epilogue: dz = (A B^T) * gelu'(BN(ysaved)) with the column sums (sum dz, sum dz*xhat)
    ysv = torch.randn(M, N, generator=g).cuda() * 1.5 + 0.2
    mean, rstd = ysv.double().mean(0), (ysv.double().var(0, unbiased=False) + 1e-5).rsqrt()
    gam, bet = (torch.rand(N, generator=g) + 0.5).cuda(), (torch.randn(N, generator=g) * 0.2).cuda()
    stat = torch.stack([mean, rstd]).float()
    xh = (ysv.double() - mean) * rstd
    pre2 = (gam.double() * xh + bet.double()).clone().requires_grad_(True); F.gelu(pre2).sum().backward()
    dz_ref = ref * pre2.grad
    dz, coef, dg, db = ops.conv_dgrad_bn_bwd(A, B, ysv, stat, gam, bet, act="gelu")
    assert rel(dz, dz_ref) < tol
    assert rel(db, dz_ref.sum(0)) < 1e-4 and rel(dg, (dz_ref * xh).sum(0)) < 1e-4
    if K <= 384:
        # the A-prologue kernels on the same tiles: BN+GELU of A while staging (+ column statistics), and the two-source affine of BatchNorm backward
        sa = torch.stack([A.double().mean(0), (A.double().var(0, unbiased=False) + 1e-5).rsqrt()]).float()
        ga, ba = (torch.rand(K, generator=g) + 0.5).cuda(), (torch.randn(K, generator=g) * 0.2).cuda()
        Ap = F.gelu((A.double() - sa[0].double()) * sa[1].double() * ga.double() + ba.double())
        c, st = ops.conv_bn_prologue(A, sa, ga, ba, B, act="gelu", colstats=True)
        refp = Ap @ B.double().t()
        assert rel(c, refp) < tol and rel(st.sum(0)[0], refp.sum(0)) < 1e-4 and rel(st.sum(0)[1], (refp * refp).sum(0)) < 1e-4
        y2 = torch.randn(M, K, generator=g).cuda(); coef = torch.stack([1.0 + 0.1 * torch.randn(K, generator=g), 0.05 * torch.randn(K, generator=g), 0.02 * torch.randn(K, generator=g)]).cuda()
        Wm = B.t().contiguous()                                          # folded_dgrad takes the conv weight [Cout = K, Cin = N]
        dx = ops.folded_dgrad(A, y2, Wm, coef, None, residual=res)
        refd = (coef[0].double() * A.double() + coef[1].double() * y2.double() + coef[2].double()) @ Wm.double() + res.double()
        assert rel(dx, refd) < tol


def test_gemm_tn_many_slabs_small_matrix(ops):
    """patch_embed.conv1's weight-gradient shape (48 x 32 from millions of rows): >= 64 split slabs of a tiny matrix take the
    column x slab-lane reduction kernel."""
    M, N, K = 70000, 48, 32
    X, dY = rnd(M, K, seed=40), rnd(M, N, seed=41, scale=0.1)
    got = ops.gemm_tn(dev(dY, BF), dev(X, BF))
    close(got, dY.to(BF).float().t() @ X.to(BF).float(), rtol=1e-3, atol=2e-2, what="tn gemm, 68 slabs")
    acc = torch.full((N, K), 2.0, device="cuda")
    got = ops.gemm_tn(dev(dY, BF), dev(X, BF), accumulate_into=acc)
    close(got, 2.0 + dY.to(BF).float().t() @ X.to(BF).float(), rtol=1e-3, atol=2e-2, what="tn gemm accumulate")


def test_gemm_splitk_and_wgrad_form(ops):
    """wgrad: dW[N,K] = dY^T X with the reduction (rows) split across blockIdx.y."""
    Mrows, N, K = 5000, 96, 160
    X, dY = rnd(Mrows, K, seed=8), rnd(Mrows, N, seed=9, scale=0.1)
    XT, dYT = ops.transpose_bf16(dev(X, BF)), ops.transpose_bf16(dev(dY, BF))
    assert XT.shape == (K, 5000) and dYT.shape == (N, 5000)
    close(XT, X.t(), rtol=0, atol=0, what="transpose")
    acc = torch.ones(N, K, device="cuda")
    got = ops.gemm_splitk(dYT, XT, split_k=7, accumulate_into=acc)
    close(got, 1.0 + dY.t() @ X, rtol=1e-4, atol=1e-2, what="split-k wgrad")
    # row-scaled transpose (DropPath) and odd row count (zero padded to a multiple of 8)
    rs = torch.tensor([2.0, 0.0, 0.5])
    Y = rnd(333, 40, seed=10)
    YT = ops.transpose_bf16(dev(Y, BF), rowscale=dev(rs), rows_per_scale=111)
    assert YT.shape == (40, 336)
    close(YT[:, :333], (Y * rs.repeat_interleave(111)[:, None]).t(), what="scaled transpose")
    assert float(YT[:, 333:].abs().max()) == 0.0
    close(ops.colsum_bf16(dev(Y, BF), rowscale=dev(rs), rows_per_scale=111), (Y * rs.repeat_interleave(111)[:, None]).sum(0),
          rtol=1e-4, atol=1e-3, what="colsum_bf16")


@pytest.mark.parametrize("M,K,N,act", [(3000, 384, 96, "gelu"), (1000, 160, 48, "gelu"), (777, 64, 200, None)])
def test_gemm_batchnorm_prologue(ops, M, K, N, act):
    """conv3(act(BN2(y2))) with BN+act applied in the GEMM's A prologue == bn_apply followed by the plain GEMM."""
    y = (rnd(M, K, seed=80, scale=1.5) + 0.4).to(BF).float()
    W = (rnd(N, K, seed=81) / K ** 0.5).to(BF).float()
    gamma, beta = 1 + 0.2 * rnd(K, seed=82), 0.3 * rnd(K, seed=83)
    mean, var = y.mean(0), y.var(0, unbiased=False)
    stat = dev(torch.stack([mean, torch.rsqrt(var + 1e-5)]))
    z = F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5)
    a = (F.gelu(z) if act else z).to(BF).float()
    ref = a @ W.T
    out, stats = ops.conv_bn_prologue(dev(y, BF), stat, dev(gamma), dev(beta), dev(W, BF), act=act, colstats=True)
    close(out, ref, rtol=2e-2, atol=2e-2, what="prologue gemm")
    s = stats.cpu().sum(0)
    oq = out.float().cpu()
    close(s[0], oq.sum(0), rtol=1e-3, atol=0.05, what="prologue colsum")
    close(s[1], (oq * oq).sum(0), rtol=1e-3, atol=0.05, what="prologue colsumsq")


@pytest.mark.parametrize("M,Cin,Cmid,Cout", [(3000, 96, 384, 96), (1111, 64, 128, 192)])
def test_convnorm_chain_backward_fused_into_gemms(ops, M, Cin, Cmid, Cout):
    """x -conv1-> y1 -BN(train)+GELU-> a1 -conv3-> y3: dgrad of conv3 carries BN backward's reduce in its epilogue, dgrad of
    conv1 reads (dz1, y1) through BN-folded weights.  Checked against fp32 autograd."""
    x = rnd(M, Cin, seed=70).to(BF).float()
    W1 = (rnd(Cmid, Cin, seed=71) / Cin ** 0.5)
    W3 = (rnd(Cout, Cmid, seed=72) / Cmid ** 0.5).to(BF).float()
    gamma, beta = 1 + 0.2 * rnd(Cmid, seed=73), 0.3 * rnd(Cmid, seed=74)
    G = rnd(M, Cout, seed=75).to(BF).float()
    skip = rnd(M, Cin, seed=76).to(BF).float()
    y1 = ((x @ W1.to(BF).float().T) + 1.5).to(BF).float()          # off-centre channels: exercises the mean correction
    yr = y1.clone().requires_grad_(True)
    g_, b_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    a1 = F.gelu(F.batch_norm(yr, None, None, g_, b_, True, 0.1, 1e-5))
    (a1 @ W3.T).backward(G)
    dx_ref = yr.grad @ W1 + skip

    mean, var = y1.mean(0), y1.var(0, unbiased=False)
    stat = dev(torch.stack([mean, torch.rsqrt(var + 1e-5)]))
    W3t = dev(W3.T.contiguous(), BF)                                 # [Cmid, Cout]
    dz, coef, dg, db = ops.conv_dgrad_bn_bwd(dev(G, BF), W3t, dev(y1, BF), stat, dev(gamma), dev(beta), act="gelu")
    dy = coef[0] * dz.float() + coef[1] * dev(y1) + coef[2]
    close(dy, yr.grad, rtol=2e-2, atol=2e-2, what="dy from gemm-epilogue dz + coef")
    close(dg, g_.grad, rtol=2e-2, atol=0.5, what="dgamma")
    close(db, b_.grad, rtol=2e-2, atol=0.5, what="dbeta")
    dx = ops.folded_dgrad(dz, dev(y1, BF), dev(W1), coef, stat, residual=dev(skip, BF))
    close(dx, dx_ref, rtol=2e-2, atol=3e-2, what="folded dgrad")


@pytest.mark.parametrize("M,N,K", [(5000, 96, 160), (4099, 48, 32), (1030, 576, 2304), (777, 1728, 576), (9000, 96, 432)])
def test_gemm_tn_weight_gradient(ops, M, N, K):
    """dW = dY^T X straight from the row-major operands (transposing LDS reads), with a DropPath row scale."""
    X, dY = rnd(M, K, seed=81), rnd(M, N, seed=82, scale=0.1)
    got = ops.gemm_tn(dev(dY, BF), dev(X, BF))
    close(got, dY.t() @ X, rtol=1e-4, atol=2e-2, what="gemm_tn")
    T = 10
    rs = (torch.arange((M + T - 1) // T) % 3).float() * 0.75
    acc = torch.full((N, K), 2.0, device="cuda")
    got = ops.gemm_tn(dev(dY, BF), dev(X, BF), rowscale=dev(rs), rows_per_scale=T, accumulate_into=acc)
    dYs = (dY * rs.repeat_interleave(T)[:M, None]).to(BF).float()
    close(got, 2.0 + dYs.t() @ X, rtol=1e-4, atol=2e-2, what="gemm_tn rowscale")


# ------------------------------------------------------------------------------------------- convolutions
@pytest.mark.parametrize("H", [20, 18])          # W % 4 == 0: LDS-staged NCHW gather; otherwise the per-pixel gather
def test_im2col_matches_conv(ops, H):
    B = 2
    x = rnd(B, 3, H, H, seed=11)
    w = rnd(16, 3, 3, 3, seed=12, scale=0.3)
    col = ops.im2col_nchw3(dev(x), stride=2)                      # [B*Ho*Ho, 32], k = (ky,kx,ci)
    wk = torch.zeros(16, 32)
    wk[:, :27] = w.permute(0, 2, 3, 1).reshape(16, 27)
    Ho = H // 2
    got = ops.gemm_nt(col, dev(wk, BF), out_f32=True).view(B, Ho, Ho, 16).permute(0, 3, 1, 2)
    close(got, F.conv2d(x, w, None, 2, 1), rtol=1e-4, atol=1e-3, what="conv1 via im2col")
    C = 16
    xh = rnd(B, H, H, C, seed=13)
    w2 = rnd(24, C, 3, 3, seed=14, scale=0.2)
    col2 = ops.im2col_nhwc(dev(xh, BF), stride=2)
    got = ops.gemm_nt(col2, dev(w2.permute(0, 2, 3, 1).reshape(24, 9 * C), BF), out_f32=True).view(B, Ho, Ho, 24).permute(0, 3, 1, 2)
    ref = F.conv2d(xh.permute(0, 3, 1, 2), w2, None, 2, 1)
    close(got, ref, rtol=1e-4, atol=1e-3, what="conv2 via im2col")
    # col2im is the adjoint of im2col: <im2col(x), d> == <x, col2im(d)>
    d = rnd(B * Ho * Ho, 9 * C, seed=15)
    dx = ops.col2im_nhwc(dev(d, BF), B, H, H, C, stride=2)
    xr = xh.clone().requires_grad_(True)
    cols_ref = F.unfold(xr.permute(0, 3, 1, 2), 3, padding=1, stride=2)             # (B, C*9, L), k = (c,ky,kx)
    cols_ref = cols_ref.view(B, C, 9, Ho * Ho).permute(0, 3, 2, 1).reshape(B * Ho * Ho, 9 * C)   # -> (ky,kx,c)
    (cols_ref * d).sum().backward()
    close(dx, xr.grad, what="col2im")


def test_im2col_fused_batchnorm_gelu(ops):
    """im2col over GELU(BatchNorm(y)) of a saved conv output == BatchNorm-apply pass followed by the plain im2col (the activation
    tensor the reference materialises, patch_embed.conv1 -> conv2: tiny_vit.py PatchEmbed); padding taps stay exactly zero."""
    B, H, C = 3, 18, 48
    y = dev(rnd(B, H, H, C, seed=16, scale=2.0), BF)
    mean, var = rnd(C, seed=17, scale=0.5), rnd(C, seed=18).abs() + 0.5
    stat = dev(torch.stack([mean, (var + 1e-5).rsqrt()]))
    gamma, beta = dev(rnd(C, seed=19) + 1.0), dev(rnd(C, seed=20, scale=0.3))
    for act in ("gelu", None):
        a = ops.bn_apply(y.view(-1, C), stat, gamma, beta, act=act).view(B, H, H, C)
        want = ops.im2col_nhwc(a, stride=2)
        got = ops.im2col_nhwc_bn(y, stat, gamma, beta, act=act, stride=2)
        assert ((want == 0) == (got == 0)).all() or (want.float() - got.float()).abs().max() < 1e-2
        close(got, want.float(), rtol=8e-3, atol=1e-3, what=f"fused im2col act={act}")      # <= 1 bf16 ulp (scalar vs paired GELU form)
        border = got.view(B, H // 2, H // 2, 9, C)[:, 0, :, 0:3]                               # ky = 0 taps of the first output row: padding
        assert (border == 0).all()


@pytest.mark.parametrize("act", ["gelu", None])
def test_col2im_fused_batchnorm_backward_reduce(ops, act):
    """col2im + BN-backward reduce in one pass (patch_embed: conv2 dgrad -> BN1/GELU backward) == col2im, then dz = da * act'(BN(y)) and the
    column sums (sum dz, sum dz * xhat) the reduce pass would produce."""
    B, H, C = 3, 18, 48
    Ho = H // 2
    dcol = dev(rnd(B * Ho * Ho, 9 * C, seed=21, scale=0.5), BF)
    y = rnd(B, H, H, C, seed=22, scale=1.5).to(BF).float()
    mean, var = rnd(C, seed=23, scale=0.3), rnd(C, seed=24).abs() + 0.5
    rstd = (var + 1e-5).rsqrt()
    gamma, beta = rnd(C, seed=25) + 1.0, rnd(C, seed=26, scale=0.3)
    da = ops.col2im_nhwc(dcol, B, H, H, C, stride=2).float().cpu()             # bf16-rounded, as the unfused path stores it
    xh = (y - mean) * rstd
    pre = (gamma * xh + beta).clone().requires_grad_(True)
    (F.gelu(pre) if act else pre).sum().backward()
    dpre = da * pre.grad
    dz, part = ops.col2im_nhwc_bnbwd(dcol, dev(y, BF), dev(torch.stack([mean, rstd])), dev(gamma), dev(beta), act=act, nparts=37)
    close(dz, dpre, rtol=8e-3, atol=1e-3, what="fused col2im dz")
    sums = part.cpu().sum(0)
    close(sums[0], dpre.sum((0, 1, 2)), rtol=2e-3, atol=2e-2, what="sum dz")
    close(sums[1], (dpre * xh).sum((0, 1, 2)), rtol=2e-3, atol=2e-2, what="sum dz*xhat")


@pytest.mark.parametrize("act,C,H", [("gelu", 48, 18), (None, 48, 18), ("gelu", 36, 11)])
def test_f32_col2im_fused_batchnorm_backward_reduce(ops, act, C, H):
    """f32 twin (gg_col2im_nhwc_bnbwd_f32): dz equals gg_col2im_nhwc_f32's da times act'(BN(y)); odd sizes cover the border taps."""
    B = 3
    Ho = (H - 1) // 2 + 1
    dcol = dev(rnd(B * Ho * Ho, 9 * C, seed=21, scale=0.5))
    y = rnd(B, H, H, C, seed=22, scale=1.5)
    mean, var = rnd(C, seed=23, scale=0.3), rnd(C, seed=24).abs() + 0.5
    rstd = (var + 1e-5).rsqrt()
    gamma, beta = rnd(C, seed=25) + 1.0, rnd(C, seed=26, scale=0.3)
    da = ops.col2im_nhwc(dcol, B, H, H, C, stride=2).cpu()
    assert da.dtype == torch.float32
    xh = (y - mean) * rstd
    pre = (gamma * xh + beta).double().clone().requires_grad_(True)
    (F.gelu(pre) if act else pre).sum().backward()
    dpre = da.double() * pre.grad
    dz, part = ops.col2im_nhwc_bnbwd(dcol, dev(y), dev(torch.stack([mean, rstd])), dev(gamma), dev(beta), act=act, nparts=37)
    close(dz, dpre.float(), rtol=2e-5, atol=2e-6, what="f32 fused col2im dz")
    sums = part.cpu().double().sum(0)
    close(sums[0].float(), dpre.sum((0, 1, 2)).float(), rtol=1e-4, atol=1e-3, what="f32 sum dz")
    close(sums[1].float(), (dpre * xh.double()).sum((0, 1, 2)).float(), rtol=1e-4, atol=1e-3, what="f32 sum dz*xhat")


@pytest.mark.parametrize("C,stride,H", [(16, 1, 12), (48, 2, 14), (384, 1, 8), (576, 2, 14), (40, 1, 7), (24, 2, 9), (64, 2, 15)])
def test_dwconv(ops, C, stride, H):
    B = 3
    x = rnd(B, H, H, C, seed=20)
    w = rnd(C, 1, 3, 3, seed=21, scale=0.4)
    taps = w.view(C, 9).t().contiguous()
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yref = F.conv2d(xr, wr, None, stride, 1, 1, C)
    y, stats = ops.dwconv3x3_fwd(dev(x, BF), dev(taps), stride=stride, colstats=True)
    close(y.permute(0, 3, 1, 2), yref, what="dwconv fwd")
    s = stats.cpu().sum(0)
    # BatchNorm partials are taken over the STORED (bf16-rounded) result, as the reference's autocast BatchNorm sees it
    yq = y.float().cpu()
    close(s[0], yq.sum((0, 1, 2)), rtol=1e-3, atol=1e-2, what="dw colsum")
    close(s[1], (yq * yq).sum((0, 1, 2)), rtol=1e-3, atol=1e-2, what="dw colsumsq")
    close(s[0], yref.sum((0, 2, 3)), rtol=5e-3, atol=0.3, what="dw colsum vs fp32 reference")
    Ho = yref.shape[-1]
    dy = rnd(B, Ho, Ho, C, seed=22)
    yref.backward(dy.permute(0, 3, 1, 2))
    dx = ops.dwconv3x3_bwd_data(dev(dy, BF), dev(taps), B, H, H, C, stride=stride)
    close(dx.permute(0, 3, 1, 2), xr.grad, what="dwconv dgrad")
    dw = ops.dwconv3x3_bwd_weight(dev(x, BF), dev(dy, BF), stride=stride)
    close(dw, wr.grad, rtol=1e-3, atol=1e-2, what="dwconv wgrad")


@pytest.mark.parametrize("C,H,act,stride", [(192, 28, "gelu", 2), (48, 15, "gelu", 2), (384, 14, None, 2),
                                            (384, 14, "gelu", 1), (16, 13, "gelu", 1), (40, 7, None, 1), (96, 30, "gelu", 1)])
def test_dwconv_with_batchnorm_gelu_on_load(ops, C, H, act, stride):
    """PatchMerging.conv2 (stride 2) / MBConv.conv2 (stride 1) over GELU(BN1(y1)) formed while loading == BatchNorm-apply pass + plain
    depthwise conv (same bf16 rounding of the activation), including the BatchNorm partial statistics of the result; odd sizes and
    widths that are not a multiple of the 4 columns a thread owns exercise the padding and tail masks."""
    B = 3
    y1 = dev(rnd(B, H, H, C, seed=50, scale=1.5), BF)
    mean, var = rnd(C, seed=51, scale=0.4), rnd(C, seed=52).abs() + 0.5
    stat = dev(torch.stack([mean, (var + 1e-5).rsqrt()]))
    gamma, beta = dev(rnd(C, seed=53) + 1.0), dev(rnd(C, seed=54, scale=0.3))
    taps = dev(rnd(9, C, seed=55, scale=0.4))
    a1 = ops.bn_apply(y1.view(-1, C), stat, gamma, beta, act=act).view(B, H, H, C)
    want, wstats = ops.dwconv3x3_fwd(a1, taps, stride=stride, colstats=True)
    got, gstats = ops.dwconv3x3_fwd_fused(y1, stat, gamma, beta, taps, act=act, stride=stride, colstats=True)
    close(got, want.float(), rtol=1e-2, atol=1e-2, what="fused dwconv")
    close(gstats.sum(0), wstats.sum(0).cpu(), rtol=2e-3, atol=5e-2, what="fused dwconv statistics")
    ref = F.conv2d(a1.float().cpu().permute(0, 3, 1, 2), taps.cpu().t().reshape(C, 1, 3, 3), None, stride, 1, 1, C)
    close(got.permute(0, 3, 1, 2), ref, what="fused dwconv vs conv2d")


# ------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("act,with_res", [(None, False), ("gelu", False), ("gelu", True)])
def test_batchnorm_train(ops, act, with_res):
    M, C, T = 840, 48, 210
    y = (rnd(M, C, seed=30, scale=2.0) + 0.5).to(BF).float()
    gamma, beta = 1 + 0.2 * rnd(C, seed=31), 0.1 * rnd(C, seed=32)
    res = rnd(M, C, seed=33)
    rs = torch.tensor([1.25, 0.0, 1.25, 1.25])
    rsr = rs.repeat_interleave(T)[:, None]
    # statistics from synthetic per-block partials
    parts = torch.stack([torch.stack([y[i:i + 128].sum(0), (y[i:i + 128] ** 2).sum(0)]) for i in range(0, M, 128)])
    rm, rv = torch.zeros(C), torch.ones(C)
    stat = ops.bn_finalize(dev(parts), M, 1e-5, 0.1, rmd := dev(rm.clone()), rvd := dev(rv.clone()))
    mean, var = y.mean(0), y.var(0, unbiased=False)
    close(stat[0], mean, rtol=1e-5, atol=1e-5, what="bn mean")
    close(stat[1], torch.rsqrt(var + 1e-5), rtol=1e-4, atol=1e-5, what="bn rstd")
    close(rmd, 0.1 * mean, rtol=1e-5, atol=1e-6, what="running_mean")
    close(rvd, 0.9 + 0.1 * y.var(0, unbiased=True), rtol=1e-4, atol=1e-6, what="running_var")

    yr = y.clone().requires_grad_(True)
    g_, b_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z = F.batch_norm(yr, None, None, g_, b_, True, 0.1, 1e-5)
    if with_res:
        z = res + rsr * z
    out_ref = F.gelu(z) if act else z
    out = ops.bn_apply(dev(y, BF), stat, dev(gamma), dev(beta), act=act, residual=dev(res, BF) if with_res else None,
                       rowscale=dev(rs) if with_res else None, rows_per_scale=T)
    close(out, out_ref, what="bn apply")
    dout = rnd(M, C, seed=34)
    out_ref.backward(dout)
    dz, dy, dg, db = ops.bn_bwd(dev(dout, BF), dev(y, BF), stat, dev(gamma), dev(beta), act=act,
                                residual=dev(res, BF) if with_res else None, rowscale=dev(rs) if with_res else None, rows_per_scale=T)
    close(dy, yr.grad, rtol=2e-2, atol=1e-2, what="bn dy")
    close(dg, g_.grad, rtol=1e-2, atol=0.5, what="bn dgamma")
    close(db, b_.grad, rtol=1e-2, atol=0.5, what="bn dbeta")
    if with_res:   # dz is the gradient w.r.t. the pre-activation = skip-path gradient of an MBConv
        zz = z.detach().clone().requires_grad_(True)
        F.gelu(zz).backward(dout)
        close(dz, zz.grad, what="bn dz (skip grad)")


@pytest.mark.parametrize("C,f32", [(192, False), (576, False), (160, False), (768, False), (576, True), (1024, False), (384, True), (192, True), (1024, True),
                                   (160, True)])
def test_layernorm(ops, C, f32):
    M = 301
    x = (rnd(M, C, seed=40, scale=1.5) + 0.3).to(BF).float()
    gamma, beta = 1 + 0.2 * rnd(C, seed=41), 0.1 * rnd(C, seed=42)
    xr, g_, b_ = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = F.layer_norm(xr, (C,), g_, b_, 1e-5)
    xin = dev(x) if f32 else dev(x, BF)
    out, mean, rstd = ops.layernorm_fwd(xin, dev(gamma), dev(beta))
    close(out, ref, rtol=1e-4 if f32 else 1e-2, atol=1e-4 if f32 else 1e-2, what="ln fwd")
    close(mean, x.mean(1), rtol=1e-5, atol=1e-5, what="ln mean")
    dout, dres = rnd(M, C, seed=43), rnd(M, C, seed=44)
    ref.backward(dout)
    dx, dg, db = ops.layernorm_bwd(dev(dout) if f32 else dev(dout, BF), xin, mean, rstd, dev(gamma), dres=dev(dres) if f32 else dev(dres, BF))
    close(dx, xr.grad + dres, rtol=1e-4 if f32 else 2e-2, atol=1e-4 if f32 else 2e-2, what="ln dx")
    close(dg, g_.grad, rtol=1e-2, atol=0.3, what="ln dgamma")
    close(db, b_.grad, rtol=1e-2, atol=0.3, what="ln dbeta")


@pytest.mark.parametrize("M,C,f32", [(9000, 384, True), (2500, 192, True), (700, 576, True), (333, 96, True), (9000, 384, False), (1200, 192, False)])
def test_layernorm_bwd_with_batchnorm_column_sums(ops, M, C, f32):
    """x = BN_train(y) -> LN(x): gg_layernorm_bwd_colsum's dx equals gg_layernorm_bwd's, and the BatchNorm-backward coefficients
    gg_bn_bwd_coef_from_x derives from its (sum dx*x, sum dx) rows reproduce torch's dL/dy (timm TinyVitBlock: local_conv.bn -> norm2)."""
    y = rnd(M, C, seed=80, scale=1.7) + 0.4
    bg, bb = rnd(C, seed=81) * 0.3 + 1.0, rnd(C, seed=82, scale=0.5)
    g, b = rnd(C, seed=83) * 0.2 + 1.0, rnd(C, seed=84, scale=0.2)
    dout, dres = rnd(M, C, seed=85), rnd(M, C, seed=86)
    if not f32:
        y, dout, dres = y.to(BF).float(), dout.to(BF).float(), dres.to(BF).float()
    yr = y.double().clone().requires_grad_(True)
    x_ref = F.batch_norm(yr, None, None, bg.double(), bb.double(), True, 0.1, 1e-5)
    (F.layer_norm(x_ref, (C,), g.double(), b.double(), 1e-5) * dout.double()).sum().backward(retain_graph=True)
    x_ref.backward(dres.double())          # the residual path: x also feeds the block output directly
    stat = torch.stack([y.double().mean(0), (y.double().var(0, unbiased=False) + 1e-5).rsqrt()]).float()
    dt = torch.float32 if f32 else BF
    x = ops.bn_apply(dev(y, dt), dev(stat), dev(bg), dev(bb))
    _, mean, rstd = ops.layernorm_fwd(x, dev(g), dev(b))
    dx0, _, _ = ops.layernorm_bwd(dev(dout, dt), x, mean, rstd, dev(g), dres=dev(dres, dt), want_param_grads=False)
    dx, part, rows = ops.layernorm_bwd_colsum(dev(dout, dt), x, mean, rstd, dev(g), dres=dev(dres, dt))
    close(dx, dx0.float(), rtol=1e-6 if f32 else 8e-3, atol=1e-6 if f32 else 1e-3, what="dx of the colsum form")      # fma contraction may differ
    coef = ops.bn_bwd_coef_from_x(part, rows, M, dev(stat), dev(bg), dev(bb)).cpu().double()
    dy = coef[0] * dx.double().cpu() + coef[1] * y.double() + coef[2]
    ref = yr.grad
    tol = 2e-5 if f32 else 3e-2
    err = float((dy - ref).abs().max() / ref.abs().max())
    assert err < tol, f"dy through coef_from_x: {err:.2e}"
    # against the existing two-pass route on the same dx
    _, dy2, _, _ = ops.bn_bwd(dx, dev(y, dt), dev(stat), dev(bg), dev(bb), want_param_grads=False)
    err2 = float((dy.float() - dy2.float().cpu()).abs().max() / ref.abs().max())
    assert err2 < (1e-5 if f32 else 2e-2), f"coef_from_x vs reduce+finalize: {err2:.2e}"


def test_pooling(ops):
    B, T, C = 5, 49, 64
    x = rnd(B * T, C, seed=50)
    close(ops.token_mean_fwd(dev(x, BF), B, T), x.view(B, T, C).mean(1), rtol=1e-5, atol=1e-5, what="token mean")
    d = rnd(B, C, seed=51)
    close(ops.token_mean_bwd(dev(d), T), (d / T).repeat_interleave(T, 0), what="token mean bwd")


@pytest.mark.parametrize("M,C", [(1000, 384), (333, 192), (77, 576), (50, 40)])
def test_layernorm_with_batchnorm_apply_on_load(ops, M, C):
    """LN(BN(y)) with the BatchNorm apply folded into the LayerNorm load == gg_bn_apply followed by gg_layernorm_fwd, bit for bit
    (TinyVitBlock: local_conv's BatchNorm -> norm2; the applied tensor is the residual stream and is written by the same kernel)."""
    y = dev(rnd(M, C, seed=70, scale=2.0), BF)
    mean, var = rnd(C, seed=71, scale=0.5), rnd(C, seed=72).abs() + 0.5
    stat = dev(torch.stack([mean, (var + 1e-5).rsqrt()]))
    bg, bb = dev(rnd(C, seed=73) + 1.0), dev(rnd(C, seed=74, scale=0.3))
    g, b = dev(rnd(C, seed=75) + 1.0), dev(rnd(C, seed=76, scale=0.2))
    x_ref = ops.bn_apply(y, stat, bg, bb)
    o_ref, m_ref, r_ref = ops.layernorm_fwd(x_ref, g, b)
    x, o, m, r = ops.layernorm_fwd_bn(y, stat, bg, bb, g, b)
    close(x, x_ref.float(), rtol=8e-3, atol=1e-3, what="bn-applied stream")          # <= 1 bf16 ulp (fma contraction may differ)
    close(o, o_ref.float(), rtol=2e-2, atol=2e-2, what="ln of bn")
    x32 = x.float().cpu()
    close(o, F.layer_norm(x32, (C,), g.cpu(), b.cpu(), 1e-5), rtol=1e-2, atol=1e-2, what="ln vs torch on the written stream")
    close(m, x32.mean(1), rtol=1e-4, atol=1e-4, what="ln mean")


@pytest.mark.parametrize("M,C", [(1000, 384), (333, 192), (77, 576), (50, 40)])
def test_f32_layernorm_with_batchnorm_apply_on_load(ops, M, C):
    """fp32 twin: gg_layernorm_fwd_bn_f32 == gg_bn_apply_f32 then gg_layernorm_fwd on f32 storage, and == torch."""
    y = rnd(M, C, seed=70, scale=2.0)
    mean, var = rnd(C, seed=71, scale=0.5), rnd(C, seed=72).abs() + 0.5
    stat = torch.stack([mean, (var + 1e-5).rsqrt()])
    bg, bb = rnd(C, seed=73) + 1.0, rnd(C, seed=74, scale=0.3)
    g, b = rnd(C, seed=75) + 1.0, rnd(C, seed=76, scale=0.2)
    x_ref = ops.bn_apply(dev(y), dev(stat), dev(bg), dev(bb))
    o_ref, m_ref, r_ref = ops.layernorm_fwd(x_ref, dev(g), dev(b))
    x, o, m, r = ops.layernorm_fwd_bn(dev(y), dev(stat), dev(bg), dev(bb), dev(g), dev(b))
    assert x.dtype == torch.float32 and o.dtype == torch.float32
    close(x, x_ref, rtol=1e-6, atol=1e-6, what="f32 bn-applied stream")
    close(o, o_ref, rtol=1e-5, atol=1e-5, what="f32 ln of bn")
    xt = (y - mean) * stat[1] * bg + bb
    close(o, F.layer_norm(xt, (C,), g, b, 1e-5), rtol=1e-4, atol=1e-4, what="f32 ln of bn vs torch")
    close(m, xt.mean(1), rtol=1e-5, atol=1e-5, what="f32 ln mean")


# ------------------------------------------------------------------------------------------- attention
def _attn_ref(qkv, nh, hd, ws, Hm, Wm, B, bias, layout):
    """fp32 reference of timm Attention (per-head interleaved qkv, window partition, rel-pos bias) / CLIP MHSA."""
    from oracle.tinyvit_ref import attention_bias_idxs
    M = qkv.shape[0]
    if layout == "tinyvit":
        x = qkv.view(B, Hm // ws, ws, Wm // ws, ws, nh, 3 * hd).permute(0, 1, 3, 5, 2, 4, 6).reshape(-1, nh, ws * ws, 3 * hd)
        q, k, v = x.split([hd, hd, hd], -1)
        s = q @ k.transpose(-1, -2) * hd ** -0.5 + bias[:, attention_bias_idxs(ws)][None]
    else:
        T = M // B
        x = qkv.view(B, T, 3, nh, hd).permute(2, 0, 3, 1, 4)
        q, k, v = x[0], x[1], x[2]
        s = q @ k.transpose(-1, -2) * hd ** -0.5
    o = s.softmax(-1) @ v
    if layout == "tinyvit":
        o = o.view(B, Hm // ws, Wm // ws, nh, ws, ws, hd).permute(0, 1, 4, 2, 5, 3, 6).reshape(M, nh * hd)
    else:
        o = o.permute(0, 2, 1, 3).reshape(M, nh * hd)
    return o


@pytest.mark.parametrize("ws,Hm,nh", [(7, 14, 2), (14, 14, 3), (7, 7, 2), (12, 12, 1), (16, 16, 1), (7, 35, 2), (7, 42, 3)])
def test_window_attention_fwd_bwd(ops, ws, Hm, nh):
    B, hd = 3, 32
    C = nh * hd
    M = B * Hm * Hm
    qkv = rnd(M, 3 * C, seed=60, scale=1.0)
    bias = rnd(nh, ws * ws, seed=61, scale=0.5)
    qr, br = qkv.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    ref = _attn_ref(qr, nh, hd, ws, Hm, Hm, B, br, "tinyvit")
    kw = dict(num_windows=B * (Hm // ws) ** 2, tokens_per_window=ws * ws, num_heads=nh, head_dim=hd, q_off=0, k_off=hd,
              v_off=2 * hd, head_stride=3 * hd, window_size=ws, map_h=Hm, map_w=Hm, bias=dev(bias))
    out, lse = ops.attention(dev(qkv, BF), want_lse=True, **kw)
    close(out, ref, rtol=2e-2, atol=2e-2, what="attn fwd")
    dout = rnd(M, C, seed=62)
    ref.backward(dout)
    dqkv, dbias = ops.attention(dev(qkv, BF), dout=dev(dout, BF), want_dbias=True, out=out, lse=lse, **kw)
    close(dqkv, qr.grad, rtol=3e-2, atol=3e-2, what="attn dqkv")
    close(dbias, br.grad, rtol=3e-2, atol=5e-2, what="attn dbias")
    # frozen-bias variant of the kernel (no dbias bins) must give the same dqkv; (7,35)/(7,42) have >= 64 windows and take the
    # multi-window kernels (75 windows = a ragged last group of 8)
    dqkv2 = ops.attention(dev(qkv, BF), dout=dev(dout, BF), out=out, lse=lse, **kw)
    dqkv2 = dqkv2[0] if isinstance(dqkv2, tuple) else dqkv2
    assert torch.equal(dqkv2, dqkv)


def test_clip_attention_fwd(ops):
    B, T, nh, hd = 4, 50, 3, 64
    D = nh * hd
    qkv = rnd(B * T, 3 * D, seed=63)
    ref = _attn_ref(qkv, nh, hd, 0, 0, 0, B, None, "clip")
    out = ops.attention(dev(qkv, BF), num_windows=B, tokens_per_window=T, num_heads=nh, head_dim=hd, q_off=0, k_off=D, v_off=2 * D,
                        head_stride=hd)
    close(out, ref, rtol=2e-2, atol=2e-2, what="clip attn")


@pytest.mark.parametrize("B,T,nh,hd", [(4, 50, 3, 64), (2, 197, 2, 64), (3, 49, 2, 32), (1, 256, 1, 64)])
def test_clip_attention_fwd_fp16(ops, B, T, nh, hd):
    """fp16 storage + v_mfma_f32_16x16x32_f16 (the CLIP tower's fp16 inference mode, BASELINE c4): 50 tokens (ViT-B/32), 197 (B/16), a 49-token and
    a full 256-token case, against fp32 attention of the fp16-representable inputs; lse against torch.logsumexp."""
    D = nh * hd
    qkv = rnd(B * T, 3 * D, seed=64).half().float()
    ref = _attn_ref(qkv, nh, hd, 0, 0, 0, B, None, "clip")
    out, lse = ops.attention(qkv.cuda().half(), num_windows=B, tokens_per_window=T, num_heads=nh, head_dim=hd, q_off=0, k_off=D, v_off=2 * D, head_stride=hd,
                             want_lse=True)
    assert out.dtype == torch.float16
    close(out, ref, rtol=2e-3, atol=2e-3, what="clip attn fp16")
    q = qkv[:, :D].view(B, T, nh, hd).permute(0, 2, 1, 3)
    k = qkv[:, D:2 * D].view(B, T, nh, hd).permute(0, 2, 1, 3)
    lse_ref = torch.logsumexp((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1).permute(0, 2, 1).reshape(B * T, nh)
    close(lse, lse_ref, rtol=1e-3, atol=1e-3, what="clip attn fp16 lse")


@pytest.mark.parametrize("M,N,K", [(300, 200, 96), (1000, 48, 40), (4099, 384, 384), (129, 1153, 1536), (64, 130, 8)])
def test_split3_gemm_is_fp32_accurate(M, N, K):
    """Experiment kernel: C = A . B^T from three bf16 planes per operand (six bf16 MFMA products, f32 accumulation).  The planes sum to the
    f32 input to 24 bits; the product matches an fp64 product at the level of the f32-MFMA GEMM (rel-L2 < 1e-6) on ragged shapes, a K tail shorter than
    the 32-element stage, and with a bias."""
    import ctypes as C
    from geoguessr_ai_amd import _lib as L, ops
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    B = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()

    def planes(x):
        out = torch.empty((3,) + tuple(x.shape), dtype=torch.bfloat16, device="cuda")
        L.check(L.lib().gg_split3_bf16(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), out.data_ptr(), L.stream()), "gg_split3_bf16")
        return out
    Ap, Bp = planes(A), planes(B)
    resid = (A.double() - Ap.double().sum(0)).abs().max() / A.abs().max()
    assert float(resid) < 2 ** -22, float(resid)                      # x1 + x2 + x3 reproduces x to (at least) 23 bits
    out = torch.empty(M, N, device="cuda")
    L.check(L.lib().gg_gemm_nt_split3(Ap.data_ptr(), K, Bp.data_ptr(), K, out.data_ptr(), N, M, N, K, bias.data_ptr(), L.stream()), "gg_gemm_nt_split3")
    ref = A.double() @ B.double().T + bias.double()
    e = float((out.double() - ref).norm() / ref.norm())
    e32 = float((ops.gemm_nt(A, B, bias=bias).double() - ref).norm() / ref.norm())
    print(f"\n[split3 {M}x{N}x{K}] rel-L2 vs fp64: split3 {e:.2e}, f32-MFMA {e32:.2e}")
    assert e < 1e-6 and e < 2.5 * e32 + 1e-7


def test_split3_gemm_epilogues_match_the_f32_gemm(ops):
    """The split GEMM with the epilogue family of the model's Linears against gg_gemm_nt_f32 with the same epilogue (both f32-accurate: agreement at 1e-5):
    fc1 (bias + GELU + pre-activation copy), fc2 / proj (bias, DropPath row scale, residual), the dgrad through GELU (x GELU'(saved pre-activation) x row
    scale); and the result leaving as three bf16 planes that sum to the f32 result to 23 bits and feed the next split GEMM."""
    import ctypes as C
    from geoguessr_ai_amd import _lib as L
    M, N, K, rps = 777, 200, 96, 49
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g).cuda(); B = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    bias = torch.randn(N, generator=g).cuda(); res = torch.randn(M, N, generator=g).cuda(); pre = torch.randn(M, N, generator=g).cuda()
    scale = (torch.rand((M + rps - 1) // rps, generator=g) > 0.3).float().cuda() / 0.7

    def planes(x):
        out = torch.empty((3,) + tuple(x.shape), dtype=torch.bfloat16, device="cuda")
        L.check(L.lib().gg_split3_bf16(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), out.data_ptr(), L.stream()), "gg_split3_bf16")
        return out
    Ap, Bp = planes(A), planes(B)

    def run(**kw):
        a = L.Split3Args()
        a.a_planes, a.lda, a.b_planes, a.ldb, a.M, a.N, a.K = Ap.data_ptr(), K, Bp.data_ptr(), K, M, N, K
        out = torch.empty(M, N, device="cuda"); a.C, a.ldc = out.data_ptr(), N
        keep = []
        for k, v in kw.items():
            if torch.is_tensor(v):
                keep.append(v); setattr(a, k, v.data_ptr())
            else:
                setattr(a, k, v)
        L.check(L.lib().gg_gemm_nt_split3_ex(C.byref(a), L.stream()), "gg_gemm_nt_split3_ex")
        return out
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
    # fc1: bias + GELU, pre-activation copy
    pre_out = torch.empty(M, N, device="cuda")
    got = run(bias=bias, act=1, preact=pre_out)
    ref, ref_pre = ops.gemm_nt(A, B, bias=bias, act="gelu", preact=True)
    assert rel(got, ref) < 1e-5 and rel(pre_out, ref_pre) < 1e-5
    # fc2 / proj: bias, row scale, residual
    got = run(bias=bias, rowscale=scale, rows_per_scale=rps, residual=res, ldr=N)
    ref = ops.gemm_nt(A, B, bias=bias, rowscale=scale, rows_per_scale=rps, residual=res)
    assert rel(got, ref) < 1e-5
    # dgrad through GELU
    got = run(dact_preact=pre, dact=1, rowscale=scale, rows_per_scale=rps)
    ref = ops.gemm_nt(A, B, dact_preact=pre, dact="gelu", rowscale=scale, rows_per_scale=rps)
    assert rel(got, ref) < 1e-5
    # plane output: sums to the f32 result, and chains into the next split GEMM
    cp = torch.empty(3, M, N, dtype=torch.bfloat16, device="cuda")
    got = run(bias=bias, c_planes=cp, ldp=N)
    assert float((cp.double().sum(0) - got.double()).abs().max() / got.abs().max()) < 2 ** -22
    W2 = (torch.randn(64, N, generator=g) * N ** -0.5).cuda()
    W2p = planes(W2)
    out2 = torch.empty(M, 64, device="cuda")
    L.check(L.lib().gg_gemm_nt_split3(cp.data_ptr(), N, W2p.data_ptr(), N, out2.data_ptr(), 64, M, 64, N, None, L.stream()), "gg_gemm_nt_split3")
    assert rel(out2, got.double() @ W2.double().T) < 1e-6


@pytest.mark.parametrize("M,N,K,lda", [(300, 200, 96, 96), (4099, 384, 384, 400), (1000, 136, 712, 712), (257, 1153, 1536, 1536), (256, 128, 8, 8), (600, 192, 96, 96), (700, 192, 768, 776),
                                       (513, 90, 416, 416)])
def test_split3_gemm_with_f32_activation_operand(ops, M, N, K, lda):
    """gg_gemm_nt_split3_af32 (the fp32_split mode's Linear): A is the f32 activation itself, split into its three bf16 terms while the kernel stages it; the
    weight comes as cached planes.  Both tile forms (K < 384: 128 x 128, two workgroups per CU; else 256 x 128), each with 128- and 96-column tiles (N = 192, 90), on ragged shapes, a K that is not a multiple
    of the 32-element stage, a strided A: the result is BIT-IDENTICAL to the plane-fed kernel on pre-split planes of the same A (same products, same order),
    f32-accurate against fp64, and the epilogue family (bias + GELU + pre-activation copy; row scale + residual; x GELU') matches gg_gemm_nt_f32."""
    import ctypes as C
    from geoguessr_ai_amd import _lib as L
    g = torch.Generator().manual_seed(M + N + K)
    Afull = torch.randn(M, lda, generator=g).cuda()
    A = Afull[:, :K]
    B = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    bias = torch.randn(N, generator=g).cuda(); res = torch.randn(M, N, generator=g).cuda(); pre = torch.randn(M, N, generator=g).cuda()
    rps = 49
    scale = (torch.rand((M + rps - 1) // rps, generator=g) > 0.3).float().cuda() / 0.7

    def planes(x):
        x = x.contiguous()
        out = torch.empty((3,) + tuple(x.shape), dtype=torch.bfloat16, device="cuda")
        L.check(L.lib().gg_split3_bf16(x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), out.data_ptr(), L.stream()), "gg_split3_bf16")
        return out
    Ap, Bp = planes(A), planes(B)

    def run(**kw):
        a = L.Split3Args()
        a.b_planes, a.ldb, a.M, a.N, a.K = Bp.data_ptr(), K, M, N, K
        out = torch.empty(M, N, device="cuda"); a.C, a.ldc = out.data_ptr(), N
        keep = []
        for k, v in kw.items():
            if torch.is_tensor(v):
                keep.append(v); setattr(a, k, v.data_ptr())
            else:
                setattr(a, k, v)
        L.check(L.lib().gg_gemm_nt_split3_af32(C.byref(a), Afull.data_ptr(), lda, 0, L.stream()), "gg_gemm_nt_split3_af32")
        return out
    got = run(bias=bias)
    fed = torch.empty(M, N, device="cuda")
    L.check(L.lib().gg_gemm_nt_split3(Ap.data_ptr(), K, Bp.data_ptr(), K, fed.data_ptr(), N, M, N, K, bias.data_ptr(), L.stream()), "gg_gemm_nt_split3")
    assert torch.equal(got, fed)                 # same products in the same order (the 32 x 32 x 16 dev form, GG_SPLIT3A_MFMA=32, agrees to rounding: tools/bench_split3a.py)
    ref = A.double() @ B.double().T + bias.double()
    e = float((got.double() - ref).norm() / ref.norm())
    e32 = float((ops.gemm_nt(A.contiguous(), B, bias=bias).double() - ref).norm() / ref.norm())
    assert e < 1e-6 and e < 1.5 * e32 + 1e-7, (e, e32)
    rel = lambda x, y: float((x.double() - y.double()).norm() / y.double().norm())
    Ac = A.contiguous()
    pre_out = torch.empty(M, N, device="cuda")
    got = run(bias=bias, act=1, preact=pre_out)
    r, r_pre = ops.gemm_nt(Ac, B, bias=bias, act="gelu", preact=True)
    assert rel(got, r) < 1e-5 and rel(pre_out, r_pre) < 1e-5
    # the compile-time epilogue classes (split3_epilogue_rows_ec: taken when N % 8 == 0 and the pitches allow the vector path) against the generic row epilogue, which a
    # plane output forces: same arithmetic in the same order, bit for bit
    cp = torch.empty(3, M, N, dtype=torch.bfloat16, device="cuda")
    pre_gen = torch.empty(M, N, device="cuda")
    assert torch.equal(got, run(bias=bias, act=1, preact=pre_gen, c_planes=cp, ldp=N)) and torch.equal(pre_out, pre_gen)
    got = run(bias=bias, rowscale=scale, rows_per_scale=rps, residual=res, ldr=N)
    assert rel(got, ops.gemm_nt(Ac, B, bias=bias, rowscale=scale, rows_per_scale=rps, residual=res)) < 1e-5
    assert torch.equal(got, run(bias=bias, rowscale=scale, rows_per_scale=rps, residual=res, ldr=N, c_planes=cp, ldp=N))
    assert torch.equal(run(bias=bias, residual=res, ldr=N), run(bias=bias, residual=res, ldr=N, c_planes=cp, ldp=N))
    got = run(dact_preact=pre, dact=1, rowscale=scale, rows_per_scale=rps)
    assert rel(got, ops.gemm_nt(Ac, B, dact_preact=pre, dact="gelu", rowscale=scale, rows_per_scale=rps)) < 1e-5
    assert torch.equal(got, run(dact_preact=pre, dact=1, rowscale=scale, rows_per_scale=rps, c_planes=cp, ldp=N))
    assert torch.equal(run(dact_preact=pre, dact=1), run(dact_preact=pre, dact=1, c_planes=cp, ldp=N))


@pytest.mark.parametrize("M,N,K", [(1000, 200, 96), (700, 192, 416), (5000, 384, 96), (300, 48, 32)])
def test_split3_gemm_batchnorm_partials(M, N, K):
    """gg_gemm_nt_split3_af32_stats: the plain-epilogue split product of a ConvNorm's dense convolution also leaves the BatchNorm partials GgGemmArgs.colstats
    defines -- per 128-row block the column sums of the result and of its square, [ceil(M / 128)][2][N] -- in both tile heights and both tile widths."""
    import ctypes as C
    from geoguessr_ai_amd import _lib as L
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g).cuda(); B = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    Bp = torch.empty(3, N, K, dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().gg_split3_bf16(B.data_ptr(), N, K, K, Bp.data_ptr(), L.stream()), "gg_split3_bf16")
    parts = (M + 127) // 128
    out = torch.empty(M, N, device="cuda"); stats = torch.full((parts, 2, N), float("nan"), device="cuda")
    a = L.Split3Args()
    a.b_planes, a.ldb, a.M, a.N, a.K, a.C, a.ldc = Bp.data_ptr(), K, M, N, K, out.data_ptr(), N
    L.check(L.lib().gg_gemm_nt_split3_af32_stats(C.byref(a), A.data_ptr(), K, 0, stats.data_ptr(), L.stream()), "gg_gemm_nt_split3_af32_stats")
    ref = A.double() @ B.double().T
    assert float((out.double() - ref).norm() / ref.norm()) < 1e-6
    pad = torch.zeros(parts * 128, N, dtype=torch.float64, device="cuda"); pad[:M] = out.double()
    blk = pad.view(parts, 128, N)
    want = torch.stack([blk.sum(1), (blk * blk).sum(1)], 1)
    assert torch.isfinite(stats).all()
    assert float((stats.double() - want).abs().max() / want.abs().max()) < 1e-5
    out2 = torch.empty(M, N, device="cuda"); stats2 = torch.empty_like(stats)
    a.C = out2.data_ptr()
    L.check(L.lib().gg_gemm_nt_split3_af32_stats(C.byref(a), A.data_ptr(), K, 0, stats2.data_ptr(), L.stream()), "gg_gemm_nt_split3_af32_stats")
    assert torch.equal(stats, stats2) and torch.equal(out, out2)            # no atomics: repeatable


@pytest.mark.parametrize("M,N,K,act", [(1000, 96, 384, 1), (700, 200, 416, 1), (513, 96, 1024, 0), (300, 128, 384, 2)])
def test_split3_gemm_batchnorm_act_prologue(M, N, K, act):
    """gg_gemm_nt_split3_af32_pro (the fp32_split mode's MBConv.conv3): C = act(BatchNorm(A)) . W^T with the transform formed in the loader, in front of the split --
    against an fp64 product of the fp64-transformed operand (exact erf GELU / QuickGELU), against gg_gemm_nt_f32's own BatchNorm prologue (GgGemmArgs.a_bn_*), and
    the BatchNorm partials of the result; 96- and 128-column tiles, a K that is not a multiple of the stage, a ragged M."""
    import ctypes as C
    from geoguessr_ai_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(M + N + K + act)
    A = (torch.randn(M, K, generator=g) * 1.5 + 0.3).cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    mean, rstd = torch.randn(K, generator=g) * 0.3, torch.rand(K, generator=g) + 0.5
    gamma, beta = torch.randn(K, generator=g), torch.randn(K, generator=g) * 0.2
    stat = torch.stack([mean, rstd]).contiguous().cuda(); gd, bd = gamma.cuda(), beta.cuda()
    Wp = torch.empty(3, N, K, dtype=torch.bfloat16, device="cuda")
    L.check(lib.gg_split3_bf16(W.data_ptr(), N, K, K, Wp.data_ptr(), L.stream()), "gg_split3_bf16")
    parts = (M + 127) // 128
    out = torch.empty(M, N, device="cuda"); stats = torch.full((parts, 2, N), float("nan"), device="cuda")
    a = L.Split3Args()
    a.b_planes, a.ldb, a.M, a.N, a.K, a.C, a.ldc = Wp.data_ptr(), K, M, N, K, out.data_ptr(), N
    L.check(lib.gg_gemm_nt_split3_af32_pro(C.byref(a), A.data_ptr(), K, 0, stat.data_ptr(), gd.data_ptr(), bd.data_ptr(), act, stats.data_ptr(), L.stream()), "gg_gemm_nt_split3_af32_pro")
    z = (A.double().cpu() - mean.double()) * (rstd.double() * gamma.double()) + beta.double()
    t = z if act == 0 else 0.5 * z * (1 + torch.erf(z / 2 ** 0.5)) if act == 1 else z * torch.sigmoid(1.702 * z)
    ref = t @ W.double().cpu().T
    e = float((out.double().cpu() - ref).norm() / ref.norm())
    # the f32-MFMA kernel's prologue on the same operands
    ga = L.GemmArgs()
    o32 = torch.empty(M, N, device="cuda")
    ga.A, ga.lda, ga.B, ga.ldb, ga.C, ga.ldc, ga.M, ga.N, ga.K = A.data_ptr(), K, W.data_ptr(), K, o32.data_ptr(), N, M, N, K
    ga.a_bn_stat, ga.a_bn_gamma, ga.a_bn_beta, ga.a_bn_act = stat.data_ptr(), gd.data_ptr(), bd.data_ptr(), act
    L.check(lib.gg_gemm_nt_f32(C.byref(ga), L.stream()), "gg_gemm_nt_f32")
    e32 = float((o32.double().cpu() - ref).norm() / ref.norm())
    print(f"\n[split3 BN prologue {M}x{N}x{K} act {act}] rel-L2 vs fp64: split {e:.2e}, f32-MFMA {e32:.2e}")
    assert e < 2e-6 and e < 1.5 * e32 + 2e-7
    pad = torch.zeros(parts * 128, N, dtype=torch.float64); pad[:M] = out.double().cpu()
    blk = pad.view(parts, 128, N)
    want = torch.stack([blk.sum(1), (blk * blk).sum(1)], 1)
    assert torch.isfinite(stats).all() and float((stats.double().cpu() - want).abs().max() / want.abs().max()) < 1e-5
    # without statistics, and refused outside its domain
    out2 = torch.empty(M, N, device="cuda"); a.C = out2.data_ptr()
    L.check(lib.gg_gemm_nt_split3_af32_pro(C.byref(a), A.data_ptr(), K, 0, stat.data_ptr(), gd.data_ptr(), bd.data_ptr(), act, None, L.stream()), "gg_gemm_nt_split3_af32_pro")
    assert torch.equal(out, out2)
    a.K = 192
    assert lib.gg_gemm_nt_split3_af32_pro(C.byref(a), A.data_ptr(), K, 0, stat.data_ptr(), gd.data_ptr(), bd.data_ptr(), act, None, L.stream()) != 0


@pytest.mark.parametrize("M,N,K,rps", [(5003, 200, 136, 7), (40, 8, 12, 0), (9000, 576, 320, 49), (4096, 260, 132, 0), (1100, 96, 432, 64)])
def test_split3_weight_gradient_gemm(M, N, K, rps):
    """gg_gemm_tn_split3 (the fp32_split mode's weight gradient): dW[N][K] = sum_m s_m dY[m][n] X[m][k] with both f32 operands split in the kernel's loader, row slabs
    reduced by gg_splitk_reduce -- against an fp64 product (ragged row counts, N / K that are not multiples of the 256 x 128 tile, a DropPath row scale with zeros,
    strided operands), at the accuracy of gg_gemm_tn_f32 and repeatable bit for bit."""
    from geoguessr_ai_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(M + N + K)
    dYf = torch.randn(M, N + 4, generator=g).cuda(); Xf = torch.randn(M, K + 8, generator=g).cuda()
    dY, X = dYf[:, :N], Xf[:, :K]
    scale = ((torch.rand((M + rps - 1) // rps, generator=g) > 0.3).float() / 0.7).cuda() if rps else None
    scratch = torch.empty(8 << 20, device="cuda")

    def run(kind, out):
        fn_s, fn = (lib.gg_gemm_tn_split3_splits, lib.gg_gemm_tn_split3) if kind == "split" else (lib.gg_gemm_tn_f32_splits, lib.gg_gemm_tn_f32)
        s = fn_s(M, N, K)
        assert s >= 1 and s * N * K <= scratch.numel()
        L.check(fn(dYf.data_ptr(), dYf.stride(0), Xf.data_ptr(), Xf.stride(0), M, N, K, scale.data_ptr() if rps else None, rps, scratch.data_ptr(), s, L.stream()), kind)
        L.check(lib.gg_splitk_reduce(scratch.data_ptr(), out.data_ptr(), N * K, s, 0, 1.0, L.stream()), "gg_splitk_reduce")
    o3, o3b, o32 = (torch.empty(N, K, device="cuda") for _ in range(3))
    run("split", o3); run("f32", o32); run("split", o3b)
    w = scale.double().repeat_interleave(rps)[:M, None] if rps else 1.0
    ref = (dY.double() * w).T @ X.double()
    e3 = float((o3.double() - ref).norm() / ref.norm()); e32 = float((o32.double() - ref).norm() / ref.norm())
    print(f"\n[tn split {M}x{N}x{K}] rel-L2 vs fp64: split {e3:.2e}, f32-MFMA {e32:.2e}")
    assert e3 < 2e-6 and e3 < 1.5 * e32 + 2e-7
    assert torch.equal(o3, o3b)


def _range_rows(M, K, g):
    """Operand rows for the range tests of the split GEMMs: row blocks of (0) N(0,1); (1) one power of ten per ROW from 1e-20 ... 1e20; (2) one power of ten per
    ELEMENT from 1e-6 ... 1e6; (3) f32 denormals (|x| ~ 1e-40); (4) the top of the f32 range, 3.39e38 ... 3.4028e38 -- where bf16(x) alone rounds to inf;
    (5) rows holding one +inf, one -inf or one NaN.  Returns (A, block index per row)."""
    A = torch.randn(M, K, generator=g)
    blk = torch.arange(M) % 6
    A[blk == 1] *= 10.0 ** torch.randint(-20, 21, (int((blk == 1).sum()), 1), generator=g).float()
    A[blk == 2] *= 10.0 ** torch.randint(-6, 7, (int((blk == 2).sum()), K), generator=g).float()
    A[blk == 3] *= 1e-40
    top = torch.rand(int((blk == 4).sum()), K, generator=g) * (3.4028e38 - 3.39e38) + 3.39e38
    A[blk == 4] = top * torch.sign(torch.randn(top.shape, generator=g))
    A[blk == 4, 0] = 3.4028234e38                                # FLT_MAX itself
    rows5 = torch.nonzero(blk == 5).flatten()
    for i, r in enumerate(rows5.tolist()):
        A[r, (7 * i) % K] = (float("inf"), float("-inf"), float("nan"))[i % 3]
    return A, blk


@pytest.mark.parametrize("M,N,K", [(516, 192, 416), (300, 130, 96), (1030, 90, 1536)])
def test_split3_gemm_operand_range(ops, M, N, K):
    """gg_gemm_nt_split3_af32 outside N(0, 1): mixed magnitudes per row and per element, f32 denormals, the top of the f32 range, +-inf / NaN (both tile
    heights and widths).  Stated bounds, per element c = sum_k a_k b_k:
      * finite operands: the result is FINITE wherever gg_gemm_nt_f32's is (the first bf16 term is clamped to the largest finite bf16: without the clamp
        |a| >= 3.3961e38 became (inf, -inf, NaN)), and |c - c_fp64| <= 1e-5 sum_k |a_k b_k| + K 2^-126 (max_k |a_k| + max_k |b_k| + 1) -- the f32 accumulation bound
        plus flush-to-zero of sub-normal operand terms and products (what lies below 2^-126 may be dropped; the f32 kernel gets the same allowance);
      * an operand row holding +-inf or NaN gives a NON-FINITE result in every column, as the f32 GEMM does (the split product gives NaN where the f32
        GEMM may keep +-inf: inf - inf in the residual of the split)."""
    import ctypes as C
    from geoguessr_ai_amd import _lib as L
    g = torch.Generator().manual_seed(M + N + K)
    A, blk = _range_rows(M, K, g)
    B = torch.randn(N, K, generator=g) * (0.1 * K ** -0.5)         # |c| stays below FLT_MAX on the top-of-range rows
    Ad, Bd = A.cuda(), B.cuda()
    Bp = torch.empty(3, N, K, dtype=torch.bfloat16, device="cuda")
    L.check(L.lib().gg_split3_bf16(Bd.data_ptr(), N, K, K, Bp.data_ptr(), L.stream()), "gg_split3_bf16")
    a = L.Split3Args()
    got = torch.empty(M, N, device="cuda")
    a.b_planes, a.ldb, a.M, a.N, a.K, a.C, a.ldc = Bp.data_ptr(), K, M, N, K, got.data_ptr(), N
    L.check(L.lib().gg_gemm_nt_split3_af32(C.byref(a), Ad.data_ptr(), K, 0, L.stream()), "gg_gemm_nt_split3_af32")
    f32 = ops.gemm_nt(Ad, Bd).cpu()
    got = got.cpu()
    fin = blk != 5
    assert torch.isfinite(f32[fin]).all() and torch.isfinite(got[fin]).all(), "finite operands must give finite results (near-FLT_MAX clamp)"
    assert not torch.isfinite(got[~fin]).any() and not torch.isfinite(f32[~fin]).any(), "an inf / NaN operand row must give non-finite results"
    ref = A[fin].double() @ B.double().T
    bound = 1e-5 * (A[fin].double().abs() @ B.double().abs().T) + K * 2.0 ** -126 * (A[fin].double().abs().amax(1, keepdim=True) + B.double().abs().amax(1)[None, :] + 1.0)
    err, err32 = (got[fin].double() - ref).abs(), (f32[fin].double() - ref).abs()
    worst = float((err / bound).max())
    print(f"\n[split3 range {M}x{N}x{K}] worst |err| / bound: split {worst:.3f}, f32-MFMA {float((err32 / bound).max()):.3f}")
    assert worst <= 1.0 and float((err32 / bound).max()) <= 1.0
    for b in range(5):                                              # per regime, in the norm the other split tests use (denormal rows: covered by the bound above)
        if b == 3: continue
        sel = blk[fin] == b
        e = float((got[fin][sel].double() - ref[sel]).norm() / ref[sel].norm())
        assert e < 2e-6, (b, e)


def test_split3_weight_gradient_operand_range():
    """gg_gemm_tn_split3 outside N(0, 1): the contraction runs over rows, so the regimes are per COLUMN of dY / X -- powers of ten from 1e-15 ... 1e15, a denormal
    column, a column holding the top of the f32 range (finite result: its partner operand is small), and columns holding +inf / NaN.  Result class per element equal to
    gg_gemm_tn_f32's (finite where it is finite, non-finite where it is not), error of the finite part <= 1e-5 sum_m |dy x| + M 2^-126 (max_m |dy| + max_m |x| + 1) (sub-normal terms may be flushed)."""
    from geoguessr_ai_amd import _lib as L
    lib = L.lib()
    M, N, K = 2100, 264, 136
    g = torch.Generator().manual_seed(77)
    dY, X = torch.randn(M, N, generator=g), torch.randn(M, K, generator=g) * 0.01
    dY[:, 8:200] *= 10.0 ** torch.randint(-15, 16, (1, 192), generator=g).float()
    X[:, 4:100] *= 10.0 ** torch.randint(-15, 16, (1, 96), generator=g).float()
    dY[:, 200] *= 1e-40; X[:, 100] *= 1e-38                           # denormal columns
    dY[:, 201] = 0.0; dY[5, 201] = 3.4028234e38; dY[900, 201] = -3.395e38      # top of the range (x stays <= ~0.05: the products are finite)
    dY[17, 202] = float("inf"); dY[1800, 203] = float("nan"); X[33, 101] = float("-inf")
    bad_n, bad_k = torch.tensor([202, 203]), torch.tensor([101])
    dYd, Xd = dY.cuda(), X.cuda()
    scratch = torch.empty(8 << 20, device="cuda")

    def run(kind):
        out = torch.empty(N, K, device="cuda")
        fn_s, fn = (lib.gg_gemm_tn_split3_splits, lib.gg_gemm_tn_split3) if kind == "split" else (lib.gg_gemm_tn_f32_splits, lib.gg_gemm_tn_f32)
        s_ = fn_s(M, N, K)
        L.check(fn(dYd.data_ptr(), N, Xd.data_ptr(), K, M, N, K, None, 0, scratch.data_ptr(), s_, L.stream()), kind)
        L.check(lib.gg_splitk_reduce(scratch.data_ptr(), out.data_ptr(), N * K, s_, 0, 1.0, L.stream()), "gg_splitk_reduce")
        return out.cpu()
    got, f32 = run("split"), run("f32")
    nonfin = torch.zeros(N, K, dtype=torch.bool); nonfin[bad_n] = True; nonfin[:, bad_k] = True
    dYc, Xc = dY.double().clone(), X.double().clone()
    dYc[:, bad_n] = 0; Xc[:, bad_k] = 0
    ref, mag = dYc.T @ Xc, dYc.abs().T @ Xc.abs()
    fin = ~nonfin & (mag < 1e37)                                      # (the top-of-range column against an X column scaled up by 1e15 overflows f32 legitimately: excluded)
    assert int((mag[201] < 1e37).sum()) >= 30                         # ... but most of its row is in range
    assert torch.isfinite(got[fin]).all() and torch.isfinite(f32[fin]).all()
    assert not torch.isfinite(got[nonfin]).any() and not torch.isfinite(f32[nonfin]).any()
    bound = 1e-5 * mag + M * 2.0 ** -126 * (dYc.abs().amax(0)[:, None] + Xc.abs().amax(0)[None, :] + 1.0)
    w3 = float(((got.double() - ref).abs() / bound)[fin].max()); w32 = float(((f32.double() - ref).abs() / bound)[fin].max())
    print(f"\n[tn split range] worst |err| / bound: split {w3:.3f}, f32-MFMA {w32:.3f}")
    assert w3 <= 1.0 and w32 <= 1.0


# ------------------------------------------------------------------------------------------- head / loss / geo
def test_geo_head_matches_oracle_and_reference_golden(ops, golden_dir, centroids):
    import os
    from oracle import geo_ref as G
    g = np.load(os.path.join(golden_dir, "geo_loss.npz"))
    labels = g["labels"]
    logits = np.random.default_rng(int(g["logits_seed"])).standard_normal((16, 12647), dtype=np.float32) * np.float32(g["logits_scale"])
    cent = dev(torch.from_numpy(centroids))
    r = ops.geo_head(dev(torch.from_numpy(logits)), cent, labels=dev(torch.from_numpy(labels)), mode=1, want_dlogits=True,
                     want_nearest=True)
    np.testing.assert_allclose(float(r["loss"]), float(g["loss"]), rtol=2e-4)            # vs the REFERENCE (fixture)
    dl = r["dlogits"].float().cpu().numpy()[:, :12647]
    np.testing.assert_allclose(dl, g["dlogits"], rtol=2e-2, atol=2e-6)                   # bf16 storage of dlogits
    near = r["nearest"].cpu().numpy()
    np.testing.assert_allclose(g["distances"][np.arange(16), near], g["distances"][np.arange(16), g["argmin"]], atol=0.05)
    d = ops.haversine_matrix(dev(torch.from_numpy(labels)), cent).cpu().numpy()
    np.testing.assert_allclose(d, g["distances"], rtol=2e-4, atol=0.05)                  # 0.05 km (SURVEY 8c tolerance)
    # predictions vs the oracle
    lp = G.log_softmax(logits)
    idx = np.argsort(-lp, axis=-1, kind="stable")[:, :5]
    np.testing.assert_array_equal(r["topk_idx"].cpu().numpy(), idx)
    np.testing.assert_allclose(r["topk_vals"].cpu().numpy(), np.exp(np.take_along_axis(lp, idx, -1)), rtol=1e-4)
    np.testing.assert_array_equal(r["preds"].cpu().numpy(), idx[:, 0])
    np.testing.assert_allclose(r["llh"].cpu().numpy(), centroids[idx[:, 0]])
    # hard CE
    r2 = ops.geo_head(dev(torch.from_numpy(logits)), cent, labels_clf=dev(torch.from_numpy(g["argmin"])), mode=2, want_dlogits=True)
    np.testing.assert_allclose(float(r2["loss"]), float(g["hard_ce"]), rtol=2e-5)
    lh, dlh = G.hard_ce(logits, g["argmin"])
    np.testing.assert_allclose(r2["dlogits"].float().cpu().numpy()[:, :12647], dlh, rtol=2e-2, atol=2e-6)
    # fp32 dlogits (the reference-precision mode of the head): SURVEY 8(c)'s fp32 class against the REFERENCE fixture
    r3 = ops.geo_head(dev(torch.from_numpy(logits)), cent, labels=dev(torch.from_numpy(labels)), mode=1, want_dlogits=True, dlogits_f32=True)
    assert r3["dlogits"].dtype == torch.float32
    dl3 = r3["dlogits"].cpu().numpy()[:, :12647]
    rel = np.linalg.norm((dl3 - g["dlogits"]).astype(np.float64)) / np.linalg.norm(g["dlogits"].astype(np.float64))
    print(f"geo_head f32: loss rel {abs(float(r3['loss']) / float(g['loss']) - 1):.2e}, dlogits rel-L2 {rel:.2e}, "
          f"max abs {np.abs(dl3 - g['dlogits']).max():.2e}")
    np.testing.assert_allclose(float(r3["loss"]), float(g["loss"]), rtol=1e-5)
    # element-wise the soft targets carry the fp32 trig rounding of the distances (d/65 km in the exponent: a 0.02 km difference between two
    # correctly rounded libms is 3e-4 relative on that element; the numpy oracle itself is at 1e-3 against the fixture), the norm does not
    np.testing.assert_allclose(dl3, g["dlogits"], rtol=1e-3, atol=1e-8)
    assert rel < 2e-5, rel                      # measured 7e-6
    r4 = ops.geo_head(dev(torch.from_numpy(logits)), cent, labels_clf=dev(torch.from_numpy(g["argmin"])), mode=2, want_dlogits=True, dlogits_f32=True)
    np.testing.assert_allclose(float(r4["loss"]), float(g["hard_ce"]), rtol=1e-5)
    np.testing.assert_allclose(r4["dlogits"].cpu().numpy()[:, :12647], dlh, rtol=1e-5, atol=1e-8)


def test_geo_head_small_k_and_edge_rows(ops):
    """ragged K (not a multiple of the block), a single row, ties broken towards the lower index."""
    from oracle import geo_ref as G
    rng = np.random.default_rng(3)
    K = 1000
    cent = np.stack([rng.uniform(-180, 180, K), rng.uniform(-90, 90, K)], 1).astype(np.float32)
    logits = rng.standard_normal((1, K), dtype=np.float32)
    logits[0, 17] = logits[0, 400] = 9.0                      # tie for the arg-max
    labels = np.asarray([[cent[5, 0], cent[5, 1]]], np.float32)
    r = ops.geo_head(dev(torch.from_numpy(logits)), dev(torch.from_numpy(cent)), labels=dev(torch.from_numpy(labels)), mode=1,
                     want_dlogits=True, want_nearest=True)
    loss, dl, _, _ = G.soft_ce(logits, labels, cent)
    np.testing.assert_allclose(float(r["loss"]), loss, rtol=1e-4)
    assert int(r["nearest"][0]) == 5 and int(r["preds"][0]) == 17 and r["topk_idx"][0, :2].tolist() == [17, 400]
    np.testing.assert_allclose(r["dlogits"].float().cpu().numpy()[:, :K], dl, rtol=2e-2, atol=1e-5)


def test_adamw_matches_torch(ops):
    n = 1003
    p, g = rnd(n, seed=70), rnd(n, seed=71, scale=0.1)
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([ref], lr=2e-5)
    pd, md, vd = dev(torch.cat([p, torch.zeros(1)])), torch.zeros(n + 1, device="cuda"), torch.zeros(n + 1, device="cuda")
    for step in range(1, 4):
        ref.grad = g * step
        opt.step()
        ops.adamw_step(pd[:n], dev(g * step), md[:n], vd[:n], step, 2e-5)
    close(pd[:n], ref.data, rtol=1e-6, atol=1e-7, what="adamw")
    assert float(pd[n]) == 0.0


def test_proto_refine_matches_oracle():
    from geoguessr_ai_amd.models.proto_refiner import ProtoRefiner
    from oracle import proto_ref as P
    rng = np.random.default_rng(11)
    Kc, D, B = 40, 64, 33
    counts = rng.poisson(2.0, Kc)
    counts[[3, 7]] = 0
    gi = np.repeat(np.arange(Kc), counts)
    emb = rng.standard_normal((len(gi), D), dtype=np.float32)
    lng, lat = rng.uniform(-180, 180, len(gi)).astype(np.float32), rng.uniform(-90, 90, len(gi)).astype(np.float32)
    ref = ProtoRefiner.from_clusters(gi, emb, lng, lat, Kc, topk=5).cuda().eval()
    q = rng.standard_normal((B, 4, D), dtype=np.float32)
    cands = np.stack([rng.permutation(Kc)[:5] for _ in range(B)]).astype(np.int64)
    cands[0, 0] = 3                                                    # a cell without prototypes
    probs = np.sort(rng.dirichlet(np.ones(5), B).astype(np.float32), 1)[:, ::-1].copy()
    init = np.stack([rng.uniform(-180, 180, B), rng.uniform(-90, 90, B)], 1).astype(np.float32)
    loss, llh, cell = ref(torch.from_numpy(q), torch.from_numpy(init), torch.from_numpy(cands), torch.from_numpy(probs))
    ptr = ref.cell_ptr.cpu().numpy()
    o_llh, o_cell, o_idx = P.refine(q, init, cands, probs, ptr, ref.proto_emb.cpu().numpy(), ref.proto_lnglat.cpu().numpy())
    assert loss is None
    np.testing.assert_array_equal(cell.cpu().numpy(), o_cell)
    np.testing.assert_allclose(llh.cpu().numpy(), o_llh)
    np.testing.assert_array_equal(ref.last_guess_index.cpu().numpy(), o_idx)


def test_proto_refine_within_cluster_matches_oracle():
    """within_cluster=True (models/proto_refiner.py:239-269): a cluster with members answers with the member at argmax of the Euclidean distances,
    an empty cluster with its centroid; (n, 4, D) member embeddings are averaged over views first.  Against the oracle's restatement."""
    from geoguessr_ai_amd.models.proto_refiner import ProtoRefiner
    from oracle import proto_ref as P
    rng = np.random.default_rng(12)
    Kc, D, B = 30, 48, 41
    counts = rng.poisson(2.0, Kc) + 1
    gi = np.repeat(np.arange(Kc), counts)
    gi = gi[rng.permutation(len(gi))]                                  # clusters arrive unsorted: the member lists must follow the sort
    R = len(gi)
    emb = rng.standard_normal((R, D), dtype=np.float32)
    lng, lat = rng.uniform(-180, 180, R).astype(np.float32), rng.uniform(-90, 90, R).astype(np.float32)
    nmem = rng.integers(0, 5, R)
    nmem[:3] = 0                                                       # clusters without members -> centroid
    mptr = np.concatenate([[0], np.cumsum(nmem)]).astype(np.int64)
    memb = rng.standard_normal((int(mptr[-1]), 4, D), dtype=np.float32)
    mll = np.stack([rng.uniform(-180, 180, int(mptr[-1])), rng.uniform(-90, 90, int(mptr[-1]))], 1).astype(np.float32)
    ref = ProtoRefiner.from_clusters(gi, emb, lng, lat, Kc, topk=5, max_refinement=30000, member_ptr=mptr, member_emb=memb, member_lnglat=mll).cuda().eval()
    assert ref.within_cluster and ref.member_emb.shape == (int(mptr[-1]), D)
    q = rng.standard_normal((B, 4, D), dtype=np.float32)
    cands = np.stack([rng.permutation(Kc)[:5] for _ in range(B)]).astype(np.int64)
    probs = np.sort(rng.dirichlet(np.ones(5), B).astype(np.float32), 1)[:, ::-1].copy()
    init = np.stack([rng.uniform(-180, 180, B), rng.uniform(-90, 90, B)], 1).astype(np.float32)
    _, llh, cell = ref(torch.from_numpy(q), torch.from_numpy(init), torch.from_numpy(cands), torch.from_numpy(probs))
    args = (q, init, cands, probs, ref.cell_ptr.cpu().numpy(), ref.proto_emb.cpu().numpy(), ref.proto_lnglat.cpu().numpy())
    o_llh, o_cell, o_idx = P.refine(*args, max_refinement=30000, member_ptr=ref.member_ptr.cpu().numpy(), member_emb=ref.member_emb.cpu().numpy(),
                                    member_lnglat=ref.member_lnglat.cpu().numpy())
    np.testing.assert_array_equal(cell.cpu().numpy(), o_cell)
    np.testing.assert_allclose(llh.cpu().numpy(), o_llh)
    np.testing.assert_array_equal(ref.last_guess_index.cpu().numpy(), o_idx)
    c_llh, _, _ = P.refine(*args, max_refinement=30000)                # centroid answers differ: the member branch really ran
    assert (np.abs(c_llh - o_llh).max(axis=1) > 1e-3).mean() > 0.5
    # the sorted member table equals a direct gather in sorted-cluster order
    order = np.argsort(gi, kind="stable")
    want = np.concatenate([memb[mptr[j]:mptr[j + 1]].mean(1) for j in order if mptr[j + 1] > mptr[j]])
    np.testing.assert_allclose(ref.member_emb.cpu().numpy(), want, rtol=1e-6, atol=1e-7)


def test_proto_refine_matches_reference_forward_golden(golden_dir):
    """``gg_proto_refine`` against the REFERENCE's own ``ProtoRefiner.forward`` run (tests/golden/proto_refine.npz, made by make_golden_r4.py):
    centroid branch and member branch, missing-prototype sentinel (incl. the all-missing sample), 1000 km gate, 40-prototype cell, (B, 4, D)
    embeddings, ``candidate_probs=None``.  Geocells and winning slots identical, coordinates equal to float32 rounding of the stored table."""
    from geoguessr_ai_amd.models.proto_refiner import ProtoRefiner
    z = np.load(os.path.join(golden_dir, "proto_refine.npz"))
    for case in ("centroid", "member"):
        g = {k.split("__", 1)[1]: z[k] for k in z.files if k.startswith(case + "__")}
        counts = np.diff(g["cell_ptr"])
        gi = np.repeat(np.arange(len(counts)), counts)
        kw = {}
        if case == "member":
            kw = dict(member_ptr=g["member_ptr"], member_emb=g["member_emb4"], member_lnglat=g["member_lnglat"])
        ref = ProtoRefiner.from_clusters(gi, g["proto_emb"], g["proto_lnglat"][:, 0], g["proto_lnglat"][:, 1], len(counts), topk=5,
                                         max_refinement=1000, temperature=1.6, **kw).cuda().eval()
        t = lambda k: torch.from_numpy(g[k])
        loss, llh, cell = ref(t("embedding"), t("initial_preds"), t("candidate_cells"), t("candidate_probs"))
        assert loss is None
        np.testing.assert_array_equal(cell.cpu().numpy(), g["preds_geocell"])
        np.testing.assert_array_equal(ref.last_guess_index.cpu().numpy(), g["guess_index"])
        np.testing.assert_allclose(llh.cpu().numpy(), g["preds_LLH"], rtol=0, atol=1e-5)
        _, llh0, cell0 = ref(t("embedding").mean(dim=1), t("initial_preds"), t("candidate_cells"), None)
        np.testing.assert_array_equal(cell0.cpu().numpy(), g["preds_geocell_noprobs"])
        np.testing.assert_allclose(llh0.cpu().numpy(), g["preds_LLH_noprobs"], rtol=0, atol=1e-5)


def test_scoring_matches_reference_golden(ops, golden_dir):
    """gg_geoguessr_score against run_benchmark.py:25-65 executed by tests/golden/make_golden_r2.py: fp64 distances, INTEGER scores
    (clamp + round-half-even) bit-exact."""
    g = np.load(os.path.join(golden_dir, "score.npz"))
    d, s = ops.geoguessr_score(dev(torch.from_numpy(g["pred"])), dev(torch.from_numpy(g["true"])))
    assert d.dtype == torch.float64 and s.dtype == torch.int32
    np.testing.assert_allclose(d.cpu().numpy(), g["dist_km"], rtol=1e-8, atol=1e-9)    # asin(sqrt(a)) near exact antipodes amplifies the last ulp
    np.testing.assert_array_equal(s.cpu().numpy(), g["score"])
    from oracle import geo_ref as G
    np.testing.assert_array_equal(s.cpu().numpy(), G.geoguessr_score(d.cpu().numpy()))


def test_scoring_summary_and_metrics_callable(golden_dir):
    """scoring.compute_summary == run_benchmark.py:67-117 (golden from the reference function itself) on GPU-computed distances / scores of
    the same 200 samples; scoring.geocell_metrics honours the metrics-callable contract of evaluate_model."""
    from geoguessr_ai_amd import scoring
    g = np.load(os.path.join(golden_dir, "score.npz"))
    d, s = scoring.score_batch(dev(torch.from_numpy(g["pred"][:200])), dev(torch.from_numpy(g["true"][:200])))
    summ = scoring.compute_summary(d, s, g["sample_top1"])
    want = dict(zip([str(k) for k in g["summary_keys"]], g["summary_vals"]))
    assert set(summ) == set(want)
    assert summ["num_samples"] == 200 and summ["avg_score"] == want["avg_score"]           # integer scores: exact
    for k in ("avg_distance_km", "median_distance_km", "avg_top1_prob"):
        np.testing.assert_allclose(summ[k], want[k], rtol=1e-9)
    with pytest.raises(ValueError):
        scoring.compute_summary([], [])
    cells = np.arange(200) % 7
    top5 = np.stack([(cells + k) % 7 for k in (1, 0, 2, 3, 4)], 1)
    m = scoring.geocell_metrics((g["pred"][:200], cells, top5, g["true"][:200], np.where(np.arange(200) % 4 == 0, cells, (cells + 5) % 7)))
    assert m["Geocell_accuracy"] == 0.25 and m["Geocell_top5_accuracy"] == 0.25
    np.testing.assert_allclose(m["Mean_score"], want["avg_score"]); np.testing.assert_allclose(m["Median_distance_km"], want["median_distance_km"], rtol=1e-9)


def test_preprocess_bilinear_matches_reference_golden(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "preprocess.npz"))
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    for name in ("pano_down", "single_up", "same_size", "resize_only"):
        size = tuple(int(v) for v in g[name + ".size"])
        norm = bool(g[name + ".norm"][0])
        y = ops.preprocess_bilinear(dev(torch.from_numpy(g[name + ".x"])), None if size[0] < 0 else size,
                                    mean if norm else None, std if norm else None)
        ref = torch.from_numpy(g[name + ".y"])
        assert tuple(y.shape) == tuple(ref.shape)
        close(y, ref, rtol=0, atol=1e-5, what="preprocess " + name)          # fp32, FMA contraction only
    u8 = torch.randint(0, 256, (2, 3, 9, 7), dtype=torch.uint8)
    from oracle import preprocess_ref as P
    close(ops.preprocess_bilinear(u8.cuda(), (12, 12), mean, std), torch.from_numpy(P.prepare_batch(u8.numpy(), (12, 12), mean, std)),
          rtol=0, atol=1e-5, what="preprocess u8")


def test_build_prototypes_matches_reference_golden(golden_dir):
    from geoguessr_ai_amd.embedding_store import build_prototypes
    g = np.load(os.path.join(golden_dir, "preprocess.npz"))
    ptr, member = g["proto.ptr"], g["proto.member"]
    cluster = np.full(g["proto.emb"].shape[0], -1, np.int64)
    for k in range(len(ptr) - 1):
        cluster[member[ptr[k]:ptr[k + 1]]] = k
    # members are accumulated in increasing panorama order; reorder the golden's member lists the same way
    from oracle import preprocess_ref as P
    order = np.concatenate([np.sort(member[ptr[k]:ptr[k + 1]]) for k in range(len(ptr) - 1)]).astype(np.int64)
    want = P.cluster_mean(g["proto.pano_vec"], ptr, order)
    protos, counts = build_prototypes(torch.from_numpy(g["proto.emb"]).cuda(), cluster, num_clusters=len(ptr) - 1)
    np.testing.assert_array_equal(counts.numpy(), np.diff(ptr))
    np.testing.assert_allclose(protos.cpu().numpy(), want, rtol=0, atol=2e-7)      # view mean is torch's, the segment sum is ordered
    import geoguessr_ai_amd.ops as O
    got = O.segment_mean(torch.from_numpy(g["proto.pano_vec"]).cuda(), torch.from_numpy(ptr).cuda(), torch.from_numpy(member).cuda())
    np.testing.assert_array_equal(got.cpu().numpy(), g["proto.out"])              # same fp32 additions in the same order: bit-exact


@pytest.mark.parametrize("C,H", [(192, 12), (40, 7)])
def test_dwconv_stride2_data_gradient_with_batchnorm_fusions(ops, C, H):
    """gg_dwconv3x3_s2_bwd_data_fused == BN-backward apply -> stride-2 data gradient -> act'(BN) * . + column sums, composed from
    fp32 torch math on the same bf16 inputs."""
    B, Ho = 2, (H - 1) // 2 + 1
    dz = rnd(B, Ho, Ho, C, seed=90).to(BF).float(); y2 = rnd(B, Ho, Ho, C, seed=91).to(BF).float()
    coef = torch.stack([1 + 0.2 * rnd(C, seed=92), 0.3 * rnd(C, seed=93), 0.1 * rnd(C, seed=94)])
    w = rnd(C, 1, 3, 3, seed=95, scale=0.4); taps = w.view(C, 9).t().contiguous()
    y1 = (rnd(B, H, H, C, seed=96) + 0.3).to(BF).float()
    gamma, beta = 1 + 0.2 * rnd(C, seed=97), 0.2 * rnd(C, seed=98)
    mean, var = y1.mean((0, 1, 2)), y1.var((0, 1, 2), unbiased=False)
    rstd = torch.rsqrt(var + 1e-5)
    dy = (coef[0] * dz + coef[1] * y2 + coef[2]).to(BF).float()
    xr = torch.zeros(B, C, H, H, requires_grad=True)
    F.conv2d(xr, w, None, 2, 1, 1, C).backward(dy.permute(0, 3, 1, 2))
    da = xr.grad.permute(0, 2, 3, 1)
    z = ((y1 - mean) * rstd * gamma + beta).requires_grad_(True)
    F.gelu(z).sum().backward()
    ref = da * z.grad
    out, part = ops.dwconv3x3_s2_bwd_data_fused(dev(dz, BF), dev(y2, BF), dev(coef), dev(taps), H, H, ep_y=dev(y1, BF),
                                                ep_stat=dev(torch.stack([mean, rstd])), ep_gamma=dev(gamma), ep_beta=dev(beta), ep_act="gelu")
    close(out, ref, rtol=2e-2, atol=2e-2, what="s2 fused dgrad")
    oq = out.float().cpu()
    s = part.cpu().sum(0)
    close(s[0], oq.sum((0, 1, 2)), rtol=1e-3, atol=2e-2, what="s2 fused sum dz")
    close(s[1], (oq * ((y1 - mean) * rstd)).sum((0, 1, 2)), rtol=1e-3, atol=5e-2, what="s2 fused sum dz*xhat")
    plain, _ = ops.dwconv3x3_s2_bwd_data_fused(dev(dy, BF), None, None, dev(taps), H, H)
    close(plain, da, rtol=2e-2, atol=2e-2, what="s2 plain dgrad")


@pytest.mark.parametrize("C,H,B", [(192, 12, 2), (40, 7, 3), (384, 28, 5)])
def test_dwconv_stride2_data_gradient_with_batchnorm_fusions_f32(ops, C, H, B):
    """gg_dwconv3x3_s2_bwd_data_fused_f32 (PatchMerging backward of the fp32 mode): the same composition in fp32 at fp32 tolerances; every
    combination of the two fusions; an odd map (7 -> 4) and enough pixels for several blocks."""
    Ho = (H - 1) // 2 + 1
    dz = rnd(B, Ho, Ho, C, seed=190); y2 = rnd(B, Ho, Ho, C, seed=191)
    coef = torch.stack([1 + 0.2 * rnd(C, seed=192), 0.3 * rnd(C, seed=193), 0.1 * rnd(C, seed=194)])
    w = rnd(C, 1, 3, 3, seed=195, scale=0.4); taps = w.view(C, 9).t().contiguous()
    y1 = rnd(B, H, H, C, seed=196) + 0.3
    gamma, beta = 1 + 0.2 * rnd(C, seed=197), 0.2 * rnd(C, seed=198)
    mean, var = y1.mean((0, 1, 2)), y1.var((0, 1, 2), unbiased=False)
    rstd = torch.rsqrt(var + 1e-5)

    def conv_t(dy):
        xr = torch.zeros(B, C, H, H, requires_grad=True)
        F.conv2d(xr, w, None, 2, 1, 1, C).backward(dy.permute(0, 3, 1, 2))
        return xr.grad.permute(0, 2, 3, 1)
    dy = coef[0] * dz + coef[1] * y2 + coef[2]
    z = ((y1 - mean) * rstd * gamma + beta).requires_grad_(True)
    F.gelu(z).sum().backward()
    stat = dev(torch.stack([mean, rstd]))
    out, part = ops.dwconv3x3_s2_bwd_data_fused(dev(dz), dev(y2), dev(coef), dev(taps), H, H, ep_y=dev(y1), ep_stat=stat, ep_gamma=dev(gamma),
                                                ep_beta=dev(beta), ep_act="gelu")
    ref = conv_t(dy) * z.grad
    close(out, ref, rtol=2e-4, atol=2e-5, what="f32 s2 fused dgrad (both fusions)")
    s = part.double().cpu().sum(0)
    close(s[0], ref.double().sum((0, 1, 2)), rtol=2e-4, atol=2e-3, what="f32 s2 fused sum dz")
    close(s[1], (ref.double() * ((y1 - mean) * rstd).double()).sum((0, 1, 2)), rtol=2e-4, atol=2e-3, what="f32 s2 fused sum dz*xhat")
    only_in, _ = ops.dwconv3x3_s2_bwd_data_fused(dev(dz), dev(y2), dev(coef), dev(taps), H, H)
    close(only_in, conv_t(dy), rtol=2e-4, atol=2e-5, what="f32 s2 dgrad, input fusion only")
    only_ep, part2 = ops.dwconv3x3_s2_bwd_data_fused(dev(dy), None, None, dev(taps), H, H, ep_y=dev(y1), ep_stat=stat, ep_gamma=dev(gamma),
                                                     ep_beta=dev(beta), ep_act="gelu")
    close(only_ep, ref, rtol=2e-4, atol=2e-5, what="f32 s2 dgrad, output fusion only")
    close(part2.double().cpu().sum(0)[0], s[0], rtol=1e-5, atol=1e-4, what="partials agree")
    plain, _ = ops.dwconv3x3_s2_bwd_data_fused(dev(dy), None, None, dev(taps), H, H)
    close(plain, conv_t(dy), rtol=2e-4, atol=2e-5, what="f32 s2 plain dgrad")
    close(plain, ops.dwconv3x3_bwd_data(dev(dy), dev(taps), B, H, H, C, stride=2), rtol=1e-5, atol=1e-6, what="agrees with the unfused gather (tap order differs)")


def test_preprocess_pil_matches_pillow_and_transformers_golden():
    """gg_preprocess_pil through training.preprocess.images_to_pixel_values against tests/golden/preprocess_pil.npz (Pillow's Image.resize and transformers'
    CLIPImageProcessorPil, run by tests/golden/make_golden_r5.py): the uint8 crop of every pipeline -- CLIPProcessor (pretrain/clip_embedder.py:51-55), timm's eval
    transform at 224 / 384 / 512-squash (pretrain/tinyvit_embedder.py:51-69), inference.py:84's Resize + CenterCrop -- is BIT-IDENTICAL; pixel_values agree to
    1e-6 (one fp32 ulp: transformers rescales in a wider type).  Images: landscape, portrait, mode L, mode RGBA, an up-scale, a no-op resize, a 2.2:1 strip."""
    import hashlib
    from oracle import preprocess_ref as P
    from geoguessr_ai_amd import _lib as L, ops
    from geoguessr_ai_amd.training.preprocess import images_to_pixel_values, TINYVIT_MEAN, TINYVIT_STD, CLIP_MEAN, CLIP_STD
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "preprocess_pil.npz"))
    checked = 0
    for n in sorted({k.split(".")[0] for k in g.files}):
        img = P.to_rgb(g[n + ".img"], "".join(chr(c) for c in g[n + ".mode"]))
        for key, pipe, size, ms, kw in (("clip", "clip", 224, (CLIP_MEAN, CLIP_STD), {}), ("timm224", "timm", 224, (TINYVIT_MEAN, TINYVIT_STD), dict(crop_pct=0.95)),
                                        ("timm512", "timm", 512, (TINYVIT_MEAN, TINYVIT_STD), dict(crop_pct=1.0, crop_mode="squash")),
                                        ("inf336", "torchvision", 336, (CLIP_MEAN, CLIP_STD), {}), ("timm384", "timm", 384, (TINYVIT_MEAN, TINYVIT_STD), dict(crop_pct=1.0))):
            if not any(f"{n}.{key}_{suf}" in g.files for suf in ("u8", "sha")):
                continue
            pv, u8 = images_to_pixel_values(img, size, ms[0], ms[1], "cuda", pipeline=pipe, return_u8=True, **kw)
            pv, u8 = pv[0].cpu().numpy(), u8[0].cpu().numpy()
            if f"{n}.{key}_u8" in g.files:
                assert np.array_equal(u8, g[f"{n}.{key}_u8"]), (n, key, int((u8 != g[f"{n}.{key}_u8"]).sum()))
            else:
                assert hashlib.sha256(np.ascontiguousarray(u8).tobytes()).digest() == g[f"{n}.{key}_sha"].tobytes(), (n, key)
            if f"{n}.{key}_pv" in g.files:
                assert np.abs(pv - g[f"{n}.{key}_pv"]).max() <= 1e-6, (n, key)
            elif f"{n}.{key}_pv_sum" in g.files:
                cs = np.asarray([pv.astype(np.float64).sum(), np.abs(pv.astype(np.float64)).sum()])
                assert np.abs(cs - g[f"{n}.{key}_pv_sum"]).max() <= 1e-6 * g[f"{n}.{key}_pv_sum"][1], (n, key)
            checked += 1
    assert checked == 19
    # the NCHW uint8 tensor form and a list give the same rows; an image that the upstream transforms would have to pad is refused
    land = P.to_rgb(g["land.img"], "RGB")
    a = images_to_pixel_values(land, 224, CLIP_MEAN, CLIP_STD, "cuda", pipeline="clip")
    b = images_to_pixel_values(torch.from_numpy(land).permute(2, 0, 1), 224, CLIP_MEAN, CLIP_STD, "cuda", pipeline="clip")
    c = images_to_pixel_values([land, land[:, ::-1].copy()], 224, CLIP_MEAN, CLIP_STD, "cuda", pipeline="clip")
    assert torch.equal(a, b) and c.shape == (2, 3, 224, 224) and torch.equal(c[:1], a)
    with pytest.raises(L.GgError, match="inside the resized image|smaller than"):
        ops.preprocess_pil(torch.zeros((50, 60, 3), dtype=torch.uint8, device="cuda"), 3, (50, 60), (0, 0), (224, 224))
