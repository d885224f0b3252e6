#!/usr/bin/env python3
"""dev: the split-bf16 fp32 attention kernels against the f32-MFMA flash kernels (GG_ATTN_NO_SPLIT=1 in a child process is not possible in-process: the switch is
read once) -- so against an fp64 torch reference on a few windows -- and their timing at the TinyViT-21M-224 / 1024-image shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from geoguessr_ai_amd import _lib as L
B = int(os.environ.get("B", "1024"))
DT = torch.bfloat16 if os.environ.get("DT") == "bf16" else torch.float32
CODE = 0 if DT == torch.bfloat16 else 1
def FWD(a):
    return L.lib().gg_attention_flash_fwd(C.byref(a), 1, L.stream()) if CODE else L.lib().gg_attention_fwd(C.byref(a), L.stream())
def BWD(a):
    return L.lib().gg_attention_flash_bwd(C.byref(a), 1, L.stream()) if CODE else L.lib().gg_attention_bwd(C.byref(a), L.stream())
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def bias_full(table, ws):
    ii = torch.arange(ws * ws, device=table.device)
    cy, cx = ii // ws, ii % ws
    idx = (cy[:, None] - cy[None, :]).abs() * ws + (cx[:, None] - cx[None, :]).abs()
    return table[:, idx]                                  # (nh, N, N)
for name, res, ws, Cc, nh in [("s1", 28, 7, 192, 6), ("s2", 14, 14, 384, 12), ("s3", 7, 7, 576, 18)]:
    M, N = B * res * res, ws * ws
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = torch.randn(M, 3 * Cc, device="cuda", generator=g).to(DT); out = torch.empty(M, Cc, device="cuda", dtype=DT); dout = torch.randn(M, Cc, device="cuda", generator=g).to(DT)
    dqkv = torch.zeros_like(qkv); lse = torch.empty(M, nh, device="cuda"); table = torch.randn(nh, N, device="cuda", generator=g) * 0.3
    dbias = torch.zeros_like(table)
    a = L.AttnArgs()
    a.qkv, a.ld, a.q_off, a.k_off, a.v_off, a.head_stride, a.head_dim = qkv.data_ptr(), 3 * Cc, 0, 32, 64, 96, 32
    a.num_heads, a.num_windows, a.tokens_per_window = nh, B * (res // ws) ** 2, N
    a.window_size, a.map_h, a.map_w = ws, res, res
    a.bias_table = table.data_ptr(); a.scale = 32 ** -0.5
    if not CODE:
        Np = L.lib().gg_attention_padded_tokens(N)
        full = torch.empty((nh, Np, Np), dtype=torch.bfloat16, device="cuda")
        L.check(L.lib().gg_attention_expand_bias(table.data_ptr(), nh, ws, C.c_float(a.scale), full.data_ptr(), L.stream()))
        a.bias = full.data_ptr()
    a.out, a.ldo, a.lse = out.data_ptr(), Cc, lse.data_ptr()
    L.check(FWD(a))
    a.dout, a.lddo, a.dqkv = dout.data_ptr(), Cc, dqkv.data_ptr()
    L.check(BWD(a))
    torch.cuda.synchronize()
    # fp64 reference on the first and last image
    nW = res // ws
    errs = {}
    for b in (0, B - 1):
        x = qkv[b * res * res:(b + 1) * res * res].double().view(nW, ws, nW, ws, nh, 96).permute(0, 2, 4, 1, 3, 5).reshape(nW * nW, nh, N, 96)
        q, k, v = x[..., :32].clone().requires_grad_(), x[..., 32:64].clone().requires_grad_(), x[..., 64:].clone().requires_grad_()
        s = (q @ k.transpose(-1, -2)) * a.scale + bias_full(table.double(), ws)[None]
        o = torch.softmax(s, -1) @ v                                           # (windows, nh, N, 32)
        go = dout[b * res * res:(b + 1) * res * res].double().view(nW, ws, nW, ws, nh, 32).permute(0, 2, 4, 1, 3, 5).reshape(nW * nW, nh, N, 32)
        o.backward(go)
        def back(t, c):
            return t.view(nW, nW, nh, ws, ws, c).permute(0, 3, 1, 4, 2, 5).reshape(res * res, nh * c)
        o_ref = back(o.detach(), 32)
        got = out[b * res * res:(b + 1) * res * res].double()
        errs.setdefault("out", []).append(float((got - o_ref).norm() / o_ref.norm()))
        dref = torch.cat([q.grad, k.grad, v.grad], -1)
        dref = back(dref, 96)
        dgot = dqkv[b * res * res:(b + 1) * res * res].double()
        errs.setdefault("dqkv", []).append(float((dgot - dref).norm() / dref.norm()))
        if os.environ.get("DBG") and b == 0:
            gq, rq = dgot.view(-1, nh, 96)[..., :32], dref.view(-1, nh, 96)[..., :32]
            print(name, "dq err by column block", [float((gq[..., c:c+8] - rq[..., c:c+8]).norm() / rq[..., c:c+8].norm()) for c in range(0, 32, 8)])
            nrow = gq.shape[0]
            print(name, "dq err by 16-row tile", [round(float((gq[t:t+16] - rq[t:t+16]).norm() / rq[t:t+16].norm()), 3) for t in range(0, min(nrow, 224), 16)])
            print(name, "ratio got/ref sample", (gq[:4, 0, :4] / rq[:4, 0, :4]).tolist())
        for nm, lo in (("dq", 0), ("dk", 32), ("dv", 64)):
            gg_, rr_ = dgot.view(-1, nh, 96)[..., lo:lo + 32], dref.view(-1, nh, 96)[..., lo:lo + 32]
            errs.setdefault(nm, []).append(float((gg_ - rr_).norm() / rr_.norm()))
    tf = timed(lambda: L.check(FWD(a)))
    tb = timed(lambda: L.check(BWD(a)))
    fl = 4.0 * M * N * Cc
    print(f"{name} ws={ws}  fwd {tf*1e3:8.1f} us ({fl/tf/1e9:6.1f} TF/s)   bwd {tb*1e3:8.1f} us ({2.5*fl/tb/1e9:6.1f} TF/s)   rel-L2 vs fp64: out {max(errs['out']):.2e}  dqkv {max(errs['dqkv']):.2e} (dq {max(errs['dq']):.1e} dk {max(errs['dk']):.1e} dv {max(errs['dv']):.1e})", flush=True)
