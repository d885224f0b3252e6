"""dev: do co-resident fp32 ring-GEMM workgroups overlap their store phase with each other's k-loops?  Per CU, from the per-workgroup
trace (gg_gemm_f32_set_trace): time share with n workgroups inside the k-loop / inside the epilogue, and the k-loop duration of a
workgroup against the share of its k-loop during which co-resident workgroups were storing.

usage: python tools/trace_gemm_phases.py [M N K [gelu]]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoguessr_ai_amd import ops, _lib as L

args = sys.argv[1:]
M, N, K = (int(args[0]), int(args[1]), int(args[2])) if len(args) >= 3 else (802816, 768, 192)
gelu = len(args) < 4 or args[3] == "gelu"
A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda") * 0.05; out = torch.empty(M, N, device="cuda")
bias = torch.randn(N, device="cuda")
bn = 64 if (N <= 64 or (N % 128 != 0 and N % 128 <= 64)) else 128
tiles = ((M + 127) // 128) * ((N + bn - 1) // bn)
buf = torch.zeros(tiles, 8, dtype=torch.int64, device="cuda")
kw = dict(bias=bias, act="gelu", preact=True) if gelu else {}
for _ in range(2): ops.gemm_nt(A, B, out=out, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.gemm_nt(A, B, out=out, **kw); e1.record(); torch.cuda.synchronize()
print(f"M={M} N={N} K={K} {'bias+gelu+preact' if gelu else 'plain'}: {e0.elapsed_time(e1) * 1e3:.0f} us untraced, {tiles} tiles")
L.lib().gg_gemm_f32_set_trace(buf.data_ptr())
ops.gemm_nt(A, B, out=out, **kw)
torch.cuda.synchronize()
L.lib().gg_gemm_f32_set_trace(None)
t = buf.cpu().numpy()
hw, xcc = t[:, 0] & 0xFFFFFFFF, (t[:, 0] >> 32) & 0xF
cu = ((xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF))
tick = 0.01   # us per 100 MHz tick
span = (t[:, 5].max() - t[:, 2].min()) * tick
loop, epi = (t[:, 4] - t[:, 3]) * tick, (t[:, 5] - t[:, 4]) * tick
print(f"kernel span {span:.0f} us; per workgroup: first data {np.median((t[:, 3] - t[:, 2]) * tick):.1f} us, k-loop {np.median(loop):.1f} (p10 {np.percentile(loop, 10):.1f}, p90 {np.percentile(loop, 90):.1f}), "
      f"epilogue+drain {np.median(epi):.1f} (p10 {np.percentile(epi, 10):.1f}, p90 {np.percentile(epi, 90):.1f})")
mhz = np.median(t[:, 1] / ((t[:, 5] - t[:, 2]) * 0.01))          # shader-clock cycles per microsecond
w_dma, w_frag = (t[:, 6] & 0xFFFFFFFF) / mhz, (t[:, 6] >> 32) / mhz
w_mfma, issue = (t[:, 7] & 0xFFFFFFFF) / mhz, (t[:, 7] >> 32) * tick
print(f"wave 0 inside the k-loop (us, medians, clock {mhz:.0f} MHz): DMA wait + barrier {np.median(w_dma):.1f}, issue next stage + LDS fragment reads {np.median(w_frag):.1f}, "
      f"MFMA issue {np.median(w_mfma):.1f};  epilogue: instructions issued after {np.median(issue):.1f} us, store drain {np.median(epi - issue):.1f} us")
share_k, share_e = np.zeros(8), np.zeros(8)
joint = np.zeros((8, 8))
rows = []
for c in np.unique(cu)[:96]:
    idx = np.where(cu == c)[0]
    ev = []
    for i in idx:
        ev += [(t[i, 3], 0, 1), (t[i, 4], 0, -1), (t[i, 4], 1, 1), (t[i, 5], 1, -1)]
    ev.sort()
    nk = ne = 0
    last = ev[0][0]
    for x, kind, d in ev:
        dt = x - last
        if dt > 0:
            share_k[min(nk, 7)] += dt; share_e[min(ne, 7)] += dt; joint[min(nk, 7), min(ne, 7)] += dt
        last = x
        if kind == 0: nk += d
        else: ne += d
    # per workgroup: share of its k-loop during which >= 1 co-resident workgroup was in its epilogue
    es, ee = t[idx, 4], t[idx, 5]
    for i in idx:
        a, b = t[i, 3], t[i, 4]
        ov = np.clip(np.minimum(ee, b) - np.maximum(es, a), 0, None)
        ov[idx == i] = 0
        rows.append(((b - a) * tick, ov.sum() * tick))
tot = share_k.sum()
print("time share with n workgroups in the k-loop :", " ".join(f"{n}:{share_k[n] / tot:.2f}" for n in range(6)))
print("time share with n workgroups in the epilogue:", " ".join(f"{n}:{share_e[n] / tot:.2f}" for n in range(6)))
print("joint (rows: in k-loop, cols: in epilogue):")
for a in range(5): print("   ", " ".join(f"{joint[a, b] / tot:.3f}" for b in range(5)))
rows = np.array(rows)
q = np.percentile(rows[:, 1], [25, 50, 75])
for lo, hi, lab in [(-1, q[0], "least"), (q[0], q[1], "q2"), (q[1], q[2], "q3"), (q[2], 1e9, "most")]:
    sel = (rows[:, 1] > lo) & (rows[:, 1] <= hi)
    if sel.any(): print(f"  k-loop of workgroups with {lab} co-resident epilogue time ({rows[sel, 1].mean():.1f} us): {rows[sel, 0].mean():.1f} us")
