"""dev: per-workgroup timeline of the fp32 ring GEMM (gg_gemm_f32_set_trace): where a workgroup's life goes (first data, k-loop, epilogue +
store drain) and how long a CU slot idles between two workgroups."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geoguessr_ai_amd import ops, _lib as L
for name, M, N, K in [("s0.conv1", 3211264 // 4, 384, 96), ("s2.qkv", 200704, 1152, 384), ("s2.fc2", 200704, 384, 1536)]:
    A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda") * 0.05; out = torch.empty(M, N, device="cuda")
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    buf = torch.zeros(tiles, 8, dtype=torch.int64, device="cuda")
    for _ in range(2): ops.gemm_nt(A, B, out=out)
    torch.cuda.synchronize()
    L.lib().gg_gemm_f32_set_trace(buf.data_ptr())
    ops.gemm_nt(A, B, out=out)
    torch.cuda.synchronize()
    L.lib().gg_gemm_f32_set_trace(None)
    t = buf.cpu().numpy()
    hw, xcc = t[:, 0] & 0xFFFFFFFF, (t[:, 0] >> 32) & 0xF
    mhz = np.median(t[:, 1] / ((t[:, 5] - t[:, 2]) * 0.01))          # s_memtime ticks per microsecond
    cu = ((xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF))
    tick = 10.0  # ns per 100 MHz tick
    first, loop, epi = (t[:, 3] - t[:, 2]) * tick / 1e3, (t[:, 4] - t[:, 3]) * tick / 1e3, (t[:, 5] - t[:, 4]) * tick / 1e3
    t0 = t[:, 2].min()
    span = (t[:, 5].max() - t0) * tick / 1e3
    print(f"{name} M={M} N={N} K={K}: {tiles} tiles, {len(np.unique(cu))} CUs seen, kernel span {span:.1f} us")
    print(f"  per workgroup (us): first data {np.median(first):.2f} (p90 {np.percentile(first, 90):.2f}), k-loop {np.median(loop):.2f} (p90 {np.percentile(loop, 90):.2f}), "
          f"epilogue+drain {np.median(epi):.2f} (p90 {np.percentile(epi, 90):.2f}), life {np.median(first + loop + epi):.2f}")
    print(f"  wave 0 inside the k-loop: waiting for its DMAs {np.median(t[:, 6]) / mhz:.2f} us, at the stage barrier {np.median(t[:, 7]) / mhz:.2f} us (s_memtime at {mhz:.0f} MHz)")
    # per CU: concurrency and slot gaps
    gaps, conc = [], []
    for c in np.unique(cu)[:64]:
        idx = np.where(cu == c)[0]
        st, en = np.sort(t[idx, 2]), np.sort(t[idx, 5])
        # k-th start happens after (k - slots)-th end: estimate slots as max concurrency
        ev = sorted([(x, 1) for x in t[idx, 2]] + [(x, -1) for x in t[idx, 5]])
        cur = mx = 0
        for _, d in ev: cur += d; mx = max(mx, cur)
        conc.append(mx)
        if len(idx) > mx:
            gaps += list((st[mx:] - en[:len(st) - mx]) * tick / 1e3)
    print(f"  per CU: max concurrent workgroups {np.median(conc):.0f}, slot hand-over gap (next start - matching end) median {np.median(gaps):.2f} us, p90 {np.percentile(gaps, 90):.2f}")
    del A, B, out, buf
