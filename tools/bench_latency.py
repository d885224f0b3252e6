#!/usr/bin/env python3
"""Launch-bound cases: the serving call on one panorama (reference inference.py:162-170: 4 headings through TinyViT-21M + the geocell head) and the
c1 training step (TinyViT-5M, batch 8).  Wall time per call with the GPU drained after every call (latency) and back to back (throughput), next to the
sum of the kernels' own durations (GG_PROF events) -- the gap is launch / host overhead.  GG_GRAPH=0 / 1 selects eager launches or the runtime's
captured HIP graphs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
from geoguessr_ai_amd.models.super_guessr import SuperGuessr
from geoguessr_ai_amd.optim import AdamW


def wall(fn, n, sync_each):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
        if sync_each: torch.cuda.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = "cuda"
    print("GG_GRAPH =", os.environ.get("GG_GRAPH", "(default)"))
    torch.manual_seed(0)
    for name, B in [("tiny_vit_21m_224", 4), ("tiny_vit_21m_224", 16), ("tiny_vit_5m_224", 4)]:
        base = TinyViTAdapter(name, pretrained=False, precision="fp32")
        model = SuperGuessr(base, panorama=True, serving=True).to(dev).eval()
        x = torch.randn(B // 4, 4, 3, 224, 224, device=dev)
        dummy = torch.zeros(B // 4, dtype=torch.long, device=dev)
        with torch.no_grad():
            f = lambda: model(pixel_values=x, labels_clf=dummy)
            lat, thr = wall(f, 50, True), wall(f, 50, False)
        print(f"serving {name} {B // 4} panorama(s) = {B} images: latency {lat:.3f} ms/call, back-to-back {thr:.3f} ms/call", flush=True)
        del model, base
    base = TinyViTAdapter("tiny_vit_5m_224", pretrained=False, precision="fp32")
    model = SuperGuessr(base, panorama=False, should_smooth_labels=False, serving=False).to(dev).train()
    opt = AdamW(model, lr=5e-5, betas=(0.9, 0.999), weight_decay=0.01)
    g = torch.Generator(device=dev).manual_seed(330)
    x = torch.randn(8, 3, 224, 224, device=dev, generator=g)
    lab = torch.stack([torch.rand(8, device=dev, generator=g) * 360 - 180, torch.rand(8, device=dev, generator=g) * 180 - 90], 1)
    clf = torch.randint(0, model.num_cells, (8,), device=dev, generator=g)

    def c1():
        o = model(pixel_values=x, labels=lab, labels_clf=clf)
        o.loss.backward(); opt.step(); opt.zero_grad()
    print(f"c1 train step (tiny_vit_5m_224, batch 8): latency {wall(c1, 30, True):.3f} ms, back-to-back {wall(c1, 30, False):.3f} ms", flush=True)

    def fwd_only():
        with torch.no_grad():
            model(pixel_values=x, labels=lab, labels_clf=clf)
    print(f"   forward only (train mode): {wall(fwd_only, 30, False):.3f} ms", flush=True)


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "profile"):
    main()


def profile_serving(name="tiny_vit_21m_224", B=4):
    """per-launch durations (GG_PROF events) of one serving call: where a launch-bound call's time goes"""
    import ctypes as C
    from geoguessr_ai_amd import _lib as L
    dev = "cuda"
    base = TinyViTAdapter(name, pretrained=False, precision="fp32")
    model = SuperGuessr(base, panorama=True, serving=True).to(dev).eval()
    x = torch.randn(B // 4, 4, 3, 224, 224, device=dev)
    dummy = torch.zeros(B // 4, dtype=torch.long, device=dev)
    lib = L.lib()
    with torch.no_grad():
        for _ in range(3): model(pixel_values=x, labels_clf=dummy)
        torch.cuda.synchronize()
        lib.gg_prof_enable(1); lib.gg_prof_reset()
        model(pixel_values=x, labels_clf=dummy)
        torch.cuda.synchronize()
    n = lib.gg_prof_count()
    cat, ms, fl, by = C.c_int(), C.c_double(), C.c_double(), C.c_double()
    rows = []
    for i in range(n):
        lib.gg_prof_record(i, C.byref(cat), C.byref(ms), C.byref(fl), C.byref(by))
        rows.append((i, cat.value, ms.value * 1e3, fl.value, by.value))
    lib.gg_prof_enable(0); lib.gg_prof_reset()
    names = ["gemm", "attn", "dwconv", "norm", "head", "optim", "move"]
    tot = sum(r[2] for r in rows)
    print(f"{name} B={B}: {n} launches, sum of launch durations {tot:.0f} us")
    per = {}
    for r in rows: per.setdefault(names[r[1]], []).append(r[2])
    for k, v in per.items(): print(f"  {k:7s} {len(v):4d} launches {sum(v):8.0f} us  (median {sorted(v)[len(v)//2]:.1f} us, max {max(v):.1f})")
    for r in sorted(rows, key=lambda r: -r[2])[:12]: print(f"    #{r[0]:3d} {names[r[1]]:7s} {r[2]:7.1f} us  {r[3]/1e9:7.2f} GF {r[4]/1e6:8.1f} MB")
    a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
    print("graph cache:", lib.gg_graph_stats(C.byref(a), C.byref(b), C.byref(c)), "keys; captures", a.value, "replays", b.value, "eager", c.value)


def profile_c1():
    import ctypes as C
    from geoguessr_ai_amd import _lib as L
    dev = "cuda"
    base = TinyViTAdapter("tiny_vit_5m_224", pretrained=False, precision="fp32")
    model = SuperGuessr(base, panorama=False, should_smooth_labels=False, serving=False).to(dev).train()
    opt = AdamW(model, lr=5e-5, betas=(0.9, 0.999), weight_decay=0.01)
    g = torch.Generator(device=dev).manual_seed(330)
    x = torch.randn(8, 3, 224, 224, device=dev, generator=g)
    lab = torch.stack([torch.rand(8, device=dev, generator=g) * 360 - 180, torch.rand(8, device=dev, generator=g) * 180 - 90], 1)
    clf = torch.randint(0, model.num_cells, (8,), device=dev, generator=g)

    def c1():
        o = model(pixel_values=x, labels=lab, labels_clf=clf)
        o.loss.backward(); opt.step(); opt.zero_grad()
    lib = L.lib()
    for _ in range(3): c1()
    torch.cuda.synchronize()
    lib.gg_prof_enable(1); lib.gg_prof_reset()
    c1()
    torch.cuda.synchronize()
    n = lib.gg_prof_count()
    cat, ms, fl, by = C.c_int(), C.c_double(), C.c_double(), C.c_double()
    rows = []
    for i in range(n):
        lib.gg_prof_record(i, C.byref(cat), C.byref(ms), C.byref(fl), C.byref(by))
        rows.append((i, cat.value, ms.value * 1e3, fl.value, by.value))
    lib.gg_prof_enable(0); lib.gg_prof_reset()
    names = ["gemm", "attn", "dwconv", "norm", "head", "optim", "move"]
    print(f"c1 step: {n} launches, sum of launch durations {sum(r[2] for r in rows):.0f} us")
    per = {}
    for r in rows: per.setdefault(names[r[1]], []).append(r[2])
    for k, v in per.items(): print(f"  {k:7s} {len(v):4d} launches {sum(v):8.0f} us  (median {sorted(v)[len(v)//2]:.1f} us, max {max(v):.1f})")
    for r in sorted(rows, key=lambda r: -r[2])[:16]: print(f"    #{r[0]:3d} {names[r[1]]:7s} {r[2]:7.1f} us  {r[3]/1e9:7.2f} GF {r[4]/1e6:8.1f} MB")


if len(sys.argv) > 1 and sys.argv[1] == "profile":
    profile_serving()
    profile_c1()
