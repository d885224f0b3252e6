#!/usr/bin/env python3
"""A/B of the 16-bit NT GEMM forms (dev tool): the register-staged 128 x 128 kernel (GG_GEMM_DMA=0) against the LDS-DMA form in its two geometries (192 x 128, two
workgroups per CU; 256 x 256, one), interleaved in ONE process on the same random operands, on the shapes of BASELINE c4 (CLIP ViT-B/32, batch 1024) and of the bf16 TinyViT-21M step.  Every shape is first checked
against a torch fp32 product of the same bf16 operands (both forms).  Run with GG_DEV_SWITCHES=1."""
import os, sys
os.environ.setdefault("GG_DEV_SWITCHES", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import ops

T = 1024 * 50
Ms2, Ms3 = 1024 * 14 * 14, 1024 * 7 * 7
shapes = [
    ("c4.qkv", T, 2304, 768, {"bias": True}), ("c4.proj", T, 768, 768, {"bias": True, "residual": True}),
    ("c4.fc1", T, 3072, 768, {"bias": True, "act": "quick_gelu"}), ("c4.fc2", T, 768, 3072, {"bias": True, "residual": True}),
    ("s2.qkv", Ms2, 1152, 384, {"bias": True}), ("s2.proj", Ms2, 384, 384, {"bias": True, "residual": True}),
    ("s2.fc1", Ms2, 1536, 384, {"bias": True, "act": "gelu", "preact": True}), ("s2.fc2", Ms2, 384, 1536, {"bias": True, "residual": True}),
    ("s2.fc2.dgrad", Ms2, 1536, 384, {"dact": True}), ("s2.fc1.dgrad", Ms2, 384, 1536, {}),
    ("s3.qkv", Ms3, 1728, 576, {"bias": True}), ("s3.fc1", Ms3, 2304, 576, {"bias": True, "act": "gelu", "preact": True}),
    ("s3.fc2", Ms3, 576, 2304, {"bias": True, "residual": True}),
    ("square4k", 4096, 4096, 4096, {}), ("square8k", 8192, 8192, 8192, {}),
    ("edge", 5000, 1000, 712, {"bias": True, "residual": True}),
]
only = sys.argv[1:] or None
rounds = 5


FORMS = {"0": ("0", "192"), "1": ("1", "192"), "2": ("1", "256")}


def run(form, A, W, out, kw):
    os.environ["GG_GEMM_DMA"], os.environ["GG_GEMM_DMA_TILE"] = FORMS[form]
    return ops.gemm_nt(A, W, out=out, **kw)


for name, M, N, K, o in shapes:
    if only and not any(name.startswith(x) for x in only):
        continue
    torch.manual_seed(0)
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    kw = {}
    if o.get("bias"): kw["bias"] = torch.randn(N, device="cuda")
    if o.get("act"): kw["act"] = o["act"]
    if o.get("preact"): kw["preact"] = True
    if o.get("residual"): kw["residual"] = torch.randn(M, N, device="cuda").bfloat16()
    if o.get("dact"): kw["dact_preact"], kw["dact"] = torch.randn(M, N, device="cuda").bfloat16(), "gelu"
    out = {f: torch.empty((M, N), dtype=torch.bfloat16, device="cuda") for f in "012"}
    # parity of both forms on a row sample against torch fp32 (same bf16 operands)
    rows = torch.randint(0, M, (512,), device="cuda")
    rows[:4] = torch.tensor([0, 1, M - 2, M - 1], device="cuda")
    ref = A[rows].float() @ W.float().t()
    if "bias" in kw: ref = ref + kw["bias"]
    ref = ref.bfloat16().float()
    if o.get("act") == "gelu": ref = torch.nn.functional.gelu(ref)
    if o.get("act") == "quick_gelu": ref = ref * torch.sigmoid(1.702 * ref)
    if o.get("dact"):
        x = kw["dact_preact"][rows].float()
        ref = ref * (0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5)
    if o.get("residual"): ref = ref.bfloat16().float() + kw["residual"][rows].float()
    errs = {}
    for f in "012":
        r = run(f, A, W, out[f], kw)
        got = (r[0] if isinstance(r, tuple) else r)[rows].float()
        errs[f] = float((got - ref).norm() / ref.norm())
    same = bool((out["0"] == out["1"]).all()) and bool((out["0"] == out["2"]).all())
    torch.cuda.synchronize()
    t = {"0": [], "1": [], "2": []}
    for _ in range(rounds):
        for f in "012":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run(f, A, W, out[f], kw)
            e0.record()
            for _ in range(3): run(f, A, W, out[f], kw)
            e1.record(); torch.cuda.synchronize()
            t[f].append(e0.elapsed_time(e1) / 3)
    med = {f: sorted(t[f])[len(t[f]) // 2] for f in "012"}
    fl = 2.0 * M * N * K
    print(f"{name:13s} M={M:7d} N={N:5d} K={K:5d}  old {med['0']*1e3:8.1f} us {fl/med['0']/1e9:6.0f} TF | 192x128 {med['1']*1e3:8.1f} us {fl/med['1']/1e9:6.0f} TF (x{med['0']/med['1']:.2f}) | "
          f"256x256 {med['2']*1e3:8.1f} us {fl/med['2']/1e9:6.0f} TF (x{med['0']/med['2']:.2f})  relerr {errs['0']:.1e} {errs['1']:.1e} {errs['2']:.1e} bit-identical {same}", flush=True)
    del A, W, out, kw
