#!/usr/bin/env python3
"""Per-launch table of one instrumented training step (dev tool): groups the libgg launch records by (category, flops, bytes)
and prints time, TF/s and GB/s for each group, largest first."""
import sys, os, ctypes as C, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import _lib as L
from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
from geoguessr_ai_amd.models.super_guessr import SuperGuessr
from geoguessr_ai_amd.optim import AdamW

CATS = ["gemm", "attention", "dwconv", "norm", "head", "optim", "move"]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
PREC = "fp32" if "--fp32" in sys.argv else "bf16"
PEAK = 157.3e9 if PREC == "fp32" else 2.5e12      # flop / ms
base = TinyViTAdapter("tiny_vit_21m_224", pretrained=False, precision=PREC)
model = SuperGuessr(base, panorama=True, should_smooth_labels=True, serving=False).to(dev).train()
if "--unfrozen" in sys.argv: base.unfreeze_all()
opt = AdamW(model, lr=5e-5)
N = 256
x = torch.randn(N, 4, 3, 224, 224, device=dev)
lab = torch.stack([torch.rand(N, device=dev) * 360 - 180, torch.rand(N, device=dev) * 180 - 90], 1)
def step():
    out = model(pixel_values=x, labels=lab)
    out.loss.backward(); opt.step(); opt.zero_grad()
for _ in range(2): step()
torch.cuda.synchronize()
lib = L.lib()
lib.gg_prof_reset(); lib.gg_prof_enable(1)
K = 3
for _ in range(K): step()
torch.cuda.synchronize()
lib.gg_prof_enable(0)
groups = collections.OrderedDict()
for i in range(lib.gg_prof_count()):
    c, ms, fl, by = C.c_int(), C.c_double(), C.c_double(), C.c_double()
    L.check(lib.gg_prof_record(i, C.byref(c), C.byref(ms), C.byref(fl), C.byref(by)), "rec")
    g = groups.setdefault((c.value & 15, fl.value, by.value), [0, 0.0, i])      # (bit 4: split-bf16 GEMM, same class)
    g[0] += 1; g[1] += ms.value
tot = sum(g[1] for g in groups.values()) / K
print(f"precision {PREC}: total {tot:.2f} ms/step of instrumented kernels")
ideal = sum(n * max(fl / PEAK, by / 8e9) for (c, fl, by), (n, ms, first) in groups.items() if c == 0) / K
gemm = sum(ms for (c, fl, by), (n, ms, first) in groups.items() if c == 0) / K
print(f"GEMM per-launch roofline: sum max(flops/peak, bytes/8TB/s) = {ideal:.2f} ms vs measured {gemm:.2f} ms -> {ideal / gemm:.3f}")
for (c, fl, by), (n, ms, first) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    per = ms / n
    if ms / K < 0.15: continue
    print(f"{CATS[c]:9s} first#{first:4d} n/step={n/K:5.1f} {ms/K:7.3f} ms/step  {per*1e3:8.1f} us/launch  {fl/per/1e9 if fl else 0:7.1f} TF/s  {by/per/1e6:7.1f} GB/s  "
          f"gflop={fl/1e9:8.2f} MB={by/1e6:8.1f}  hbm-floor {by/8e6:7.1f} us  mfma-floor {fl/PEAK*1e3:7.1f} us")
