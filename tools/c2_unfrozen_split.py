#!/usr/bin/env python3
"""dev: c2 with EVERY parameter trainable (TinyViT-21M-224, 256 panoramas, fwd + bwd + AdamW) in the fp32_split and the fp32 mode: ms per step, the first losses, and the
relative difference of the two modes' gradients after one step from the same state (the unfrozen schedule sends every block's weight gradient through gg_gemm_tn_split3)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
from geoguessr_ai_amd.models.super_guessr import SuperGuessr
from geoguessr_ai_amd.optim import AdamW
dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device=dev).manual_seed(1234)
x = torch.randn(N, 4, 3, 224, 224, device=dev, generator=g)
lab = torch.stack([torch.rand(N, device=dev, generator=g) * 360 - 180, torch.rand(N, device=dev, generator=g) * 180 - 90], 1)
grads = {}
for prec in ("fp32_split", "fp32"):
    torch.manual_seed(0)
    base = TinyViTAdapter("tiny_vit_21m_224", pretrained=False, precision=prec, drop_path_rate=0.0)
    model = SuperGuessr(base, panorama=True, should_smooth_labels=True).to(dev).train()
    base.unfreeze_all()
    opt = AdamW(model, lr=5e-5)
    losses = []
    def step():
        o = model(pixel_values=x, labels=lab); o.loss.backward()
        if not grads.get(prec + "_done"):
            grads[prec] = base.backbone.flat_grads().clone(); grads[prec + "_done"] = True
        opt.step(); opt.zero_grad(); losses.append(float(o.loss.detach()))
    step(); step(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"{prec:10s} unfrozen, {N} panoramas: {dt * 1e3:7.1f} ms per step, {4 * N / dt:7.0f} images/s, losses {[round(l, 4) for l in losses]}", flush=True)
    del model, base, opt
    import gc; gc.collect(); torch.cuda.empty_cache()
a, b = grads["fp32_split"].double(), grads["fp32"].double()
print(f"first-step flat gradient, fp32_split vs fp32: rel-L2 {float((a - b).norm() / b.norm()):.2e}, all finite {bool(torch.isfinite(a).all())}")
