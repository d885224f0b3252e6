#!/usr/bin/env python3
"""dev: which (N, K, epilogue) of the bf16 TinyViT step makes the forward non-repeatable under the LDS-DMA GEMM."""
import os, sys
os.environ["GG_DEV_SWITCHES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
N_IMG = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
ad = TinyViTAdapter("tiny_vit_21m_224", pretrained=False, precision="bf16").cuda()
x = torch.randn(N_IMG, 3, 224, 224, device="cuda")
ad.train(); bb = ad.backbone
for n, p in bb.named_parameters(): p.requires_grad = not n.startswith(("stages.0", "stages.1", "stages.2"))
g = torch.Generator(device="cuda").manual_seed(5)
drop = bb.make_drop_scales(N_IMG, generator=g)
d_out = torch.randn(N_IMG, 576, device="cuda", generator=g) * 1e-3
def run():
    bb.flat_grads().zero_()
    out = bb.forward_hip(x, training=True, drop_scales=drop).clone()
    bb.backward_hip(d_out)
    return out, bb.flat_grads().clone()
cands = [(n, k, e) for (n, k) in [(192, 192), (576, 192), (768, 192), (192, 768), (384, 384), (1152, 384), (1536, 384), (384, 1536), (576, 576), (1728, 576), (2304, 576), (576, 2304), (384, 768), (768,384), (192,384), (384, 192)] for e in (0, 1, 2, 4)]
os.environ["GG_GEMM_DMA_ONLY"] = "0,0"
o1, g1 = run(); o2, g2 = run()
print("no DMA: fwd equal", torch.equal(o1, o2))
del os.environ["GG_GEMM_DMA_ONLY"]
o1, g1 = run(); o2, g2 = run()
print("all DMA: fwd equal", torch.equal(o1, o2), "max diff", float((o1.float() - o2.float()).abs().max()))
for n, k, e in cands:
    os.environ["GG_GEMM_DMA_ONLY"] = f"{n},{k},{e}"
    o1, g1 = run(); o2, g2 = run()
    if not torch.equal(o1, o2) or not torch.allclose(g1, g2, rtol=1e-3, atol=1e-7):
        print(f"N={n} K={k} epi={e}: fwd equal {torch.equal(o1, o2)} grads close {torch.allclose(g1, g2, rtol=1e-3, atol=1e-7)}", flush=True)
print("done")
