#!/usr/bin/env python3
"""c1 (TinyViT-5M-224, batch 8 train step) with the whole step captured in a HIP graph: the step is ~600 launches of a static
schedule, launch-bound at this batch size.  Prints eager vs graph-replay ms/step."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geoguessr_ai_amd import _lib as L
from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
from geoguessr_ai_amd.models.super_guessr import SuperGuessr
from geoguessr_ai_amd.optim import AdamW

L.require_gpu()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
name = sys.argv[2] if len(sys.argv) > 2 else "tiny_vit_5m_224"
pano = len(sys.argv) > 3 and sys.argv[3] == "pano"            # headline form: n panoramas x 4 headings, soft labels
base = TinyViTAdapter(name, pretrained=False)
model = SuperGuessr(base, panorama=pano, should_smooth_labels=pano, serving=False).to(dev).train()
opt = AdamW(model, lr=5e-5)
g = torch.Generator(device=dev).manual_seed(1234)
x = torch.randn((n, 4, 3, 224, 224) if pano else (n, 3, 224, 224), device=dev, generator=g)
lab = torch.stack([torch.rand(n, device=dev, generator=g) * 360 - 180, torch.rand(n, device=dev, generator=g) * 180 - 90], 1)
clf = torch.randint(0, model.num_cells, (n,), device=dev, generator=g)


def step():
    out = model(pixel_values=x, labels=lab) if pano else model(pixel_values=x, labels=lab, labels_clf=clf)
    out.loss.backward(); opt.step(); opt.zero_grad()
    return out.loss


def timed(fn, steps=20 if pano else 50, warmup=5):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


eager = timed(step)
res = dict(case=name, images_per_step=n * (4 if pano else 1), eager_ms=round(eager, 3))
try:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss = step()
    gms = timed(graph.replay)
    res.update(graph_ms=round(gms, 3), images_per_s_graph=round(n / gms * 1e3, 1), loss=float(loss))
except Exception as e:
    res.update(graph_error=repr(e)[:300])
print(json.dumps(res))
