#!/usr/bin/env python3
"""Secondary numbers of SURVEY.md 8(d) on one MI355X (the headline c2 line is bench.py's):
  c1  TinyViT-5M-224, batch 8, forward + geocell hard-CE + backward + AdamW (panorama=False)
  c2u TinyViT-21M-224, 256 panoramas, every parameter trainable
  c4  CLIP ViT-B/32 vision tower, batch 1024, inference (random weights; fp32 = the reference's precision, bf16 = 16-bit MFMA operands)
  c4-train  SuperGuessr on the CLIP ViT-B/32 base, 64 panoramas, reference freeze policy with the head file present, forward + backward + AdamW
  c5  SuperGuessr head (serving) + ProtoRefiner on precomputed embeddings, batch 4096
Prints one JSON object per case.  Synthetic inputs as in bench.py."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from geoguessr_ai_amd import _lib as L
from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
from geoguessr_ai_amd.models.super_guessr import SuperGuessr
from geoguessr_ai_amd.models.proto_refiner import ProtoRefiner
from geoguessr_ai_amd.optim import AdamW

L.require_gpu()
dev = torch.device("cuda", 0)
cases = [a for a in sys.argv[1:] if not a.startswith("--")] or ["c1", "c2u", "c4", "c5"]
precisions = ["fp32", "bf16"] if "--both" in sys.argv or not any(a.startswith("--") for a in sys.argv[1:]) else [a[2:] for a in sys.argv[1:] if a in ("--fp32", "--bf16")]


def timed(fn, steps, warmup):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def train_case(name, model_name, n, panorama, unfrozen, smooth, steps, warmup, precision):
    torch.manual_seed(0)
    base = TinyViTAdapter(model_name, pretrained=False, precision=precision)
    model = SuperGuessr(base, panorama=panorama, should_smooth_labels=smooth, serving=False).to(dev).train()
    if unfrozen: base.unfreeze_all()
    opt = AdamW(model, lr=5e-5, betas=(0.9, 0.999), weight_decay=0.01)
    g = torch.Generator(device=dev).manual_seed(1234)
    x = torch.randn((n, 4, 3, 224, 224) if panorama else (n, 3, 224, 224), device=dev, generator=g)
    lab = torch.stack([torch.rand(n, device=dev, generator=g) * 360 - 180, torch.rand(n, device=dev, generator=g) * 180 - 90], 1)
    clf = torch.randint(0, model.num_cells, (n,), device=dev, generator=g)
    def step():
        out = model(pixel_values=x, labels=lab) if smooth else model(pixel_values=x, labels=lab, labels_clf=clf)   # soft / hard CE
        out.loss.backward(); opt.step(); opt.zero_grad()
    dt = timed(step, steps, warmup)
    imgs = n * (4 if panorama else 1)
    print(json.dumps(dict(case=name, precision=precision, model=model_name, images_per_step=imgs, ms_per_step=round(dt * 1e3, 3), images_per_s=round(imgs / dt, 1),
                          trainable="all" if unfrozen else "freeze_all_but_last_stage")))
    del model, base, opt, x
    import gc; gc.collect(); torch.cuda.empty_cache()


for prec in precisions:
    if "c1" in cases:
        train_case("c1", "tiny_vit_5m_224", 8, False, False, False, 20, 5, prec)
    if "c2u" in cases:
        train_case("c2-unfrozen", "tiny_vit_21m_224", 256, True, True, True, 5, 2, prec)
if "c4" in cases:
    from geoguessr_ai_amd.pretrain.clip_embedder import CLIPVisionTower
    for prec in precisions + ["fp16"]:  # fp32 = the reference's precision (pretrain/clip_embedder.py:51-66 runs the tower in fp32), fp16 = BASELINE c4's, bf16
        tower = CLIPVisionTower("openai/clip-vit-base-patch32", precision=prec).to(dev).eval()
        for p_ in tower.parameters():
            p_.requires_grad = False
        x = torch.randn(1024, 3, 224, 224, device=dev)
        with torch.no_grad():
            dt = timed(lambda: tower(pixel_values=x, return_last_hidden=False), 5, 2)
        print(json.dumps(dict(case="c4", precision=prec, model="CLIP ViT-B/32 vision tower (random weights), inference", images_per_step=1024,
                              ms_per_step=round(dt * 1e3, 3), images_per_s=round(1024 / dt, 1), tflops=round(1024 / dt * 8.82e9 / 1e12, 1))))
        del tower, x
        import gc; gc.collect(); torch.cuda.empty_cache()
if "c4t" in cases or "c4" in cases:
    # SuperGuessr on a CLIP ViT-B/32 base, 4-heading panoramas, last encoder layer + embeddings trainable (models/super_guessr.py:134-150), fwd+bwd+AdamW
    from geoguessr_ai_amd.pretrain.clip_embedder import CLIPVisionTower
    for prec in precisions:
        torch.manual_seed(0)
        tower = CLIPVisionTower("openai/clip-vit-base-patch32", precision=prec)
        model = SuperGuessr(tower, panorama=True, should_smooth_labels=True).to(dev).train()
        for layer in list(tower.vision_model.encoder.layers)[:-1]:
            for p_ in layer.parameters():
                p_.requires_grad = False
        opt = AdamW(model, lr=2e-5)
        n = 64
        x = torch.randn(n, 4, 3, 224, 224, device=dev)
        lab = torch.stack([torch.rand(n, device=dev) * 360 - 180, torch.rand(n, device=dev) * 180 - 90], 1)
        def step():
            out = model(pixel_values=x, labels=lab)
            out.loss.backward(); opt.step(); opt.zero_grad()
        dt = timed(step, 5, 2)
        print(json.dumps(dict(case="c4-train", precision=prec, model="SuperGuessr on CLIP ViT-B/32, encoder layers[:-1] frozen (embeddings + last layer + head train)",
                              images_per_step=4 * n, ms_per_step=round(dt * 1e3, 3), images_per_s=round(4 * n / dt, 1))))
        del model, tower, opt, x
        import gc; gc.collect(); torch.cuda.empty_cache()
if "c5" in cases:
    torch.manual_seed(0)
    Bq, D, K = 4096, 576, 12647
    head = SuperGuessr(None, panorama=True, serving=True, embed_dim=D, precision="fp32").to(dev).eval()
    rng = np.random.default_rng(0)
    counts = rng.poisson(4.0, K)
    gi = np.repeat(np.arange(K), counts)
    refiner = ProtoRefiner.from_clusters(gi, rng.standard_normal((len(gi), D), dtype=np.float32), rng.uniform(-180, 180, len(gi)).astype(np.float32),
                                         rng.uniform(-90, 90, len(gi)).astype(np.float32), K, topk=5).to(dev).eval()
    emb = torch.randn(Bq, 4, D, device=dev)
    def full():
        with torch.no_grad():
            llh, topk, e = head(embedding=emb)               # serving eval: (pred_LLH, TopK(values, indices), embedding)
            refiner(e, llh, topk.indices, topk.values)
    dt = timed(full, 10, 3)
    print(json.dumps(dict(case="c5", what="SuperGuessr serving head + ProtoRefiner on precomputed embeddings", batch=Bq, prototypes=int(len(gi)),
                          ms_per_step=round(dt * 1e3, 3), samples_per_s=round(Bq / dt, 1))))
