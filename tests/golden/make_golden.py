#!/usr/bin/env python3
"""Generates the golden fixtures in this directory by IMPORTING the reference
(``/root/reference``, read-only) -- run only in the build container, never on the GPU box:

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Outputs are data only (inputs + expected outputs); no reference source is copied.
Import recipe: SURVEY.md Appendix D (``config.py`` cannot be imported under transformers>=5,
so its constants above the ``# Training arguments`` marker are exec'd into a stub module).

Big random operands are regenerated from seeds with ``numpy.random.default_rng`` (same
image on the GPU box => same stream); a checksum of each regenerated operand is stored so
a stream mismatch is detected rather than misread as a parity failure.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _import_reference():
    os.chdir(REF)
    sys.path.insert(0, REF)
    src = open("config.py").read()
    head = src.split("# Training arguments")[0].replace("from transformers import TrainingArguments", "")
    cfg = types.ModuleType("config")
    exec(head, cfg.__dict__)
    cfg.TRAIN_ARGS = cfg.PRETRAIN_ARGS = cfg.PRETAIN_ARGS = None
    sys.modules["config"] = cfg


def edge_labels():
    """(lon, lat) deg: identical-to-centroid, antipodes, poles, +-180 wrap, equator, random."""
    rng = np.random.default_rng(330)
    pts = [
        (20.126657, 41.71182),      # == centroid 0 (distance 0 -> soft target peak)
        (-159.873343, -41.71182),   # its antipode
        (0.0, 90.0), (0.0, -90.0),  # poles
        (180.0, 0.0), (-180.0, 0.0), (179.9999, 10.0), (-179.9999, 10.0),
        (0.0, 0.0), (10.75, 59.91), (-73.98, 40.75), (139.69, 35.68),
    ]
    while len(pts) < 16:
        pts.append((rng.uniform(-180, 180), rng.uniform(-90, 90)))
    return np.asarray(pts, np.float32)


def main():
    _import_reference()
    import torch
    torch.manual_seed(0)
    from models.super_guessr import SuperGuessr
    from models.utils import haversine_matrix, smooth_labels
    from preprocessing.geo_utils import haversine

    m = SuperGuessr(base_model=None, panorama=True, should_smooth_labels=True, embed_dim=576)
    K = m.num_cells
    cent = m.geocell_centroid_coords.detach().numpy().copy()
    np.save(os.path.join(HERE, "centroids_12647x2_f32.npy"), cent)

    # ---- haversine / smooth / soft-CE -------------------------------------------------
    labels = edge_labels()
    rng = np.random.default_rng(1234)
    logits = rng.standard_normal((16, K), dtype=np.float32) * 2.0
    lt = torch.from_numpy(labels)
    zt = torch.from_numpy(logits).requires_grad_(True)
    d = haversine_matrix(lt, m.geocell_centroid_coords.data.t())
    soft = smooth_labels(d)
    soft_n = soft / soft.sum(dim=-1, keepdim=True).clamp_min(1e-12)
    lp = torch.nn.functional.log_softmax(zt, dim=-1)
    loss = -(soft_n * lp).sum(dim=-1).mean()
    loss.backward()
    targets = torch.argmin(d, dim=-1)
    hard = torch.nn.CrossEntropyLoss()(zt.detach(), targets)
    np.savez_compressed(
        os.path.join(HERE, "geo_loss.npz"),
        labels=labels, logits_seed=1234, logits_scale=2.0, logits_checksum=np.float64(logits.astype(np.float64).sum()),
        distances=d.numpy(), soft_rowsum=soft.sum(-1).numpy(), soft_argmax=soft_n.argmax(-1).numpy(),
        loss=np.float32(loss.item()), dlogits=zt.grad.numpy(), argmin=targets.numpy(),
        hard_ce=np.float32(hard.item()))

    # ---- head forward / loss through the real SuperGuessr ------------------------------
    rng = np.random.default_rng(77)
    W = (rng.standard_normal((K, 576), dtype=np.float32) * 0.05)
    b = (rng.standard_normal((K,), dtype=np.float32) * 0.1)
    emb = rng.standard_normal((32, 4, 576), dtype=np.float32)
    lab = np.stack([rng.uniform(-180, 180, 32), rng.uniform(-90, 90, 32)], 1).astype(np.float32)
    with torch.no_grad():
        m.cell_layer.weight.copy_(torch.from_numpy(W)); m.cell_layer.bias.copy_(torch.from_numpy(b))
    m.train()
    lab_t = torch.from_numpy(lab)
    clf = torch.argmin(haversine_matrix(lab_t, m.geocell_centroid_coords.data.t()), dim=-1)
    e = torch.from_numpy(emb).requires_grad_(True)
    out = m(embedding=e, labels=lab_t, labels_clf=clf)
    out.loss.backward()
    m.should_smooth_labels = False
    out_h = m(embedding=torch.from_numpy(emb), labels=lab_t, labels_clf=clf)
    m.should_smooth_labels = True
    np.savez_compressed(
        os.path.join(HERE, "head.npz"),
        seed=77, W_checksum=np.float64(W.astype(np.float64).sum()), b_checksum=np.float64(b.astype(np.float64).sum()),
        emb_checksum=np.float64(emb.astype(np.float64).sum()), labels=lab, labels_clf=clf.numpy(),
        loss=np.float32(out.loss.item()), loss_hard=np.float32(out_h.loss.item()),
        preds_geocell=out.preds_geocell.numpy(), preds_LLH=out.preds_LLH.numpy(),
        top5_vals=out.top5_geocells.values.detach().numpy(), top5_idx=out.top5_geocells.indices.numpy(),
        demb=e.grad.numpy(),
        dW_rows=m.cell_layer.weight.grad.numpy()[clf.numpy()[:8]],   # 8 rows of dW (rows of the true cells)
        dW_abs_sum=np.float64(m.cell_layer.weight.grad.abs().double().sum().item()),
        db_abs_sum=np.float64(m.cell_layer.bias.grad.abs().double().sum().item()))
    m.zero_grad()

    # ---- 3-step training trace (legacy loop contract: AdamW(lr=2e-5), config.py:96) -------
    torch.manual_seed(0)
    rng = np.random.default_rng(2024)
    W0 = (rng.standard_normal((K, 576), dtype=np.float32) * 0.02)
    with torch.no_grad():
        m.cell_layer.weight.copy_(torch.from_numpy(W0)); m.cell_layer.bias.zero_()
    opt = torch.optim.AdamW(m.parameters(), lr=2e-5)
    emb_t = rng.standard_normal((3, 64, 4, 576), dtype=np.float32)
    lab3 = np.stack([rng.uniform(-180, 180, (3, 64)), rng.uniform(-90, 90, (3, 64))], -1).astype(np.float32)
    losses = []
    for s in range(3):
        lt3 = torch.from_numpy(lab3[s])
        clf3 = torch.argmin(haversine_matrix(lt3, m.geocell_centroid_coords.data.t()), dim=-1)
        o = m(embedding=torch.from_numpy(emb_t[s]), labels=lt3, labels_clf=clf3)
        o.loss.backward(); opt.step(); opt.zero_grad()
        losses.append(o.loss.item())
    np.savez_compressed(
        os.path.join(HERE, "train_trace.npz"), seed=2024, lr=2e-5, losses=np.asarray(losses, np.float32),
        W_final_checksum=np.float64(m.cell_layer.weight.double().sum().item()),
        W_delta_abs_sum=np.float64((m.cell_layer.weight.detach() - torch.from_numpy(W0)).abs().double().sum().item()),
        b_final=m.cell_layer.bias.detach().numpy()[:64].copy())

    # ---- fp64-radius pairwise haversine (refiner gate) -----------------------------------
    a = torch.from_numpy(edge_labels()); bpts = torch.from_numpy(edge_labels()[::-1].copy())
    hv = haversine(a, bpts).numpy()

    # ---- ProtoRefiner helpers (class imported behind MagicMock stubs) -------
    import datasets, transformers  # noqa: F401  (must be imported before the stubs)
    from unittest.mock import MagicMock
    for name in ["timm", "timm.data", "timm.data.transforms_factory", "torchvision", "torchvision.transforms",
                 "loguru", "wandb", "dotenv", "boto3", "botocore", "botocore.config", "botocore.exceptions", "s3fs"]:
        sys.modules.setdefault(name, MagicMock())
    from models.proto_refiner import ProtoRefiner
    rng = np.random.default_rng(5)
    M = rng.standard_normal((7, 576), dtype=np.float32); v = rng.standard_normal((576,), dtype=np.float32)
    ed = ProtoRefiner._euclidean_distance(None, torch.from_numpy(M), torch.from_numpy(v)).numpy()
    obj = types.SimpleNamespace(temperature=torch.tensor(1.6))
    xs = np.asarray([-3.0, -2.5, -100000.0, -7.25, -2.75], np.float32)
    ts = ProtoRefiner._temperature_softmax(obj, torch.from_numpy(xs)).numpy()
    np.savez_compressed(os.path.join(HERE, "proto_helpers.npz"), M=M, v=v, euclid=ed, ts_in=xs, ts_out=ts,
                        hav_a=a.numpy(), hav_b=bpts.numpy(), hav_km=hv)

    # ---- tiny CLIP vision tower through transformers --------------------------------------
    from transformers import CLIPVisionConfig, CLIPVisionModel
    torch.manual_seed(0)
    tcfg = dict(hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                image_size=64, patch_size=32)
    clip = CLIPVisionModel(CLIPVisionConfig(**tcfg)).eval()
    sd = {k.replace("vision_model.", ""): v.detach().numpy().copy() for k, v in clip.state_dict().items()
          if "position_ids" not in k}
    x = torch.randn(3, 3, 64, 64)
    with torch.no_grad():
        o = clip(pixel_values=x)
        y = o.last_hidden_state.mean(dim=1)
    np.savez_compressed(os.path.join(HERE, "clip_tiny.npz"), x=x.numpy(), y=y.numpy(),
                        last_hidden_state=o.last_hidden_state.numpy(),
                        cfg=np.asarray([tcfg[k] for k in ("hidden_size", "intermediate_size", "num_hidden_layers",
                                                          "num_attention_heads", "image_size", "patch_size")]),
                        **{"w." + k: v for k, v in sd.items()})
    print("fixtures written to", HERE)
    for f in sorted(os.listdir(HERE)):
        print(f"  {f:40s} {os.path.getsize(os.path.join(HERE, f)) / 1024:.1f} KB")


if __name__ == "__main__":
    main()
