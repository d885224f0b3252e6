#!/usr/bin/env python3
"""Round-3 golden fixture, produced by RUNNING the reference's own ``SuperGuessr`` on a ``transformers.CLIPVisionModel`` base -- build
container only, never on the GPU box:

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r3.py

* ``clip_train.npz`` -- one training forward + backward of ``SuperGuessr(base_model=CLIPVisionModel(tiny config), panorama=True,
  should_smooth_labels=True)`` (``models/super_guessr.py:20-395``; the CLIP branch of ``_freeze_params`` :134-150 with the pretrained head
  file absent, i.e. every layer trainable -- the configuration ``main_coordinator_idun_s3.py:183-203`` builds).  The tower's weights are
  the ones already committed in ``clip_tiny.npz``; the head weight and the inputs are regenerated from seeds (checksums stored).
  Stored: loss, the (N,4,C) embedding, the gradient of EVERY vision-tower parameter -- small tensors (biases, LayerNorm, class /
  position embeddings) in full, weight matrices as their Frobenius norm plus the two seeded random projections ``G @ r`` and ``l @ G``
  (every entry of G enters both) --, and rows of the head's weight gradient.
Outputs are data only (inputs + expected outputs)."""
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


def projections(name, G):
    """Frobenius norm + G @ r and l @ G for seeded unit-variance r, l (the seed is a function of the tensor name and shape)."""
    G2 = G.reshape(G.shape[0], -1).astype(np.float64)
    seed = (sum(ord(c) for c in name) * 7919 + G2.shape[0] * 31 + G2.shape[1]) % (2 ** 31)
    rng = np.random.default_rng(seed)
    r, l = rng.standard_normal(G2.shape[1]), rng.standard_normal(G2.shape[0])
    return np.float64(np.linalg.norm(G2)), (G2 @ r).astype(np.float64), (l @ G2).astype(np.float64)


def main():
    _import_reference()
    import torch
    torch.manual_seed(0)
    from transformers import CLIPVisionConfig, CLIPVisionModel
    from models.super_guessr import SuperGuessr

    g = np.load(os.path.join(HERE, "clip_tiny.npz"))
    hs, inter, nl, nh, img, ps = [int(v) for v in g["cfg"]]
    clip = CLIPVisionModel(CLIPVisionConfig(hidden_size=hs, intermediate_size=inter, num_hidden_layers=nl, num_attention_heads=nh,
                                            image_size=img, patch_size=ps))
    sd = clip.state_dict()
    prefix = "vision_model." if any(k.startswith("vision_model.") for k in sd) else ""
    with torch.no_grad():
        for k in g.files:
            if k.startswith("w."):
                sd[prefix + k[2:]].copy_(torch.from_numpy(g[k]))
    clip.config._name_or_path = "openai/clip-vit-tiny-golden"          # -> the CLIP branch of _freeze_params; no head file -> nothing frozen
    model = SuperGuessr(base_model=clip, panorama=True, should_smooth_labels=True)
    assert all(p.requires_grad for p in clip.parameters())
    K = model.num_cells
    rng = np.random.default_rng(4242)
    W = rng.standard_normal((K, hs), dtype=np.float32) * np.float32(0.05)
    b = rng.standard_normal((K,), dtype=np.float32) * np.float32(0.1)
    x = rng.standard_normal((3, 4, 3, img, img), dtype=np.float32)
    labels = np.stack([rng.uniform(-180, 180, 3), rng.uniform(-90, 90, 3)], 1).astype(np.float32)
    with torch.no_grad():
        model.cell_layer.weight.copy_(torch.from_numpy(W)); model.cell_layer.bias.copy_(torch.from_numpy(b))
    model.train()
    from models.utils import haversine_matrix
    lab_t = torch.from_numpy(labels)
    clf = torch.argmin(haversine_matrix(lab_t, model.geocell_centroid_coords.data.t()), dim=-1)
    out = model(pixel_values=torch.from_numpy(x), labels=lab_t, labels_clf=clf)
    out.loss.backward()
    res = dict(seed=4242, W_checksum=np.float64(W.astype(np.float64).sum()), x_checksum=np.float64(x.astype(np.float64).sum()),
               labels=labels, labels_clf=clf.numpy(), loss=np.float32(out.loss.item()), embedding=out.embedding.detach().numpy(),
               preds_geocell=out.preds_geocell.numpy(), dW_rows=model.cell_layer.weight.grad.numpy()[clf.numpy()],
               dW_abs_sum=np.float64(model.cell_layer.weight.grad.abs().double().sum().item()))
    names = []
    for k, p in clip.named_parameters():
        name = k[len(prefix):] if prefix and k.startswith(prefix) else k
        G = p.grad.detach().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        names.append(name)
        if G.ndim >= 2 and G.size > 4096:
            fro, gr, lg = projections(name, G)
            res["gn." + name], res["gr." + name], res["gl." + name] = fro, gr, lg
        else:
            res["g." + name] = G
    res["param_names"] = np.asarray(names)
    np.savez_compressed(os.path.join(HERE, "clip_train.npz"), **res)
    print("clip_train.npz:", os.path.getsize(os.path.join(HERE, "clip_train.npz")) / 1024, "KB; loss", float(out.loss),
          "params", len(names))


if __name__ == "__main__":
    main()
