"""CPU reference of one SuperGuessr-on-TinyViT training step (torch autograd over the oracle restatements).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Mirrors ``SuperGuessr.forward`` (models/super_guessr.py:309-395)
on top of ``oracle.tinyvit_ref.forward`` and the live step of ``main_coordinator_idun_s3.py:384-424``
(nearest-centroid labels, soft-CE, AdamW)."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import tinyvit_ref as R

LABEL_SMOOTHING_CONSTANT = 65.0


def haversine_matrix_t(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """models/utils.py:39-57 (x (N,2) lon/lat deg, y (2,M))."""
    x_rad, y_rad = torch.deg2rad(x), torch.deg2rad(y)
    delta = x_rad.unsqueeze(2) - y_rad
    p = torch.cos(x_rad[:, 1]).unsqueeze(1) * torch.cos(y_rad[1, :]).unsqueeze(0)
    a = torch.sin(delta[:, 1, :] / 2) ** 2 + p * torch.sin(delta[:, 0, :] / 2) ** 2
    c = 2 * torch.arcsin(torch.sqrt(a.clamp(max=1.0)))
    return (6378137.0 * c) / 1000


def head_loss(embedding: torch.Tensor, W: torch.Tensor, b: torch.Tensor, centroids: torch.Tensor, labels: torch.Tensor,
              smooth: bool = True, labels_clf: Optional[torch.Tensor] = None, emulate_bf16: bool = False):
    q = (lambda t: t.to(torch.bfloat16).float()) if emulate_bf16 else (lambda t: t)
    x = embedding.mean(dim=1) if embedding.dim() == 3 else embedding
    logits = F.linear(q(x), q(W), b)
    if smooth:
        d = haversine_matrix_t(labels, centroids.t())
        s = torch.exp(-(d - d.min(dim=-1, keepdim=True)[0]) / LABEL_SMOOTHING_CONSTANT)
        s = torch.nan_to_num(s, nan=0.0, posinf=0.0, neginf=0.0)
        s = s / s.sum(dim=-1, keepdim=True).clamp_min(1e-12)
        loss = -(s * F.log_softmax(logits, dim=-1)).sum(dim=-1).mean()
    else:
        loss = F.cross_entropy(logits, labels_clf)
    return loss, logits


def train_step(cfg: R.TinyVitConfig, state: Dict[str, torch.Tensor], W: torch.Tensor, b: torch.Tensor, centroids: torch.Tensor,
               pixel_values: torch.Tensor, labels: torch.Tensor, drop_masks: Optional[List[Optional[torch.Tensor]]] = None,
               trainable: Optional[List[str]] = None, emulate_bf16: bool = False, update_running: bool = False):
    """pixel_values (N,4,3,H,W) -> dict(loss, embedding (N,4,C), logits, grads{name: tensor})."""
    n, v = pixel_values.shape[:2]
    st = {}
    for k, t in state.items():
        if t.is_floating_point() and "running" not in k and (trainable is None or k in trainable):
            st[k] = t.clone().requires_grad_(True)
        else:
            st[k] = t.clone()
    Wg, bg = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    emb = R.forward(cfg, st, pixel_values.reshape(n * v, *pixel_values.shape[2:]), training=True, emulate_bf16=emulate_bf16,
                    drop_masks=drop_masks, update_running=update_running)
    emb = emb.view(n, v, -1)
    loss, logits = head_loss(emb, Wg, bg, centroids, labels, emulate_bf16=emulate_bf16)
    loss.backward()
    grads = {k: t.grad for k, t in st.items() if t.requires_grad and t.grad is not None}
    grads["cell_layer.weight"], grads["cell_layer.bias"] = Wg.grad, bg.grad
    return dict(loss=loss.detach(), embedding=emb.detach(), logits=logits.detach(), grads=grads, state=st)
