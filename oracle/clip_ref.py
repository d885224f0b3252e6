"""CPU restatement (plain torch fp32) of the CLIP vision tower as the reference uses it:
``clip_model.base_model(pixel_values).last_hidden_state.mean(dim=1)``
(``pretrain/clip_embedder.py:63-65``; ``models/super_guessr.py:323-325``).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  PINNED against
``transformers.CLIPVisionModel`` (the third-party library holding the arithmetic;
reference pins transformers 4.57.1 in ``uv.lock``, this image has 5.x -- same math) on the
tiny config committed under ``tests/golden/clip_tiny.npz``.

State-dict keys follow HF ``CLIPVisionModel`` *without* the ``vision_model.`` prefix
(transformers 5 is flat; 4.57 nests under ``vision_model.`` -- the host loader strips it).
``last_hidden_state`` is the encoder output WITHOUT ``post_layernorm``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


@dataclass
class ClipVisionConfig:
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    image_size: int = 224
    patch_size: int = 32
    layer_norm_eps: float = 1e-5

    @property
    def num_tokens(self) -> int:
        return (self.image_size // self.patch_size) ** 2 + 1


def param_spec(cfg: ClipVisionConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    D, I, P = cfg.hidden_size, cfg.intermediate_size, cfg.patch_size
    spec = [
        ("embeddings.class_embedding", (D,)),
        ("embeddings.patch_embedding.weight", (D, 3, P, P)),
        ("embeddings.position_embedding.weight", (cfg.num_tokens, D)),
        ("pre_layrnorm.weight", (D,)), ("pre_layrnorm.bias", (D,)),
    ]
    for i in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{i}"
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            spec += [(f"{p}.self_attn.{n}.weight", (D, D)), (f"{p}.self_attn.{n}.bias", (D,))]
        spec += [(f"{p}.layer_norm1.weight", (D,)), (f"{p}.layer_norm1.bias", (D,)),
                 (f"{p}.mlp.fc1.weight", (I, D)), (f"{p}.mlp.fc1.bias", (I,)),
                 (f"{p}.mlp.fc2.weight", (D, I)), (f"{p}.mlp.fc2.bias", (D,)),
                 (f"{p}.layer_norm2.weight", (D,)), (f"{p}.layer_norm2.bias", (D,))]
    spec += [("post_layernorm.weight", (D,)), ("post_layernorm.bias", (D,))]
    return spec


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def forward(cfg: ClipVisionConfig, st: Dict[str, torch.Tensor], x: torch.Tensor,
            emulate_bf16: bool = False) -> torch.Tensor:
    """(B,3,H,W) -> mean over all T tokens of last_hidden_state: (B, D)."""
    q = (lambda t: t.to(torch.bfloat16).to(torch.float32)) if emulate_bf16 else (lambda t: t)
    B = x.shape[0]
    D, nh = cfg.hidden_size, cfg.num_attention_heads
    hd = D // nh
    eps = cfg.layer_norm_eps
    pe = F.conv2d(q(x), q(st["embeddings.patch_embedding.weight"]), None, cfg.patch_size)
    pe = pe.flatten(2).transpose(1, 2)                                  # (B, T-1, D)
    cls = st["embeddings.class_embedding"].expand(B, 1, D)
    h = torch.cat([cls, pe], dim=1) + st["embeddings.position_embedding.weight"][None]
    h = q(h)
    h = q(F.layer_norm(h, (D,), st["pre_layrnorm.weight"], st["pre_layrnorm.bias"], eps))
    for i in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{i}"
        a = q(F.layer_norm(h, (D,), st[f"{p}.layer_norm1.weight"], st[f"{p}.layer_norm1.bias"], eps))
        qq = q(F.linear(a, q(st[f"{p}.self_attn.q_proj.weight"]), st[f"{p}.self_attn.q_proj.bias"]))
        kk = q(F.linear(a, q(st[f"{p}.self_attn.k_proj.weight"]), st[f"{p}.self_attn.k_proj.bias"]))
        vv = q(F.linear(a, q(st[f"{p}.self_attn.v_proj.weight"]), st[f"{p}.self_attn.v_proj.bias"]))
        T = h.shape[1]
        qq, kk, vv = (t.view(B, T, nh, hd).transpose(1, 2) for t in (qq, kk, vv))
        s = (qq @ kk.transpose(-2, -1)) * (hd ** -0.5)
        if emulate_bf16:
            m = s.amax(-1, keepdim=True)
            pexp = torch.exp(s - m)
            o = (q(pexp) @ vv) / pexp.sum(-1, keepdim=True)
        else:
            o = s.softmax(-1) @ vv
        o = q(o.transpose(1, 2).reshape(B, T, D))
        h = q(h + F.linear(o, q(st[f"{p}.self_attn.out_proj.weight"]), st[f"{p}.self_attn.out_proj.bias"]))
        m_ = q(F.layer_norm(h, (D,), st[f"{p}.layer_norm2.weight"], st[f"{p}.layer_norm2.bias"], eps))
        m_ = q(quick_gelu(q(F.linear(m_, q(st[f"{p}.mlp.fc1.weight"]), st[f"{p}.mlp.fc1.bias"]))))
        h = q(h + F.linear(m_, q(st[f"{p}.mlp.fc2.weight"]), st[f"{p}.mlp.fc2.bias"]))
    return h.mean(dim=1)


def train_step(cfg: ClipVisionConfig, st: Dict[str, torch.Tensor], W: torch.Tensor, b: torch.Tensor, centroids: torch.Tensor,
               pixel_values: torch.Tensor, labels: torch.Tensor, trainable=None, emulate_bf16: bool = False):
    """One SuperGuessr-on-CLIP training forward + backward (models/super_guessr.py:309-383 with a CLIPVisionModel base: token mean of
    last_hidden_state per view, 4-view mean, Linear, haversine-smoothed soft CE) by torch autograd over ``forward``.  pixel_values
    (N,4,3,H,W) -> dict(loss, embedding (N,4,D), grads{name}).  PINNED against the reference run in ``tests/golden/clip_train.npz``."""
    from . import step_ref as S
    n, v = pixel_values.shape[:2]
    stg = {k: (t.clone().requires_grad_(True) if (t.is_floating_point() and (trainable is None or k in trainable)) else t.clone()) for k, t in st.items()}
    Wg, bg = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    emb = forward(cfg, stg, pixel_values.reshape(n * v, *pixel_values.shape[2:]), emulate_bf16=emulate_bf16).view(n, v, -1)
    loss, logits = S.head_loss(emb, Wg, bg, centroids, labels, emulate_bf16=emulate_bf16)
    loss.backward()
    grads = {k: t.grad for k, t in stg.items() if t.requires_grad and t.grad is not None}
    grads["cell_layer.weight"], grads["cell_layer.bias"] = Wg.grad, bg.grad
    return dict(loss=loss.detach(), embedding=emb.detach(), logits=logits.detach(), grads=grads)
