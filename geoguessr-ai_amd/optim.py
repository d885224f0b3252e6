"""AdamW + LR schedule of the reference's step (``main_coordinator_idun_s3.py:286-294,423-424,544``;
``training/train_eval_loop.py:188-190``) on the fused HIP kernel ``gg_adamw_step``.

The TinyViT backbone's parameters live in one flat fp32 buffer, so its trainable tensors are updated with one launch per
contiguous trainable range (two under ``freeze_all_but_last_stage``); other parameters (the geocell head) get one launch
each.  Under data parallelism the gradient average is folded into the kernel (``grad_scale = 1/world``) after a
sum all-reduce over RCCL -- gradients are the only thing exchanged."""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.distributed as dist

from . import ops


def cosine_warm_restarts_lr(epoch: float, base_lr: float, T_0: int = 10, T_mult: int = 2, eta_min: float = 1e-6) -> float:
    """Closed form of ``CosineAnnealingWarmRestarts(T_0, T_mult, eta_min).step(epoch)``."""
    if epoch >= T_0:
        if T_mult == 1:
            t_cur, t_i = epoch % T_0, T_0
        else:
            n = int(math.log(epoch / T_0 * (T_mult - 1) + 1, T_mult))
            t_cur = epoch - T_0 * (T_mult ** n - 1) / (T_mult - 1)
            t_i = T_0 * T_mult ** n
    else:
        t_cur, t_i = epoch, T_0
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t_cur / t_i)) / 2


class AdamW:
    """``torch.optim.AdamW(model.parameters(), lr, betas, weight_decay)`` semantics (decoupled decay on every trainable
    parameter, bias-corrected) for a SuperGuessr / TinyViTAdapter built on flat storage."""

    def __init__(self, model: torch.nn.Module, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.01):
        self.model = model
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        self.step_count = 0
        self.backbones = [m for m in model.modules() if hasattr(m, "trainable_ranges") and hasattr(m, "flat_params")]
        flat_ids = set()
        for bb in self.backbones:
            flat_ids.update(id(p) for p in bb._params.values())
        self.loose: List[torch.nn.Parameter] = [p for p in model.parameters() if p.requires_grad and id(p) not in flat_ids]
        self.state = {}

    def _st(self, key, like):
        if key not in self.state or self.state[key][0].device != like.device:
            self.state[key] = (torch.zeros_like(like), torch.zeros_like(like))
        return self.state[key]

    def grad_buffers(self) -> List[torch.Tensor]:
        """Every gradient tensor an all-reduce must cover (flat trainable ranges + loose parameter grads)."""
        out = []
        for bb in self.backbones:
            fg = bb.flat_grads()
            out += [fg[s:e] for s, e in bb.trainable_ranges()]
        out += [p.grad for p in self.loose if p.grad is not None]
        return out

    def allreduce_grads(self, async_op: bool = False):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return []
        return [dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=async_op) for g in self.grad_buffers()]

    def step(self, grad_scale: Optional[float] = None):
        g = self.param_groups[0]
        if grad_scale is None:
            grad_scale = 1.0
            if dist.is_available() and dist.is_initialized():
                grad_scale = 1.0 / dist.get_world_size()
        self.step_count += 1
        kw = dict(step=self.step_count, lr=g["lr"], beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"],
                  weight_decay=g["weight_decay"], grad_scale=grad_scale)
        for i, bb in enumerate(self.backbones):
            fp, fg = bb.flat_params, bb.flat_grads()
            m, v = self._st(("bb", i), fp)
            for s, e in bb.trainable_ranges():
                ops.adamw_step(fp[s:e], fg[s:e], m[s:e], v[s:e], **kw)
            bb.mark_params_dirty()
        for p in self.loose:
            if p.grad is None:
                continue
            m, v = self._st(id(p), p.data)
            ops.adamw_step(p.data.view(-1), p.grad.contiguous().view(-1), m.view(-1), v.view(-1), **kw)
        if hasattr(self.model, "mark_params_dirty"):
            self.model.mark_params_dirty()

    def zero_grad(self, set_to_none: bool = True):
        for bb in self.backbones:
            if bb._flat_grad is not None:
                bb._flat_grad.zero_()
        for p in self.loose:
            p.grad = None

    def state_dict(self):
        return dict(step=self.step_count, param_groups=self.param_groups,
                    state={str(k): (m.cpu(), v.cpu()) for k, (m, v) in self.state.items()})
