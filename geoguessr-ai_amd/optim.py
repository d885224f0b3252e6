"""AdamW + LR schedule of the reference's step (``main_coordinator_idun_s3.py:286-294,423-424,544``;
``training/train_eval_loop.py:188-190``) on the fused HIP kernel ``gg_adamw_step``.

The TinyViT backbone's parameters live in one flat fp32 buffer, so its trainable tensors are updated with one launch per
contiguous trainable range (two under ``freeze_all_but_last_stage``); other parameters (the geocell head) get one launch
each.  Under data parallelism the gradient average is folded into the kernel (``grad_scale = 1/world``) after a
sum all-reduce over RCCL -- gradients are the only thing exchanged."""
from __future__ import annotations

import math
import contextlib
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


class _NativeWork:
    """Work handle of the buckets sent through the C-ABI communicator (``async_op=True``): ``wait()`` orders the current compute stream
    behind the communicator's stream, the same contract as a ``torch.distributed`` work object on the GPU."""

    def __init__(self, nc):
        self._nc = nc

    def wait(self):
        self._nc.wait()
        return True

    def is_completed(self):
        return not self._nc._pending


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
        self._flat_ids = flat_ids
        self.loose: List[torch.nn.Parameter] = [p for p in model.parameters() if p.requires_grad and id(p) not in flat_ids]
        self.state = {}
        self._inflight, self._covered = {}, {}

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

    # ---- data-parallel exchange (the reference gets all of it from Accelerate -> DistributedDataParallel,
    #      training/train_eval_loop.py:184-187,200-202,234) ------------------------------------------------------------
    @staticmethod
    def _world() -> int:
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

    def _native(self):
        """The gg_comm (RCCL behind the C-ABI) communicator when GG_NATIVE_COMM=1 and the tensors live on the GPU; else None
        (torch.distributed carries the same collectives)."""
        from . import comm as _comm
        if not _comm.enabled() or not any(bb.flat_params.is_cuda for bb in self.backbones) and not any(p.is_cuda for p in self.loose):
            return None
        if getattr(self, "_ncomm", None) is None:
            self._ncomm = _comm.NativeComm()
        return self._ncomm

    def _bcast(self, t: torch.Tensor, src: int):
        nc = self._native()
        if nc is not None and nc.accepts(t, any_dtype=True):
            nc.broadcast_(t, src)
        else:
            dist.broadcast(t, src)

    def broadcast_params(self, src: int = 0):
        """DDP construction semantics: every rank starts from rank ``src``'s parameters and buffers."""
        if self._world() == 1:
            return
        for bb in self.backbones:
            self._bcast(bb.flat_params, src)
            bb.mark_params_dirty()
        for p in self.model.parameters():
            if id(p) not in self._flat_ids:
                self._bcast(p.data, src)
        self.broadcast_buffers(src)
        if hasattr(self.model, "mark_params_dirty"):
            self.model.mark_params_dirty()

    def broadcast_buffers(self, src: int = 0):
        """``DistributedDataParallel(broadcast_buffers=True)`` (the default the reference runs with): BatchNorm running
        statistics and ``num_batches_tracked`` of rank ``src`` overwrite every other rank's before a training forward."""
        if self._world() == 1:
            return
        for bb in self.backbones:
            for t in (bb._flat_buf, bb._counters):
                if t.numel() > 0:          # (the CLIP tower has no buffers)
                    self._bcast(t, src)
        nc = self._native()
        if nc is not None:
            nc.wait()

    def _launch(self, key, tensor):
        if key in self._inflight or tensor is None or tensor.numel() == 0:
            return
        nc = self._native()
        if nc is not None and nc.accepts(tensor):
            nc.allreduce_sum_(tensor)                   # on the communicator's own stream, behind an event on the compute stream
            self._inflight[key] = None
        else:
            self._inflight[key] = dist.all_reduce(tensor, op=dist.ReduceOp.SUM, async_op=True)

    def _bucket_ready(self, bb_index: int, lo: int, hi: int):
        """Called (through the backward pass's stage callback) when flat gradient floats [lo, hi) of a backbone are final."""
        bb = self.backbones[bb_index]
        fg = bb.flat_grads()
        for s, e in bb.trainable_ranges():
            a, b = max(s, lo), min(e, hi)
            if a < b:
                self._launch(("bb", bb_index, a, b), fg[a:b])
                self._covered.setdefault(bb_index, []).append((a, b))

    @contextlib.contextmanager
    def overlap_allreduce(self, enabled: bool = True):
        """Inside this context, gradient buckets are all-reduced (async, on RCCL's own stream) as soon as the backward pass
        has finished them: the geocell head's weight/bias right after the head's backward, the encoder's last stage while
        the earlier stages are still running.  ``allreduce_grads()`` afterwards sends what is left and waits for all of it."""
        if not enabled or self._world() == 1:
            yield
            return
        handles = []
        for p in self.loose:
            handles.append(p.register_post_accumulate_grad_hook(lambda q: self._launch(("loose", id(q)), q.grad)))
        for i, bb in enumerate(self.backbones):
            bb._grad_ready_hook = (lambda lo, hi, i=i: self._bucket_ready(i, lo, hi))
        try:
            yield
        finally:
            for h in handles:
                h.remove()
            for bb in self.backbones:
                bb._grad_ready_hook = None

    def allreduce_grads(self, async_op: bool = False):
        """Sum every trainable gradient over the ranks (RCCL over xGMI under backend "nccl"; gloo in the CPU tests).  Buckets
        already launched by ``overlap_allreduce`` are not sent twice; returns after the current stream is ordered behind
        every reduction (``async_op=True``: returns the work handles instead)."""
        if self._world() == 1:
            return []
        for i, bb in enumerate(self.backbones):
            fg = bb.flat_grads()
            done = sorted(self._covered.get(i, []))
            for s, e in bb.trainable_ranges():
                cur = s
                for a, b in done:
                    if b <= cur or a >= e:
                        continue
                    if a > cur:
                        self._launch(("bb", i, cur, a), fg[cur:a])
                    cur = max(cur, b)
                if cur < e:
                    self._launch(("bb", i, cur, e), fg[cur:e])
        for p in self.loose:
            if p.grad is not None:
                self._launch(("loose", id(p)), p.grad)
        works = [w for w in self._inflight.values() if w is not None]
        native_pending = any(w is None for w in self._inflight.values())
        self._inflight, self._covered = {}, {}
        nc = self._native() if native_pending else None
        if async_op:
            if nc is not None:                          # buckets that went through gg_comm_*: their handle orders the compute stream behind them
                works.append(_NativeWork(nc))
            return works
        for w in works:
            w.wait()
        if nc is not None:
            nc.wait()                                   # the compute stream is ordered behind the native collectives
        return []

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
            bb.mark_params_dirty(only=bb.trainable_mask())          # only these tensors' cached forms need rebuilding
        for p in self.loose:
            if p.grad is None:
                continue
            m, v = self._st(id(p), p.data)
            ops.adamw_step(p.data.view(-1), p.grad.contiguous().view(-1), m.view(-1), v.view(-1), **kw)
        if hasattr(self.model, "mark_params_dirty"):
            try:
                self.model.mark_params_dirty(backbone=False)          # the head's cached weight copies; the backbones were marked above
            except TypeError:
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
