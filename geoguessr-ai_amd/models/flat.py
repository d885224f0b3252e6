"""Flat parameter storage shared by the encoder runtimes (TinyViT, CLIP): every parameter of the module tree is a view into ONE flat fp32
device buffer described by the runtime's tensor table (``gg_*_tensor_info``), gradients are views into one flat gradient buffer.  That
layout is what the RCCL gradient all-reduce and the fused AdamW kernel (``optim.AdamW``) operate on; the module nesting is rebuilt from the
dotted tensor names so that state-dict keys come out with the upstream library's names (timm / transformers)."""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch
import torch.nn as nn


class _Tree(nn.Module):
    """Anonymous container mirroring timm's module nesting so that state-dict keys come out with timm names."""

    def __iter__(self):
        return iter(self.children())

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, i):
        return list(self.children())[i]


class FlatStore(_Tree):
    """Needs from the subclass, before ``_register_table``: ``self.table`` (list of dicts name / offset / numel / shape / kind with kind
    0 = parameter, 1 = float buffer, 2 = int64 counter), ``self.param_floats``, ``self.buffer_floats``, ``self.num_counters``."""

    def _register_table(self, init: Callable[[str, tuple], torch.Tensor]):
        self._flat = torch.zeros(self.param_floats)
        self._flat_buf = torch.zeros(self.buffer_floats)
        self._counters = torch.zeros(self.num_counters, dtype=torch.int64)
        self._flat_grad: Optional[torch.Tensor] = None
        self._params: Dict[str, nn.Parameter] = {}
        for t in self.table:
            parent, leaf = self._walk(t["name"])
            if t["kind"] == 0:
                view = self._flat[t["offset"]:t["offset"] + t["numel"]].view(t["shape"])
                view.copy_(init(t["name"], t["shape"]))
                p = nn.Parameter(view)
                parent.register_parameter(leaf, p)
                self._params[t["name"]] = p
            elif t["kind"] == 1:
                view = self._flat_buf[t["offset"]:t["offset"] + t["numel"]].view(t["shape"])
                view.fill_(1.0 if leaf == "running_var" else 0.0)
                parent.register_buffer(leaf, view)
            else:
                parent.register_buffer(leaf, self._counters[t["offset"]])

    # -- module tree helpers ---------------------------------------------------------------------------
    def _walk(self, name: str):
        parts = name.split(".")
        mod = self
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, _Tree())
            mod = mod._modules[p]
        return mod, parts[-1]

    def _apply(self, fn, *a, **k):
        """``.to(device)`` moves every parameter separately; re-point them into fresh flat buffers afterwards."""
        super()._apply(fn, *a, **k)
        self._reflatten()
        return self

    def _reflatten(self):
        some = next(iter(self._params.values()))
        dev = some.device
        flat = torch.zeros(self.param_floats, device=dev)
        buf = torch.zeros(self.buffer_floats, device=dev)
        cnt = torch.zeros(self.num_counters, dtype=torch.int64, device=dev)
        grad = None
        if any(p.grad is not None for p in self._params.values()):
            grad = torch.zeros(self.param_floats, device=dev)
        for t in self.table:
            parent, leaf = self._walk(t["name"])
            sl = slice(t["offset"], t["offset"] + t["numel"])
            if t["kind"] == 0:
                p = parent._parameters[leaf]
                flat[sl].view(t["shape"]).copy_(p.data.to(torch.float32))
                p.data = flat[sl].view(t["shape"])
                if p.grad is not None:
                    grad[sl].view(t["shape"]).copy_(p.grad)
                    p.grad = grad[sl].view(t["shape"])
            elif t["kind"] == 1:
                buf[sl].view(t["shape"]).copy_(parent._buffers[leaf].to(torch.float32))
                parent._buffers[leaf] = buf[sl].view(t["shape"])
            else:
                cnt[t["offset"]] = parent._buffers[leaf].to(torch.int64)
                parent._buffers[leaf] = cnt[t["offset"]]
        self._flat, self._flat_buf, self._counters, self._flat_grad = flat, buf, cnt, grad
        self._wcache, self._wcache_version, self._ws = None, -1, {}

    # -- flat views used by the optimizer / all-reduce ---------------------------------------------------
    @property
    def flat_params(self) -> torch.Tensor:
        return self._flat

    def flat_grads(self) -> torch.Tensor:
        if self._flat_grad is None or self._flat_grad.device != self._flat.device:
            self._flat_grad = torch.zeros_like(self._flat)
        return self._flat_grad

    def attach_grads(self, zero_missing: bool = True):
        """Make every trainable parameter's ``.grad`` a view of the flat gradient buffer.  If a ``zero_grad(set_to_none)``
        dropped the views the buffer is zeroed first (its contents belonged to the previous step)."""
        fg = self.flat_grads()
        dropped = any(p.requires_grad and p.grad is None for p in self._params.values())
        if dropped and zero_missing:
            fg.zero_()
        for t in self.table:
            if t["kind"] != 0:
                continue
            p = self._params[t["name"]]
            if p.requires_grad:
                if p.grad is None or p.grad.data_ptr() != fg.data_ptr() + 4 * t["offset"]:
                    p.grad = fg[t["offset"]:t["offset"] + t["numel"]].view(t["shape"])
        return fg

    def trainable_mask(self) -> bytes:
        return bytes(int(t["kind"] == 0 and self._params[t["name"]].requires_grad) for t in self.table)

    def trainable_ranges(self):
        """Contiguous [start, end) float ranges of the flat buffer covering runs of trainable tensors."""
        ranges, cur = [], None
        for t in self.table:
            if t["kind"] != 0:
                continue
            tr = self._params[t["name"]].requires_grad
            end = t["offset"] + (t["numel"] + 7) // 8 * 8
            if tr:
                cur = [t["offset"], end] if cur is None else [cur[0], end]
            elif cur is not None:
                ranges.append(tuple(cur)); cur = None
        if cur is not None:
            ranges.append(tuple(cur))
        return ranges

    def mark_params_dirty(self, only: Optional[bytes] = None):
        """A raw-pointer writer (the fused AdamW kernel, a broadcast, a checkpoint load) changed parameters behind torch's version counters: the
        weight cache must be rebuilt.  ``only`` = the trainable mask the writer went by (one byte per tensor): then only those tensors' cached
        forms are rebuilt (masks of several writes are OR-ed); ``None`` = anything may have changed."""
        self._wcache_version = -1
        if only is None:
            self._dirty_only = None
            self._dirty_all = True
        elif not getattr(self, "_dirty_all", True):
            prev = getattr(self, "_dirty_only", None)
            self._dirty_only = bytes(only) if prev is None else bytes(a | b for a, b in zip(prev, only))

    def _param_version(self):
        """Changes whenever any parameter is written through torch (``torch.optim`` steps, ``load_state_dict``, ``p.copy_``):
        after ``_reflatten`` every Parameter is its own view with its own version counter, so the flat buffer's counter alone
        misses those writes.  Raw-pointer writers (the fused AdamW kernel) call ``mark_params_dirty`` instead."""
        return (self._flat._version, sum(p._version for p in self._params.values()), self._flat.data_ptr())

