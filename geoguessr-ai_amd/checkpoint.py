"""Checkpoint compatibility with the reference's training coordinator.

* ``make_state``: the dict ``main_coordinator_idun_s3.py:562-573`` saves (``epoch, global_step, model_state_dict,
  optimizer_state_dict, scheduler_state_dict, best_value, monitored_value, config``); model keys are the reference's
  (``base_model.backbone.<timm name>``, ``cell_layer.*``), the optimizer dict is ``torch.optim.AdamW``'s layout.
* ``CheckpointKeeper``: ``last.pt`` every epoch, ``epoch_{epoch:04d}_{value:.6f}.pt`` only while among the top
  ``keep_last_n`` by the monitored value (min or max mode), worse files pruned, ``best.pt`` on improvement
  (``main_coordinator_idun_s3.py:575-712``).
* ``load_model_state``: unwrap ``model_state_dict``, keep only keys whose shapes match, ``strict=False``
  (``inference.py:126-156``) -- reference-trained weights drop in and vice versa.
* ``adamw_state_to_torch`` / ``adamw_state_from_torch``: this package's flat-buffer AdamW <-> ``torch.optim.AdamW.state_dict()``."""
import os
from typing import Any, Dict, List, Optional, Tuple

import torch


def make_state(model, optimizer, scheduler, epoch: int, global_step: int, best_value: float, monitored_value: float,
               config: Optional[dict] = None) -> Dict[str, Any]:
    opt_sd = adamw_state_to_torch(optimizer) if hasattr(optimizer, "backbones") else optimizer.state_dict()
    state = {"epoch": epoch, "global_step": global_step, "model_state_dict": model.state_dict(), "optimizer_state_dict": opt_sd,
             "scheduler_state_dict": scheduler.state_dict() if scheduler is not None else {}, "best_value": best_value,
             "monitored_value": monitored_value, "config": dict(config or {})}
    # extra key (the reference's loaders ignore unknown keys): DropPath stream position of every flat-storage backbone, so a resumed run continues the
    # mask sequence instead of replaying it from the start
    dp = {n: m.drop_path_state() for n, m in model.named_modules() if hasattr(m, "drop_path_state")}
    if dp:
        state["gg_drop_path_state"] = dp
    return state


def restore_drop_path_state(model, state: Dict[str, Any]) -> None:
    """Counterpart of the ``gg_drop_path_state`` key written by :func:`make_state` (no-op for checkpoints without it, e.g. the reference's own)."""
    dp = (state or {}).get("gg_drop_path_state") or {}
    mods = dict(model.named_modules())
    for n, st in dp.items():
        if n in mods and hasattr(mods[n], "load_drop_path_state"):
            mods[n].load_drop_path_state(st)


def _value_from_filename(name: str) -> Optional[float]:
    try:
        return float(name[:-3].split("_")[-1])
    except Exception:
        return None


class CheckpointKeeper:
    def __init__(self, checkpoint_dir: str, keep_last_n: int = 3, save_every_epochs: int = 1, monitor_mode: str = "min",
                 save_fn=torch.save):
        self.dir = checkpoint_dir
        self.k = max(0, int(keep_last_n))
        self.every = max(1, int(save_every_epochs))
        self.is_min = monitor_mode == "min"
        self.best_value = float("inf") if self.is_min else float("-inf")
        self.save = save_fn
        os.makedirs(self.dir, exist_ok=True)

    def _existing(self) -> List[Tuple[str, float]]:
        out = []
        for f in os.listdir(self.dir):
            if not (f.startswith("epoch_") and f.endswith(".pt")):
                continue
            v = _value_from_filename(f)
            if v is None:       # unknown value: worst side, pruned first
                v = float("inf") if self.is_min else float("-inf")
            out.append((f, v))
        return out

    def update(self, state: Dict[str, Any], epoch: int, current_value: float) -> Dict[str, Any]:
        """Apply one epoch's saving policy; returns {"last": path, "epoch": path|None, "best": path|None, "improved": bool}."""
        os.makedirs(self.dir, exist_ok=True)
        res = {"last": os.path.join(self.dir, "last.pt"), "epoch": None, "best": None, "improved": False}
        self.save(state, res["last"])
        if (epoch + 1) % self.every == 0:
            existing = self._existing()
            if self.k == 0:
                should = False
            elif len(existing) < self.k:
                should = True
            else:
                worst = max(v for _, v in existing) if self.is_min else min(v for _, v in existing)
                should = current_value < worst if self.is_min else current_value > worst
            if should:
                res["epoch"] = os.path.join(self.dir, f"epoch_{epoch:04d}_{current_value:.6f}.pt")
                self.save(state, res["epoch"])
                existing = self._existing()
                existing.sort(key=lambda t: t[1], reverse=not self.is_min)       # best first
                for f, _ in existing[self.k:]:
                    try:
                        os.remove(os.path.join(self.dir, f))
                    except FileNotFoundError:
                        pass
                if not os.path.exists(res["epoch"]):
                    res["epoch"] = None
        improved = current_value < self.best_value if self.is_min else current_value > self.best_value
        if improved:
            self.best_value = current_value
            state["best_value"] = self.best_value
            res["best"] = os.path.join(self.dir, "best.pt")
            self.save(state, res["best"])
        res["improved"] = improved
        return res


def load_model_state(model: torch.nn.Module, path_or_state, map_location="cpu") -> Dict[str, List[str]]:
    """inference.py:126-156: unwrap ``model_state_dict``, keep shape-matching keys, load non-strictly."""
    raw = torch.load(path_or_state, map_location=map_location, weights_only=False) if isinstance(path_or_state, (str, os.PathLike)) \
        else path_or_state
    sd = raw.get("model_state_dict", raw) if isinstance(raw, dict) else raw
    own = model.state_dict()
    kept = {k: v for k, v in sd.items() if k in own and tuple(own[k].shape) == tuple(v.shape)}
    skipped = [k for k in sd if k not in kept]
    model.load_state_dict(kept, strict=False)
    if hasattr(model, "mark_params_dirty"):
        model.mark_params_dirty()
    if isinstance(raw, dict):
        restore_drop_path_state(model, raw)          # (a checkpoint written by make_state: the DropPath stream continues where it stopped)
    return {"loaded": sorted(kept), "skipped": sorted(skipped), "missing": sorted(k for k in own if k not in kept)}


def _param_list(optimizer) -> List[torch.nn.Parameter]:
    return list(optimizer.model.parameters())


def adamw_state_to_torch(optimizer) -> Dict[str, Any]:
    """This package's AdamW (moments in flat buffers) -> ``torch.optim.AdamW(model.parameters()).state_dict()`` layout."""
    params = _param_list(optimizer)
    g = optimizer.param_groups[0]
    state: Dict[int, Dict[str, torch.Tensor]] = {}
    flat_index = {}
    for bi, bb in enumerate(optimizer.backbones):
        for t in bb.table:
            if t["kind"] == 0:
                flat_index[id(bb._params[t["name"]])] = (bi, t["offset"], t["numel"])
    for i, p in enumerate(params):
        if not p.requires_grad or optimizer.step_count == 0:
            continue
        if id(p) in flat_index:
            bi, off, n = flat_index[id(p)]
            mv = optimizer.state.get(("bb", bi))
            if mv is None:
                continue
            m, v = (mv[0][off:off + n].view(p.shape).detach().cpu().clone(), mv[1][off:off + n].view(p.shape).detach().cpu().clone())
        else:
            mv = optimizer.state.get(id(p))
            if mv is None:
                continue
            m, v = mv[0].detach().cpu().clone(), mv[1].detach().cpu().clone()
        state[i] = {"step": torch.tensor(float(optimizer.step_count)), "exp_avg": m, "exp_avg_sq": v}
    group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": g["weight_decay"], "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(len(params)))}
    return {"state": state, "param_groups": [group]}


def adamw_state_from_torch(optimizer, sd: Dict[str, Any]) -> None:
    """Load a ``torch.optim.AdamW`` state dict (reference checkpoint) into this package's AdamW."""
    params = _param_list(optimizer)
    g = sd["param_groups"][0]
    optimizer.param_groups[0].update(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g["weight_decay"])
    order = g["params"]
    flat_index = {}
    for bi, bb in enumerate(optimizer.backbones):
        for t in bb.table:
            if t["kind"] == 0:
                flat_index[id(bb._params[t["name"]])] = (bi, t["offset"], t["numel"])
    steps = []
    for pos, idx in enumerate(order):
        st = sd["state"].get(idx, sd["state"].get(str(idx)))
        if st is None or pos >= len(params):
            continue
        p = params[pos]
        steps.append(int(float(st["step"])))
        if id(p) in flat_index:
            bi, off, n = flat_index[id(p)]
            bb = optimizer.backbones[bi]
            m, v = optimizer._st(("bb", bi), bb.flat_params)
            m[off:off + n].copy_(st["exp_avg"].reshape(-1).to(m.device)); v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(v.device))
        else:
            m, v = optimizer._st(id(p), p.data)
            m.copy_(st["exp_avg"].to(m.device)); v.copy_(st["exp_avg_sq"].to(v.device))
    if steps:
        optimizer.step_count = max(steps)
