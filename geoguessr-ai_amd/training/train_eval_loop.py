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


def _shard(n: int, shuffle: bool, seed: int, rank: int, world: int) -> torch.Tensor:
    """Row indices of this rank.  Same contract as ``torch.utils.data.DistributedSampler`` (what Accelerate wraps the
    reference's DataLoader in, training/train_eval_loop.py:200-202): the order is padded by wrapping around to a multiple of
    ``world`` and dealt out strided, so EVERY rank gets exactly ceil(n / world) rows -- equal batch counts, hence equal
    numbers of gradient all-reduces (a short rank would deadlock the collective)."""
    order = torch.randperm(n, generator=torch.Generator().manual_seed(seed)) if shuffle else torch.arange(n)
    if world <= 1:
        return order
    per = (n + world - 1) // world
    total = per * world
    if total > n:
        reps = (total + n - 1) // n
        order = order.repeat(reps)[:total]
    return order[rank:total:world]


def _num_batches(n: int, batch_size: int, world: int) -> int:
    per = (n + world - 1) // world if world > 1 else n
    return (per + batch_size - 1) // batch_size


def _batches(ds, batch_size: int, shuffle: bool, seed: int, rank: int, world: int):
    mine = _shard(_num_rows(ds), shuffle, seed, rank, world)
    for i in range(0, len(mine), batch_size):
        yield _take(ds, mine[i:i + batch_size])


MODEL_KEYS = ("pixel_values", "embedding", "labels", "labels_clf", "index")


def _gather_rows(arrays, rank: int, world: int):
    """Concatenate per-rank result lists in rank order (each rank evaluated one contiguous slice of the dataset)."""
    if world <= 1:
        return arrays
    box = [None] * world
    dist.all_gather_object(box, arrays)
    return [np.concatenate([b[i] for b in box], 0) for i in range(len(arrays))]


def evaluate_model(model, dataset, metrics: Callable, train_args, refiner=None, writer=None, step: int = 0) -> float:
    """Reference contract (:37-155): returns ``-metrics(results)["Geocell_accuracy"]`` with
    ``results = (preds_LLH, preds_geocell, top5_geocells, labels_lla, labels_cell)`` over the WHOLE validation set in dataset
    order.  Under data parallelism each rank evaluates one contiguous slice and the slices are all-gathered (the reference
    has every rank evaluate everything, :62-68 -- same numbers, world times the work)."""
    logger.warning("Starting evaluation ...")
    rank, world = _rank_world()
    model.eval()
    if refiner is not None:
        refiner.eval()
    n = _num_rows(dataset)
    per = (n + world - 1) // world
    lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
    ebs = train_args.per_device_eval_batch_size
    preds, cells, top5c, top5p, lab_lla, lab_cell = [], [], [], [], [], []
    loss_sum, n_seen = 0.0, 0
    with torch.no_grad():
        for s0 in range(lo, hi, ebs):
            data = _take(dataset, torch.arange(s0, min(hi, s0 + ebs)))
            outputs = model(**{k: v for k, v in data.items() if k in MODEL_KEYS})
            bs = _num_rows(data)
            if outputs.loss is not None:
                loss_sum += float(outputs.loss) * bs          # weighted by rows (the reference uses len(dict), C11)
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
    cat = lambda xs, shape, dt: np.concatenate(xs, 0) if xs else np.zeros(shape, dt)
    k = int(getattr(model, "num_candidates", 5))
    local = [cat(preds, (0, 2), np.float32), cat(cells, (0,), np.int64), cat(top5c, (0, k), np.int64),
             cat(lab_lla, (0, 2), np.float32), cat(lab_cell, (0,), np.int64), np.asarray([loss_sum, n_seen], np.float64)[None]]
    full = _gather_rows(local, rank, world)
    results = tuple(full[:5])
    eval_dict = metrics(results)
    tot = full[5].sum(0)
    if writer is not None and tot[1] > 0:
        writer("Loss/val", float(tot[0] / tot[1]), step)
        for key, v in eval_dict.items():
            writer(key, v, step)
    model.train()
    logger.warning("Back to training ...")
    return -eval_dict["Geocell_accuracy"]


def train_model(loaded_model: Any, dataset, on_embeddings: bool, train_args, metrics: Callable, patience: int = None,
                should_profile: bool = True, refiner=None, log_fn: Optional[Callable] = None, save_path: str = CURRENT_SAVE_PATH,
                broadcast_buffers: bool = True):
    """Reference contract (:158-274; ``should_profile`` defaults to True as there, :165: the torch.profiler schedule of ``generate_profiler`` writes its
    traces under ``runs/profile``).  DDP start-up semantics of ``accelerator.prepare`` (:200-202) are kept: parameters and
    buffers are broadcast from rank 0 once, BatchNorm running statistics are re-broadcast from rank 0 before every training
    forward (``DistributedDataParallel(broadcast_buffers=True)``, the default), gradients are summed over ranks before each
    optimizer step (overlapped with the encoder's backward pass) and averaged inside the AdamW kernel."""
    rank, world = _rank_world()
    model = loaded_model
    optimizer = AdamW(model, lr=train_args.learning_rate)         # torch defaults: betas (.9,.999), eps 1e-8, wd 1e-2
    optimizer.broadcast_params()                                  # no-op for world == 1
    grad_acc_steps = getattr(train_args, "gradient_accumulation_steps", None) or 1
    logging_steps = getattr(train_args, "logging_steps", 0) or 0
    bs = train_args.per_device_train_batch_size
    steps = _num_batches(_num_rows(dataset["train"]), bs, world)  # == len(train_data) after accelerator.prepare: identical on every rank
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
                last = i % grad_acc_steps == (grad_acc_steps - 1) or (i + 1) == steps
                if broadcast_buffers:
                    optimizer.broadcast_buffers()
                with optimizer.overlap_allreduce(enabled=last):      # gradient buckets leave as the backward pass completes them
                    output = model(**{k: v for k, v in data.items() if k in MODEL_KEYS})
                    output.loss.backward()
                if last:
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
