"""GeoGuessr scoring of predicted coordinates -- the "next" row f1 (reference run_benchmark.py:25-117 and the ``metrics``
callable contract of training/train_eval_loop.py:130-137,155).

* :func:`score_batch`      -- ``haversine_np`` + ``geoguessr_score_from_distance`` for a whole batch in one launch
  (``gg_geoguessr_score``: fp64 haversine, ``int(round(clamp(5000 * exp(-d / 1492.7), 0, 5000)))`` -> int32).
* :func:`compute_summary`  -- ``_compute_summary_from_data`` (run_benchmark.py:67-117): same keys, same arithmetic order
  (plain sums / n, numpy median) so the numbers match the reference's JSON summary.
* :func:`geocell_metrics`  -- a ``metrics(results)`` callable for ``evaluate_model`` / ``train_model``: the 5-tuple
  ``(preds, preds_geocells, top5_geocells, labels_lla, labels_cell)`` -> ``{"Geocell_accuracy": ..., ...}``.
"""
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import ops


def score_batch(pred_llh: torch.Tensor, true_llh: torch.Tensor):
    """(distance_km float64 (N,), score int32 (N,)) on the GPU; rows are (lon, lat) in degrees like the reference's arrays.  float64 rows stay
    float64 (``haversine_np`` runs on float64 arrays, run_benchmark.py:28-47: the integer scores match it bit for bit only then); the model's
    own float32 predictions are read as float32, which is exact."""
    return ops.geoguessr_score(pred_llh, true_llh)


def compute_summary(distance_km: Sequence[float], score: Sequence[float], top1_prob: Optional[Sequence[float]] = None, strict: bool = False) -> Dict[str, float]:
    """run_benchmark.py:67-117.  ``top1_prob[i] < 0`` (or ``top1_prob=None``) = the sample has no top-5 list (counts as 0.0)."""
    d = np.asarray(torch.as_tensor(distance_km).cpu() if torch.is_tensor(distance_km) else distance_km, np.float64)
    s = np.asarray(torch.as_tensor(score).cpu() if torch.is_tensor(score) else score, np.float64)
    if d.ndim != 1 or d.size == 0:
        raise ValueError("Expected a non-empty list of samples for summary computation")
    p = np.zeros_like(d) if top1_prob is None else np.asarray(torch.as_tensor(top1_prob).cpu() if torch.is_tensor(top1_prob) else top1_prob,
                                                              np.float64)
    n = n_total = d.size
    # gg_geoguessr_score marks a non-finite coordinate pair with distance NaN / score -1.  The reference's haversine_np propagates such a NaN into every
    # mean and carries on (a validation epoch never aborts a run): do the same for the run, but keep the sentinels out of the averages -- they are
    # counted and reported (a warning), the means are over the valid samples (``strict=True`` raises instead).  ``num_samples`` is always the TOTAL, as in the
    # reference; ``num_valid`` / ``num_invalid`` say how many the averages are over
    bad = ~np.isfinite(d) | (s < 0)
    n_bad = int(bad.sum())
    if n_bad:
        msg = f"compute_summary: {n_bad} of {n} samples have a non-finite distance / sentinel score (first at index {int(np.argmax(bad))})"
        if strict:
            raise ValueError(msg)
        import warnings
        warnings.warn(msg + "; they are left out of the averages")
        if n_bad == n:
            return {"num_samples": n_total, "num_valid": 0, "num_invalid": n_bad, "avg_distance_km": float("nan"), "median_distance_km": float("nan"), "avg_top1_prob": float("nan"),
                    "avg_score": float("nan")}
        d, s, p = d[~bad], s[~bad], p[~bad]
        n = d.size
    total_distance = total_score = total_top = 0.0
    for i in range(n):                       # the reference accumulates sample by sample in Python floats: keep its summation order
        total_distance += float(d[i]); total_score += float(s[i]); total_top += float(p[i]) if p[i] >= 0 else 0.0
    out = {"num_samples": n_total, "avg_distance_km": total_distance / n, "median_distance_km": float(np.median(d)),
           "avg_top1_prob": total_top / n, "avg_score": total_score / n}
    if n_bad:
        out["num_valid"], out["num_invalid"] = n, n_bad
    return out


def geocell_metrics(results) -> Dict[str, float]:
    """``metrics`` callable: ``results = (preds (N,2) lon/lat, preds_geocells (N,), top5_geocells (N,k), labels_lla (N,2), labels_cell (N,))``
    as numpy arrays (what ``evaluate_model`` concatenates, training/train_eval_loop.py:125-137).  Distances and scores run on the GPU."""
    preds, cells, topk, labels_lla, labels_cell = (np.asarray(r) for r in results)
    out = {"Geocell_accuracy": float((cells == labels_cell).mean()),
           "Geocell_top5_accuracy": float((topk == labels_cell[:, None]).any(1).mean())}
    d, s = score_batch(torch.as_tensor(preds).cuda(), torch.as_tensor(labels_lla).cuda())
    summ = compute_summary(d, s)
    out.update(Mean_distance_km=summ["avg_distance_km"], Median_distance_km=summ["median_distance_km"], Mean_score=summ["avg_score"])
    return out
