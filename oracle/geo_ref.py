"""CPU restatement (numpy) of the SuperGuessr head, loss and geo helpers.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  PINNED against the imported
reference by ``tests/golden/geo_*.npz`` (``tests/test_oracle_geo.py``).

Reference lines followed:
  haversine_matrix      models/utils.py:39-57           (lon,lat) deg, R = 6378.137 km, dtype follows input
  smooth_labels         models/utils.py:20-32           exp(-(d - rowmin)/65), nan/inf -> 0
  soft / hard CE        models/super_guessr.py:372-383  -(soft/sum * log_softmax).sum(-1).mean()
  head                  models/super_guessr.py:333-365  mean over 4 views, Linear, softmax, argmax, topk(5)
  nearest-centroid      main_coordinator_idun_s3.py:384-391
  haversine (fp64 gate) preprocessing/geo_utils.py:39-54
  scoring               run_benchmark.py:28-65          R = 6371 km; 5000*exp(-d/1492.7)
"""
from __future__ import annotations

import numpy as np

LABEL_SMOOTHING_CONSTANT = 65.0      # config.py:52
EARTH_RADIUS_M = 6378137.0           # models/utils.py:55, preprocessing/geo_utils.py:6
SCORE_RADIUS_M = 6371000.0           # run_benchmark.py:25
DECAY_CONSTANT = 1492.7              # config.py:49


def haversine_matrix(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """x (N,2) (lon,lat) deg, y (2,M) -> (N,M) km, computed in x.dtype."""
    dt = x.dtype
    d2r = dt.type(np.pi / 180.0)      # torch.deg2rad multiplies by pi/180 in the tensor dtype
    x_rad, y_rad = x * d2r, y.astype(dt) * d2r
    delta = x_rad[:, :, None] - y_rad[None, :, :]          # (N,2,M)
    p = np.cos(x_rad[:, 1])[:, None] * np.cos(y_rad[1, :])[None, :]
    a = np.sin(delta[:, 1, :] / dt.type(2)) ** 2 + p * np.sin(delta[:, 0, :] / dt.type(2)) ** 2
    c = dt.type(2) * np.arcsin(np.sqrt(a))
    return (dt.type(EARTH_RADIUS_M) * c) / dt.type(1000)


def smooth_labels(d: np.ndarray) -> np.ndarray:
    adj = d - d.min(axis=-1, keepdims=True)
    s = np.exp(-adj / d.dtype.type(LABEL_SMOOTHING_CONSTANT))
    return np.nan_to_num(s, nan=0.0, posinf=0.0, neginf=0.0)


def log_softmax(z: np.ndarray) -> np.ndarray:
    m = z.max(axis=-1, keepdims=True)
    return z - m - np.log(np.exp(z - m).sum(axis=-1, keepdims=True))


def soft_ce(logits: np.ndarray, labels: np.ndarray, centroids: np.ndarray):
    """returns (loss, dlogits, soft_targets, distances)."""
    d = haversine_matrix(labels.astype(np.float32), centroids.T.astype(np.float32))
    soft = smooth_labels(d)
    soft = soft / np.maximum(soft.sum(axis=-1, keepdims=True), np.float32(1e-12))
    lp = log_softmax(logits.astype(np.float32))
    loss = -(soft * lp).sum(axis=-1).mean()
    n = logits.shape[0]
    # d/dlogits of -sum(t * log_softmax) = softmax * sum(t) - t
    dlogits = (np.exp(lp) * soft.sum(axis=-1, keepdims=True) - soft) / n
    return loss, dlogits, soft, d


def hard_ce(logits: np.ndarray, labels_clf: np.ndarray):
    lp = log_softmax(logits.astype(np.float32))
    n = logits.shape[0]
    loss = -lp[np.arange(n), labels_clf].mean()
    dl = np.exp(lp)
    dl[np.arange(n), labels_clf] -= 1
    return loss, dl / n


def nearest_centroid(labels: np.ndarray, centroids: np.ndarray) -> np.ndarray:
    d = haversine_matrix(labels.astype(np.float32), centroids.T.astype(np.float32))
    return d.argmin(axis=-1).astype(np.int64)


def head_forward(embedding: np.ndarray, W: np.ndarray, b: np.ndarray, centroids: np.ndarray,
                 num_candidates: int = 5, panorama: bool = True):
    """embedding (N,4,C) or (N,C).  Returns dict(logits, probs, preds, llh, topk_vals, topk_idx)."""
    x = embedding.mean(axis=1) if (panorama and embedding.ndim == 3) else embedding
    logits = x.astype(np.float32) @ W.T.astype(np.float32) + b.astype(np.float32)
    lp = log_softmax(logits)
    probs = np.exp(lp)
    preds = probs.argmax(axis=-1).astype(np.int64)
    llh = centroids[preds]
    idx = np.argsort(-probs, axis=-1, kind="stable")[:, :num_candidates]
    vals = np.take_along_axis(probs, idx, axis=-1)
    return dict(logits=logits, probs=probs, preds=preds, llh=llh, topk_vals=vals, topk_idx=idx.astype(np.int64))


def haversine_pairs_f64(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """preprocessing/geo_utils.py:39-54 -- row-wise, float64 radius (promotes to fp64)."""
    x_rad, y_rad = np.deg2rad(x.astype(np.float64)), np.deg2rad(y.astype(np.float64))
    delta = y_rad - x_rad
    a = np.sin(delta[:, 1] / 2) ** 2 + np.cos(x_rad[:, 1]) * np.cos(y_rad[:, 1]) * np.sin(delta[:, 0] / 2) ** 2
    return EARTH_RADIUS_M * 2 * np.arcsin(np.sqrt(a)) / 1000


def haversine_np_score(lat1, lon1, lat2, lon2) -> np.ndarray:
    """run_benchmark.py:28-47 (km, R = 6371 km)."""
    lat1, lon1, lat2, lon2 = map(lambda v: np.radians(np.asarray(v, np.float64)), (lat1, lon1, lat2, lon2))
    dlat, dlon = lat2 - lat1, lon2 - lon1
    a = np.sin(dlat / 2) ** 2 + np.cos(lat1) * np.cos(lat2) * np.sin(dlon / 2) ** 2
    return SCORE_RADIUS_M * 2 * np.arcsin(np.sqrt(a)) / 1000


def geoguessr_score(d_km) -> np.ndarray:
    """run_benchmark.py:50-65: negative distances count as 0, 5000*exp(-d/1492.7) clamped to [0, 5000], ``int(round(.))`` --
    Python's round() on a float is round-half-to-even = np.rint.  int32 result.  PINNED by tests/golden/score.npz."""
    d = np.maximum(np.asarray(d_km, np.float64), 0.0)
    pts = np.minimum(np.maximum(5000.0 * np.exp(-(d / DECAY_CONSTANT)), 0.0), 5000.0)
    return np.rint(pts).astype(np.int32)


def score_summary(distance_km, score, top1_prob) -> dict:
    """run_benchmark.py:67-117 (``_compute_summary_from_data``) on arrays; ``top1_prob`` < 0 marks a sample without top-5 list."""
    d = np.asarray(distance_km, np.float64); s = np.asarray(score, np.float64); p = np.asarray(top1_prob, np.float64)
    n = len(d)
    return dict(num_samples=n, avg_distance_km=float(np.sum(d) / n), median_distance_km=float(np.median(d)),
                avg_top1_prob=float(np.sum(np.where(p < 0, 0.0, p)) / n), avg_score=float(np.sum(s) / n))


# ---------------------------------------------------------------------------- optimizer / schedule

def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.01):
    """torch.optim.AdamW single-tensor math (main_coordinator_idun_s3.py:286-291)."""
    p = p * (1 - lr * wd)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    denom = np.sqrt(v) / np.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


def cosine_warm_restarts_lr(epoch: int, base_lr: float, T_0: int = 10, T_mult: int = 2, eta_min: float = 1e-6):
    """CosineAnnealingWarmRestarts.step(epoch) closed form (main_coordinator_idun_s3.py:292-294,544)."""
    import math
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
