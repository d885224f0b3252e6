"""CPU restatement (numpy) of ``ProtoRefiner.forward`` (``models/proto_refiner.py:129-237``)
over a CSR prototype table.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Helpers PINNED by
``tests/golden/proto_helpers.npz`` (reference ``_euclidean_distance`` :364-376,
``_temperature_softmax`` :378-389, ``preprocessing/geo_utils.haversine`` :39-54).  The
reference ``forward`` itself cannot run as shipped (hard-coded "cuda", undefined
``self.dataset`` at :254 -- SURVEY.md C10), so the loop below restates it line by line; the
within-cluster step (:239-269) answers with the cluster centroid for an empty cluster (the only branch that runs as shipped) and, when
a member table is supplied, with the member the reference's code selects.

Prototype table: ``cell_ptr`` (K+1,) int, prototypes of cell c are rows
``cell_ptr[c]:cell_ptr[c+1]`` of ``proto_emb`` (P,D) f32 / ``proto_lnglat`` (P,2) f32.
An empty range is the reference's ``self.protos[cell_id] is None`` case (:180-187).
"""
from __future__ import annotations

import numpy as np

from .geo_ref import EARTH_RADIUS_M

MISSING_LOGIT = -100000.0      # models/proto_refiner.py:185


def euclidean_distance(matrix: np.ndarray, vector: np.ndarray) -> np.ndarray:
    return np.sqrt(((matrix.astype(np.float32) - vector.astype(np.float32)[None]) ** 2).sum(-1))


def temperature_softmax(x: np.ndarray, temperature: float = 1.6) -> np.ndarray:
    ex = np.exp(x.astype(np.float32) / np.float32(temperature))
    return ex / ex.sum(axis=0)


def haversine_gate_km(a: np.ndarray, b: np.ndarray) -> float:
    """geo_utils.haversine on two (lon,lat) fp32 points: fp32 trig, fp64 radius."""
    a = a.astype(np.float32); b = b.astype(np.float32)
    d2r = np.float32(np.pi / 180)
    ar, br = a * d2r, b * d2r
    delta = br - ar
    h = np.sin(delta[1] / np.float32(2)) ** 2 + np.cos(ar[1]) * np.cos(br[1]) * np.sin(delta[0] / np.float32(2)) ** 2
    c = np.float32(2) * np.arcsin(np.sqrt(h))
    return float(np.float64(EARTH_RADIUS_M) * np.float64(c) / 1000.0)


def within_cluster(emb, j, proto_lnglat, member_ptr, member_emb, member_lnglat):
    """models/proto_refiner.py:239-269 for prototype row j: count == 0 (no members) -> the cluster centroid (:251-252); otherwise the labels of the
    member at ``torch.argmax`` of the Euclidean distances between the member embeddings and the query (:262-268 -- the reference takes the
    LARGEST distance; restated as written)."""
    if member_ptr is None or member_ptr[j + 1] == member_ptr[j]:
        return float(proto_lnglat[j, 0]), float(proto_lnglat[j, 1])
    lo, hi = int(member_ptr[j]), int(member_ptr[j + 1])
    k = int(np.argmax(euclidean_distance(member_emb[lo:hi], emb)))
    return float(member_lnglat[lo + k, 0]), float(member_lnglat[lo + k, 1])


def refine(embedding, initial_preds, candidate_cells, candidate_probs, cell_ptr, proto_emb,
           proto_lnglat, topk: int = 5, max_refinement: float = 1000.0, temperature: float = 1.6,
           member_ptr=None, member_emb=None, member_lnglat=None):
    """Returns (preds_LLH (B,2) f32, preds_geocell (B,) i64, guess_index (B,) i64)."""
    if embedding.ndim == 3:
        embedding = embedding.mean(axis=1)                      # :150-151
    B = embedding.shape[0]
    if candidate_probs is None:                                 # :154-156
        candidate_probs = np.zeros(candidate_cells.shape, np.float32)
        candidate_probs[:, 0] = 1
    out_llh = np.zeros((B, 2), np.float32)
    out_cell = np.zeros((B,), np.int64)
    out_idx = np.zeros((B,), np.int64)
    for i in range(B):
        emb = embedding[i].astype(np.float32)
        top_d, top_p = [], []
        for cell in candidate_cells[i, :topk]:
            lo, hi = int(cell_ptr[cell]), int(cell_ptr[cell + 1])
            if hi == lo:
                top_d.append(MISSING_LOGIT); top_p.append((0.0, 0.0))
                continue
            logits = -euclidean_distance(proto_emb[lo:hi], emb)  # :190
            j = int(np.argmax(logits))                           # :194
            top_d.append(float(logits[j]))
            top_p.append(within_cluster(emb, lo + j, proto_lnglat, member_ptr, member_emb, member_lnglat))      # :201-202
        probs = temperature_softmax(np.asarray(top_d, np.float32), temperature)   # :205-206
        c = candidate_probs[i, :topk].astype(np.float32)
        final = c * probs                                        # :210
        refined = int(np.argmax(final))
        dist = haversine_gate_km(initial_preds[i], np.asarray(top_p[refined], np.float32))  # :216-219
        if dist > max_refinement:                                # :220-223
            final = c
        k = int(np.argmax(final))                                # :225
        out_idx[i] = k
        out_llh[i] = top_p[k]
        out_cell[i] = candidate_cells[i, k]
    return out_llh, out_cell, out_idx
