"""Bulk embedding store and prototype building -- the steps on either side of the encoder/refiner path.

* ``EmbeddingWriter`` / ``read_embeddings``: the reference's precomputed-embedding SQLite layout
  (``backend/s3bucket.py:846-861`` table ``samples``; rows written as in ``:910-957`` -- float32 little-endian ``embedding``
  BLOB + ``embedding_dim``, ``INSERT OR REPLACE`` keyed by ``(location_id, heading)``; same PRAGMAs).
* ``embed_and_store``: batched embedding on the GPU (any module with the reference embedders' ``_get_embedding`` /
  ``forward`` contract) -> rows.
* ``build_prototypes``: per-cluster mean over member panoramas (``models/proto_refiner.py:461-517``: each panorama's (V, D)
  embedding is averaged over views, clusters keep the running fp32 mean in member order) on the GPU via ``gg_view_mean`` /
  ``gg_segment_mean``; feeds ``ProtoRefiner.from_clusters``.
The S3 upload / snapshot download around these (``backend/s3bucket.py``) stays out of scope."""
import sqlite3
from typing import Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import ops

SCHEMA = """
CREATE TABLE IF NOT EXISTS samples (
  location_id TEXT NOT NULL,
  lat REAL NOT NULL,
  lon REAL NOT NULL,
  heading INTEGER NOT NULL,
  capture_date TEXT,
  pano_id TEXT,
  batch_date TEXT,
  embedding BLOB NOT NULL,
  embedding_dim INTEGER NOT NULL,
  PRIMARY KEY (location_id, heading)
) WITHOUT ROWID;
"""
_INSERT = ("INSERT OR REPLACE INTO samples (location_id, lat, lon, heading, capture_date, pano_id, batch_date, embedding, "
           "embedding_dim) VALUES (?, ?, ?, ?, ?, ?, ?, ?, ?)")


class EmbeddingWriter:
    """Append embedding rows to a reference-format SQLite file (``backend/s3bucket.py:840-861,950-957``)."""

    def __init__(self, db_path: str, commit_every: int = 2000):
        self.conn = sqlite3.connect(db_path)
        cur = self.conn.cursor()
        for pragma in ("journal_mode=WAL", "synchronous=NORMAL", "temp_store=MEMORY", "mmap_size=268435456"):
            cur.execute(f"PRAGMA {pragma};")
        cur.execute(SCHEMA)
        self.conn.commit()
        self.commit_every = commit_every
        self._pending = 0
        self.rows_written = 0

    def write_batch(self, records: Sequence[dict], embeddings) -> int:
        """records: dicts with location_id, lat, lon, heading (+ optional capture_date, pano_id, batch_date);
        embeddings: (len(records), D) tensor / array -- stored as float32 little-endian bytes."""
        emb = embeddings.detach().to(torch.float32).cpu().numpy() if torch.is_tensor(embeddings) else np.asarray(embeddings)
        emb = np.ascontiguousarray(emb, dtype="<f4")
        assert emb.ndim == 2 and emb.shape[0] == len(records), "one embedding row per record"
        dim = int(emb.shape[1])
        rows = [(r.get("location_id"), None if r.get("lat") is None else float(r["lat"]),
                 None if r.get("lon") is None else float(r["lon"]), None if r.get("heading") is None else int(r["heading"]),
                 r.get("capture_date"), r.get("pano_id"), r.get("batch_date"), sqlite3.Binary(e.tobytes()), dim)
                for r, e in zip(records, emb)]
        self.conn.executemany(_INSERT, rows)
        self._pending += len(rows)
        self.rows_written += len(rows)
        if self._pending >= self.commit_every:
            self.conn.commit()
            self._pending = 0
        return len(rows)

    def close(self):
        self.conn.commit()
        self.conn.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def read_embeddings(db_path: str):
    """-> (records list of dicts, embeddings float32 (N, D)) in primary-key order."""
    conn = sqlite3.connect(db_path)
    cur = conn.execute("SELECT location_id, lat, lon, heading, capture_date, pano_id, batch_date, embedding, embedding_dim "
                       "FROM samples ORDER BY location_id, heading")
    recs, embs = [], []
    for loc, lat, lon, heading, cap, pano, batch, blob, dim in cur:
        v = np.frombuffer(blob, dtype="<f4")
        assert v.size == dim, f"row ({loc}, {heading}): blob holds {v.size} floats, embedding_dim says {dim}"
        recs.append(dict(location_id=loc, lat=lat, lon=lon, heading=heading, capture_date=cap, pano_id=pano, batch_date=batch))
        embs.append(v)
    conn.close()
    return recs, (np.stack(embs).astype(np.float32) if embs else np.zeros((0, 0), np.float32))


@torch.no_grad()
def embed_and_store(embedder, batches: Iterable, db_path: str) -> int:
    """``batches`` yields (records, pixel_values (B,3,H,W) already conditioned); every image's embedding row is written.
    ``embedder(pixel_values)`` must return (B, D) (``TinyViTEmbedding`` / ``CLIPEmbedding`` of this package, both HIP-backed)."""
    n = 0
    with EmbeddingWriter(db_path) as w:
        for records, pixel_values in batches:
            emb = embedder(pixel_values)
            emb = getattr(emb, "pooler_output", emb)
            n += w.write_batch(records, emb)
    return n


def build_prototypes(panorama_embeddings: torch.Tensor, cluster_of_panorama: Sequence[int], num_clusters: Optional[int] = None):
    """Mean embedding per cluster (``models/proto_refiner.py:461-517``).  panorama_embeddings: (P, V, D) or (P, D) on the GPU;
    ``cluster_of_panorama[p]`` = cluster id (negative = skip, the reference's invalid-row guard).  Members are accumulated in
    increasing panorama order.  Returns (prototypes (K, D) float32 on the GPU, counts (K,))."""
    emb = panorama_embeddings.to(torch.float32)
    if emb.dim() == 3:                                   # `vec.mean(dim=0)` over the views of a panorama
        emb = emb.mean(dim=1)
    cl = np.asarray(cluster_of_panorama, np.int64)
    assert cl.shape[0] == emb.shape[0]
    K = int(cl.max()) + 1 if num_clusters is None else int(num_clusters)
    valid = np.nonzero(cl >= 0)[0]
    order = valid[np.argsort(cl[valid], kind="stable")]
    counts = np.bincount(cl[valid], minlength=K).astype(np.int64)
    ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    dev = emb.device
    protos = ops.segment_mean(emb.contiguous(), torch.from_numpy(ptr).to(dev), torch.from_numpy(order.astype(np.int64)).to(dev))
    return protos, torch.from_numpy(counts)


def build_prototypes_from_members(panorama_embeddings: torch.Tensor, ptr, member, latlon_by_index=None) -> torch.Tensor:
    """``Embeddings.generate_embeddings`` (models/proto_refiner.py:461-517) for every cluster of a CSR member list, on the GPU:
    members outside [0, P) or whose (lat, lon) row is not finite are skipped (:469-474), each member's (V, D) embedding is
    averaged over its views (:483-485, ``gg_view_mean_fwd``), a cluster's vectors are summed in LIST order in fp32 and divided
    by the number of valid members, an empty cluster is the zero vector (``gg_segment_mean``).  Bit-exact against the reference's
    running CPU sum (tests/golden/proto_mean.npz).  Returns (num_clusters, D) float32 on the GPU."""
    emb = panorama_embeddings.to(torch.float32).contiguous()
    if emb.dim() == 3:
        emb = ops.view_mean_f32(emb)
    P = emb.shape[0]
    ptr = np.asarray(ptr, np.int64); member = np.asarray(member, np.int64)
    ok = (member >= 0) & (member < P)
    if latlon_by_index is not None:
        ll = np.asarray(latlon_by_index, np.float64)
        ok &= member < len(ll)
        fin = np.zeros(member.shape, bool)
        fin[ok] = np.isfinite(ll[member[ok]]).all(axis=1)
        ok &= fin
    seg = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
    counts = np.bincount(seg[ok], minlength=len(ptr) - 1)
    ptr2 = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    dev = emb.device
    return ops.segment_mean(emb, torch.from_numpy(ptr2).to(dev), torch.from_numpy(member[ok]).to(dev))


def read_panorama_embeddings(db_path: str):
    """Embedding database -> ((P, V, D) float32 array, (P, 2) (lat, lon)): panorama i = the i-th distinct ``location_id`` in key
    order, its views ordered by heading (every location must have the same number of views)."""
    recs, emb = read_embeddings(db_path)
    if not recs:
        raise ValueError(f"no embeddings in {db_path}")
    locs: List[str] = []
    for r in recs:
        if not locs or locs[-1] != r["location_id"]:
            locs.append(r["location_id"])
    V = len(recs) // len(locs)
    assert V * len(locs) == len(recs), "every location needs the same number of heading views"
    latlon = np.asarray([[recs[i * V]["lat"], recs[i * V]["lon"]] for i in range(len(locs))], np.float64)
    return emb.reshape(len(locs), V, -1), latlon
