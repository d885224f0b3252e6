"""Drop-in for the reference's ``models/proto_refiner.py`` (``ProtoRefiner``, :30-389): same constructor knobs
(``topk=5, max_refinement=1000, temperature=1.6``) and the same ``forward(embedding, initial_preds, candidate_cells,
candidate_probs) -> (loss, preds_LLH, preds_geocell)`` contract, executed as ONE HIP launch over the whole batch
(``gg_proto_refine``) instead of the reference's Python double loop with a ``.item()`` sync per candidate.

The prototype store is a CSR table on the device (``cell_ptr``, ``proto_emb``, ``proto_lnglat``) holding what the reference
keeps as one HF ``Dataset`` per geocell (:104-113).  The constructor builds it the reference's way -- ``proto_path`` CSV ->
``ProtoDataManager`` -> per-cluster mean embeddings (built on the GPU by ``gg_segment_mean`` or loaded from
``data/geocells/protos/proto_{i}``) -- or from a ready table (``from_clusters``).

Within-cluster refinement (:239-269): by default a cluster answers with its centroid -- the only branch that can execute in the
reference as shipped (``self.dataset`` is undefined at :254).  ``within_cluster=True`` runs the branch the reference's code
describes: a per-cluster member table (view-averaged member embeddings + their (lng, lat) labels, built from the same per-panorama
embeddings the prototypes are built from, or passed in) rides along, and a cluster with members answers with the member at ``argmax`` of
the Euclidean distances (:262-268, restated as written), inside the same launch.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch
from torch import nn, Tensor
from torch.nn.parameter import Parameter

from .. import _lib as L


PROTO_PATH = "data/geocells/proto_df.csv"            # models/proto_refiner.py:25
PROTOS_DIR = "data/geocells/protos"                   # models/proto_refiner.py:101,107 (proto_{i} HF datasets)


class ProtoRefiner(nn.Module):
    """Constructor = the reference's (:33-118): ``proto_path`` is read into a ``ProtoDataManager``; ``protos=None`` BUILDS the
    prototypes (per-cluster mean embedding of the member panoramas, :271-345,461-517) and saves them under ``protos_dir`` as one
    HF dataset per geocell, any other ``protos`` LOADS them from there (:104-113) -- or takes a ready list of per-cell tables.
    The per-panorama embeddings the build needs come from ``embeddings=`` ((P, V, D) / (P, D), row = the sample index the
    ``indices`` column refers to) or from the backend's embedding database (``clip_db_path`` / ``tinyvit_db_path``, the
    reference's precomputed-embedding SQLite layout; panorama i = i-th location in key order); the reference instead re-embeds
    the images on the fly (``Embeddings``, :409-459), which needs its image database.
    Extra keyword-only ways in: ``cell_ptr/proto_emb/proto_lnglat`` (a ready CSR table) and ``from_clusters``."""

    def __init__(self, topk: int = 5, max_refinement: int = 1000, temperature: float = 1.6, proto_path: str = PROTO_PATH,
                 protos=None, verbose: bool = False, clip_db_path: str = "data/sqlite/clip/dataset.sqlite",
                 tinyvit_db_path: str = "data/sqlite/tinyvit/dataset.sqlite", backend: str = "clip", cell_ptr=None, proto_emb=None,
                 proto_lnglat=None, embeddings=None, latlon_by_index=None, protos_dir: str = PROTOS_DIR, save_protos: bool = True,
                 within_cluster: bool = False, member_ptr=None, member_emb=None, member_lnglat=None):
        super().__init__()
        self.topk = topk
        self.max_refinement = max_refinement
        self.verbose = verbose
        backend = (backend or "clip").lower()
        if backend not in ("clip", "tinyvit"):
            raise ValueError("backend must be 'clip' or 'tinyvit'")          # :413-414
        self.backend = backend
        self.temperature = Parameter(torch.tensor(float(temperature)), requires_grad=False)
        self.geo_scaling = Parameter(torch.tensor(20.0), requires_grad=False)
        self.within_cluster = bool(within_cluster)
        if cell_ptr is None:
            import pandas as pd
            from .utils import ProtoDataManager
            self.proto_df = pd.read_csv(proto_path)                           # FileNotFoundError like the reference (:79)
            self.proto_manager = ProtoDataManager(self.proto_df)
            self.proto_df["geocell_index"] = self.proto_df["geocell_index"].astype(int)
            num_geocells = int(self.proto_df["geocell_index"].max()) + 1
            if protos is None:
                tab = self._build_prototypes(embeddings, latlon_by_index, clip_db_path if backend == "clip" else tinyvit_db_path)
                if save_protos and protos_dir:
                    self._save_protos(tab, num_geocells, protos_dir)
            elif isinstance(protos, (list, tuple)):
                if len(protos) != num_geocells:
                    raise ValueError("Number of loaded prototypes does not match number of geocells.")      # :92-95
                tab = self._table_from_cells(list(protos))
            else:
                tab = self._table_from_cells(self._load_protos(num_geocells, protos_dir))
                if verbose:
                    print("Loaded prototypes from disk.")
            gi = tab["geocell_index"]
            order = np.argsort(gi, kind="stable")
            cell_ptr = np.concatenate([[0], np.cumsum(np.bincount(gi, minlength=num_geocells))]).astype(np.int64)
            proto_emb = np.asarray(tab["embedding"], np.float32)[order]
            proto_lnglat = np.stack([tab["centroid_lng"], tab["centroid_lat"]], 1).astype(np.float32)[order]
            if self.within_cluster and member_ptr is None:
                member_ptr, member_emb, member_lnglat = self._member_table(order, embeddings, latlon_by_index,
                                                                           clip_db_path if backend == "clip" else tinyvit_db_path)
        self.register_buffer("cell_ptr", torch.as_tensor(np.asarray(cell_ptr), dtype=torch.int64).contiguous())
        self.register_buffer("proto_emb", torch.as_tensor(np.asarray(proto_emb), dtype=torch.float32).contiguous())
        self.register_buffer("proto_lnglat", torch.as_tensor(np.asarray(proto_lnglat), dtype=torch.float32).contiguous())
        self.num_geocells = self.cell_ptr.numel() - 1
        assert self.proto_emb.shape[0] == self.proto_lnglat.shape[0] == int(self.cell_ptr[-1])
        if self.within_cluster:
            if member_ptr is None:
                raise ValueError("within_cluster=True needs the member table (member_ptr / member_emb / member_lnglat) or per-panorama embeddings")
            mp = torch.as_tensor(np.asarray(member_ptr), dtype=torch.int64).contiguous()
            assert mp.numel() == self.proto_emb.shape[0] + 1, "member_ptr must have one entry per prototype row + 1"
            me = torch.as_tensor(np.asarray(member_emb) if not torch.is_tensor(member_emb) else member_emb, dtype=torch.float32)
            me = (me.mean(dim=1) if me.dim() == 3 else me).contiguous()                  # (n, 4, D) member embeddings are averaged over views (:258-260)
            assert me.shape[0] == int(mp[-1]) and me.shape[1] == self.proto_emb.shape[1]
            self.register_buffer("member_ptr", mp)
            self.register_buffer("member_emb", me)
            self.register_buffer("member_lnglat", torch.as_tensor(np.asarray(member_lnglat), dtype=torch.float32).contiguous())

    # ---- prototype build / load (models/proto_refiner.py:271-345) ---------------------------------------------------------
    def _build_prototypes(self, embeddings, latlon_by_index, db_path):
        from ..embedding_store import build_prototypes_from_members, read_panorama_embeddings
        L.require_gpu()
        if embeddings is None:
            import os
            if not os.path.exists(db_path):
                raise FileNotFoundError(f"ProtoRefiner(protos=None) builds prototypes from per-panorama embeddings: pass embeddings= or "
                                        f"provide the {self.backend} embedding database at '{db_path}'")
            embeddings, latlon_by_index = read_panorama_embeddings(db_path)
        tab = self.proto_manager.cluster_table()
        emb = torch.as_tensor(np.asarray(embeddings) if not torch.is_tensor(embeddings) else embeddings).to("cuda", torch.float32)
        protos = build_prototypes_from_members(emb, tab["ptr"], tab["member"], latlon_by_index)
        tab["embedding"] = protos.cpu().numpy()
        return tab

    def _member_table(self, order, embeddings, latlon_by_index, db_path):
        """CSR member lists per prototype row (rows in the order the prototype table ends up in): the proto_df ``indices`` of each cluster that
        point at a valid sample, with the samples' view-averaged embeddings and (lng, lat) labels -- what ``self.dataset["train"][cluster["indices"]]``
        would hand the reference's ``_within_cluster_refinement`` (:254-268)."""
        if embeddings is None:
            from ..embedding_store import read_panorama_embeddings
            embeddings, latlon_by_index = read_panorama_embeddings(db_path)
        if latlon_by_index is None:
            raise ValueError("within_cluster=True needs latlon_by_index (the samples' (lat, lon) labels)")
        emb = torch.as_tensor(np.asarray(embeddings) if not torch.is_tensor(embeddings) else embeddings, dtype=torch.float32)
        emb = emb.mean(dim=1) if emb.dim() == 3 else emb
        ll = np.asarray(latlon_by_index, np.float64)
        tab = self.proto_manager.cluster_table()
        ptr, member = np.asarray(tab["ptr"], np.int64), np.asarray(tab["member"], np.int64)
        lists = []
        for j in order:
            m = member[ptr[j]:ptr[j + 1]]
            m = m[(m >= 0) & (m < min(len(ll), emb.shape[0]))]
            lists.append(m[np.isfinite(ll[m]).all(axis=1)] if m.size else m)
        mptr = np.concatenate([[0], np.cumsum([len(m) for m in lists])]).astype(np.int64)
        flat = np.concatenate(lists) if lists and mptr[-1] > 0 else np.zeros((0,), np.int64)
        return mptr, emb[torch.from_numpy(flat)], np.stack([ll[flat, 1], ll[flat, 0]], 1).astype(np.float32)       # labels as (lng, lat)

    @staticmethod
    def _table_from_cells(cells):
        """cells[i]: None or a table (HF Dataset / dict of columns) with ``embedding`` (P_i, D), ``centroid_lng``, ``centroid_lat``."""
        gi, emb, lng, lat = [], [], [], []
        for i, c in enumerate(cells):
            if c is None:
                continue
            e = np.asarray(c["embedding"], np.float32)
            e = e.reshape(len(c["centroid_lng"]), -1)
            gi += [i] * e.shape[0]; emb.append(e)
            lng += [float(v) for v in c["centroid_lng"]]; lat += [float(v) for v in c["centroid_lat"]]
        return dict(geocell_index=np.asarray(gi, np.int64), embedding=np.concatenate(emb, 0) if emb else np.zeros((0, 1), np.float32),
                    centroid_lng=np.asarray(lng, np.float32), centroid_lat=np.asarray(lat, np.float32))

    @staticmethod
    def _load_protos(num_geocells: int, protos_dir: str):
        from datasets import Dataset
        cells = [None] * num_geocells
        for i in range(num_geocells):
            try:
                cells[i] = Dataset.load_from_disk(f"{protos_dir}/proto_{i}").with_format("numpy")
            except FileNotFoundError:
                cells[i] = None
        return cells

    @staticmethod
    def _save_protos(tab, num_geocells: int, protos_dir: str):
        """One HF dataset per geocell with the proto_df columns + ``embedding`` (:97-101)."""
        from datasets import Dataset
        import os
        os.makedirs(protos_dir, exist_ok=True)
        gi = tab["geocell_index"]
        for i in range(num_geocells):
            rows = np.nonzero(gi == i)[0]
            if rows.size == 0:
                continue
            Dataset.from_dict(dict(geocell_index=[int(i)] * rows.size, count=[int(v) for v in tab["count"][rows]],
                                   centroid_lat=[float(v) for v in tab["centroid_lat"][rows]],
                                   centroid_lng=[float(v) for v in tab["centroid_lng"][rows]],
                                   embedding=[tab["embedding"][r].tolist() for r in rows])).save_to_disk(f"{protos_dir}/proto_{i}")

    @classmethod
    def from_clusters(cls, geocell_index: Sequence[int], embeddings, centroid_lng, centroid_lat, num_geocells: int, **kw):
        """One row per cluster (the rows of ``proto_df.csv``: geocell_index, centroid_lng/lat + its mean embedding)."""
        gi = np.asarray(geocell_index, np.int64)
        order = np.argsort(gi, kind="stable")
        counts = np.bincount(gi, minlength=num_geocells)
        ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        emb = np.asarray(embeddings, np.float32)[order]
        ll = np.stack([np.asarray(centroid_lng, np.float32), np.asarray(centroid_lat, np.float32)], 1)[order]
        if kw.get("member_ptr") is not None:          # member lists arrive in the caller's cluster order: carry them through the sort
            mp = np.asarray(kw["member_ptr"], np.int64)
            me, ml = np.asarray(kw["member_emb"], np.float32), np.asarray(kw["member_lnglat"], np.float32)
            rows = [np.arange(mp[j], mp[j + 1]) for j in order]
            flat = np.concatenate(rows) if rows else np.zeros((0,), np.int64)
            kw.update(member_ptr=np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64), member_emb=me[flat], member_lnglat=ml[flat],
                      within_cluster=True)
        return cls(cell_ptr=ptr, proto_emb=emb, proto_lnglat=ll, **kw)

    def __str__(self):
        return (f"ProtoRefiner(\n\ttopk\t\t= {self.topk}\n\tmax_refinement\t= {self.max_refinement}\n"
                f"\ttemperature\t= {self.temperature.data.item()}\n\tgeo_scaling\t= {self.geo_scaling.data.item()}\n)")

    def forward(self, embedding: Tensor = None, initial_preds: Tensor = None, candidate_cells: Tensor = None,
                candidate_probs: Tensor = None):
        assert self.topk <= candidate_cells.size(1), \
            '"topk" parameter must be smaller or equal to the number of geocell candidates passed into the forward function.'
        L.require_gpu()
        dev = self.proto_emb.device
        if not self.proto_emb.is_cuda:
            raise L.GgError("ProtoRefiner tables are on the CPU; call .to('cuda') -- there is no CPU fallback")
        emb = embedding.detach().to(device=dev, dtype=torch.float32).contiguous()
        if emb.dim() == 3:
            B, V, D = emb.shape
        else:
            (B, D), V = emb.shape, 1
        assert D == self.proto_emb.shape[1], f"embedding dim {D} != prototype dim {self.proto_emb.shape[1]}"
        init = initial_preds.detach().to(device=dev, dtype=torch.float32).contiguous()
        cells = candidate_cells.detach().to(device=dev, dtype=torch.int64).contiguous()
        probs = None if candidate_probs is None else candidate_probs.detach().to(device=dev, dtype=torch.float32).contiguous()
        out_llh = torch.empty((B, 2), dtype=torch.float32, device=dev)
        out_cell = torch.empty((B,), dtype=torch.int64, device=dev)
        out_idx = torch.empty((B,), dtype=torch.int64, device=dev)
        a = L.ProtoRefineArgs()
        a.embedding, a.B, a.V, a.D = L.ptr(emb), B, V, D
        a.initial_preds, a.candidate_cells, a.candidate_probs = L.ptr(init), L.ptr(cells), L.ptr(probs)
        a.num_candidates, a.topk = cells.shape[1], self.topk
        a.cell_ptr, a.num_cells = L.ptr(self.cell_ptr), self.num_geocells
        a.proto_emb, a.proto_lnglat = L.ptr(self.proto_emb), L.ptr(self.proto_lnglat)
        a.max_refinement, a.temperature = float(self.max_refinement), float(self.temperature.item())
        a.out_llh, a.out_cell, a.out_idx = L.ptr(out_llh), L.ptr(out_cell), L.ptr(out_idx)
        if self.within_cluster:
            a.member_ptr, a.member_emb, a.member_lnglat = L.ptr(self.member_ptr), L.ptr(self.member_emb), L.ptr(self.member_lnglat)
        L.check(L.lib().gg_proto_refine(C.byref(a), L.stream()), "gg_proto_refine")
        self.last_guess_index = out_idx
        if self.verbose:
            perc_changed = (out_idx != 0).sum() / out_idx.size(0)
            print(f"Changed geocell predictions of {perc_changed * 100:.1f} % of guesses.")
        loss = 0 if self.training else None
        return loss, out_llh, out_cell
