"""Drop-in for the reference's ``models/proto_refiner.py`` (``ProtoRefiner``, :30-389): same constructor knobs
(``topk=5, max_refinement=1000, temperature=1.6``) and the same ``forward(embedding, initial_preds, candidate_cells,
candidate_probs) -> (loss, preds_LLH, preds_geocell)`` contract, executed as ONE HIP launch over the whole batch
(``gg_proto_refine``) instead of the reference's Python double loop with a ``.item()`` sync per candidate.

The prototype store is a CSR table on the device (``cell_ptr``, ``proto_emb``, ``proto_lnglat``) built from whatever the
caller has: a ``proto_df``-like table + per-cluster mean embeddings (``from_clusters``), i.e. the data the reference
keeps as one HF ``Dataset`` per geocell (:104-113).  Building prototypes by embedding the training set (:271-345,
:409-517) is the offline job of SURVEY.md row f2 and is out of scope here.

Within-cluster refinement (:239-269) uses the cluster centroid -- the only branch that can execute in the reference as
shipped (``self.dataset`` is undefined at :254).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch
from torch import nn, Tensor
from torch.nn.parameter import Parameter

from .. import _lib as L


class ProtoRefiner(nn.Module):
    def __init__(self, topk: int = 5, max_refinement: int = 1000, temperature: float = 1.6, proto_path: str = None,
                 protos=None, verbose: bool = False, clip_db_path: str = None, tinyvit_db_path: str = None,
                 backend: str = "clip", cell_ptr=None, proto_emb=None, proto_lnglat=None):
        super().__init__()
        self.topk = topk
        self.max_refinement = max_refinement
        self.verbose = verbose
        self.backend = backend
        self.temperature = Parameter(torch.tensor(float(temperature)), requires_grad=False)
        self.geo_scaling = Parameter(torch.tensor(20.0), requires_grad=False)
        if cell_ptr is None:
            raise L.GgError("ProtoRefiner needs a prototype table: pass cell_ptr/proto_emb/proto_lnglat or use "
                            "ProtoRefiner.from_clusters(...) (building prototypes from the image database is offline work)")
        self.register_buffer("cell_ptr", torch.as_tensor(np.asarray(cell_ptr), dtype=torch.int64).contiguous())
        self.register_buffer("proto_emb", torch.as_tensor(np.asarray(proto_emb), dtype=torch.float32).contiguous())
        self.register_buffer("proto_lnglat", torch.as_tensor(np.asarray(proto_lnglat), dtype=torch.float32).contiguous())
        self.num_geocells = self.cell_ptr.numel() - 1
        assert self.proto_emb.shape[0] == self.proto_lnglat.shape[0] == int(self.cell_ptr[-1])

    @classmethod
    def from_clusters(cls, geocell_index: Sequence[int], embeddings, centroid_lng, centroid_lat, num_geocells: int, **kw):
        """One row per cluster (the rows of ``proto_df.csv``: geocell_index, centroid_lng/lat + its mean embedding)."""
        gi = np.asarray(geocell_index, np.int64)
        order = np.argsort(gi, kind="stable")
        counts = np.bincount(gi, minlength=num_geocells)
        ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        emb = np.asarray(embeddings, np.float32)[order]
        ll = np.stack([np.asarray(centroid_lng, np.float32), np.asarray(centroid_lat, np.float32)], 1)[order]
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
        L.check(L.lib().gg_proto_refine(C.byref(a), L.stream()), "gg_proto_refine")
        self.last_guess_index = out_idx
        if self.verbose:
            perc_changed = (out_idx != 0).sum() / out_idx.size(0)
            print(f"Changed geocell predictions of {perc_changed * 100:.1f} % of guesses.")
        loss = 0 if self.training else None
        return loss, out_llh, out_cell
