#!/usr/bin/env python3
"""Round-4 golden fixture, produced by RUNNING the reference's own ``ProtoRefiner.forward`` -- build container only, never on the GPU box:

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r4.py

``proto_refine.npz`` -- ``ProtoRefiner.forward`` + ``_within_cluster_refinement`` + ``_euclidean_distance`` + ``_temperature_softmax``
(``models/proto_refiner.py:129-269,364-389``) and the fp64-radius gate ``preprocessing/geo_utils.haversine`` (``:39-54``), executed
UNMODIFIED on the CPU.  The class is imported behind the import stubs of SURVEY.md App. D; the instance is made with ``object.__new__`` +
``nn.Module.__init__`` (the shipped constructor needs ``proto_df.csv`` -- a missing large blob --, S3 and two embedders) and given exactly
the attributes ``forward`` reads: ``topk``, ``max_refinement``, ``verbose``, ``temperature`` (the frozen Parameter of ``:117``), ``protos``
(per geocell ``None`` or a torch-formatted table: ``["embedding"]`` -> (P, D) tensor, ``[j]`` -> row dict ``indices / count / centroid_lat /
centroid_lng``) and -- for the member case only -- ``dataset`` (the attribute ``:254`` reads and the shipped constructor leaves commented out,
``:75-77``).  The forward's hard-coded ``"cuda"`` placements (``:185,:205,:216,:231-236`` and the ``.to("cuda")`` calls) are neutralised for the
duration of the call by mapping the device string to ``"cpu"`` in ``torch.tensor`` / ``Tensor.to``; no arithmetic is touched.

Two cases are stored, inputs + the reference's outputs only (no reference text):
  case "centroid": every cluster row has ``count == 0`` -> the only branch of ``_within_cluster_refinement`` that runs as shipped (:251-252)
  case "member":   rows with ``count > 0`` go through ``:254-268`` (member embeddings (n, 4, D) averaged over the views, ``argmax`` of the
                   Euclidean DISTANCES, labels (lng, lat) of that member)
Both contain: cells without prototypes (the -100000 sentinel with prediction (0, 0)), refinements cancelled by the 1000 km gate, a cell
with 40 prototypes (``torch.cdist`` switches to its matmul form above 25 rows), 3-D (B, 4, D) embeddings, and a call with
``candidate_probs=None``.  Prototype geometry is generated with clear margins so the selections do not hinge on the last float32 bit."""
import contextlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _import_reference():
    os.chdir(REF)
    sys.path.insert(0, REF)
    src = open("config.py").read()
    head = src.split("# Training arguments")[0].replace("from transformers import TrainingArguments", "")
    cfg = types.ModuleType("config")
    exec(head, cfg.__dict__)
    cfg.TRAIN_ARGS = cfg.PRETRAIN_ARGS = cfg.PRETAIN_ARGS = None
    sys.modules["config"] = cfg
    import datasets, transformers, accelerate  # noqa: F401  (must be imported before the stubs)
    import transformers.models.clip.modeling_clip  # noqa: F401
    from unittest.mock import MagicMock
    for name in ["timm", "timm.data", "timm.data.transforms_factory", "torchvision", "torchvision.transforms",
                 "loguru", "wandb", "dotenv", "boto3", "botocore", "botocore.config", "botocore.exceptions", "s3fs"]:
        sys.modules.setdefault(name, MagicMock())


@contextlib.contextmanager
def _cuda_means_cpu():
    """Map the device string "cuda" to "cpu" in torch.tensor(..., device=) and Tensor.to(...) while the reference's forward runs."""
    import torch
    real_tensor, real_to = torch.tensor, torch.Tensor.to

    def fix(d):
        return "cpu" if (isinstance(d, str) and d.startswith("cuda")) else d

    def tensor(*a, **k):
        if "device" in k:
            k["device"] = fix(k["device"])
        return real_tensor(*a, **k)

    def to(self, *a, **k):
        a = tuple(fix(x) for x in a)
        if "device" in k:
            k["device"] = fix(k["device"])
        return real_to(self, *a, **k)

    torch.tensor, torch.Tensor.to = tensor, to
    try:
        yield
    finally:
        torch.tensor, torch.Tensor.to = real_tensor, real_to


class _TorchTable:
    """What a torch-formatted ``datasets.Dataset`` gives the forward: ``t["col"]`` -> the stacked column, ``t[j]`` -> the row as a dict of tensors,
    ``t[index_tensor]`` -> the selected rows as a dict of stacked columns."""

    def __init__(self, cols):
        self.cols = cols

    def __getitem__(self, key):
        import torch
        if isinstance(key, str):
            return self.cols[key]
        if isinstance(key, int):
            return {k: (v[key] if not isinstance(v, list) else v[key]) for k, v in self.cols.items()}
        idx = torch.as_tensor(key, dtype=torch.long)
        return {k: v[idx] for k, v in self.cols.items() if not isinstance(v, list)}


def _geometry(seed, member):
    """Synthetic prototype table (CSR over geocells) + member table + queries, with margins."""
    rng = np.random.default_rng(seed)
    K, D, B, NC = 30, 64, 56, 6                       # geocells, embedding dim, samples, candidates passed (>= topk = 5)
    counts = rng.integers(1, 6, K)
    counts[[4, 11, 23]] = 0                            # cells without prototypes -> the reference's `None` entries
    counts[7] = 40                                     # > 25 rows: torch.cdist takes its matmul form
    cell_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    P = int(cell_ptr[-1])
    proto_emb = (rng.standard_normal((P, D)) * 1.0).astype(np.float32)
    # geocells come in 6 regions of 5 neighbouring cells (a few hundred km apart) so that a sample's candidates are mostly within the 1000 km gate
    region = np.stack([rng.uniform(-160, 160, 6), rng.uniform(-60, 60, 6)], 1)
    cell_centre = np.repeat(region, 5, axis=0) + rng.normal(0, 2.0, (K, 2))
    proto_lnglat = (np.repeat(cell_centre, counts, axis=0) + rng.normal(0, 1.5, (P, 2))).astype(np.float32)
    # members: per prototype row 0..4 training panoramas, (4, D) view embeddings around the prototype, labels (lng, lat) near the centroid
    if member:
        mcount = rng.integers(0, 5, P)
        mcount[rng.random(P) < 0.3] = 0
    else:
        mcount = np.zeros(P, np.int64)
    member_ptr = np.concatenate([[0], np.cumsum(mcount)]).astype(np.int64)
    M = int(member_ptr[-1])
    member_emb4 = (np.repeat(proto_emb, mcount, axis=0)[:, None, :] + rng.standard_normal((M, 4, D)) * 0.8).astype(np.float32)
    member_lnglat = (np.repeat(proto_lnglat, mcount, axis=0) + rng.normal(0, 0.4, (M, 2))).astype(np.float32)
    # queries: near a random prototype of one of their candidate cells (so the refinement has something to find)
    cand = np.zeros((B, NC), np.int64)
    for i in range(B):
        if rng.random() < 0.25:
            cand[i] = rng.choice(K, NC, replace=False)                       # candidates scattered over the globe
        else:
            r = rng.integers(0, 6)
            other = rng.choice([c for c in range(K) if c // 5 != r])
            cand[i] = np.concatenate([rng.permutation(np.arange(5 * r, 5 * r + 5)), [other]])
    cand[0, :5] = [4, 11, 23, 4, 11]                   # a sample whose first five candidates have no prototypes at all
    cand[1, 0] = 7                                     # the 40-prototype cell in front
    emb = np.zeros((B, D), np.float32)
    for i in range(B):
        c = cand[i, rng.integers(0, 5)]
        lo, hi = cell_ptr[c], cell_ptr[c + 1]
        base = proto_emb[rng.integers(lo, hi)] if hi > lo else rng.standard_normal(D)
        emb[i] = base + rng.standard_normal(D) * 0.6
    emb4 = (emb[:, None, :] + rng.standard_normal((B, 4, D)) * 0.3).astype(np.float32)
    probs = rng.dirichlet(np.ones(NC) * 1.5, B).astype(np.float32)
    probs = -np.sort(-probs, axis=1)                   # top-k of a softmax comes sorted
    # initial prediction = centroid of the first candidate cell; a third of the samples start far away so the 1000 km gate fires
    initial = cell_centre[cand[:, 0]].astype(np.float32)
    far = rng.random(B) < 0.2
    initial[far] = np.stack([rng.uniform(-170, 170, far.sum()), rng.uniform(-70, 70, far.sum())], 1).astype(np.float32)
    return dict(cell_ptr=cell_ptr, proto_emb=proto_emb, proto_lnglat=proto_lnglat, member_ptr=member_ptr, member_emb4=member_emb4,
                member_lnglat=member_lnglat, embedding=emb4, candidate_cells=cand, candidate_probs=probs, initial_preds=initial)


def _run_reference(g, member):
    import torch
    from torch import nn
    from models.proto_refiner import ProtoRefiner
    K = len(g["cell_ptr"]) - 1
    protos = []
    for c in range(K):
        lo, hi = int(g["cell_ptr"][c]), int(g["cell_ptr"][c + 1])
        if hi == lo:
            protos.append(None)
            continue
        mp = g["member_ptr"]
        protos.append(_TorchTable(dict(
            embedding=torch.from_numpy(g["proto_emb"][lo:hi]),
            indices=[torch.arange(int(mp[j]), int(mp[j + 1])) for j in range(lo, hi)],
            count=torch.from_numpy((mp[lo + 1:hi + 1] - mp[lo:hi]).astype(np.int64)),
            centroid_lat=torch.from_numpy(g["proto_lnglat"][lo:hi, 1].copy()),
            centroid_lng=torch.from_numpy(g["proto_lnglat"][lo:hi, 0].copy()))))
    obj = object.__new__(ProtoRefiner)
    nn.Module.__init__(obj)
    obj.topk, obj.max_refinement, obj.verbose = 5, 1000, False
    obj.protos = protos
    obj.temperature = nn.Parameter(torch.tensor(1.6), requires_grad=False)           # models/proto_refiner.py:117
    obj.geo_scaling = nn.Parameter(torch.tensor(20.0), requires_grad=False)          # :118
    if member:
        obj.dataset = {"train": _TorchTable(dict(embedding=torch.from_numpy(g["member_emb4"]), labels=torch.from_numpy(g["member_lnglat"])))}
    obj.eval()
    t = lambda k: torch.from_numpy(g[k])
    out = {}
    with _cuda_means_cpu(), torch.no_grad():
        loss, llh, cell = ProtoRefiner.forward(obj, t("embedding"), t("initial_preds"), t("candidate_cells"), t("candidate_probs"))
        assert loss is None
        out["preds_LLH"], out["preds_geocell"] = llh.numpy().astype(np.float32), cell.numpy().astype(np.int64)
        # candidate_probs=None (:154-156: integer zeros with a 1 in front): same loop, first candidate only
        _, llh0, cell0 = ProtoRefiner.forward(obj, t("embedding").mean(dim=1), t("initial_preds"), t("candidate_cells"), None)
        out["preds_LLH_noprobs"], out["preds_geocell_noprobs"] = llh0.numpy().astype(np.float32), cell0.numpy().astype(np.int64)
    # which candidate slot won (what the forward prints as "changed"): recover it from the returned geocell
    out["guess_index"] = np.asarray([int(np.where(g["candidate_cells"][i, :5] == out["preds_geocell"][i])[0][0]) for i in range(len(cell))], np.int64)
    return out


def main():
    _import_reference()
    store = {}
    for name, seed, member in (("centroid", 41, False), ("member", 42, True)):
        g = _geometry(seed, member)
        out = _run_reference(g, member)
        changed = float((out["guess_index"] != 0).mean())
        print(f"proto_refine[{name}]: B={len(out['guess_index'])}, changed {100 * changed:.1f} %, slots {np.bincount(out['guess_index'], minlength=5).tolist()}")
        for k, v in {**g, **out}.items():
            store[f"{name}__{k}"] = v
    np.savez_compressed(os.path.join(HERE, "proto_refine.npz"), **store)
    print("proto_refine.npz:", os.path.getsize(os.path.join(HERE, "proto_refine.npz")), "bytes")


if __name__ == "__main__":
    main()
