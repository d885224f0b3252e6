#!/usr/bin/env python3
"""Round-2 golden fixtures, produced by RUNNING the reference's own code -- build container only, never on the GPU box:

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r2.py

* ``score.npz``      -- ``haversine_np``, ``geoguessr_score_from_distance`` and ``_compute_summary_from_data`` of
                        ``run_benchmark.py:25-117``.  The module itself cannot be imported here (torchvision, boto3, ...), so the
                        three function definitions and the ``rad_np`` constant are located with ``ast`` in the reference file at run
                        time, compiled and executed on our inputs; nothing of them is stored in this repository.
* ``proto_mean.npz`` -- ``Embeddings.generate_embeddings`` (``models/proto_refiner.py:461-517``), the running-mean prototype
                        builder, imported (behind the import stubs of SURVEY.md App. D) and called unbound on a stand-in object whose
                        embedder returns rows of a seeded table.
* ``proto_df_small.csv`` + ``proto_manager.json`` -- a ``proto_df.csv``-format table (authored here: the reference's own file is a
                        missing large blob) and what the reference's ``ProtoDataManager`` (``models/utils.py:98-181``) parses out of it.
* ``hier.npz``       -- ``SuperGuessr(hierarchical=True)`` of the reference (``models/super_guessr.py:89-99,340-345``,
                        ``models/layers/positional_encoder.py``) in eval mode (dropout off): forward outputs and gradients.
Outputs are data only (inputs + expected outputs)."""
import ast
import json
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _reference_functions(path, names, assigns=()):
    """Compile the named top-level function definitions (+ simple assignments) of a reference file into a fresh namespace."""
    tree = ast.parse(open(path).read(), path)
    keep = [n for n in tree.body if (isinstance(n, ast.FunctionDef) and n.name in names) or
            (isinstance(n, ast.Assign) and any(isinstance(t, ast.Name) and t.id in assigns for t in n.targets))]
    assert len(keep) == len(names) + len(assigns), [getattr(n, "name", None) for n in keep]
    from typing import Any, Dict, List, Optional
    env = dict(np=np, math=math, List=List, Dict=Dict, Any=Any, Optional=Optional)
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), env)
    return env


def _import_reference():
    os.chdir(REF)
    sys.path.insert(0, REF)
    src = open("config.py").read()
    head = src.split("# Training arguments")[0].replace("from transformers import TrainingArguments", "")
    cfg = types.ModuleType("config")
    exec(head, cfg.__dict__)
    cfg.TRAIN_ARGS = cfg.PRETRAIN_ARGS = cfg.PRETAIN_ARGS = None
    sys.modules["config"] = cfg


def make_score():
    env = _reference_functions(os.path.join(REF, "run_benchmark.py"),
                               ("haversine_np", "geoguessr_score_from_distance", "_compute_summary_from_data"), ("rad_np",))
    rng = np.random.default_rng(11)
    n = 4096
    pred = np.stack([rng.uniform(-180, 180, n), rng.uniform(-90, 90, n)], 1).astype(np.float32)
    true = np.stack([rng.uniform(-180, 180, n), rng.uniform(-90, 90, n)], 1).astype(np.float32)
    # near misses (short distances -> scores close to 5000 where rounding matters), identical points, antipodes, poles, wrap
    true[:1024] = pred[:1024] + rng.normal(0, 0.5, (1024, 2)).astype(np.float32)
    true[1024:1536] = pred[1024:1536] + rng.normal(0, 0.01, (512, 2)).astype(np.float32)
    true[:, 1] = np.clip(true[:, 1], -90, 90)
    true[1536] = pred[1536]
    pred[1537] = (20.126657, 41.71182); true[1537] = (-159.873343, -41.71182)
    pred[1538] = (0, 90); true[1538] = (0, -90)
    pred[1539] = (179.9999, 10); true[1539] = (-179.9999, 10)
    # the reference feeds Python floats / float64 arrays (inference.py hands it .item()'d values): float64 of the f32 inputs
    d = env["haversine_np"](pred.astype(np.float64), true.astype(np.float64))
    score = np.asarray([env["geoguessr_score_from_distance"](float(x)) for x in d], np.int32)
    # direct distances incl. exact .5 boundaries and out-of-range values for the clamp / round-half-even behaviour
    dk = np.concatenate([np.linspace(0, 25000, 501), [-5.0, -0.0, 1e-9, 1e9],
                         [-1492.7 * math.log((k + 0.5) / 5000.0) for k in (0, 1, 2, 2499, 4998)]]).astype(np.float64)
    sk = np.asarray([env["geoguessr_score_from_distance"](float(x)) for x in dk], np.int32)
    samples = [dict(distance_km=float(d[i]), score=int(score[i]),
                    top5_geocells=[dict(probability=float(rng.uniform()))] if i % 7 else []) for i in range(200)]
    summary = env["_compute_summary_from_data"](samples)
    np.savez_compressed(os.path.join(HERE, "score.npz"), pred=pred, true=true, dist_km=d, score=score, direct_km=dk, direct_score=sk,
                        sample_top1=np.asarray([s["top5_geocells"][0]["probability"] if s["top5_geocells"] else -1.0 for s in samples]),
                        summary_keys=np.asarray(sorted(summary)), summary_vals=np.asarray([float(summary[k]) for k in sorted(summary)]))
    print("score.npz: %d pairs, score range %d..%d" % (n, score.min(), score.max()))


def _stub_modules():
    import datasets, transformers  # noqa: F401  (must be imported before the stubs)
    from unittest.mock import MagicMock
    for name in ["timm", "timm.data", "timm.data.transforms_factory", "torchvision", "torchvision.transforms",
                 "loguru", "wandb", "dotenv", "boto3", "botocore", "botocore.config", "botocore.exceptions", "s3fs"]:
        sys.modules.setdefault(name, MagicMock())


def make_proto_mean():
    import torch
    _stub_modules()
    from models.proto_refiner import Embeddings
    g = torch.Generator().manual_seed(21)
    npts, V, D = 40, 4, 96
    table = torch.randn(npts, V, D, generator=g)
    latlon = np.stack([np.linspace(-60, 60, npts), np.linspace(-170, 170, npts)], 1)
    latlon[5] = (np.nan, 3.0)                               # malformed row: skipped by the reference
    by_key = {(float(latlon[i, 0]), float(latlon[i, 1])): i for i in range(npts) if np.isfinite(latlon[i]).all()}
    obj = types.SimpleNamespace(
        lat_lon_by_index=latlon, backend="tinyvit", device="cpu",
        data=types.SimpleNamespace(get_tensor_of_panorama_images_from_point=lambda d: by_key[(d["lat"], d["lon"])]),
        model_tiny=types.SimpleNamespace(_get_embedding=lambda key: table[key] if not torch.is_tensor(key) else torch.zeros(V, D)),
        model_clip=None)
    clusters = [[3, 7, 9], [0], [5, 6, 8], [39, 1, 2, 4, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22], [-1, 99, 23, 24], [5], []]
    outs = [Embeddings.generate_embeddings(obj, idxs).numpy() for idxs in clusters]
    np.savez_compressed(os.path.join(HERE, "proto_mean.npz"), table=table.numpy(), latlon=latlon,
                        ptr=np.cumsum([0] + [len(c) for c in clusters]).astype(np.int64),
                        member=np.asarray([i for c in clusters for i in c], np.int64), out=np.stack(outs))
    print("proto_mean.npz:", np.stack(outs).shape)


def make_proto_manager():
    import pandas as pd
    from models.utils import ProtoDataManager
    rows = [
        (0, "[3, 7, 9]", 3, 41.2, 20.1), (0, "[0]", 1, 41.9, 19.8), (2, "(5, 6, 8)", 3, -33.5, 151.0),
        (2, "10,11 , 12", 3, -34.0, 150.5), (5, "", 0, 59.9, 10.7), (3, "[ ]", 0, 48.8, 2.3), (3, "17", 1, 48.9, 2.4),
        (7, "[18, 'x', 19.0, ' 20 ']", 3, 35.6, 139.6), (1, "[21, 22, 23, 24, 25]", 5, 40.7, -74.0),
    ]
    df = pd.DataFrame(rows, columns=["geocell_index", "indices", "count", "centroid_lat", "centroid_lng"])
    path = os.path.join(HERE, "proto_df_small.csv")
    df.to_csv(path, index=False)
    mgr = ProtoDataManager(pd.read_csv(path))
    cells = {}
    for cid in range(9):
        sub = mgr.get_indices_for_cell(cid)
        cells[str(cid)] = dict(indices=[list(map(int, v)) for v in sub["indices"].tolist()] if len(sub) else [],
                               count=[int(v) for v in sub["count"].tolist()] if len(sub) else [],
                               centroid_lat=[float(v) for v in sub["centroid_lat"].tolist()] if len(sub) else [],
                               centroid_lng=[float(v) for v in sub["centroid_lng"].tolist()] if len(sub) else [])
    parse_cases = [[1, 2], (3,), "", "  ", "[4,5]", "6;7", "8, 9", float("nan"), 12, "(13,)", "{14, 15}", "[16, [17]]", "abc"]
    parsed = [ProtoDataManager._parse_indices_value(c) for c in parse_cases]
    json.dump(dict(cells=cells, parse_cases=[c if not (isinstance(c, float) and c != c) else "__nan__" for c in map(lambda c: list(c) if isinstance(c, tuple) else c, parse_cases)],
                   parse_tuple=[isinstance(c, tuple) for c in parse_cases], parsed=parsed,
                   geocell_keys=sorted(int(k) for k in mgr.geocell_indices)), open(os.path.join(HERE, "proto_manager.json"), "w"), indent=1)
    print("proto_manager.json:", {k: len(v["indices"]) for k, v in cells.items()})


def make_hier():
    """SuperGuessr(hierarchical=True) of the reference in eval mode (dropouts off): forward + backward on embeddings."""
    import torch
    from models.super_guessr import SuperGuessr
    from models.utils import haversine_matrix
    torch.manual_seed(0)
    m = SuperGuessr(base_model=None, panorama=True, hierarchical=True, should_smooth_labels=True, embed_dim=576).eval()
    K, C = m.num_cells, 576
    rng = np.random.default_rng(99)
    W = rng.standard_normal((K, C), dtype=np.float32) * np.float32(0.05)
    b = rng.standard_normal((K,), dtype=np.float32) * np.float32(0.1)
    in_w = rng.standard_normal((3 * C, C), dtype=np.float32) * np.float32(0.04)
    in_b = rng.standard_normal((3 * C,), dtype=np.float32) * np.float32(0.1)
    out_w = rng.standard_normal((C, C), dtype=np.float32) * np.float32(0.04)
    out_b = rng.standard_normal((C,), dtype=np.float32) * np.float32(0.1)
    emb = rng.standard_normal((24, 4, C), dtype=np.float32)
    lab = np.stack([rng.uniform(-180, 180, 24), rng.uniform(-90, 90, 24)], 1).astype(np.float32)
    with torch.no_grad():
        m.cell_layer.weight.copy_(torch.from_numpy(W)); m.cell_layer.bias.copy_(torch.from_numpy(b))
        m.self_attn.in_proj_weight.copy_(torch.from_numpy(in_w)); m.self_attn.in_proj_bias.copy_(torch.from_numpy(in_b))
        m.self_attn.out_proj.weight.copy_(torch.from_numpy(out_w)); m.self_attn.out_proj.bias.copy_(torch.from_numpy(out_b))
    lab_t = torch.from_numpy(lab)
    clf = torch.argmin(haversine_matrix(lab_t, m.geocell_centroid_coords.data.t()), dim=-1)
    e = torch.from_numpy(emb).requires_grad_(True)
    out = m(embedding=e, labels=lab_t, labels_clf=clf)
    out.loss.backward()
    a = m.self_attn
    np.savez_compressed(os.path.join(HERE, "hier.npz"), seed=99, heads=a.num_heads, labels=lab, labels_clf=clf.numpy(),
                        checks=np.asarray([W.astype(np.float64).sum(), in_w.astype(np.float64).sum(), emb.astype(np.float64).sum()]),
                        loss=np.float32(out.loss.item()), preds_geocell=out.preds_geocell.numpy(), top5_idx=out.top5_geocells.indices.numpy(),
                        top5_vals=out.top5_geocells.values.detach().numpy(), demb=e.grad.numpy(),
                        d_in_w_abs=np.float64(a.in_proj_weight.grad.abs().double().sum().item()), d_in_w_rows=a.in_proj_weight.grad.numpy()[[0, 577, 1200]],
                        d_in_b=a.in_proj_bias.grad.numpy(), d_out_w_rows=a.out_proj.weight.grad.numpy()[[0, 100, 575]], d_out_b=a.out_proj.bias.grad.numpy(),
                        pe_rows=m.pos_encoder.pos_encoding.detach().numpy()[[0, 1, 7, 999], 0, :8],
                        state_keys=np.asarray(sorted(k for k in m.state_dict() if k.startswith(("pos_encoder", "self_attn")))))
    print("hier.npz: loss", out.loss.item())


def main():
    _import_reference()
    if "--hier-only" in sys.argv:
        make_hier()
        return
    make_score()
    make_proto_manager()
    make_proto_mean()
    make_hier()
    for f in ("score.npz", "proto_mean.npz", "proto_df_small.csv", "proto_manager.json", "hier.npz"):
        print(f"  {f:30s} {os.path.getsize(os.path.join(HERE, f)) / 1024:.1f} KB")


if __name__ == "__main__":
    main()
