"""Drop-in for the reference's ``models/utils.py``: ``ModelOutput`` (:12-17), ``smooth_labels`` (:20-32),
``haversine_matrix`` (:39-57), ``predict`` (:60-71), tolerant ``load_state_dict`` (:74-95), ``ProtoDataManager`` (:98-181).  The two geo functions run as HIP kernels
(``csrc/geo.hip``); inside the training step they are not called at all -- the fused head kernel computes
distances, argmin, soft targets and the loss in one pass."""
from __future__ import annotations

import ast
from collections import namedtuple
from typing import Any, Dict, List, Tuple

import numpy as np
import pandas as pd
import torch
from torch import Tensor
from torch.nn.parameter import Parameter

from .. import ops
from ..config import LABEL_SMOOTHING_CONSTANT

ModelOutput = namedtuple("ModelOutput", "loss loss_clf preds_LLH preds_geocell top5_geocells embedding")
TopK = namedtuple("TopK", "values indices")       # stands in for torch.return_types.topk


def haversine_matrix(x: Tensor, y: Tensor) -> Tensor:
    """x (N,2) (lon,lat) deg, y (2,M) -> (N,M) km (fp32), R = 6378.137 km."""
    cent = y.t().contiguous().to(torch.float32)
    return ops.haversine_matrix(x.to(torch.float32).contiguous(), cent)


def smooth_labels(distances: Tensor) -> Tensor:
    """exp(-(d - rowmin)/65) with nan/inf -> 0.  Elementwise glue for API parity only (not on the step path)."""
    adj = distances - distances.min(dim=-1, keepdim=True)[0]
    return torch.nan_to_num(torch.exp(-adj / LABEL_SMOOTHING_CONSTANT), nan=0.0, posinf=0.0, neginf=0.0)


def load_state_dict(self, state_dict: Dict, embedder: bool = False):
    """Tolerant loader with the behaviour of models/utils.py:74-95: every entry whose name exists in ``self.state_dict()`` is copied in place
    (a serialized ``Parameter`` contributes its ``.data``), every other entry is reported with the reference's message and skipped.  With
    ``embedder=True`` names containing ``base_model`` lose their first dotted component before the lookup."""
    targets = self.state_dict()
    for key, value in state_dict.items():
        if embedder and "base_model" in key:
            key = key.partition(".")[2]
        dst = targets.get(key)
        if dst is None:
            print(f"Parameter {key} not in model's state.")
        else:
            dst.copy_(value.data if isinstance(value, Parameter) else value)


PredictionOutput = namedtuple("PredictionOutput", "predictions label_ids metrics")


def predict(model: Any, dataset, batch_size: int = 8) -> Tuple:
    """models/utils.py:60-71 (``Trainer(model=model).predict(dataset)``) without the HF Trainer: the model is run in eval mode over
    the dataset in order (Trainer's default eval batch size is 8) and the per-batch outputs are concatenated.  Returns
    ``PredictionOutput(predictions, label_ids, metrics)`` like ``Trainer.predict``: ``predictions`` = the model output fields after
    the loss as numpy arrays (Trainer drops the loss entry when labels are given), ``label_ids`` = (labels, labels_clf) where
    present, ``metrics`` = {"test_loss": row-weighted mean loss} when the model returned one."""
    from ..training.train_eval_loop import MODEL_KEYS, _num_rows, _take
    was_training = model.training
    model.eval()
    cols, loss_sum, n_seen = None, 0.0, 0
    n = _num_rows(dataset)
    with torch.no_grad():
        for s0 in range(0, n, batch_size):
            data = _take(dataset, torch.arange(s0, min(n, s0 + batch_size)))
            out = model(**{k: v for k, v in data.items() if k in MODEL_KEYS})
            if isinstance(out, tuple) and not hasattr(out, "_fields"):           # serving tuple (pred_LLH, topk, embedding)
                fields = [out[0], out[1].values, out[1].indices, out[2]]
            else:
                if out.loss is not None:
                    loss_sum += float(out.loss) * _num_rows(data); n_seen += _num_rows(data)
                fields = [out.loss_clf if out.loss_clf is None else out.loss_clf.reshape(1), out.preds_LLH, out.preds_geocell,
                          out.top5_geocells.values, out.top5_geocells.indices, out.embedding][1:]
            arrs = [f.detach().cpu().numpy() for f in fields]
            cols = [[a] for a in arrs] if cols is None else [c + [a] for c, a in zip(cols, arrs)]
    model.train(was_training)
    preds = tuple(np.concatenate(c, 0) for c in cols) if cols else ()
    labels = tuple(np.asarray(dataset[k]) if isinstance(dataset, dict) and k in dataset else None for k in ("labels", "labels_clf"))
    labels = tuple(l for l in labels if l is not None) or None
    return PredictionOutput(preds, labels, {"test_loss": loss_sum / n_seen} if n_seen else {})


class ProtoDataManager:
    """Manages prototype data for geocell prototypes: the rows of ``proto_df.csv`` (``geocell_index, indices, count, centroid_lat,
    centroid_lng``), grouped per geocell, with the ``indices`` column parsed into lists of ints (models/utils.py:98-181; same
    attributes ``proto_df`` / ``geocell_indices`` and the same tolerant parsing, pinned by tests/golden/proto_manager.json)."""

    def __init__(self, proto_data: pd.DataFrame):
        self.proto_df = proto_data.copy()
        if "geocell_index" in self.proto_df.columns:
            self.proto_df["geocell_index"] = self.proto_df["geocell_index"].astype(int)
        if "indices" in self.proto_df.columns:
            self.proto_df["indices"] = self.proto_df["indices"].apply(self._parse_indices_value)
        self.geocell_indices = self._make_geocell_indices_list()

    @staticmethod
    def _as_int(item):
        """int(item), retried on the stripped text form; None when neither converts (such entries are dropped, :146-153)."""
        for form in (item, str(item).strip()):
            try:
                return int(form)
            except Exception:
                pass
        return None

    @classmethod
    def _parse_indices_value(cls, indices_val) -> List[int]:
        """One cell of the ``indices`` column -> list[int] (behaviour of models/utils.py:118-154, pinned by tests/golden/proto_manager.json):
        sequences are taken item by item; NaN / None / "" give []; text is read as a Python literal ("[1, 2]", "(1, 2)", "7") and, when that
        fails ("1, 2", "[3, x, 4]"), split on commas inside its outermost brackets; any other scalar is a one-item list."""
        if isinstance(indices_val, (list, tuple)):
            items = indices_val
        elif isinstance(indices_val, str):
            text = indices_val.strip()
            if not text:
                items = ()
            else:
                try:
                    literal = ast.literal_eval(text)
                    items = literal if isinstance(literal, (list, tuple)) else (literal,)
                except Exception:
                    items = [tok for tok in text.strip("[](){}").split(",") if tok]
        elif pd.isna(indices_val):
            items = ()
        else:
            items = (indices_val,)
        return [v for v in map(cls._as_int, items) if v is not None]

    def _make_geocell_indices_list(self) -> Dict[int, pd.DataFrame]:
        """geocell id -> its rows of ``proto_df`` (re-indexed from 0), {} when the table has no ``geocell_index`` column (:156-168)."""
        if "geocell_index" not in self.proto_df:
            return {}
        groups = self.proto_df.groupby("geocell_index").groups          # id -> row labels, in file order
        return {int(cell): self.proto_df.loc[rows].reset_index(drop=True) for cell, rows in groups.items()}

    def get_indices_for_cell(self, cell_id: int) -> pd.DataFrame:
        """Rows of one geocell; an empty frame with the table's columns for an unknown id (:170-181)."""
        try:
            return self.geocell_indices[cell_id]
        except KeyError:
            return pd.DataFrame(columns=list(self.proto_df.columns))

    # ---- flat views for the device-side prototype table (not in the reference) -------------------------------------------
    def cluster_table(self):
        """Clusters in (geocell_index, file order): -> dict(geocell_index (R,), centroid_lng (R,), centroid_lat (R,), count (R,),
        ptr (R+1,), member (sum,)) -- the CSR member lists ``gg_segment_mean`` consumes."""
        df = self.proto_df.sort_values("geocell_index", kind="stable")
        lists = list(df["indices"]) if "indices" in df.columns else [[] for _ in range(len(df))]
        ptr = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(np.int64)
        member = np.asarray([i for l in lists for i in l], np.int64)
        col = lambda name, dt: np.asarray(df[name], dt) if name in df.columns else np.zeros(len(df), dt)
        return dict(geocell_index=col("geocell_index", np.int64), centroid_lng=col("centroid_lng", np.float32),
                    centroid_lat=col("centroid_lat", np.float32), count=col("count", np.int64), ptr=ptr, member=member)
