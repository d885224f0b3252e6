"""Drop-in for the reference's ``models/utils.py``: ``ModelOutput`` (:12-17), ``smooth_labels`` (:20-32),
``haversine_matrix`` (:39-57), tolerant ``load_state_dict`` (:74-95).  The two geo functions run as HIP kernels
(``csrc/geo.hip``); inside the training step they are not called at all -- the fused head kernel computes
distances, argmin, soft targets and the loss in one pass."""
from __future__ import annotations

from collections import namedtuple
from typing import Dict

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
    """Loads parameters wherever names match (models/utils.py:74-95)."""
    own_state = self.state_dict()
    for name, param in state_dict.items():
        if embedder and "base_model" in name:
            name = ".".join(name.split(".")[1:])
        if name not in own_state:
            print(f"Parameter {name} not in model's state.")
            continue
        if isinstance(param, Parameter):
            param = param.data
        own_state[name].copy_(param)
