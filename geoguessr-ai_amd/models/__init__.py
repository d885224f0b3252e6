"""Mirror of the reference's ``models`` package (``from models import ProtoRefiner`` in training/train_eval_loop.py:14)."""
from .utils import ModelOutput, haversine_matrix, smooth_labels          # noqa: F401
from .tinyvit import TinyViTAdapter                                      # noqa: F401
from .super_guessr import SuperGuessr                                    # noqa: F401
from .proto_refiner import ProtoRefiner                                  # noqa: F401
