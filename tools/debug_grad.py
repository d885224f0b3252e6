"""dev: per-tensor gradient errors of one fp32 train-step case (run on the GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_precision import _train_step_case, relerr
cent = np.load(os.path.join(ROOT, "tests", "golden", "centroids_12647x2_f32.npy"))
case = _train_step_case(sys.argv[1] if len(sys.argv) > 1 else "tiny_vit_21m_224", "fp32", int(sys.argv[2]) if len(sys.argv) > 2 else 1, cent, True, seed=11, drop_path_rate=0.1)
bb, model = case["bb"], case["model"]
for name, gref in case["grads"].items():
    p = model.cell_layer.weight if name == "cell_layer.weight" else model.cell_layer.bias if name == "cell_layer.bias" else bb._params[name]
    e = relerr(p.grad, gref)
    if e > 1e-4:
        print(f"{name:50s} rel {e:.3e} |g| {float(p.grad.norm()):.4e} |ref| {float(gref.norm()):.4e} first {p.grad.flatten()[:4].tolist()} ref {gref.flatten()[:4].tolist()}")
