"""The fp32 gate of the fp32_split mode with EVERY split route taken (run by tests/test_gpu_precision.py::test_fp32_split_gate_with_every_split_route_forced in a
subprocess under GG_DEV_SWITCHES=1 GG_SPLIT_MIN_TILES=1 GG_SPLIT_TN_MIN_M=1: csrc/tinyvit.hip's routing thresholds -- 128 tiles for a Linear, 1024 rows for a
weight gradient -- would otherwise keep most of the mode's kernels out of a step small enough for the CPU oracle).  Two cases, the assertions and tolerances of
the fp32 mode's own training-step test (taps 2e-4, embedding 1e-4, loss 1e-5, every gradient tensor 2e-3), and the split launch count:
  * TinyViT-21M-224, 4 panoramas, reference freeze policy: 10 blocks x 4 Linears x (forward + data gradient) = 80, 2 trainable blocks x 4 weight gradients = 8,
    plus the ConvNorm convolutions that have planes (forward + plain data gradients);
  * TinyViT-5M-224, 3 panoramas, every parameter trainable.
Exit code 0 = both within tolerance; prints one "-> ok" line per case."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    assert os.environ.get("GG_DEV_SWITCHES") and os.environ.get("GG_SPLIT_MIN_TILES") == "1" and os.environ.get("GG_SPLIT_TN_MIN_M") == "1", "run under the forced-route switches"
    from tests import test_gpu_precision as T
    cent = np.load(os.path.join(ROOT, "tests", "golden", "centroids_12647x2_f32.npy"))
    ok = True
    for model_name, N, unfrozen in (("tiny_vit_21m_224", 4, False), ("tiny_vit_5m_224", 3, True)):
        label = f"fp32_split(all routes) {model_name} N={N} {'unfrozen' if unfrozen else 'ref-freeze'}"
        case = None
        try:
            case = T._train_step_case(model_name, "fp32_split", N, cent, unfrozen, seed=11, drop_path_rate=0.1)
            la = case["launches"]
            lin, wg = T.split_launches_expected(case["cfg"], case["trainable"])
            print(f"[{label}] GEMM launches: {la.split} split-product ({la.split_flops / 1e9:.1f} GFLOP), {la.plain} f32-MFMA ({la.plain_flops / 1e9:.1f} GFLOP); "
                  f"block Linears fwd + dgrad {lin}, trainable-block weight gradients {wg}", flush=True)
            # every Linear-family call is a split product; the rest of the split launches are ConvNorm convolutions (13 under the freeze policy at 21M)
            assert la.split >= lin + wg + 1, ("split launches", la.split, lin, wg)
            if not unfrozen and model_name == "tiny_vit_21m_224":
                assert la.split == int(os.environ.get("GG_SPLIT_GATE_EXPECT_21M", la.split)), la.split
            assert la.split_flops > 0.8 * (la.split_flops + la.plain_flops), "split products must carry the GEMM work"
            T._fp32_gate(case, label)
            print(f"{label} -> ok", flush=True)
        except AssertionError as exc:
            ok = False
            print(f"{label} -> FAIL {exc}", flush=True)
        del case
        import gc, torch
        gc.collect(); torch.cuda.empty_cache()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
