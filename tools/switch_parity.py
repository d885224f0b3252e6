"""Parity of one training step (TinyViT-5M-224 + geocell head, EVERY parameter trainable, DropPath masks injected) against the CPU oracle in both
arithmetic modes -- run by tests/test_gpu_switches.py in a subprocess per group of GG_* development switches (they are read once per process):
every kernel / schedule a switch selects is also a shape fallback of the default path, so each gets the same parity check the default path gets.
Exit code 0 = within tolerance; prints one line per mode."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
    from geoguessr_ai_amd.models.super_guessr import SuperGuessr
    from oracle import step_ref as S, tinyvit_ref as R
    cent = np.load(os.path.join(ROOT, "tests", "golden", "centroids_12647x2_f32.npy"))[:512]
    N = 2
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, 4, 3, 224, 224, generator=g)
    labels = torch.stack([torch.rand(N, generator=g) * 360 - 180, torch.rand(N, generator=g) * 180 - 90], 1)
    cfg = R.config_for("tiny_vit_5m_224", drop_path_rate=0.1)
    frozen = os.environ.get("GG_SWITCH_PARITY_FROZEN") == "1"          # reference freeze policy: the fused frozen-chain schedules
    ok = True
    ref = None
    for precision, tol_emb, tol_grad in (("fp32", 1e-4, 2e-3), ("bf16", 5e-2, 0.25)):
        torch.manual_seed(0)
        base = TinyViTAdapter("tiny_vit_5m_224", pretrained=False, drop_path_rate=0.1, precision=precision)
        model = SuperGuessr(base, panorama=True, should_smooth_labels=True, centroids=cent).cuda().train()
        if not frozen:
            base.unfreeze_all()
        bb = base.backbone
        keep = torch.rand(bb.num_drop_slots, 4 * N, generator=torch.Generator().manual_seed(9)) > 0.3
        scales = (keep.float() / (1 - torch.tensor(bb.drop_rates).unsqueeze(1))).contiguous().cuda()
        bb.make_drop_scales = lambda batch, generator=None: scales
        st = {k: v.detach().cpu().clone() for k, v in bb.state_dict().items()}
        W, b = model.cell_layer.weight.detach().cpu().clone(), model.cell_layer.bias.detach().cpu().clone()
        trainable = [n for n, p in bb.named_parameters() if p.requires_grad]
        out = model(pixel_values=x.cuda(), labels=labels.cuda())
        out.loss.backward()
        torch.cuda.synchronize()
        if ref is None:
            ref = S.train_step(cfg, st, W, b, torch.from_numpy(cent), x, labels, drop_masks=[keep[s] for s in range(bb.num_drop_slots)], trainable=trainable)
        emb, eref = out.embedding.detach().cpu().double(), ref["embedding"].double()
        rel_emb = float((emb - eref).norm() / eref.norm())
        rel_loss = abs(float(out.loss.detach()) - float(ref["loss"])) / float(ref["loss"])
        floor = 1e-4 * float(np.median([float(v.norm()) for v in ref["grads"].values()]))
        errs = {n: float((bb._params[n].grad.cpu().double() - gr.double()).norm() / (gr.double().norm() + floor))
                for n, gr in ref["grads"].items() if n in bb._params}
        worst = max(errs, key=errs.get)
        med = float(np.median(list(errs.values())))
        # bf16 against the PURE fp32 oracle: small BatchNorm-weight gradients carry the rounding of every stored activation (the bf16-emulating
        # oracle of tests/test_gpu_precision.py is the tight check); here: median and a loose worst case
        good = rel_emb < tol_emb and rel_loss < (1e-5 if precision == "fp32" else 5e-3) and (errs[worst] < tol_grad if precision == "fp32" else
                                                                                             (med < 0.1 and errs[worst] < 0.8))
        ok &= good
        print(f"switch_parity[{precision}{' frozen' if frozen else ''}]: embedding rel-L2 {rel_emb:.2e}, loss rel {rel_loss:.2e}, {len(errs)} gradients worst {worst} "
              f"{errs[worst]:.2e}, median {med:.2e} -> {'ok' if good else 'FAIL'}", flush=True)
        del model, base, out
        torch.cuda.empty_cache()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
