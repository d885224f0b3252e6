"""CLIP vision tower on the GPU: both arithmetic modes, inference and fine-tuning, against
* the transformers golden ``clip_tiny.npz`` (forward) and the REFERENCE's SuperGuessr-on-CLIP run ``clip_train.npz`` (loss, embedding,
  every parameter gradient) -- the one encoder whose backward parity is pinned by the reference itself;
* the pinned CPU oracle ``oracle/clip_ref.py`` at the real ViT-B/32 and ViT-L/14-336 shapes.
Tolerances: fp32 mode = SURVEY 8(c)'s fp32 class (loss rel 1e-5, activations / gradients rel-L2 1e-4 ... 1e-3); bf16 mode = the bf16 class."""
import os

import numpy as np
import pytest
import torch

from tests import clip_golden as CG

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = torch.as_tensor(a).flatten().double().cpu(), torch.as_tensor(b).flatten().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


def _tiny_tower(case, precision, name="openai/clip-vit-tiny-golden"):
    from geoguessr_ai_amd.pretrain.clip_embedder import CLIPVisionTower
    hs, inter, nl, nh, img, ps = case["cfg"]
    tower = CLIPVisionTower(name, hidden_size=hs, intermediate_size=inter, num_layers=nl, num_heads=nh, image_size=img, patch_size=ps,
                            precision=precision)
    tower.load_hf_state_dict(case["weights"])
    return tower


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
def test_clip_forward_matches_transformers_golden(golden_dir, precision):
    case = CG.load(golden_dir)
    g = np.load(os.path.join(golden_dir, "clip_tiny.npz"))
    tower = _tiny_tower(case, precision).cuda().eval()
    assert {k for k in tower.state_dict()} == {"vision_model." + n for n in case["names"]}          # HF (transformers 4.x) state-dict keys
    if precision == "fp16":          # BASELINE config c4's precision: inference only
        from geoguessr_ai_amd import _lib as L
        with pytest.raises(L.GgError, match="inference-only"):
            tower(pixel_values=torch.from_numpy(g["x"]).cuda())
        for p in tower.parameters():
            p.requires_grad = False
    out = tower(pixel_values=torch.from_numpy(g["x"]).cuda())
    y, lh = out.pooled_mean.detach().cpu().numpy(), out.last_hidden_state.detach().cpu().numpy()
    e_y, e_lh = _rel(y, g["y"]), _rel(lh, g["last_hidden_state"])
    print(f"\n[CLIP tiny {precision}] pooled rel-L2 {e_y:.2e} (max abs {np.abs(y - g['y']).max():.2e}), last_hidden rel-L2 {e_lh:.2e}")
    if precision == "fp32":
        assert e_y < 1e-4 and e_lh < 1e-4
        np.testing.assert_allclose(y, g["y"], rtol=1e-4, atol=2e-5)
    elif precision == "fp16":        # 11-bit significand storage, f32 accumulation: an order of magnitude tighter than bf16
        assert np.abs(y - g["y"]).max() < 4e-3 and e_lh < 2e-3
    else:
        assert np.abs(y - g["y"]).max() < 3e-2 and e_lh < 2e-2


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("policy", ["all_layers", "last_layer"])
def test_superguessr_on_clip_training_matches_reference_golden(golden_dir, centroids, precision, policy):
    """SuperGuessr(base_model=CLIPVisionTower, panorama, smooth labels).train(): forward + backward against the reference's own run
    (models/super_guessr.py:134-150,309-383 on a transformers CLIPVisionModel; clip_train.npz).  all_layers = the no-head-file policy the golden
    was produced under; last_layer = what the reference does when the pretrained head exists (layers[:-1] frozen): same gradients for the
    tensors that stay trainable, none for the frozen ones."""
    from geoguessr_ai_amd.models.super_guessr import SuperGuessr
    case = CG.load(golden_dir)
    g = case["g"]
    tower = _tiny_tower(case, precision)
    model = SuperGuessr(base_model=tower, panorama=True, should_smooth_labels=True)
    assert model.mode == "transformer" and model.hidden_size == case["cfg"][0] and model.precision == precision
    assert all(p.requires_grad for p in tower.parameters())               # CLIP branch without the head file: nothing frozen
    if policy == "last_layer":
        for layer in list(tower.vision_model.encoder.layers)[:-1]:
            for p in layer.parameters():
                p.requires_grad = False
    with torch.no_grad():
        model.cell_layer.weight.copy_(case["W"]); model.cell_layer.bias.copy_(case["b"])
    model = model.cuda().train()
    out = model(pixel_values=case["x"].cuda(), labels=case["labels"].cuda(), labels_clf=torch.from_numpy(g["labels_clf"]).cuda())
    out.loss.backward()
    torch.cuda.synchronize()
    loss_rel = abs(float(out.loss.detach()) - float(g["loss"])) / float(g["loss"])
    emb_rel = _rel(out.embedding.detach(), g["embedding"])
    vm = tower.vision_model
    grads = {n: (p.grad if p.grad is not None else None) for n, p in vm._params.items()}
    nl = case["cfg"][2]
    frozen = [n for n, p in vm._params.items() if not p.requires_grad]
    if policy == "last_layer":
        assert frozen and all(n.startswith("encoder.layers.") and int(n.split(".")[2]) < nl - 1 for n in frozen)
        assert all(grads[n] is None for n in frozen)
    live = {n: t for n, t in grads.items() if n not in frozen}
    case_live = dict(case, names=[n for n in case["names"] if n not in frozen])
    errs = CG.grad_errors(case_live, {n: (t if t is not None else torch.zeros_like(vm._params[n])) for n, t in live.items()})
    worst = max(errs, key=errs.get)
    dW_rel = _rel(model.cell_layer.weight.grad[torch.from_numpy(g["labels_clf"]).cuda()], g["dW_rows"])
    print(f"\n[SuperGuessr on CLIP tiny, {precision}, {policy}] loss rel {loss_rel:.2e}, embedding rel-L2 {emb_rel:.2e}, head dW rows {dW_rel:.2e}, "
          f"{len(errs)} tower gradients: worst {worst} {errs[worst]:.2e}, median {float(np.median(list(errs.values()))):.2e}")
    if precision == "fp32":
        assert loss_rel < 1e-5 and emb_rel < 1e-4 and dW_rel < 1e-4
        assert errs[worst] < 1e-4, sorted(errs.items(), key=lambda kv: -kv[1])[:5]          # measured 8e-6
        np.testing.assert_array_equal(out.preds_geocell.cpu().numpy(), g["preds_geocell"])
    else:
        assert loss_rel < 5e-3 and emb_rel < 3e-2 and dW_rel < 5e-2
        assert errs[worst] < 0.15 and float(np.median(list(errs.values()))) < 5e-2, sorted(errs.items(), key=lambda kv: -kv[1])[:5]


def _oracle_case(model_name, cfg_tuple, n_pano, seed, trainable_from, centroids):
    """Random-weight tower of a real configuration + the pinned oracle's training step with the same weights."""
    from geoguessr_ai_amd.pretrain.clip_embedder import CLIPVisionTower
    from oracle import clip_ref as CR
    tower = CLIPVisionTower(model_name, seed=seed, precision="fp32")
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():          # non-trivial LayerNorm affine parameters and biases
        for n, p in tower.vision_model._params.items():
            if n.endswith(("norm.weight", "norm1.weight", "norm2.weight")):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif n.endswith(".bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    st = {k: v.detach().clone() for k, v in tower.named_views().items()}
    cfg = CR.ClipVisionConfig(*cfg_tuple)
    x = torch.randn(n_pano, 4, 3, cfg.image_size, cfg.image_size, generator=g)
    labels = torch.stack([torch.rand(n_pano, generator=g) * 360 - 180, torch.rand(n_pano, generator=g) * 180 - 90], 1)
    W, b = torch.randn(12647, cfg.hidden_size, generator=g) * 0.03, torch.randn(12647, generator=g) * 0.1
    names = [n for n in st if n.startswith("encoder.layers.") and int(n.split(".")[2]) >= trainable_from]
    ref = CR.train_step(cfg, st, W, b, torch.from_numpy(centroids), x, labels, trainable=names)
    return tower, cfg, x, labels, W, b, names, ref


def test_clip_base_patch32_last_layer_finetune_matches_oracle(centroids):
    """The reference's CLIP fine-tune at the real ViT-B/32 shapes (768 wide, 12 layers, 50 tokens): last encoder layer trainable (the policy of
    models/super_guessr.py:140-146), fp32 mode against the pinned oracle: loss, embedding, the 16 gradient tensors of layer 11; then two AdamW
    steps move exactly the trainable range."""
    from geoguessr_ai_amd.models.super_guessr import SuperGuessr
    from geoguessr_ai_amd.optim import AdamW
    tower, cfg, x, labels, W, b, names, ref = _oracle_case("openai/clip-vit-base-patch32", (768, 3072, 12, 12, 224, 32), 2, 5, 11, centroids)
    model = SuperGuessr(base_model=tower, panorama=True, should_smooth_labels=True)
    for n, p in tower.vision_model._params.items():
        p.requires_grad = n in names
    with torch.no_grad():
        model.cell_layer.weight.copy_(W); model.cell_layer.bias.copy_(b)
    model = model.cuda().train()
    opt = AdamW(model, lr=1e-3)
    assert len(opt.backbones) == 1 and opt.backbones[0] is tower.vision_model
    before = tower.vision_model.flat_params.clone()
    out = model(pixel_values=x.cuda(), labels=labels.cuda())
    out.loss.backward()
    errs = {n: _rel(tower.vision_model._params[n].grad, ref["grads"][n]) for n in names if not n.endswith("k_proj.bias")}
    worst = max(errs, key=errs.get)
    loss_rel = abs(float(out.loss.detach()) - float(ref["loss"])) / float(ref["loss"])
    print(f"\n[CLIP B/32 fp32, layer 11 trainable] loss rel {loss_rel:.2e}, embedding rel-L2 {_rel(out.embedding.detach(), ref['embedding']):.2e}, "
          f"{len(errs)} gradients: worst {worst} {errs[worst]:.2e}")
    assert loss_rel < 1e-5 and _rel(out.embedding.detach(), ref["embedding"]) < 1e-4 and errs[worst] < 2e-4      # measured 2e-5
    assert _rel(model.cell_layer.weight.grad, ref["grads"]["cell_layer.weight"]) < 1e-4
    opt.step(); opt.zero_grad()
    out2 = model(pixel_values=x.cuda(), labels=labels.cuda())
    out2.loss.backward(); opt.step()
    torch.cuda.synchronize()
    after = tower.vision_model.flat_params
    (lo, hi), = tower.vision_model.trainable_ranges()
    assert torch.equal(after[:lo], before[:lo]) and torch.equal(after[hi:], before[hi:]) and not torch.equal(after[lo:hi], before[lo:hi])
    assert float(out2.loss.detach()) < float(out.loss.detach())


def test_clip_large_patch14_336_forward_and_finetune_fp32(centroids):
    """openai/clip-vit-large-patch14-336 -- the reference's CLIP_MODEL (config.py:6): 577 tokens (online-softmax attention, resident form off),
    patch 14 (contraction 588 -> padded 592), 24 layers -- in the reference's precision against the (non-emulating) pinned oracle: forward at
    rtol 1e-4, and the last layer's gradients."""
    from geoguessr_ai_amd.models.super_guessr import SuperGuessr
    tower, cfg, x, labels, W, b, names, ref = _oracle_case("openai/clip-vit-large-patch14-336", (1024, 4096, 24, 16, 336, 14), 1, 3, 23, centroids)
    model = SuperGuessr(base_model=tower, panorama=True, should_smooth_labels=True)
    for n, p in tower.vision_model._params.items():
        p.requires_grad = n in names
    with torch.no_grad():
        model.cell_layer.weight.copy_(W); model.cell_layer.bias.copy_(b)
    model = model.cuda().train()
    out = model(pixel_values=x.cuda(), labels=labels.cuda())
    out.loss.backward()
    emb_rel = _rel(out.embedding.detach(), ref["embedding"])
    errs = {n: _rel(tower.vision_model._params[n].grad, ref["grads"][n]) for n in names if not n.endswith("k_proj.bias")}
    worst = max(errs, key=errs.get)
    print(f"\n[CLIP L/14-336 fp32] embedding rel-L2 {emb_rel:.2e}, loss rel {abs(float(out.loss.detach()) / float(ref['loss']) - 1):.2e}, "
          f"layer-23 gradients worst {worst} {errs[worst]:.2e}")
    assert emb_rel < 1e-4 and abs(float(out.loss.detach()) / float(ref["loss"]) - 1) < 1e-5 and errs[worst] < 2e-4
    model.eval()
    with torch.no_grad():
        o = tower(pixel_values=x[0].cuda())
    assert o.last_hidden_state.shape == (4, 577, 1024) and _rel(o.pooled_mean, ref["embedding"][0]) < 1e-4


def test_clip_embedding_wrapper_tensor_and_raw_image_inputs():
    """CLIPEmbedding (pretrain/clip_embedder.py:10-101): float tensors are pixel_values (:58-59), panorama kwargs stack on dim 1 (:94-101); raw
    uint8 images (PIL-style HWC arrays, a list of them, or an NCHW uint8 tensor) go through the processor's pipeline on the device: Pillow bicubic resize
    of the shortest edge, centre crop, * 1/255, CLIP mean / std (bit-identical uint8 stage: tests/test_gpu_kernels.py::test_preprocess_pil_matches_...)."""
    from geoguessr_ai_amd.pretrain.clip_embedder import CLIPEmbedding, CLIP_MEAN, CLIP_STD
    e = CLIPEmbedding("openai/clip-vit-base-patch32", device="cuda", panorama=True, precision="bf16")
    assert not any(p.requires_grad for p in e.parameters()) and not e.training
    xs = [torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(10 + i)) for i in range(4)]
    pano = e(xs[0].cuda(), image_2=xs[1].cuda(), image_3=xs[2].cuda(), image_4=xs[3].cuda())
    assert pano.shape == (2, 4, 768)
    single = e(xs[1].cuda())
    assert single.shape == (2, 768) and torch.allclose(single, pano[:, 1], atol=2e-3)
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (300, 448, 3), dtype=np.uint8)                     # landscape HWC image
    from oracle import preprocess_ref as P                                         # (the Pillow-pinned restatement of the processor)
    _, want = P.raw_image_pixel_values(img, "clip", 224, CLIP_MEAN, CLIP_STD)
    want = torch.from_numpy(want).unsqueeze(0)
    from geoguessr_ai_amd.pretrain.clip_embedder import clip_preprocess
    got = clip_preprocess(img, 224, "cuda")
    assert got.shape == (1, 3, 224, 224) and torch.allclose(got.cpu(), want, atol=1e-6)
    a = e(img)
    b_ = e(want.cuda())
    assert a.shape == (1, 768) and torch.allclose(a, b_, atol=1e-3)
    two = e([img, img[:, ::-1].copy()])
    assert two.shape == (2, 768) and torch.allclose(two[0], a[0], atol=1e-6)
    u8 = torch.from_numpy(img).permute(2, 0, 1)
    assert torch.allclose(e(u8), a, atol=1e-6)


def test_clip_base_patch32_fp16_inference_matches_oracle():
    """BASELINE config c4 (CLIP ViT-B/32 embedder, inference, "MFMA fp16") at the real shapes, batch 8: the fp16 mode (fp16 storage, v_mfma_f32_16x16x32_f16
    GEMMs, f32 accumulation / LayerNorm statistics / softmax) against the pinned fp32 oracle."""
    from geoguessr_ai_amd.pretrain.clip_embedder import CLIPEmbedding
    from oracle import clip_ref as CR
    e = CLIPEmbedding("openai/clip-vit-base-patch32", device="cuda", precision="fp16")
    assert e.clip_model.precision == "fp16"
    st = {k: v.detach().cpu().clone() for k, v in e.clip_model.named_views().items()}
    x = torch.randn(8, 3, 224, 224, generator=torch.Generator().manual_seed(2))
    got = e(x.cuda()).cpu()
    with torch.no_grad():
        ref = CR.forward(CR.ClipVisionConfig(), st, x)
    rel = _rel(got, ref)
    print(f"\n[CLIP B/32 fp16 inference] pooled embedding rel-L2 vs fp32 oracle {rel:.2e}, max abs {float((got - ref).abs().max()):.2e}")
    assert rel < 3e-3


def test_clip_layernorm_gradients_respect_a_half_frozen_pair(golden_dir):
    """A LayerNorm whose weight is frozen and whose bias trains (or the other way round): the trainable tensor's gradient equals the all-trainable
    run's, and NOTHING is written into the frozen tensor's region of the flat gradient buffer (the kernel forms dgamma and dbeta together: the
    frozen half goes to a dump row)."""
    case = CG.load(golden_dir)
    x = torch.from_numpy(np.load(os.path.join(golden_dir, "clip_tiny.npz"))["x"]).cuda()

    def run(freeze):
        tower = _tiny_tower(case, "fp32").cuda().train()
        for n, p in tower.vision_model._params.items():
            p.requires_grad = n not in freeze
        out = tower(pixel_values=x)
        out.pooled_mean.square().sum().backward()
        torch.cuda.synchronize()
        vm = tower.vision_model
        flat = vm.flat_grads().clone()
        return {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in vm._params.items()}, flat, {t["name"]: t for t in vm.table}

    full, _, _ = run(())
    fz = ("encoder.layers.0.layer_norm1.weight", "encoder.layers.1.layer_norm2.bias", "pre_layrnorm.weight")
    part, flat, table = run(fz)
    for n in fz:
        t = table[n]
        assert float(flat[t["offset"]:t["offset"] + t["numel"]].abs().max()) == 0.0, n          # the frozen tensor's slice of the flat buffer is untouched
        twin = n.replace(".weight", ".bias") if n.endswith(".weight") else n.replace(".bias", ".weight")
        assert _rel(part[twin].cpu().numpy(), full[twin].cpu().numpy()) < 1e-5, twin
    other = "encoder.layers.1.mlp.fc1.weight"
    assert _rel(part[other].cpu().numpy(), full[other].cpu().numpy()) < 1e-5
