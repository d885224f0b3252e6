"""Model-level parity on the GPU: whole TinyViT forward / backward, SuperGuessr step, CLIP tower and the drop-in call
surface, against the CPU oracle and the reference-generated golden fixtures.  Everything goes through libgg.so.

Tolerances (stated per SURVEY.md 8c), per arithmetic mode.  fp32 mode (the reference's precision) against the
REFERENCE-generated fixtures: loss rel <= 1e-5, predicted geocell and top-5 indices identical except where two logits lie within
1e-4 of each other, gradients rel-L2 <= 1e-4.  bf16 mode (bf16 storage / MFMA operands, fp32 accumulation): vs the pure-fp32 oracle
embedding |err| <= 6e-2 on unit-variance LayerNorm outputs, loss rel 1e-2 and gradient cosine >= 0.98; vs the bf16-storage-emulating
oracle the same checks are ~3x tighter."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["fp32", "bf16"])
def _precision_mode(request, monkeypatch):
    """Every model-level test of this file runs in both arithmetic modes (the modules read $GG_PRECISION when no precision= is given).
    The reference-golden tests branch on the mode (fp32: SURVEY 8(c)'s fp32 class; bf16: the bf16 class); the oracle-vs-TinyViT tests keep
    one bf16-sized bound here, their tight fp32 bounds are tests/test_gpu_precision.py."""
    monkeypatch.setenv("GG_PRECISION", request.param)
    yield request.param


def _state_from(bb):
    return {k: v.detach().cpu().clone() for k, v in bb.state_dict().items()}


def _randomize(bb, seed=0):
    """Non-trivial norm affine params / attention biases so indexing bugs cannot hide behind 1/0 initial values."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in bb.named_parameters():
            if name.endswith(("bn.weight", "norm.weight")):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith(".bias") and p.dim() == 1:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith("attention_biases"):
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            elif name.endswith(".weight") and p.dim() == 2:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))


def _rel(a, b):
    a, b = torch.as_tensor(a).flatten().double(), torch.as_tensor(b).flatten().double()
    return float((a - b).norm() / (b.norm() + 1e-300))


def _cos(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


_ADAPTERS = {}


@pytest.fixture
def adapter5m(_precision_mode):
    """One TinyViT-5M per arithmetic mode, shared by the tests of that mode (gradients of an earlier test are cleared)."""
    from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
    if _precision_mode not in _ADAPTERS:
        torch.manual_seed(0)
        m = TinyViTAdapter("tiny_vit_5m_224", pretrained=False, drop_path_rate=0.1, precision=_precision_mode)
        _randomize(m.backbone, 1)
        _ADAPTERS[_precision_mode] = m.cuda()
    m = _ADAPTERS[_precision_mode]
    assert m.backbone.precision == _precision_mode
    m.zero_grad(set_to_none=True)
    m.backbone.flat_grads().zero_()
    return m


def test_state_dict_keys_match_timm_table(adapter5m):
    from oracle import tinyvit_ref as R
    cfg = R.config_for("tiny_vit_5m_224")
    want = {n: tuple(s) for n, s, _ in R.param_spec(cfg)}
    got = {k: tuple(v.shape) for k, v in adapter5m.backbone.state_dict().items()}
    assert got == want
    assert all(k.startswith("backbone.") for k in adapter5m.state_dict())
    assert sum(p.numel() for p in adapter5m.parameters()) == 5071764


def test_tinyvit_eval_forward_matches_oracle(adapter5m):
    from oracle import tinyvit_ref as R
    cfg = R.config_for("tiny_vit_5m_224")
    st = _state_from(adapter5m.backbone)
    # give the running statistics realistic values first (one train-mode forward on both sides is avoided: set directly)
    g = torch.Generator().manual_seed(5)
    for k in st:
        if k.endswith("running_mean"):
            st[k] = 0.1 * torch.randn(st[k].shape, generator=g)
        elif k.endswith("running_var"):
            st[k] = 0.5 + torch.rand(st[k].shape, generator=g)
    adapter5m.backbone.load_state_dict(st)
    x = torch.randn(3, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    adapter5m.eval()
    with torch.no_grad():
        out = adapter5m(pixel_values=x.cuda())
    assert out.pooler_output.shape == (3, 320) and out.last_hidden_state.shape == (3, 1, 320)
    ref = R.forward(cfg, st, x, training=False)
    emu = R.forward(cfg, st, x, training=False, emulate_bf16=True)
    got = out.pooler_output.cpu()
    assert float((got - emu).abs().max()) < 4e-2, float((got - emu).abs().max())
    assert float((got - ref).abs().max()) < 8e-2, float((got - ref).abs().max())
    assert _cos(got, ref) > 0.999


@pytest.mark.parametrize("unfrozen", [False, True])
def test_tinyvit_train_step_matches_oracle(adapter5m, centroids, unfrozen):
    """fwd (batch-stat BN, DropPath masks as inputs) + bwd under the reference freeze policy (and with every parameter
    trainable: the unfused BatchNorm / weight-gradient paths of all stages), SuperGuessr head on top."""
    from geoguessr_ai_amd.models.super_guessr import SuperGuessr
    from oracle import tinyvit_ref as R
    from oracle import step_ref as S
    cfg = R.config_for("tiny_vit_5m_224", drop_path_rate=0.1)
    adapter5m.unfreeze_all()
    torch.manual_seed(3)
    model = SuperGuessr(adapter5m, panorama=True, should_smooth_labels=True).cuda().train()
    if unfrozen:
        adapter5m.unfreeze_all()
    trainable = [n for n, p in adapter5m.backbone.named_parameters() if p.requires_grad]
    if unfrozen:
        assert any(n.startswith("stages.0") for n in trainable) and any(n.startswith("stages.2") for n in trainable)
    else:
        assert not any(n.startswith(("stages.0", "stages.1", "stages.2")) for n in trainable)     # freeze_all_but_last_stage
    assert any(n.startswith("patch_embed") for n in trainable) and "head.norm.weight" in trainable
    N = 3
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, 4, 3, 224, 224, generator=g)
    labels = torch.stack([torch.rand(N, generator=g) * 360 - 180, torch.rand(N, generator=g) * 180 - 90], 1)
    bb = adapter5m.backbone
    keep = (torch.rand(bb.num_drop_slots, 4 * N, generator=g) > 0.3)
    rates = torch.tensor(bb.drop_rates).unsqueeze(1)
    scales = (keep.float() / (1 - rates)).contiguous().cuda()
    bb.make_drop_scales = lambda batch, generator=None: scales               # inject the masks the oracle will use
    st = _state_from(bb)
    W, b = model.cell_layer.weight.detach().cpu().clone(), model.cell_layer.bias.detach().cpu().clone()
    out = model(pixel_values=x.cuda(), labels=labels.cuda(), labels_clf=None)
    out.loss.backward()
    torch.cuda.synchronize()
    masks = [keep[s] for s in range(bb.num_drop_slots)]
    ref = S.train_step(cfg, st, W, b, torch.from_numpy(centroids), x, labels, drop_masks=masks, trainable=trainable)
    emb = out.embedding.detach().cpu()
    assert emb.shape == (N, 4, 320)
    assert float((emb - ref["embedding"]).abs().max()) < 0.15, float((emb - ref["embedding"]).abs().max())
    assert _cos(emb, ref["embedding"]) > 0.998
    assert abs(float(out.loss) - float(ref["loss"])) / float(ref["loss"]) < 1e-2
    bad = []
    for name, gref in ref["grads"].items():
        p = model.cell_layer.weight if name == "cell_layer.weight" else model.cell_layer.bias if name == "cell_layer.bias" else bb._params[name]
        assert p.grad is not None, name
        c = _cos(p.grad.cpu(), gref)
        ratio = float(p.grad.norm().cpu() / (gref.norm() + 1e-30))
        if float(gref.norm()) > 1e-7 and (c < 0.97 or not (0.9 < ratio < 1.1)):
            bad.append((name, round(c, 4), round(ratio, 3)))
    assert not bad, bad
    # frozen tensors got no gradient
    assert all(bb._params[n].grad is None for n in bb._params if n not in trainable)
    if unfrozen:
        return
    # BN running statistics moved like torch's (momentum 0.1, unbiased variance)
    ref2 = R.forward(cfg, st2 := {k: v.clone() for k, v in st.items()}, x.view(-1, 3, 224, 224), training=True, drop_masks=masks, update_running=True)
    got_rm = bb.state_dict()["patch_embed.conv1.bn.running_mean"].cpu()
    assert torch.allclose(got_rm, st2["patch_embed.conv1.bn.running_mean"], rtol=2e-2, atol=2e-3)
    assert int(bb.state_dict()["patch_embed.conv1.bn.num_batches_tracked"]) == 1


def test_superguessr_head_matches_reference_golden(golden_dir, centroids, _precision_mode):
    """Embeddings-only SuperGuessr (config c5) against outputs of the REAL reference (tests/golden/head.npz; models/super_guessr.py:347-383).
    fp32 mode: SURVEY 8(c)'s fp32 class -- loss rel 1e-5, geocell / top-5 indices identical outside logit gaps < 1e-4, gradients rel-L2 1e-4."""
    from geoguessr_ai_amd.models.super_guessr import SuperGuessr
    from oracle import geo_ref as G
    f32 = _precision_mode == "fp32"
    g = np.load(os.path.join(golden_dir, "head.npz"))
    rng = np.random.default_rng(int(g["seed"]))
    W = rng.standard_normal((12647, 576), dtype=np.float32) * np.float32(0.05)
    b = rng.standard_normal((12647,), dtype=np.float32) * np.float32(0.1)
    emb = rng.standard_normal((32, 4, 576), dtype=np.float32)
    model = SuperGuessr(base_model=None, panorama=True, should_smooth_labels=True, embed_dim=576).cuda().train()
    assert model.precision == _precision_mode
    with torch.no_grad():
        model.cell_layer.weight.copy_(torch.from_numpy(W)); model.cell_layer.bias.copy_(torch.from_numpy(b))
    e = torch.from_numpy(emb).cuda().requires_grad_(True)
    out = model(embedding=e, labels=torch.from_numpy(g["labels"]).cuda(), labels_clf=torch.from_numpy(g["labels_clf"]).cuda())
    out.loss.backward()
    loss_rel = abs(float(out.loss) - float(g["loss"])) / float(g["loss"])
    preds, top5 = out.preds_geocell.cpu().numpy(), out.top5_geocells.indices.cpu().numpy()
    dW = model.cell_layer.weight.grad.cpu().numpy()
    demb_rel = _rel(e.grad.cpu(), g["demb"])
    dW_rel = _rel(dW[g["labels_clf"][:8]], g["dW_rows"])
    print(f"head[{_precision_mode}]: loss rel {loss_rel:.2e}, demb rel-L2 {demb_rel:.2e}, dW rows rel-L2 {dW_rel:.2e}")
    if f32:
        assert loss_rel < 1e-5, loss_rel
        # rows whose six largest reference logits are separated by >= 1e-4 must agree exactly (the pinned oracle supplies the gaps)
        z = np.sort(G.head_forward(emb, W, b, centroids)["logits"], axis=-1)[:, ::-1][:, :6]
        clear = (z[:, :-1] - z[:, 1:]).min(-1) >= 1e-4
        assert clear.mean() > 0.9
        np.testing.assert_array_equal(preds[clear], g["preds_geocell"][clear])
        np.testing.assert_array_equal(top5[clear], g["top5_idx"][clear])
        np.testing.assert_allclose(out.top5_geocells.values.detach().cpu().numpy()[clear], g["top5_vals"][clear], rtol=1e-4)
        assert demb_rel < 1e-4 and dW_rel < 1e-4, (demb_rel, dW_rel)
        sum_tol = 1e-4
    else:
        assert loss_rel < 2e-3
        agree = (preds == g["preds_geocell"]).mean()
        assert agree >= 0.9, agree                                   # bf16 logits: near-ties may flip
        ov = np.mean([len(set(a) & set(r)) for a, r in zip(top5, g["top5_idx"])])
        assert ov >= 4.5, ov
        assert _cos(e.grad.cpu(), torch.from_numpy(g["demb"])) > 0.995
        assert _cos(torch.from_numpy(dW[g["labels_clf"][:8]]), torch.from_numpy(g["dW_rows"])) > 0.995
        sum_tol = 2e-2
    same = preds == g["preds_geocell"]
    np.testing.assert_allclose(out.preds_LLH.cpu().numpy()[same], g["preds_LLH"][same])
    np.testing.assert_allclose(np.abs(dW).astype(np.float64).sum(), float(g["dW_abs_sum"]), rtol=sum_tol)
    np.testing.assert_allclose(np.abs(model.cell_layer.bias.grad.cpu().numpy()).astype(np.float64).sum(), float(g["db_abs_sum"]), rtol=sum_tol)
    # hard-label CE and serving return
    model.should_smooth_labels = False
    out_h = model(embedding=torch.from_numpy(emb).cuda(), labels=torch.from_numpy(g["labels"]).cuda(), labels_clf=torch.from_numpy(g["labels_clf"]).cuda())
    assert abs(float(out_h.loss) - float(g["loss_hard"])) / float(g["loss_hard"]) < (1e-5 if f32 else 2e-3)
    model.serving = True
    model.eval()
    llh, topk, embedding = model(embedding=torch.from_numpy(emb).cuda(), labels_clf=None)
    assert llh.shape == (32, 2) and topk.indices.shape == (32, 5) and embedding.shape == (32, 4, 576)


def test_training_trace_matches_reference_golden(golden_dir, _precision_mode):
    """3 steps of the legacy loop contract (AdamW lr 2e-5, training/train_eval_loop.py:188-190,233-242) on embeddings vs the reference
    trace (train_trace.npz).  fp32 mode: losses rel 1e-5, the weight movement and the final bias values at fp32 rounding."""
    from geoguessr_ai_amd.models.super_guessr import SuperGuessr
    from geoguessr_ai_amd.optim import AdamW
    f32 = _precision_mode == "fp32"
    g = np.load(os.path.join(golden_dir, "train_trace.npz"))
    rng = np.random.default_rng(int(g["seed"]))
    W0 = rng.standard_normal((12647, 576), dtype=np.float32) * np.float32(0.02)
    emb_t = rng.standard_normal((3, 64, 4, 576), dtype=np.float32)
    lab3 = np.stack([rng.uniform(-180, 180, (3, 64)), rng.uniform(-90, 90, (3, 64))], -1).astype(np.float32)
    model = SuperGuessr(base_model=None, panorama=True, should_smooth_labels=True, embed_dim=576).cuda().train()
    with torch.no_grad():
        model.cell_layer.weight.copy_(torch.from_numpy(W0)); model.cell_layer.bias.zero_()
    opt = AdamW(model, lr=float(g["lr"]))
    losses = []
    for s in range(3):
        out = model(embedding=torch.from_numpy(emb_t[s]).cuda(), labels=torch.from_numpy(lab3[s]).cuda())
        out.loss.backward(); opt.step(); opt.zero_grad()
        losses.append(float(out.loss))
    Wf = model.cell_layer.weight.detach().cpu().numpy()
    delta = np.abs(Wf - W0).astype(np.float64).sum()
    bf = model.cell_layer.bias.detach().cpu().numpy()[:64]
    print(f"trace[{_precision_mode}]: loss rel {np.abs(np.asarray(losses) / g['losses'] - 1).max():.2e}, "
          f"|dW| rel {abs(delta / float(g['W_delta_abs_sum']) - 1):.2e}, b_final rel-L2 {_rel(bf, g['b_final']):.2e}")
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-5 if f32 else 2e-3)
    np.testing.assert_allclose(delta, float(g["W_delta_abs_sum"]), rtol=1e-3 if f32 else 5e-2)
    if f32:
        assert _rel(bf, g["b_final"]) < 1e-3
        assert abs(float(Wf.astype(np.float64).sum()) - float(g["W_final_checksum"])) < 1e-3 * delta


def test_no_cpu_fallback():
    from geoguessr_ai_amd import _lib as L
    from geoguessr_ai_amd.models.super_guessr import SuperGuessr
    m = SuperGuessr(base_model=None, panorama=True, embed_dim=576)          # parameters left on the CPU
    with pytest.raises(L.GgError):
        m(embedding=torch.zeros(2, 4, 576), labels_clf=torch.zeros(2, dtype=torch.int64))


def test_checkpoint_round_trip_in_reference_format(adapter5m, centroids, tmp_path):
    """make_state / load_model_state / AdamW <-> torch.optim.AdamW state-dict layout (reference checkpoint compatibility, f4):
    a model + optimizer restored from the saved dict continue on the same trajectory (the attention-bias gradient is summed
    with atomics, so two runs agree to rounding, not bit for bit)."""
    from geoguessr_ai_amd.checkpoint import make_state, load_model_state, adamw_state_from_torch, CheckpointKeeper
    from geoguessr_ai_amd.models.super_guessr import SuperGuessr
    from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
    from geoguessr_ai_amd.optim import AdamW

    def build():
        torch.manual_seed(3)
        base = TinyViTAdapter("tiny_vit_5m_224", pretrained=False)
        return SuperGuessr(base, panorama=True, should_smooth_labels=True, centroids=centroids[:64]).cuda().train()

    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(2, 4, 3, 224, 224, device="cuda", generator=g)
    lab = torch.tensor([[10.0, 50.0], [-70.0, -20.0]], device="cuda")

    def step(model, opt):
        out = model(pixel_values=x, labels=lab)
        out.loss.backward(); opt.step(); opt.zero_grad()

    m1 = build(); o1 = AdamW(m1, lr=1e-3)
    for _ in range(2):
        step(m1, o1)
    state = make_state(m1, o1, None, epoch=0, global_step=2, best_value=1.0, monitored_value=1.0, config={"lr": 1e-3})
    sd = state["optimizer_state_dict"]
    assert set(sd) == {"state", "param_groups"} and sd["param_groups"][0]["params"] == list(range(len(list(m1.parameters()))))
    trainable = [i for i, p in enumerate(m1.parameters()) if p.requires_grad]
    assert sorted(sd["state"]) == trainable and all(set(v) == {"step", "exp_avg", "exp_avg_sq"} for v in sd["state"].values())
    assert all(k.startswith(("base_model.backbone.", "cell_layer.", "geocell_centroid_coords")) for k in state["model_state_dict"])
    keeper = CheckpointKeeper(str(tmp_path), keep_last_n=1)
    paths = keeper.update(state, 0, 1.0)
    m2 = build()
    with torch.no_grad():
        for p in m2.parameters():
            p.add_(0.5) if p.requires_grad else None          # make sure the load really restores values
    rep = load_model_state(m2, paths["last"])
    assert not rep["skipped"]
    o2 = AdamW(m2, lr=5.0)
    adamw_state_from_torch(o2, torch.load(paths["last"], weights_only=False)["optimizer_state_dict"])
    assert o2.step_count == 2 and o2.param_groups[0]["lr"] == 1e-3
    step(m1, o1); step(m2, o2)
    for (n1, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.allclose(p1, p2, rtol=1e-4, atol=1e-6), n1


def test_split_serving_head_matches_the_fp32_head(centroids):
    """SuperGuessr(precision="fp32_split") serving head on precomputed embeddings (BASELINE c5's shape: 4096 x 4 x 576 -> 12647 cells): the geocell Linear runs as a
    split-bf16 product (weight planes cached per parameter version); the top-5 probabilities agree with the f32-MFMA head to 1e-4 relative, the top cell and its
    coordinates are identical wherever the two leading probabilities differ by more than 0.1 %, and an in-place weight update is picked up."""
    from geoguessr_ai_amd.models.super_guessr import SuperGuessr
    torch.manual_seed(3)
    emb = torch.randn(4096, 4, 576, device="cuda")
    heads = {}
    for prec in ("fp32", "fp32_split"):
        torch.manual_seed(5)
        heads[prec] = SuperGuessr(None, panorama=True, serving=True, embed_dim=576, precision=prec).cuda().eval()
    heads["fp32_split"].load_state_dict(heads["fp32"].state_dict())
    assert heads["fp32_split"].split and not heads["fp32"].split
    for it in range(2):
        with torch.no_grad():
            ref = heads["fp32"](embedding=emb); got = heads["fp32_split"](embedding=emb)
        llh_r, top_r = ref[0].float(), ref[1]                  # predicted (lon, lat) of the top cell; top-5 probabilities and cells
        llh_g, top_g = got[0].float(), got[1]
        assert float(((top_g.values - top_r.values).abs() / top_r.values).max()) < 1e-4
        clear = (top_r.values[:, 0] - top_r.values[:, 1]) > 1e-3 * top_r.values[:, 0]
        assert float(clear.float().mean()) > 0.9
        assert bool((top_g.indices[clear, 0] == top_r.indices[clear, 0]).all()) and torch.equal(llh_g[clear], llh_r[clear])
        with torch.no_grad():                                   # an optimizer-style in-place update: both caches must follow
            for h in heads.values():
                h.cell_layer.weight.mul_(1.5)

