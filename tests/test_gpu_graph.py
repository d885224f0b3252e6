"""Captured HIP graphs for launch-bound calls (csrc/graph.cpp): a replay is the same launch sequence with the same arguments, so its results are bit-identical
to the eager call's -- inference and a whole training step, with the inputs' CONTENTS changing between replays (a graph holds addresses, not values)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _stats():
    from geoguessr_ai_amd import _lib as L
    a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
    n = L.lib().gg_graph_stats(C.byref(a), C.byref(b), C.byref(c))
    return n, a.value, b.value, c.value


@pytest.fixture
def graph_mode():
    from geoguessr_ai_amd import _lib as L
    L.lib().gg_graph_clear()
    yield lambda m: L.check(L.lib().gg_graph_set_mode(m), "gg_graph_set_mode")
    L.lib().gg_graph_set_mode(-1)
    L.lib().gg_graph_clear()


def test_serving_forward_replays_bit_identically(graph_mode):
    from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
    torch.manual_seed(0)
    m = TinyViTAdapter("tiny_vit_5m_224", pretrained=False, precision="fp32").cuda().eval()
    xs = [torch.randn(4, 3, 224, 224, device="cuda", generator=torch.Generator(device="cuda").manual_seed(s)) for s in range(3)]
    x = torch.empty_like(xs[0])
    out = torch.empty(4, m.backbone.num_features if hasattr(m.backbone, "num_features") else 320, device="cuda")
    graph_mode(0)
    ref = []
    with torch.no_grad():
        for v in xs:
            x.copy_(v)
            ref.append(m(pixel_values=x).pooler_output.clone())
    graph_mode(-1)                                       # auto: 4 images are launch-bound
    _, cap0, rep0, _ = _stats()
    got = []
    with torch.no_grad():
        for rnd in range(3):                             # round 0: eager (first sighting) / capture, later rounds replay -- same addresses, new contents
            for v in xs:
                x.copy_(v)
                o = m(pixel_values=x).pooler_output
                got.append(o.clone())
                del o                                    # the output block returns to the allocator: the next call gets the same address
    _, cap1, rep1, _ = _stats()
    assert cap1 - cap0 >= 1 and rep1 - rep0 >= 6, (cap0, cap1, rep0, rep1)
    for i, o in enumerate(got):
        assert torch.equal(o, ref[i % 3]), i


def test_split_mode_forward_replays_bit_identically(graph_mode):
    """The fp32_split mode at a size where its Linears run on the split-bf16 GEMMs (96 images: thousands of tiles per launch), forced into graphs: the captured
    call and its replays return the eager call's bits (the split kernels hold no state that a capture would freeze: weight planes live in the weight cache)."""
    from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
    torch.manual_seed(0)
    m = TinyViTAdapter("tiny_vit_5m_224", pretrained=False, precision="fp32_split").cuda().eval()
    xs = [torch.randn(96, 3, 224, 224, device="cuda", generator=torch.Generator(device="cuda").manual_seed(s)) for s in range(2)]
    x = torch.empty_like(xs[0])
    graph_mode(0)
    ref = []
    with torch.no_grad():
        for v in xs:
            x.copy_(v)
            ref.append(m(pixel_values=x).pooler_output.clone())
    graph_mode(1)
    _, cap0, rep0, _ = _stats()
    got = []
    with torch.no_grad():
        for rnd in range(3):
            for v in xs:
                x.copy_(v)
                o = m(pixel_values=x).pooler_output
                got.append(o.clone())
                del o
    _, cap1, rep1, _ = _stats()
    assert cap1 - cap0 >= 1 and rep1 - rep0 >= 4, (cap0, cap1, rep0, rep1)
    for i, o in enumerate(got):
        assert torch.equal(o, ref[i % 2]), i


def test_training_step_replays_bit_identically(graph_mode, centroids):
    """forward + backward of the encoder as two graphs (the head, the loss and AdamW are single launches): losses and every parameter after three steps
    equal the eager run's bit for bit (BatchNorm running statistics and the batch counter included; the attention-bias tables are frozen, their gradient
    is accumulated with atomics)."""
    from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
    from geoguessr_ai_amd.models.super_guessr import SuperGuessr
    from geoguessr_ai_amd.optim import AdamW

    def run(mode):
        graph_mode(mode)
        torch.manual_seed(0)
        base = TinyViTAdapter("tiny_vit_5m_224", pretrained=False, precision="fp32", drop_path_rate=0.0)
        model = SuperGuessr(base, panorama=False, should_smooth_labels=False, serving=False, centroids=centroids).cuda().train()
        base.freeze_all_but_last_stage()
        for n, p_ in base.backbone._params.items():
            if n.endswith("attention_biases"):
                p_.requires_grad_(False)             # their gradient is the one atomic accumulation of the step: everything else repeats bit for bit
        opt = AdamW(model, lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)
        g = torch.Generator(device="cuda").manual_seed(7)
        x = torch.empty(8, 3, 224, 224, device="cuda")
        lab = torch.stack([torch.rand(8, device="cuda", generator=g) * 360 - 180, torch.rand(8, device="cuda", generator=g) * 180 - 90], 1)
        clf = torch.randint(0, model.num_cells, (8,), device="cuda", generator=g)
        losses = []
        for step in range(4):
            x.copy_(torch.randn(8, 3, 224, 224, device="cuda", generator=g))
            o = model(pixel_values=x, labels=lab, labels_clf=clf)
            o.loss.backward(); opt.step(); opt.zero_grad()
            losses.append(float(o.loss))
            del o
        bb = base.backbone
        return losses, bb._flat.clone(), bb._flat_buf.clone(), bb._counters.clone()
    l0, p0, b0, c0 = run(0)
    _, cap0, rep0, _ = _stats()
    l1, p1, b1, c1 = run(-1)
    _, cap1, rep1, _ = _stats()
    assert cap1 - cap0 >= 2 and rep1 - rep0 >= 4, (cap0, cap1, rep0, rep1)       # a forward and a backward graph, replayed on steps 2 and 3
    assert l0 == l1 and torch.equal(p0, p1) and torch.equal(b0, b1) and torch.equal(c0, c1)


def test_event_timing_turns_graphs_off(graph_mode):
    from geoguessr_ai_amd import _lib as L
    from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
    m = TinyViTAdapter("tiny_vit_5m_224", pretrained=False, precision="fp32").cuda().eval()
    x = torch.randn(4, 3, 224, 224, device="cuda")
    graph_mode(1)
    L.lib().gg_prof_enable(1)
    try:
        n0 = _stats()[0]
        with torch.no_grad():
            for _ in range(3):
                m(pixel_values=x)
        assert _stats()[0] == n0 and L.lib().gg_prof_count() > 0
    finally:
        L.lib().gg_prof_enable(0); L.lib().gg_prof_reset()


def test_address_churn_stops_capturing(graph_mode):
    """A caller whose buffers change address all the time (30 input buffers, each used twice in a row: every second call captures, then the key is never
    seen again) must not pay for captures forever: once 8 captured graphs have been evicted unused the cache stops capturing; results stay those of the
    eager path throughout."""
    from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
    torch.manual_seed(0)
    m = TinyViTAdapter("tiny_vit_5m_224", pretrained=False, precision="fp32").cuda().eval()
    src = torch.randn(4, 3, 224, 224, device="cuda")
    graph_mode(0)
    with torch.no_grad():
        ref = m(pixel_values=src).pooler_output.clone()
    graph_mode(-1)
    bufs = [src.clone() for _ in range(30)]
    _, cap0, _, _ = _stats()
    with torch.no_grad():
        for b in bufs:
            for _ in range(2):
                o = m(pixel_values=b).pooler_output
                assert torch.equal(o, ref)
                del o
    _, cap1, _, _ = _stats()
    assert 16 <= cap1 - cap0 <= 16 + 8, (cap0, cap1)
