"""CPU tests of the "next" rows: oracle vs the reference-generated golden, the embedding store's
on-disk format, the checkpoint retention policy.  No GPU, no libgg compute calls."""
import os
import sqlite3

import numpy as np
import torch

from oracle import preprocess_ref as P

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def test_oracle_preprocess_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "preprocess.npz"))
    for name in ("pano_down", "single_up", "same_size", "resize_only"):
        size = tuple(int(v) for v in g[name + ".size"])
        norm = bool(g[name + ".norm"][0])
        y = P.prepare_batch(g[name + ".x"], None if size[0] < 0 else size, MEAN if norm else None, STD if norm else None)
        assert y.shape == g[name + ".y"].shape
        np.testing.assert_allclose(y, g[name + ".y"], rtol=0, atol=1e-5, err_msg=name)       # fp32 association only
    np.testing.assert_array_equal(P.cluster_mean(g["proto.pano_vec"], g["proto.ptr"], g["proto.member"]), g["proto.out"])


def test_oracle_preprocess_edges():
    x = np.arange(2 * 3 * 1 * 1, dtype=np.float32).reshape(2, 3, 1, 1)
    y = P.prepare_batch(x, (4, 5))                                                           # 1x1 source: constant planes
    assert y.shape == (2, 3, 4, 5) and np.all(y == x)
    u8 = np.full((1, 3, 2, 2), 255, np.uint8)
    np.testing.assert_allclose(P.prepare_batch(u8, None, MEAN, STD)[0, :, 0, 0], (1 - np.asarray(MEAN)) / np.asarray(STD), rtol=1e-6)


def test_embedding_store_layout_matches_reference_schema(tmp_path):
    from geoguessr_ai_amd.embedding_store import EmbeddingWriter, read_embeddings
    db = str(tmp_path / "emb.sqlite")
    rng = np.random.default_rng(0)
    emb = rng.standard_normal((5, 576)).astype(np.float32)
    recs = [dict(location_id=f"loc{i // 2}", lat=10.0 + i, lon=-20.0 - i, heading=(i % 2) * 90, pano_id=f"p{i}") for i in range(5)]
    with EmbeddingWriter(db) as w:
        assert w.write_batch(recs, torch.from_numpy(emb)) == 5
        w.write_batch([dict(recs[0], lat=99.0)], emb[4:5])                                   # INSERT OR REPLACE on (location_id, heading)
    conn = sqlite3.connect(db)
    cols = [r[1] for r in conn.execute("PRAGMA table_info(samples)")]
    assert cols == ["location_id", "lat", "lon", "heading", "capture_date", "pano_id", "batch_date", "embedding", "embedding_dim"]
    blob, dim = conn.execute("SELECT embedding, embedding_dim FROM samples WHERE location_id='loc1' AND heading=90").fetchone()
    assert dim == 576 and bytes(blob) == emb[3].astype("<f4").tobytes()                      # raw little-endian float32
    assert conn.execute("SELECT COUNT(*) FROM samples").fetchone()[0] == 5
    conn.close()
    r2, e2 = read_embeddings(db)
    assert len(r2) == 5 and e2.shape == (5, 576)
    first = next(i for i, r in enumerate(r2) if r["location_id"] == "loc0" and r["heading"] == 0)
    assert r2[first]["lat"] == 99.0 and np.array_equal(e2[first], emb[4])


def test_checkpoint_keeper_topk_policy(tmp_path):
    from geoguessr_ai_amd.checkpoint import CheckpointKeeper
    d = str(tmp_path / "ckpt")
    saved = []
    keeper = CheckpointKeeper(d, keep_last_n=2, save_every_epochs=1, monitor_mode="min",
                              save_fn=lambda state, path: (saved.append(os.path.basename(path)), open(path, "w").write("x")))
    values = [0.9, 0.7, 0.8, 0.5, 0.95]
    results = [keeper.update({"epoch": e}, e, v) for e, v in enumerate(values)]
    files = sorted(f for f in os.listdir(d) if f.startswith("epoch_"))
    assert files == ["epoch_0001_0.700000.pt", "epoch_0003_0.500000.pt"]                      # the two lowest losses survive
    assert results[2]["epoch"] is None or not os.path.exists(results[2]["epoch"])             # 0.8 displaced 0.9, then was pruned by 0.5
    assert results[4]["epoch"] is None                                                       # worse than the kept set: not written
    assert [r["improved"] for r in results] == [True, True, False, True, False]
    assert os.path.exists(os.path.join(d, "last.pt")) and os.path.exists(os.path.join(d, "best.pt"))
    assert keeper.best_value == 0.5
    # max mode + save_every_epochs
    d2 = str(tmp_path / "ckpt2")
    k2 = CheckpointKeeper(d2, keep_last_n=1, save_every_epochs=2, monitor_mode="max", save_fn=lambda s, p: open(p, "w").write("x"))
    for e, v in enumerate([0.1, 0.3, 0.9, 0.2]):
        k2.update({}, e, v)
    assert sorted(f for f in os.listdir(d2) if f.startswith("epoch_")) == ["epoch_0001_0.300000.pt"]   # epochs 1 and 3 are eligible
    assert k2.best_value == 0.9


def test_load_model_state_filters_by_shape():
    from geoguessr_ai_amd.checkpoint import load_model_state
    m = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    sd = {"0.weight": torch.ones(3, 4), "0.bias": torch.ones(5), "extra": torch.zeros(1)}
    rep = load_model_state(m, {"model_state_dict": sd})
    assert rep["loaded"] == ["0.weight"] and "0.bias" in rep["skipped"] and "extra" in rep["skipped"]
    assert torch.equal(m[0].weight.data, torch.ones(3, 4))


def test_compute_summary_keeps_sentinel_scores_out_of_the_averages():
    """gg_geoguessr_score marks non-finite coordinate pairs with distance NaN / score -1: the summary must not average them, and -- like the
    reference, whose epoch carries on with NaN means -- must not abort a run either (strict=True raises)."""
    import pytest
    from geoguessr_ai_amd.scoring import compute_summary
    ok = compute_summary([10.0, 20.0], [4000, 3000])
    assert ok["avg_score"] == 3500.0 and ok["num_samples"] == 2 and "num_invalid" not in ok
    with pytest.warns(UserWarning, match="sentinel"):
        part = compute_summary([10.0, float("nan"), 30.0], [4000, -1, 2000], [0.5, 0.9, 0.1])
    # num_samples is the TOTAL, as in the reference (run_benchmark.py:67-117); the averages are over num_valid
    assert part["num_samples"] == 3 and part["num_valid"] == 2 and part["num_invalid"] == 1 and part["avg_score"] == 3000.0 and part["avg_distance_km"] == 20.0
    assert abs(part["avg_top1_prob"] - 0.3) < 1e-12
    with pytest.warns(UserWarning):
        none = compute_summary([float("nan")], [-1])
    assert none["num_samples"] == 1 and none["num_valid"] == 0 and none["num_invalid"] == 1 and none["avg_score"] != none["avg_score"]
    with pytest.raises(ValueError, match="sentinel"):
        compute_summary([10.0, 5.0], [4000, -1], strict=True)


def test_raw_image_inputs_are_converted_to_rgb():
    """PIL images of any mode go through ``convert("RGB")`` like CLIPProcessor / timm's transform; arrays of another layout are refused by name."""
    import numpy as np
    import pytest
    from geoguessr_ai_amd.training.preprocess import _rgb_hwc
    from geoguessr_ai_amd._lib import GgError
    class FakePIL:                      # the two attributes the path reads (Pillow is optional in this image)
        mode = "L"
        def __init__(self): self.converted = None
        def convert(self, mode):
            self.converted = mode
            return np.zeros((5, 7, 3), np.uint8)
    im = FakePIL()
    t = _rgb_hwc(im)
    assert im.converted == "RGB" and tuple(t.shape) == (5, 7, 3)
    with pytest.raises(GgError, match="raw image"):
        _rgb_hwc(np.zeros((5, 7), np.uint8))
    with pytest.raises(GgError, match="raw image"):
        _rgb_hwc(np.zeros((5, 7, 4), np.uint8))


def test_drop_path_state_travels_with_the_checkpoint():
    from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
    from geoguessr_ai_amd import checkpoint as CK
    import torch
    m = TinyViTAdapter("tiny_vit_21m_224", pretrained=False)
    bb = m.backbone
    bb._drop_seed, bb._drop_counter = 1234567, 42
    st = CK.make_state(m, torch.optim.SGD(m.parameters(), lr=0.1), None, 0, 0, 0.0, 0.0)
    assert "gg_drop_path_state" in st and not any("extra_state" in k for k in st["model_state_dict"])       # state_dict keys stay the timm contract
    m2 = TinyViTAdapter("tiny_vit_21m_224", pretrained=False)
    CK.restore_drop_path_state(m2, st)
    assert (m2.backbone._drop_seed, m2.backbone._drop_counter) == (1234567, 42)
    CK.restore_drop_path_state(m2, {"epoch": 1})            # a checkpoint without the key (the reference's own) is fine


def test_raw_image_geometry_matches_the_oracle_rules():
    """The host arithmetic of the three raw-image pipelines (shortest edge, long edge truncation, crop rounding) is the oracle's, which the Pillow / transformers
    fixture pins (tests/test_oracle_geo.py::test_pil_pipelines_match_pillow_and_transformers)."""
    from geoguessr_ai_amd.training.preprocess import raw_image_geometry
    from oracle import preprocess_ref as P
    for h, w in ((301, 452), (381, 233), (224, 224), (226, 230), (240, 531), (1000, 333), (225, 4000)):
        for pipe, size, kw in (("clip", 224, {}), ("timm", 224, dict(crop_pct=0.95)), ("timm", 512, dict(crop_pct=1.0, crop_mode="squash")), ("timm", 384, dict(crop_pct=1.0)),
                               ("torchvision", 336, {}), ("torchvision", 512, {})):
            flt, resized, crop = raw_image_geometry(h, w, pipe, size, **kw)
            oflt, oresized, ocrop, _ = P.raw_image_geometry(h, w, pipe, size, **kw)
            assert (flt, resized, crop) == (oflt, oresized, ocrop), (h, w, pipe, size)
