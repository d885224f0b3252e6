"""CPU-only checks of the boundary: libgg.so builds for gfx950, loads without a GPU, exports every symbol that
include/gg.h declares, reports consistent parameter tables, and refuses to run without a device (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from geoguessr_ai_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib


def test_header_symbols_exported_and_bound(L):
    hdr = open(os.path.join(ROOT, "include", "gg.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)          # declarations only: comments may mention entry points by name
    declared = set(re.findall(r"\b(gg_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(L.SYMBOLS)
    lib = L.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.gg_version() == 1
    # argument counts of the ctypes signatures agree with the header
    for n in L.SYMBOLS:
        m = re.search(r"\b" + n + r"\s*\(([^;]*?)\)\s*;", hdr, re.S)
        args = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
        cnt = 0 if args.strip() in ("void", "") else len(args.split(","))
        assert cnt == len(L.SIGNATURES[n][1]), n


def test_struct_layouts_match_header(L):
    """sizeof() of the ctypes mirrors == sizeof of the C structs (compiled with the host compiler)."""
    import subprocess, tempfile
    src = '#include <stdio.h>\n#include "gg.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n",sizeof(GgGemmArgs),sizeof(GgAttnArgs),' \
          'sizeof(GgGeoHeadArgs),sizeof(GgProtoRefineArgs),sizeof(GgTinyVitCfg),sizeof(GgClipCfg),sizeof(GgSplit3Args));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        sizes = [int(v) for v in subprocess.check_output([os.path.join(d, "t")]).split()]
    got = [C.sizeof(x) for x in (L.GemmArgs, L.AttnArgs, L.GeoHeadArgs, L.ProtoRefineArgs, L.TinyVitCfg, L.ClipCfg, L.Split3Args)]
    assert got == sizes


def test_tinyvit_table_matches_oracle_spec(L):
    from geoguessr_ai_amd.models.tinyvit import make_cfg, _tensor_table
    from oracle import tinyvit_ref as R
    for name in ("tiny_vit_5m_224", "tiny_vit_11m_224", "tiny_vit_21m_224"):
        cfg, _, _ = make_cfg(name)
        table = _tensor_table(cfg)
        spec = R.param_spec(R.config_for(name))
        assert [(t["name"], t["shape"]) for t in table] == [(n, tuple(s)) for n, s, _ in spec]
        assert sum(t["numel"] for t in table if t["kind"] == 0) == R.num_params(R.config_for(name))
        offs = [t["offset"] for t in table if t["kind"] == 0]
        assert offs == sorted(offs) and all(o % 8 == 0 for o in offs)
        ws_train = L.lib().gg_tinyvit_workspace_bytes(C.byref(cfg), 8, 1)
        ws_eval = L.lib().gg_tinyvit_workspace_bytes(C.byref(cfg), 8, 0)
        assert 0 < ws_eval < ws_train
    # the reference's own default (config.py:9) and the 384 variant: 32x32 / 24x24-token windows run on the online-softmax kernels
    for name, ws2 in (("tiny_vit_21m_512", 1024), ("tiny_vit_21m_384", 576)):
        for prec in ("bf16", "fp32"):
            cfg, _, _ = make_cfg(name, precision=prec)
            assert L.lib().gg_tinyvit_num_tensors(C.byref(cfg)) == 294
            t = [t for t in _tensor_table(cfg) if t["name"] == "stages.2.blocks.0.attn.attention_biases"][0]
            assert t["shape"] == (12, ws2)
            assert L.lib().gg_tinyvit_workspace_bytes(C.byref(cfg), 2, 1) > 0
    # fp32 activations double the workspace and the weight cache
    c16, _, _ = make_cfg("tiny_vit_21m_224", precision="bf16"); c32, _, _ = make_cfg("tiny_vit_21m_224", precision="fp32")
    assert L.lib().gg_tinyvit_workspace_bytes(C.byref(c32), 256, 1) > 1.9 * L.lib().gg_tinyvit_workspace_bytes(C.byref(c16), 256, 1)     # activation-dominated size
    bad, _, _ = make_cfg("tiny_vit_21m_224"); bad.act_dtype = 7
    assert L.lib().gg_tinyvit_num_tensors(C.byref(bad)) < 0 and b"act_dtype" in L.lib().gg_last_error()


def test_adapter_call_surface_on_cpu(L):
    """Construction, config shim, freezing policy and state-dict names need no GPU; running does."""
    from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
    from geoguessr_ai_amd.models.super_guessr import SuperGuessr
    m = TinyViTAdapter("tiny_vit_21m_224", pretrained=False)
    assert m.config.hidden_size == 576 and m.config.hidden_sizes == [576] and "tiny" in m.config._name_or_path
    assert len(m.vision_model.encoder.layers) == 4
    sg = SuperGuessr(m, panorama=True, should_smooth_labels=True)
    assert sg.mode == "transformer" and sg.num_cells == 12647 and sg.geocell_centroid_coords.shape == (12647, 2)
    n_train = sum(p.numel() for p in sg.parameters() if p.requires_grad)
    assert n_train == 15895163
    assert sum(p.numel() for p in sg.parameters()) == 27918887 + 12647 * 2   # + frozen centroid table
    keys = set(sg.state_dict())
    assert {"geocell_centroid_coords", "cell_layer.weight", "cell_layer.bias", "base_model.backbone.head.norm.weight",
            "base_model.backbone.stages.2.blocks.5.attn.attention_biases",
            "base_model.backbone.patch_embed.conv1.bn.num_batches_tracked"} <= keys
    ranges = m.backbone.trainable_ranges()
    assert len(ranges) == 2 and ranges[0][0] == 0                            # patch_embed | stage 3 + head.norm
    m.freeze_all()
    assert not m.training and m.train(True) is m and not m.training          # models/tinyvit.py:113-120
    m.unfreeze_all()
    assert all(p.requires_grad for p in m.parameters())
    # state-dict round trip through the flat buffer
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["backbone.head.norm.weight"] += 1.0
    m.load_state_dict(sd)
    off = [t for t in m.backbone.table if t["name"] == "head.norm.weight"][0]["offset"]
    assert torch.equal(m.backbone.flat_params[off:off + 576], sd["backbone.head.norm.weight"])
    with pytest.raises(L.GgError):
        m(pixel_values=torch.zeros(1, 3, 224, 224))                           # no GPU here -> loud failure, no fallback


def test_missing_library_fails_loudly(L, monkeypatch):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libgg.so")
    with pytest.raises(L.GgError, match="no CPU fallback"):
        L.lib()


def test_lr_schedule_and_loop_helpers():
    from geoguessr_ai_amd.optim import cosine_warm_restarts_lr
    from oracle import geo_ref as G
    for ep in range(0, 80):
        assert abs(cosine_warm_restarts_lr(ep, 5e-5) - G.cosine_warm_restarts_lr(ep, 5e-5)) < 1e-15
    from geoguessr_ai_amd.training.train_eval_loop import _batches, _shard, _num_batches
    ds = dict(a=torch.arange(10), b=torch.arange(10) * 2)
    seen = torch.cat([b["a"] for r in range(2) for b in _batches(ds, 2, True, 0, r, 2)])
    assert sorted(seen.tolist()) == list(range(10))                           # the two ranks partition the epoch
    # every rank gets the same number of rows / batches for ANY (n, world, batch) -- a short rank would deadlock the all-reduce
    from torch.utils.data import DistributedSampler
    for n, world, bs in [(10, 4, 1), (13, 2, 3), (9, 4, 2), (5, 8, 2), (64, 8, 8), (1, 2, 1)]:
        shards = [_shard(n, False, 0, r, world) for r in range(world)]
        assert len({len(sh) for sh in shards}) == 1 and len(shards[0]) == -(-n // world)
        assert set(torch.cat(shards).tolist()) == set(range(n))
        for r in range(world):        # the exact index lists of torch's DistributedSampler (what Accelerate gives the reference)
            assert shards[r].tolist() == list(DistributedSampler(range(n), num_replicas=world, rank=r, shuffle=False))
        counts = [sum(1 for _ in _batches(dict(a=torch.arange(n)), bs, True, 3, r, world)) for r in range(world)]
        assert counts == [_num_batches(n, bs, world)] * world


def test_weight_cache_dirty_mask_bookkeeping():
    """``mark_params_dirty(only=...)`` (fused optimizer step) accumulates tensor masks until the next refresh; an unmasked mark (checkpoint load,
    broadcast) means everything.  Host logic only: the backbone is constructed on the CPU and never run."""
    from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
    m = TinyViTAdapter("tiny_vit_5m_224", pretrained=False)
    m.freeze_all_but_last_stage()
    bb = m.backbone
    mask = bb.trainable_mask()
    assert len(mask) == len(bb.table) and 0 < sum(mask) < len(mask)
    bb._dirty_all, bb._dirty_only = False, None                       # state right after a refresh
    a = bytes(1 if i == 3 else 0 for i in range(len(mask)))
    bb.mark_params_dirty(only=a)
    bb.mark_params_dirty(only=mask)
    assert bb._dirty_all is False and bb._dirty_only == bytes(x | y for x, y in zip(a, mask)) and bb._wcache_version == -1
    bb.mark_params_dirty()                                            # an unmasked writer in between: full rebuild
    assert bb._dirty_all is True and bb._dirty_only is None
    bb.mark_params_dirty(only=mask)                                   # ... and a later masked mark must not narrow it again
    assert bb._dirty_all is True


def test_training_workspace_follows_the_trainable_mask():
    """gg_tinyvit_workspace_bytes_masked: under the reference's freeze policy the inputs of frozen Linears / depthwise convs are temporaries (a
    two-slot ring shared with the first gradient buffers), so the plan shrinks by > 20 %; an all-ones mask equals the unmasked plan; a temporary is
    refused by gg_tinyvit_activation_info_masked while the tensors backward needs keep distinct, in-range regions."""
    import ctypes as C
    from geoguessr_ai_amd import _lib as L
    from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
    m = TinyViTAdapter("tiny_vit_21m_224", pretrained=False, precision="fp32")
    bb = m.backbone
    lib = L.lib()
    B = 64
    full = lib.gg_tinyvit_workspace_bytes(C.byref(bb.cfg), B, 1)
    ones = bytes([1] * len(bb.table))
    assert lib.gg_tinyvit_workspace_bytes_masked(C.byref(bb.cfg), B, 1, ones) == full
    m.freeze_all_but_last_stage()
    mask = bb.trainable_mask()
    frz = lib.gg_tinyvit_workspace_bytes_masked(C.byref(bb.cfg), B, 1, mask)
    assert 0 < frz < 0.8 * full, (frz, full)
    assert lib.gg_tinyvit_workspace_bytes_masked(C.byref(bb.cfg), B, 0, mask) == lib.gg_tinyvit_workspace_bytes(C.byref(bb.cfg), B, 0)     # inference: no mask
    off, nb = C.c_int64(), C.c_int64()
    for name in ("stages.1.blocks.0.x1", "stages.2.blocks.3.ln2", "stages.2.blocks.3.fc1.act", "stages.0.blocks.0.act2"):
        assert lib.gg_tinyvit_activation_info_masked(C.byref(bb.cfg), B, name.encode(), mask, C.byref(off), C.byref(nb)) != 0
        assert b"not retained" in lib.gg_last_error()
        assert lib.gg_tinyvit_activation_info_masked(C.byref(bb.cfg), B, name.encode(), ones, C.byref(off), C.byref(nb)) == 0      # kept when everything trains
    spans = []
    for name in ("stages.1.blocks.0.x2", "stages.1.blocks.0.qkv", "stages.1.blocks.0.attn.out", "stages.1.blocks.0.fc1.pre", "stages.3.blocks.1.x1",
                 "stages.3.blocks.1.fc1.act", "stages.0.blocks.1.out", "scratch.G2"):
        assert lib.gg_tinyvit_activation_info_masked(C.byref(bb.cfg), B, name.encode(), mask, C.byref(off), C.byref(nb)) == 0, name
        assert 0 <= off.value and off.value + nb.value <= frz
        spans.append((off.value, off.value + nb.value, name))
    spans.sort()
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])), spans           # retained tensors never overlap
