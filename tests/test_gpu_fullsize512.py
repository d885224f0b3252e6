"""The reference's default model at the card's capacity (a module of its own: the 1024-image fixtures of test_gpu_fullsize.py are released when that module
ends, and this step needs 270 of the 288 GiB)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_default_512_model_trains_512_images_per_gpu_in_fp32():
    """The reference's default model (tiny_vit_21m_512, config.py:9) at 512 images per GPU and step in fp32 under the reference's freeze policy: the
    mask-aware plan is 290e9 bytes (270 GiB of the card's 288 GiB; tensors of 3.2e9 elements -- past 2^31 -- in stage 0).  The oracle cannot run this
    size; the property that must hold at any size: a batch made of 64 copies of the same 8 images has the batch statistics of the 8 images, so every copy's
    train-mode embedding and every parameter gradient of a batch-mean loss equal those of the 8-image step (different tiles / partial sums, same
    arithmetic; the copies at the end of the batch sit at the highest addresses)."""
    import ctypes as C
    import gc
    import warnings
    from geoguessr_ai_amd import _lib as L
    from geoguessr_ai_amd.models.tinyvit import TinyViTAdapter
    gc.collect(); torch.cuda.empty_cache()
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = TinyViTAdapter(precision="fp32", drop_path_rate=0.0)
    bb = m.backbone
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for name, p in bb.named_parameters():
            if name.endswith(("bn.weight", "norm.weight")): p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith(".bias") and p.dim() == 1: p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith(".weight") and p.dim() == 2: p.copy_(0.05 * torch.randn(p.shape, generator=g))
    m = m.cuda().train()
    m.freeze_all_but_last_stage()
    B, REP = 512, 64
    need = L.lib().gg_tinyvit_workspace_bytes_masked(C.byref(bb.cfg), B, 1, bb.trainable_mask())
    free, _ = torch.cuda.mem_get_info()
    assert need < 295e9
    if free < need + 8e9:
        pytest.skip(f"needs an idle MI355X (288 GiB): free {free / 2**30:.1f} GiB, torch reserved {torch.cuda.memory_reserved() / 2**30:.1f} / allocated {torch.cuda.memory_allocated() / 2**30:.1f} GiB, plan {need / 2**30:.1f} GiB")
    x8 = torch.randn(B // REP, 3, 512, 512, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    names = ["stages.3.blocks.1.mlp.fc2.weight", "stages.3.blocks.0.attn.qkv.weight", "patch_embed.conv1.conv.weight", "patch_embed.conv2.conv.weight"]

    def step(x):
        for p in bb._params.values():
            p.grad = None
        if bb._flat_grad is not None:
            bb._flat_grad.zero_()
        out = m(pixel_values=x).pooler_output
        out.square().mean().backward()
        torch.cuda.synchronize()
        return out.detach().clone(), [bb._params[n].grad.clone() for n in names]
    o8, g8 = step(x8)
    assert torch.isfinite(o8).all() and all(torch.isfinite(t).all() and float(t.abs().sum()) > 0 for t in g8)
    x = x8.repeat(REP, 1, 1, 1)
    o, gr = step(x)
    assert bb._ws[True].numel() == need
    del x
    o = o.view(REP, B // REP, -1)
    err = float((o - o8[None]).abs().max() / o8.abs().max())
    assert err < 1e-4, err                                         # every copy, first to last
    for n, a, b in zip(names, g8, gr):
        rel = float((a - b).norm() / a.norm())
        assert rel < 1e-3, (n, rel)
    bb.release_workspaces() if hasattr(bb, "release_workspaces") else None
    del m, bb
    gc.collect(); torch.cuda.empty_cache()
