"""CPU restatement (plain torch fp32) of the TinyViT encoder the reference
instantiates through ``timm.create_model(name, num_classes=0, global_pool="avg")``
(reference call sites: ``models/tinyvit.py:48-53,135``;
``pretrain/tinyvit_embedder.py:32-36,80``).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  PARITY UNPINNED: timm 1.0.21
is not available in this image; this follows the published architecture
(TinyViT, Wu et al. ECCV 2022) and is self-checked by
``tests/test_oracle_models.py`` (parameter totals 20 621 568 / 5 071 764,
state-dict key table, MAC totals 4.250 / 1.255 GMAC).

Everything is functional: parameters and buffers live in a ``dict`` keyed by the
timm state-dict names, so torch autograd on that dict gives reference gradients.

``emulate_bf16=True`` rounds GEMM/conv weights and every activation that the
HIP path stores in HBM to bf16 at the same points (DESIGN.md "Numerics"), which
lets tests separate algorithmic mismatches from storage-precision effects.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class TinyVitConfig:
    name: str = "tiny_vit_21m_224"
    img_size: int = 224
    in_chans: int = 3
    embed_dims: Tuple[int, ...] = (96, 192, 384, 576)
    depths: Tuple[int, ...] = (2, 2, 6, 2)
    num_heads: Tuple[int, ...] = (3, 6, 12, 18)
    window_sizes: Tuple[int, ...] = (7, 7, 14, 7)
    mlp_ratio: float = 4.0
    mbconv_expand_ratio: float = 4.0
    drop_path_rate: float = 0.2
    bn_eps: float = 1e-5
    ln_eps: float = 1e-5
    bn_momentum: float = 0.1

    @property
    def num_features(self) -> int:
        return self.embed_dims[-1]


VARIANTS = {
    # SURVEY.md App. A.1
    "tiny_vit_5m_224": dict(embed_dims=(64, 128, 160, 320), num_heads=(2, 4, 5, 10),
                            window_sizes=(7, 7, 14, 7), img_size=224, drop_path_rate=0.0),
    "tiny_vit_11m_224": dict(embed_dims=(64, 128, 256, 448), num_heads=(2, 4, 8, 14),
                             window_sizes=(7, 7, 14, 7), img_size=224, drop_path_rate=0.1),
    "tiny_vit_21m_224": dict(embed_dims=(96, 192, 384, 576), num_heads=(3, 6, 12, 18),
                             window_sizes=(7, 7, 14, 7), img_size=224, drop_path_rate=0.2),
    "tiny_vit_21m_384": dict(embed_dims=(96, 192, 384, 576), num_heads=(3, 6, 12, 18),
                             window_sizes=(12, 12, 24, 12), img_size=384, drop_path_rate=0.1),
    "tiny_vit_21m_512": dict(embed_dims=(96, 192, 384, 576), num_heads=(3, 6, 12, 18),
                             window_sizes=(16, 16, 32, 16), img_size=512, drop_path_rate=0.1),
}


def config_for(name: str, **overrides) -> TinyVitConfig:
    base = name.split(".")[0]
    kw = dict(VARIANTS[base])
    kw.update(overrides)
    return TinyVitConfig(name=base, **kw)


# ----------------------------------------------------------------------------
# parameter table (timm state-dict names)
# ----------------------------------------------------------------------------

def _convnorm_spec(prefix, cin, cout, ks, groups=1):
    return [
        (f"{prefix}.conv.weight", (cout, cin // groups, ks, ks), "param"),
        (f"{prefix}.bn.weight", (cout,), "param"),
        (f"{prefix}.bn.bias", (cout,), "param"),
        (f"{prefix}.bn.running_mean", (cout,), "buffer"),
        (f"{prefix}.bn.running_var", (cout,), "buffer"),
        (f"{prefix}.bn.num_batches_tracked", (), "buffer"),
    ]


def param_spec(cfg: TinyVitConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) in timm registration order."""
    d = cfg.embed_dims
    spec = []
    spec += _convnorm_spec("patch_embed.conv1", cfg.in_chans, d[0] // 2, 3)
    spec += _convnorm_spec("patch_embed.conv2", d[0] // 2, d[0], 3)
    mid = int(d[0] * cfg.mbconv_expand_ratio)
    for i in range(cfg.depths[0]):
        p = f"stages.0.blocks.{i}"
        spec += _convnorm_spec(f"{p}.conv1", d[0], mid, 1)
        spec += _convnorm_spec(f"{p}.conv2", mid, mid, 3, groups=mid)
        spec += _convnorm_spec(f"{p}.conv3", mid, d[0], 1)
    for s in range(1, len(d)):
        p = f"stages.{s}.downsample"
        spec += _convnorm_spec(f"{p}.conv1", d[s - 1], d[s], 1)
        spec += _convnorm_spec(f"{p}.conv2", d[s], d[s], 3, groups=d[s])
        spec += _convnorm_spec(f"{p}.conv3", d[s], d[s], 1)
        C, nh, ws = d[s], cfg.num_heads[s], cfg.window_sizes[s]
        hid = int(C * cfg.mlp_ratio)
        for i in range(cfg.depths[s]):
            p = f"stages.{s}.blocks.{i}"
            spec += [
                (f"{p}.attn.attention_biases", (nh, ws * ws), "param"),
                (f"{p}.attn.norm.weight", (C,), "param"),
                (f"{p}.attn.norm.bias", (C,), "param"),
                (f"{p}.attn.qkv.weight", (3 * C, C), "param"),
                (f"{p}.attn.qkv.bias", (3 * C,), "param"),
                (f"{p}.attn.proj.weight", (C, C), "param"),
                (f"{p}.attn.proj.bias", (C,), "param"),
                (f"{p}.mlp.norm.weight", (C,), "param"),
                (f"{p}.mlp.norm.bias", (C,), "param"),
                (f"{p}.mlp.fc1.weight", (hid, C), "param"),
                (f"{p}.mlp.fc1.bias", (hid,), "param"),
                (f"{p}.mlp.fc2.weight", (C, hid), "param"),
                (f"{p}.mlp.fc2.bias", (C,), "param"),
            ]
            spec += _convnorm_spec(f"{p}.local_conv", C, C, 3, groups=C)
    spec += [("head.norm.weight", (d[-1],), "param"), ("head.norm.bias", (d[-1],), "param")]
    return spec


def num_params(cfg: TinyVitConfig) -> int:
    return sum(math.prod(s) for _, s, k in param_spec(cfg) if k == "param")


def init_state(cfg: TinyVitConfig, seed: int = 0, randomize_norms: bool = False) -> Dict[str, torch.Tensor]:
    """timm's init: Linear trunc_normal(.02)/0, LayerNorm 1/0, Conv2d torch default
    (kaiming_uniform a=sqrt(5)), BN 1/0 except MBConv.conv3 BN weight 0, biases 0.
    ``randomize_norms`` perturbs norm affine params / attention biases / running stats so
    that tests exercise them (init values 1/0 hide indexing bugs)."""
    g = torch.Generator().manual_seed(seed)
    st: Dict[str, torch.Tensor] = {}
    for name, shape, kind in param_spec(cfg):
        if name.endswith("num_batches_tracked"):
            t = torch.zeros((), dtype=torch.int64)
        elif name.endswith("running_mean"):
            t = torch.zeros(shape)
        elif name.endswith("running_var"):
            t = torch.ones(shape)
        elif name.endswith("conv.weight"):
            fan_in = shape[1] * shape[2] * shape[3]
            bound = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif name.endswith("bn.weight"):
            is_mb3 = name.startswith("stages.0.") and ".conv3." in name
            t = torch.zeros(shape) if is_mb3 else torch.ones(shape)
        elif name.endswith("attention_biases"):
            t = torch.zeros(shape)
        elif name.endswith("norm.weight"):
            t = torch.ones(shape)
        elif name.endswith(".bias"):
            t = torch.zeros(shape)
        elif name.endswith(".weight"):  # Linear
            t = torch.empty(shape)
            torch.nn.init.trunc_normal_(t, std=0.02, generator=g)
        else:
            raise AssertionError(name)
        if randomize_norms:
            if name.endswith(("bn.weight", "norm.weight")):
                t = 1.0 + 0.2 * torch.randn(shape, generator=g)
            elif name.endswith((".bias",)) and t.dim() == 1:
                t = 0.1 * torch.randn(shape, generator=g)
            elif name.endswith("attention_biases"):
                t = 0.5 * torch.randn(shape, generator=g)
            elif name.endswith("running_mean"):
                t = 0.1 * torch.randn(shape, generator=g)
            elif name.endswith("running_var"):
                t = 1.0 + 0.2 * torch.rand(shape, generator=g)
        st[name] = t
    return st


def attention_bias_idxs(ws: int) -> torch.Tensor:
    """timm Attention: offsets (|dr|,|dc|) numbered in first-seen order while iterating
    p1, p2 over product(range(ws), range(ws)).  p1=(0,0) sees every offset in row-major
    order, so idx = |dr|*ws + |dc|; built here the slow way on purpose."""
    import itertools
    points = list(itertools.product(range(ws), range(ws)))
    offsets: Dict[Tuple[int, int], int] = {}
    idxs = []
    for p1 in points:
        for p2 in points:
            off = (abs(p1[0] - p2[0]), abs(p1[1] - p2[1]))
            if off not in offsets:
                offsets[off] = len(offsets)
            idxs.append(offsets[off])
    n = len(points)
    return torch.tensor(idxs, dtype=torch.long).view(n, n)


def drop_path_rates(cfg: TinyVitConfig) -> List[float]:
    n = sum(cfg.depths)
    return [float(x) for x in torch.linspace(0, cfg.drop_path_rate, n)]


# ----------------------------------------------------------------------------
# forward
# ----------------------------------------------------------------------------

class _Ctx:
    def __init__(self, cfg, st, training, emulate_bf16, update_running, taps):
        self.cfg, self.st, self.training = cfg, st, training
        self.emu = emulate_bf16
        self.update_running = update_running
        self.taps = taps

    def q(self, x):  # storage rounding of the HIP path
        if self.emu:
            return x.to(torch.bfloat16).to(torch.float32)
        return x

    def w(self, name):  # GEMM / dense-conv weight as the MFMA sees it
        return self.q(self.st[name])

    def tap(self, name, x):
        if self.taps is not None:
            self.taps[name] = x


def _bn(c: _Ctx, x, prefix):
    """BatchNorm2d on NCHW.  Train: batch stats (biased var) + running update
    (unbiased var, momentum .1); eval: running stats."""
    st = c.st
    w, b = st[f"{prefix}.bn.weight"], st[f"{prefix}.bn.bias"]
    if c.training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if c.update_running:
            n = x.numel() / x.shape[1]
            with torch.no_grad():
                m = c.cfg.bn_momentum
                st[f"{prefix}.bn.running_mean"].mul_(1 - m).add_(m * mean.detach())
                st[f"{prefix}.bn.running_var"].mul_(1 - m).add_(m * var.detach() * n / max(n - 1, 1))
                st[f"{prefix}.bn.num_batches_tracked"] += 1
    else:
        mean, var = st[f"{prefix}.bn.running_mean"], st[f"{prefix}.bn.running_var"]
    rstd = torch.rsqrt(var + c.cfg.bn_eps)
    return (x - mean[None, :, None, None]) * (rstd * w)[None, :, None, None] + b[None, :, None, None]


def _convnorm(c: _Ctx, x, prefix, stride=1, pad=0, groups=1, dense=True):
    wname = f"{prefix}.conv.weight"
    w = c.w(wname) if dense else c.st[wname]   # depthwise taps stay fp32 on the HIP path
    y = c.q(F.conv2d(x, w, None, stride, pad, 1, groups))   # pre-BN output is stored bf16
    return _bn(c, y, prefix)


def _patch_merging(c: _Ctx, x, p):
    out = c.st[f"{p}.conv1.conv.weight"].shape[0]
    x = c.q(F.gelu(_convnorm(c, x, f"{p}.conv1")))
    x = c.q(F.gelu(_convnorm(c, x, f"{p}.conv2", 2, 1, out, dense=False)))
    x = c.q(_convnorm(c, x, f"{p}.conv3"))
    return x


def _attention(c: _Ctx, x, p, nh, ws):
    """x: (B', N, C) windows.  Per-head interleaved [q|k|v] split."""
    C = x.shape[-1]
    st = c.st
    xn = c.q(F.layer_norm(x, (C,), st[f"{p}.attn.norm.weight"], st[f"{p}.attn.norm.bias"], c.cfg.ln_eps))
    return _attention_core(c, xn, p, nh, ws)


def _attention_core(c: _Ctx, xn, p, nh, ws):
    """Everything of timm's TinyViT ``Attention`` behind its LayerNorm: qkv Linear, per-head [q|k|v] split, scaled scores + relative-position bias gathered through
    the first-seen offset table, softmax, P.V, head merge (tap ``attn.out``), proj.  timm's class is LeViT's attention (Graham et al. 2021) with a LayerNorm in front
    and plain Linears; the part up to the head merge is PINNED against the LeViT implementation that ships with ``transformers``
    (``transformers.models.levit.modeling_levit.LevitAttention``: tests/test_oracle_models.py::test_tinyvit_attention_core_matches_transformers_levit)."""
    Bw, N, C = xn.shape
    st = c.st
    hd = C // nh
    qkv = c.q(F.linear(xn, c.w(f"{p}.attn.qkv.weight"), st[f"{p}.attn.qkv.bias"]))
    q, k, v = qkv.view(Bw, N, nh, 3 * hd).split([hd, hd, hd], dim=3)
    q, k, v = q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3)
    bias = st[f"{p}.attn.attention_biases"][:, attention_bias_idxs(ws)]  # (nh, N, N)
    attn = (q @ k.transpose(-2, -1)) * (hd ** -0.5) + bias
    if c.emu:
        # HIP path: P = exp(s - max) rounded to bf16 for the PV MFMA, row sum in fp32
        m = attn.amax(dim=-1, keepdim=True)
        pexp = torch.exp(attn - m)
        l = pexp.sum(dim=-1, keepdim=True)
        o = (c.q(pexp) @ v) / l
    else:
        o = attn.softmax(dim=-1) @ v
    o = c.q(o.transpose(1, 2).reshape(Bw, N, C))
    c.tap(f"{p}.attn.out", o)
    return F.linear(o, c.w(f"{p}.attn.proj.weight"), st[f"{p}.attn.proj.bias"])


def forward(cfg: TinyVitConfig, st: Dict[str, torch.Tensor], x: torch.Tensor, *,
            training: bool = False, emulate_bf16: bool = False,
            drop_masks: Optional[List[Optional[torch.Tensor]]] = None,
            update_running: bool = False, taps: Optional[dict] = None) -> torch.Tensor:
    """TinyVit.forward with num_classes=0: (B,3,H,W) fp32 -> (B, C_last) fp32.

    ``drop_masks``: one (B,) keep-mask per block (sum(depths) entries) -- timm draws an
    independent mask for drop_path1 and drop_path2 of a TinyVitBlock; the HIP path and this
    oracle both take them as inputs, indexed [2*k] / [2*k+1] for TinyVitBlocks when a list
    of length ``n_mb + 2*n_vit`` is given, or shared when length is sum(depths).
    """
    c = _Ctx(cfg, st, training, emulate_bf16, update_running, taps)
    masks = _MaskFeeder(cfg, drop_masks) if (training and drop_masks is not None) else None
    x = c.q(x)
    # PatchEmbed
    x = c.q(F.gelu(_convnorm(c, x, "patch_embed.conv1", 2, 1)))
    x = c.q(_convnorm(c, x, "patch_embed.conv2", 2, 1))
    c.tap("patch_embed", x)
    blk = 0
    for i in range(cfg.depths[0]):
        x = _mbconv_m(c, x, f"stages.0.blocks.{i}", masks, blk)
        blk += 1
    c.tap("stages.0", x)
    for s in range(1, len(cfg.embed_dims)):
        x = _patch_merging(c, x, f"stages.{s}.downsample")
        c.tap(f"stages.{s}.downsample.out", x)
        x = x.permute(0, 2, 3, 1)
        for i in range(cfg.depths[s]):
            x = _tinyvit_block_m(c, x, f"stages.{s}.blocks.{i}", cfg.num_heads[s], cfg.window_sizes[s], masks, blk)
            blk += 1
        x = x.permute(0, 3, 1, 2)
        c.tap(f"stages.{s}", x)
    # head: global avg pool -> LayerNorm2d over C -> flatten (fc = Identity)
    x = x.mean(dim=(2, 3))
    x = F.layer_norm(x, (x.shape[1],), st["head.norm.weight"], st["head.norm.bias"], cfg.ln_eps)
    return x


class _MaskFeeder:
    """Maps (block index, which drop_path) -> per-sample scale mask/(1-p)."""

    def __init__(self, cfg, masks):
        self.cfg, self.masks = cfg, masks
        self.rates = drop_path_rates(cfg)
        self.n_mb = cfg.depths[0]

    def scale(self, blk: int, which: int):
        if blk < self.n_mb:
            m = self.masks[blk]
        else:
            m = self.masks[self.n_mb + 2 * (blk - self.n_mb) + which]
        if m is None:
            return None
        return m.to(torch.float32) / (1.0 - self.rates[blk])


def _mbconv_m(c, x, p, masks, blk):
    B = x.shape[0]
    sc = x
    mid = c.st[f"{p}.conv1.conv.weight"].shape[0]
    x = c.q(F.gelu(_convnorm(c, x, f"{p}.conv1")))
    x = c.q(F.gelu(_convnorm(c, x, f"{p}.conv2", 1, 1, mid, dense=False)))
    x = _convnorm(c, x, f"{p}.conv3")
    s = masks.scale(blk, 0) if masks is not None else None
    if s is not None:
        x = x * s[:, None, None, None]
    x = c.q(F.gelu(sc + x))
    c.tap(f"{p}.out", x)
    return x


def _tinyvit_block_m(c, x, p, nh, ws, masks, blk):
    B, H, W, C = x.shape
    L = H * W
    sc = x
    if H == ws and W == ws:
        a = _attention(c, x.reshape(B, L, C), p, nh, ws).view(B, H, W, C)
    else:
        assert H % ws == 0 and W % ws == 0, "padding path never taken at 224/384/512"
        nH, nW = H // ws, W // ws
        xw = x.view(B, nH, ws, nW, ws, C).transpose(2, 3).reshape(B * nH * nW, ws * ws, C)
        a = _attention(c, xw, p, nh, ws)
        a = a.view(B, nH, nW, ws, ws, C).transpose(2, 3).reshape(B, H, W, C)
    s1 = masks.scale(blk, 0) if masks is not None else None
    if s1 is not None:
        a = a * s1[:, None, None, None]
    x = c.q(sc + a)
    c.tap(f"{p}.x1", x)
    x = x.permute(0, 3, 1, 2)
    x = c.q(_convnorm(c, x, f"{p}.local_conv", 1, 1, C, dense=False))
    x = x.reshape(B, C, L).transpose(1, 2)
    c.tap(f"{p}.x2", x)
    st = c.st
    h = c.q(F.layer_norm(x, (C,), st[f"{p}.mlp.norm.weight"], st[f"{p}.mlp.norm.bias"], c.cfg.ln_eps))
    h = F.linear(h, c.w(f"{p}.mlp.fc1.weight"), st[f"{p}.mlp.fc1.bias"])
    h = c.q(F.gelu(c.q(h)))
    h = F.linear(h, c.w(f"{p}.mlp.fc2.weight"), st[f"{p}.mlp.fc2.bias"])
    s2 = masks.scale(blk, 1) if masks is not None else None
    if s2 is not None:
        h = h * s2[:, None, None]
    x = c.q(x + h)
    c.tap(f"{p}.out", x)
    return x.view(B, H, W, C)


# ----------------------------------------------------------------------------
# self-checks used by the tests
# ----------------------------------------------------------------------------

def macs_per_image(cfg: TinyVitConfig) -> Dict[str, float]:
    """Dense MACs per image by group."""
    d, r = cfg.embed_dims, cfg.img_size
    out: Dict[str, float] = {}
    h1, h0 = r // 2, r // 4
    out["patch_embed"] = h1 * h1 * (d[0] // 2) * cfg.in_chans * 9 + h0 * h0 * d[0] * (d[0] // 2) * 9
    mid = int(d[0] * cfg.mbconv_expand_ratio)
    out["stage0_1x1"] = cfg.depths[0] * h0 * h0 * (2 * d[0] * mid)
    out["stage0_dw"] = cfg.depths[0] * h0 * h0 * mid * 9
    pm = qkv = att = proj = lc = mlp = 0
    h = h0
    for s in range(1, 4):
        C, Cp = d[s], d[s - 1]
        pm += h * h * Cp * C + (h // 2) ** 2 * (C * 9 + C * C)
        h //= 2
        L, ws = h * h, cfg.window_sizes[s]
        n = cfg.depths[s]
        hid = int(C * cfg.mlp_ratio)
        qkv += n * L * C * 3 * C
        proj += n * L * C * C
        att += n * L * (ws * ws) * C * 2
        lc += n * L * C * 9
        mlp += n * L * 2 * C * hid
    out.update(patch_merging=pm, qkv=qkv, attention=att, proj=proj, local_conv=lc, mlp=mlp)
    out["total"] = sum(out.values())
    return out
