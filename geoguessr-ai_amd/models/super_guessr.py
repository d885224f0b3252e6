"""Drop-in for the reference's ``models/super_guessr.py`` (``SuperGuessr``, :20-395): same constructor kwargs,
same ``forward(pixel_values=|embedding=, labels=, labels_clf=, index=)`` and the same returns
(``ModelOutput`` in train / non-serving eval, ``(pred_LLH, topk, embedding)`` in serving eval).

Everything after the encoder -- 4-view mean (:347), ``cell_layer`` Linear (:354), softmax / argmax / centroid gather /
top-k (:355-365), haversine-smoothed soft cross-entropy or hard CE (:372-383) and their backward -- runs as three HIP
launches: view-mean, one MFMA GEMM, and the fused per-row head kernel (``csrc/geo.hip``).

Divergences from the reference, on purpose: the centroid table is data shipped with the package or
passed in (C7/C8: no 29 s unpickling, explicit ordering); ``labels_clf=None`` is accepted (C6).  The hierarchical combine
(``hierarchical=True``: PositionalEncoder + MultiheadAttention, token 0) runs as HIP launches too (``_HierFn``).
"""
from __future__ import annotations

import math
import os
from typing import Optional

import numpy as np
import torch
from torch import nn, Tensor
from torch.nn.parameter import Parameter

from .. import _lib as L
from .. import ops
from ..config import CLIP_EMBED_DIM, CLIP_PRETRAINED_HEAD, LABEL_SMOOTHING_CONSTANT, NUM_ATTENTION_HEADS
from .utils import ModelOutput, TopK

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


def default_centroids() -> torch.Tensor:
    """(12647, 2) float32 (lng, lat): ``SuperGuessr.geocell_centroid_coords`` of the reference as built from its shipped
    geocell pickles (``_build_centroids_from_manager``, models/super_guessr.py:421-451), exported once as data."""
    path = os.environ.get("GG_CENTROIDS", os.path.join(_DATA, "centroids_12647x2_f32.npy"))
    return torch.from_numpy(np.load(path).astype(np.float32))


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model: "SuperGuessr", embedding: Tensor, weight: Tensor, bias: Tensor, labels, labels_clf, mode: int):
        L.require_gpu()
        emb = embedding.to(torch.float32).contiguous()
        if emb.dim() == 3:
            N, V, Cc = emb.shape
        else:
            (N, Cc), V = emb.shape, 1
        K = weight.shape[0]
        Kp = (K + 7) // 8 * 8
        dev = emb.device
        f32 = model.precision == "fp32"
        if f32:      # reference precision: f32 view mean, f32 MFMA logits, f32 dlogits (models/super_guessr.py:347,354 are fp32)
            xm = ops.view_mean_f32(emb.view(N, V, Cc))
            wn, wt = weight.detach(), None
        else:
            xm = torch.empty((N, Cc), dtype=torch.bfloat16, device=dev)
            L.check(L.lib().gg_view_mean_fwd(L.ptr(emb), L.ptr(xm), Cc, N, V, Cc, L.stream()), "gg_view_mean_fwd")
            wn, wt = model._weight_cache()
        logits = torch.empty((N, Kp), dtype=torch.float32, device=dev)
        if f32 and getattr(model, "split", False) and N >= 1024 and Cc % 8 == 0:
            # "fp32_split": logits = xm . W^T + b as a split product (the f32 view mean split in the GEMM's loader, the weight as cached planes)
            import ctypes as C
            a = L.Split3Args()
            a.b_planes, a.ldb, a.M, a.N, a.K = L.ptr(model._weight_planes()), Cc, N, K, Cc
            a.C, a.ldc, a.bias = L.ptr(logits), Kp, L.ptr(bias.detach(), torch.float32, "cell_layer.bias")
            L.check(L.lib().gg_gemm_nt_split3_af32(C.byref(a), L.ptr(xm, torch.float32, "view mean"), Cc, 0, L.stream()), "gg_gemm_nt_split3_af32")
        else:
            ops.gemm_nt(xm, wn, bias=bias.detach(), out_f32=True, out=logits, N=K, ldc=Kp)
        need_grad = mode != 0 and any(ctx.needs_input_grad)      # grad mode is off inside Function.forward
        r = ops.geo_head(logits, model.geocell_centroid_coords.data, labels=labels, labels_clf=labels_clf, mode=mode,
                         smoothing_km=float(LABEL_SMOOTHING_CONSTANT), want_dlogits=need_grad,
                         num_candidates=model.num_candidates, K=K, dlogits_f32=f32)
        ctx.model, ctx.dims, ctx.need, ctx.f32 = model, (N, V, Cc, K, Kp, embedding.dim()), need_grad, f32
        if need_grad:
            ctx.save_for_backward(xm, r["dlogits"])
        outs = (r["loss"].view(()), r["preds"], r["llh"], r["topk_vals"], r["topk_idx"])
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, g_loss, *_):
        N, V, Cc, K, Kp, edim = ctx.dims
        if not ctx.need:
            raise L.GgError("SuperGuessr head: backward requested but the forward ran without labels / with grad disabled")
        xm, dlogits = ctx.saved_tensors
        model = ctx.model
        g = g_loss.to(torch.float32).reshape(1).contiguous()      # upstream scalar, applied as a per-row scale on device
        demb = torch.empty((N, V, Cc), dtype=torch.float32, device=xm.device)
        dW = db = None
        if ctx.f32:
            wt = model._weight_t_f32()                                                  # (C, Kp) f32
            dxm = ops.gemm_nt(dlogits, wt, rowscale=g, rows_per_scale=N, K=Kp)            # (N, C) f32
            L.check(L.lib().gg_view_mean_bwd_f32(L.ptr(dxm), Cc, L.ptr(demb), N, V, Cc, L.stream()), "gg_view_mean_bwd_f32")
            if model.cell_layer.weight.requires_grad:
                dW = ops.gemm_tn(dlogits, xm, rowscale=g, rows_per_scale=N)[:K]          # (K, C) f32: dlogits^T . xm, no transposed copies
                db = ops.colsum_bf16(dlogits, rowscale=g, rows_per_scale=N)[:K]
        else:
            wn, wt = model._weight_cache()
            dxm = ops.gemm_nt(dlogits, wt, rowscale=g, rows_per_scale=N, K=Kp)            # (N, C) bf16
            L.check(L.lib().gg_view_mean_bwd(L.ptr(dxm), Cc, L.ptr(demb), N, V, Cc, L.stream()), "gg_view_mean_bwd")
            if model.cell_layer.weight.requires_grad:
                dl = dlogits[:, :K]
                dlT = ops.transpose_bf16(dl, rowscale=g, rows_per_scale=N)                  # (K, Np)
                xT = ops.transpose_bf16(xm)                                                 # (C, Np)
                dW = ops.gemm_nt(dlT, xT, out_f32=True)                                      # (K, C) f32
                db = ops.colsum_bf16(dl, rowscale=g, rows_per_scale=N)
        if edim == 2:
            demb = demb.view(N, Cc)
        return None, demb, dW, db, None, None, None


class _HierFn(torch.autograd.Function):
    """The hierarchical combine (models/super_guessr.py:340-345): ``self_attn(pos_encoder(x), ...)[0][:, 0]`` as HIP launches -- PE add (+dropout),
    in_proj (f32 MFMA GEMM), one softmax row per (sample, head) for query token 0 (the only one that reaches the output), out_proj on the
    token-0 rows.  fp32 in both precision modes."""

    @staticmethod
    def forward(ctx, model: "SuperGuessr", emb: Tensor, in_w: Tensor, in_b: Tensor, out_w: Tensor, out_b: Tensor):
        L.require_gpu()
        x = emb.to(torch.float32).contiguous()
        N, V, Cc = x.shape
        H = model.self_attn.num_heads
        if N > model.pos_encoder.pos_encoding.shape[0]:
            raise L.GgError(f"hierarchical SuperGuessr: batch {N} exceeds PositionalEncoder max_len {model.pos_encoder.pos_encoding.shape[0]} "
                            "(the reference indexes the encoding by BATCH position, models/layers/positional_encoder.py:44)")
        dev = x.device
        pe = model.pos_encoder.pos_encoding.detach()[:N, 0, :].contiguous()
        mask = pmask = None
        if model.training:          # nn.Dropout(0.1) of the encoder and of the attention weights (scale_by_keep)
            p1, p2 = model.pos_encoder.dropout_p, model.self_attn.dropout
            if p1 > 0:
                mask = ((torch.rand((N, V, Cc), device=dev) >= p1).to(torch.float32) / (1.0 - p1)).contiguous()
            if p2 > 0:
                pmask = ((torch.rand((N, H, V), device=dev) >= p2).to(torch.float32) / (1.0 - p2)).contiguous()
        mask = getattr(model, "_hier_masks", (mask, pmask))[0] if hasattr(model, "_hier_masks") else mask          # tests inject masks
        pmask = model._hier_masks[1] if hasattr(model, "_hier_masks") else pmask
        lib = L.lib()
        xin = torch.empty((N * V, Cc), dtype=torch.float32, device=dev)
        L.check(lib.gg_pe_add_f32(L.ptr(x), L.ptr(pe), L.ptr(mask), L.ptr(xin), N, V, Cc, L.stream()), "gg_pe_add_f32")
        qkv = ops.gemm_nt(xin, in_w.detach().contiguous(), bias=in_b.detach())
        o0 = torch.empty((N, Cc), dtype=torch.float32, device=dev)
        probs = torch.empty((N, H, V), dtype=torch.float32, device=dev)
        L.check(lib.gg_mha_q0_fwd(L.ptr(qkv), L.ptr(pmask), L.ptr(o0), L.ptr(probs), N, V, Cc, H, L.stream()), "gg_mha_q0_fwd")
        out = ops.gemm_nt(o0, out_w.detach().contiguous(), bias=out_b.detach())
        ctx.save_for_backward(xin, qkv, probs, o0, in_w, out_w)
        ctx.masks, ctx.dims = (mask, pmask), (N, V, Cc, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        xin, qkv, probs, o0, in_w, out_w = ctx.saved_tensors
        mask, pmask = ctx.masks
        N, V, Cc, H = ctx.dims
        lib = L.lib()
        dout = dout.to(torch.float32).contiguous()
        do0 = ops.gemm_nt(dout, out_w.detach().t().contiguous())                      # (N, C): dout . W_out
        d_out_w = ops.gemm_tn(dout, o0) if ctx.needs_input_grad[4] else None
        d_out_b = ops.colsum_bf16(dout) if ctx.needs_input_grad[5] else None
        dqkv = torch.empty_like(qkv)
        L.check(lib.gg_mha_q0_bwd(L.ptr(qkv), L.ptr(probs), L.ptr(pmask), L.ptr(do0), L.ptr(dqkv), N, V, Cc, H, L.stream()), "gg_mha_q0_bwd")
        d_in_w = ops.gemm_tn(dqkv, xin) if ctx.needs_input_grad[2] else None
        d_in_b = ops.colsum_bf16(dqkv) if ctx.needs_input_grad[3] else None
        demb = None
        if ctx.needs_input_grad[1]:
            dxin = ops.gemm_nt(dqkv, in_w.detach().t().contiguous())                  # (N*V, C): dqkv . W_in
            if mask is not None:
                demb = torch.empty_like(dxin)
                L.check(lib.gg_pe_add_f32(L.ptr(dxin), None, L.ptr(mask), L.ptr(demb), N, V, Cc, L.stream()), "gg_pe_add_f32")
            else:
                demb = dxin
            demb = demb.view(N, V, Cc)
        return None, demb, d_in_w, d_in_b, d_out_w, d_out_b


class _PositionalEncoder(nn.Module):
    """Parameter holder of the reference's ``PositionalEncoder`` (models/layers/positional_encoder.py:5-44): the sinusoidal table
    ``pos_encoding`` (max_len, 1, C) as a frozen parameter (state-dict key ``pos_encoder.pos_encoding``), dropout 0.1."""

    def __init__(self, dim_model: int, dropout_p: float = 0.1, max_len: int = 1000):
        super().__init__()
        self.dropout_p = dropout_p
        pos = torch.arange(max_len, dtype=torch.float32).unsqueeze(1)
        freq = torch.exp(torch.arange(0, dim_model, 2, dtype=torch.float32) * (-math.log(10000.0) / dim_model))
        table = torch.zeros(max_len, dim_model)
        table[:, 0::2] = torch.sin(pos * freq)
        table[:, 1::2] = torch.cos(pos * freq)
        self.register_parameter("pos_encoding", nn.Parameter(table.unsqueeze(1).contiguous(), requires_grad=False))


class _CellLayer(nn.Module):
    """Parameter holder with nn.Linear's state-dict keys (``cell_layer.weight`` (K,C), ``cell_layer.bias`` (K,))."""

    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        lin = nn.Linear(in_features, out_features)      # torch's default init, as the reference gets
        self.weight = Parameter(lin.weight.detach().clone())
        self.bias = Parameter(lin.bias.detach().clone())


class SuperGuessr(nn.Module):
    def __init__(self, base_model: Optional[nn.Module], panorama: bool = False, hierarchical: bool = False,
                 should_smooth_labels: bool = False, serving: bool = False, freeze_base: bool = False,
                 num_candidates: int = 5, embed_dim: int = CLIP_EMBED_DIM, centroids=None, precision: Optional[str] = None, **kwargs):
        """``centroids`` / ``precision`` are not in the reference: the geocell centroid table as data (C7/C8) and the arithmetic of
        the head ("bf16" | "fp32"; default: the base model's, else ``$GG_PRECISION``, else fp32)."""
        super().__init__()
        from .tinyvit import PRECISIONS, default_precision
        bb_prec = getattr(getattr(base_model, "backbone", None), "precision", None)
        self.precision = precision or bb_prec or default_precision()
        if self.precision not in PRECISIONS:
            raise ValueError(f"precision='{self.precision}' (known: bf16, fp32)")
        # "fp32_split" (asked for, or the base model's mode): the geocell Linear's forward as an f32-accurate split-bf16 product from 1024 rows on
        self.split = PRECISIONS[self.precision] == 3 or (precision is None and bool(getattr(getattr(base_model, "backbone", None), "split", False)))
        self.precision = "fp32" if PRECISIONS[self.precision] in (1, 3) else "bf16"
        if len(kwargs) > 0:
            print(f"Not using keyword arguments: {list(kwargs.keys())}")
        self.base_model = base_model
        self.panorama = panorama
        self.hidden_size = embed_dim
        self.serving = serving
        self.should_smooth_labels = should_smooth_labels
        self.freeze_base = freeze_base
        self.hierarchical = hierarchical
        self.num_candidates = num_candidates
        self._set_hidden_size()
        cent = default_centroids() if centroids is None else torch.as_tensor(np.asarray(centroids), dtype=torch.float32)
        assert cent.dim() == 2 and cent.shape[1] == 2, "centroids must be (num_cells, 2) in (lng, lat)"
        self.geocell_centroid_coords = nn.Parameter(cent.contiguous(), requires_grad=False)
        self.num_cells = cent.size(0)
        self.input_dim = self.hidden_size
        if self.hierarchical:                                   # models/super_guessr.py:89-99
            print("Number of attention heads:", NUM_ATTENTION_HEADS)
            self.heading_pad = 0
            self.pos_encoder = _PositionalEncoder(self.input_dim + self.heading_pad)
            # parameter container with nn.MultiheadAttention's init and state-dict keys (in_proj_weight, in_proj_bias, out_proj.*); its
            # torch forward is never called -- the arithmetic is _HierFn
            self.self_attn = nn.MultiheadAttention(self.input_dim + self.heading_pad, NUM_ATTENTION_HEADS, dropout=0.1, batch_first=True)
            self.relu = nn.ReLU()
        self.cell_layer = _CellLayer(self.input_dim, self.num_cells)
        self._wc = None
        self._wc_version = None
        self._freeze_params()
        print(f"Initialized SuperGuessr classification model with {self.num_cells} geocells.")

    # ---- reference helpers (behaviour of models/super_guessr.py:113-150,240-245,385-395; own text) -------------------------
    def _set_hidden_size(self):
        """Embedding width and ``mode`` from the base model's config: ``config.hidden_size`` -> "transformer" (CLIP, TinyViT adapter), else
        the last entry of ``config.hidden_sizes`` -> "convnext".  No base model: ``embed_dim`` stands."""
        cfg = getattr(self.base_model, "config", None)
        if self.base_model is None:
            return
        if hasattr(cfg, "hidden_size"):
            self.hidden_size, self.mode = cfg.hidden_size, "transformer"
        else:
            self.hidden_size, self.mode = cfg.hidden_sizes[-1], "convnext"       # AttributeError here = not a usable base model (as upstream)

    def _freeze_params(self):
        """Training policy by base-model family: ``freeze_base`` freezes the whole encoder; a CLIP tower that is not serving loads the
        pretrained head when the file exists and then trains only its LAST encoder layer (without the file: warning, nothing frozen); a TinyViT
        that is not serving keeps patch_embed + last stage + head trainable (``freeze_all_but_last_stage``)."""
        base = self.base_model
        if base is None:
            return
        family = getattr(base.config, "_name_or_path", "")
        if self.freeze_base:
            for p in base.parameters():
                p.requires_grad = False
            return
        if self.serving:
            return
        if "clip-vit" in family:
            if not os.path.exists(CLIP_PRETRAINED_HEAD):
                print(f"Warning: pretrained head not found at '{CLIP_PRETRAINED_HEAD}'. Proceeding without loading and without freezing base layers.")
                return
            self.load_state(CLIP_PRETRAINED_HEAD)
            print(f"Initialized model parameters from model: {CLIP_PRETRAINED_HEAD}")
            for layer in list(base.vision_model.encoder.layers)[:-1]:
                for p in layer.parameters():
                    p.requires_grad = False
        elif "tiny" in family:
            base.freeze_all_but_last_stage()

    def load_state(self, path: str):
        """Copy every tensor of the checkpoint at ``path`` whose name this model has; the others are reported and skipped (the tolerant loader
        of models/utils.py, on the device the model will run on)."""
        from .utils import load_state_dict
        load_state_dict(self, torch.load(path, map_location="cuda" if torch.cuda.is_available() else "cpu"))

    def _assert_requirements(self, pixel_values=None, embedding=None):
        have_base = self.base_model is not None
        given, name = (pixel_values, "pixel_values") if have_base else (embedding, "embedding")
        assert given is not None, (f'Parameter "{name}" must be supplied if model has a base model.' if have_base else
                                   f'Parameter "{name}" must be supplied if model does not have a base model.')

    # ---- bf16 copies of the head weight: Wn (K, C) for the forward GEMM, Wt (C, Kpad) for dgrad --------------
    def _weight_cache(self):
        w = self.cell_layer.weight
        ver = (w._version, w.data_ptr())
        if self._wc is None or self._wc_version != ver or getattr(self, "_wc_dirty", False):
            K, Cc = w.shape
            Kp = (K + 7) // 8 * 8
            if self._wc is None or self._wc[0].device != w.device:
                self._wc = (torch.empty((K, Cc), dtype=torch.bfloat16, device=w.device),
                            torch.zeros((Cc, Kp), dtype=torch.bfloat16, device=w.device))
            L.check(L.lib().gg_cast_transpose_f32(L.ptr(w.detach(), torch.float32, "cell_layer.weight"), K, Cc, L.ptr(self._wc[0]), Cc,
                                                  L.ptr(self._wc[1]), Kp, L.stream()), "gg_cast_transpose_f32")
            self._wc_version, self._wc_dirty = ver, False
        return self._wc

    def _weight_planes(self):
        """bf16 planes [3][K][C] of the head weight (w = w1 + w2 + w3 to 24 bits): the cached operand of the split forward GEMM of the "fp32_split" mode."""
        w = self.cell_layer.weight
        ver = (w._version, w.data_ptr())
        if getattr(self, "_wpl", None) is None or self._wpl_version != ver or getattr(self, "_wpl_dirty", False) or self._wpl.device != w.device:
            K, Cc = w.shape
            if getattr(self, "_wpl", None) is None or self._wpl.device != w.device:
                self._wpl = torch.empty((3, K, Cc), dtype=torch.bfloat16, device=w.device)
            L.check(L.lib().gg_split3_bf16(L.ptr(w.detach(), torch.float32, "cell_layer.weight"), K, Cc, Cc, L.ptr(self._wpl), L.stream()), "gg_split3_bf16")
            self._wpl_version, self._wpl_dirty = ver, False
        return self._wpl

    def _weight_t_f32(self):
        """(C, Kpad) f32 transpose of the head weight for the f32 dgrad (dlogits . W as an NT GEMM)."""
        w = self.cell_layer.weight
        ver = (w._version, w.data_ptr())
        if getattr(self, "_wt32", None) is None or self._wt32_version != ver or getattr(self, "_wc_dirty", False) or self._wt32.device != w.device:
            K, Cc = w.shape
            Kp = (K + 7) // 8 * 8
            if getattr(self, "_wt32", None) is None or self._wt32.device != w.device:
                self._wt32 = torch.zeros((Cc, Kp), dtype=torch.float32, device=w.device)
            L.check(L.lib().gg_transpose_f32(L.ptr(w.detach(), torch.float32, "cell_layer.weight"), K, Cc, L.ptr(self._wt32), Kp, L.stream()),
                    "gg_transpose_f32")                       # pad columns K..Kp-1 stay zero
            self._wt32_version, self._wc_dirty = ver, False
        return self._wt32

    def mark_params_dirty(self, backbone: bool = True):
        self._wc_dirty = True
        self._wpl_dirty = True
        bb = getattr(self.base_model, "backbone", None) if backbone else None
        if bb is not None and hasattr(bb, "mark_params_dirty"):
            bb.mark_params_dirty()

    # ---- forward ---------------------------------------------------------------------------------------
    def forward(self, pixel_values: Tensor = None, embedding: Tensor = None, labels: Tensor = None,
                labels_clf: Tensor = None, index: Tensor = None):
        self._assert_requirements(pixel_values, embedding)
        dev = self.cell_layer.weight.device
        if not self.cell_layer.weight.is_cuda:
            raise L.GgError("SuperGuessr parameters are on the CPU; call .to('cuda') -- there is no CPU fallback")
        mv = lambda t, dt=None: None if t is None else t.to(device=dev, dtype=dt if dt is not None else t.dtype)
        pixel_values, embedding = mv(pixel_values), mv(embedding)
        labels = mv(labels, torch.float32)
        labels_clf = mv(labels_clf, torch.int64)
        if labels_clf is not None and labels_clf.dim() == 0:
            labels_clf = labels_clf.view(1)

        if self.panorama and pixel_values is not None:
            assert pixel_values.dim() == 5, "panorama=True expects (B, 4, C, H, W)"
            n, v, c, h, w = pixel_values.shape
            pixel_values = pixel_values.reshape(n * v, c, h, w)
        if self.base_model is not None and pixel_values is not None:
            if pixel_values.dim() > 4:
                pixel_values = pixel_values.squeeze(1)
            if hasattr(self.base_model, "forward_hip") and hasattr(self.base_model, "num_tokens"):
                outs = self.base_model(pixel_values=pixel_values, return_last_hidden=False)      # HIP CLIP tower: only the token mean is needed
            else:
                outs = self.base_model(pixel_values=pixel_values)
            if hasattr(outs, "pooled_mean"):                      # HIP CLIP tower: token mean already taken on device
                embedding = outs.pooled_mean
            elif hasattr(outs, "last_hidden_state") and self.mode == "transformer" and outs.last_hidden_state.shape[1] != 1:
                embedding = outs.last_hidden_state.mean(dim=1)
            elif hasattr(outs, "pooler_output"):
                embedding = outs.pooler_output                    # TinyViT: mean over the fake length-1 axis is a no-op (C4)
            else:
                embedding = outs
            if self.panorama:
                embedding = embedding.view(n, v, -1)

        if labels is not None and getattr(self, "should_smooth_labels", False):
            mode = 1
        elif labels_clf is not None:
            mode = 2
        else:
            mode = 0
        if not self.training and self.serving:
            mode = 0
        head_in = embedding
        if self.panorama and self.hierarchical:                 # (N, 4, C) -> self_attn(pos_encoder(x))[:, 0]  (:340-345)
            assert embedding.dim() == 3, "hierarchical=True expects (N, 4, C) embeddings"
            a = self.self_attn
            head_in = _HierFn.apply(self, embedding, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias)
        loss, preds, llh, tv, ti = _HeadFn.apply(self, head_in, self.cell_layer.weight, self.cell_layer.bias,
                                                 labels.contiguous() if labels is not None else None,
                                                 labels_clf.contiguous() if labels_clf is not None else None, mode)
        topk = TopK(tv, ti)
        if not self.training and self.serving:
            return llh, topk, embedding
        if mode == 0:
            loss = None
        return ModelOutput(loss, loss, llh, preds, topk, embedding)

    def __str__(self):
        rows = [("base_model", self.base_model is not None), ("panorama", self.panorama), ("hierarchical", self.hierarchical),
                ("embedding_size", self.hidden_size), ("input_dim", self.input_dim), ("num_geocells", self.num_cells),
                ("label_smoothing", self.should_smooth_labels), ("freeze_base", self.freeze_base), ("serving", self.serving)]
        # one tab after the name, two after the one name shorter than a tab stop (the reference's layout)
        body = "".join(f"\t{k}{chr(9) * (2 if len(k) < 8 else 1)}= {v}\n" for k, v in rows)
        return f"SuperGuessr(\n{body})"
