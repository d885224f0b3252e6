"""ctypes binding of ``libgg.so`` (C-ABI in ``include/gg.h``).  No torch types cross the boundary: tensors are
passed as raw device pointers + sizes.  There is no fallback path: a missing library or a host tensor raises."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GG_LIB") or os.path.join(_HERE, "lib", "libgg.so")   # GG_LIB: dev override (A/B builds)
_lib = None


class GgError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64), ("C", C.c_void_p),
                ("ldc", C.c_int64), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("bias", C.c_void_p),
                ("act", C.c_int), ("preact", C.c_void_p), ("rowscale", C.c_void_p), ("rows_per_scale", C.c_int),
                ("residual", C.c_void_p), ("ldr", C.c_int64), ("dact_preact", C.c_void_p), ("dact", C.c_int),
                ("colstats", C.c_void_p), ("out_f32", C.c_int), ("split_k", C.c_int), ("A2", C.c_void_p), ("k_split", C.c_int),
                ("bn_y", C.c_void_p), ("bn_stat", C.c_void_p), ("bn_gamma", C.c_void_p), ("bn_beta", C.c_void_p),
                ("bn_act", C.c_int), ("a_bn_stat", C.c_void_p), ("a_bn_gamma", C.c_void_p), ("a_bn_beta", C.c_void_p),
                ("a_bn_act", C.c_int)]


class Split3Args(C.Structure):          # GgSplit3Args (experiment: fp32-accurate GEMM from three bf16 planes per operand)
    _fields_ = [("a_planes", C.c_void_p), ("lda", C.c_int64), ("b_planes", C.c_void_p), ("ldb", C.c_int64), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("C", C.c_void_p), ("ldc", C.c_int64), ("c_planes", C.c_void_p), ("ldp", C.c_int64), ("bias", C.c_void_p), ("act", C.c_int),
                ("preact", C.c_void_p), ("rowscale", C.c_void_p), ("rows_per_scale", C.c_int), ("residual", C.c_void_p), ("ldr", C.c_int64),
                ("dact_preact", C.c_void_p), ("dact", C.c_int)]


class AttnArgs(C.Structure):
    _fields_ = [("qkv", C.c_void_p), ("ld", C.c_int64), ("q_off", C.c_int), ("k_off", C.c_int), ("v_off", C.c_int),
                ("head_stride", C.c_int), ("head_dim", C.c_int), ("num_heads", C.c_int), ("num_windows", C.c_int),
                ("tokens_per_window", C.c_int), ("window_size", C.c_int), ("map_h", C.c_int), ("map_w", C.c_int),
                ("bias", C.c_void_p), ("scale", C.c_float), ("out", C.c_void_p), ("ldo", C.c_int64),
                ("dout", C.c_void_p), ("lddo", C.c_int64), ("dqkv", C.c_void_p), ("dbias", C.c_void_p), ("dbias_scratch", C.c_void_p),
                ("lse", C.c_void_p), ("bias_table", C.c_void_p), ("ds_scratch", C.c_void_p)]


class GeoHeadArgs(C.Structure):
    _fields_ = [("logits", C.c_void_p), ("ldl", C.c_int64), ("N", C.c_int), ("K", C.c_int), ("labels", C.c_void_p),
                ("centroids", C.c_void_p), ("labels_clf", C.c_void_p), ("mode", C.c_int), ("smoothing_km", C.c_float),
                ("grad_scale", C.c_float), ("loss_rows", C.c_void_p), ("loss", C.c_void_p), ("dlogits", C.c_void_p),
                ("ldd", C.c_int64), ("preds", C.c_void_p), ("llh", C.c_void_p), ("topk_vals", C.c_void_p),
                ("topk_idx", C.c_void_p), ("num_candidates", C.c_int), ("nearest", C.c_void_p), ("dlogits_f32", C.c_int)]


class ProtoRefineArgs(C.Structure):
    _fields_ = [("embedding", C.c_void_p), ("B", C.c_int), ("V", C.c_int), ("D", C.c_int), ("initial_preds", C.c_void_p),
                ("candidate_cells", C.c_void_p), ("candidate_probs", C.c_void_p), ("num_candidates", C.c_int),
                ("topk", C.c_int), ("cell_ptr", C.c_void_p), ("num_cells", C.c_int), ("proto_emb", C.c_void_p),
                ("proto_lnglat", C.c_void_p), ("max_refinement", C.c_float), ("temperature", C.c_float),
                ("out_llh", C.c_void_p), ("out_cell", C.c_void_p), ("out_idx", C.c_void_p),
                ("member_ptr", C.c_void_p), ("member_emb", C.c_void_p), ("member_lnglat", C.c_void_p)]


class TinyVitCfg(C.Structure):
    _fields_ = [("img_size", C.c_int), ("in_chans", C.c_int), ("embed_dims", C.c_int * 4), ("depths", C.c_int * 4),
                ("num_heads", C.c_int * 4), ("window_sizes", C.c_int * 4), ("mlp_ratio", C.c_float),
                ("mbconv_expand_ratio", C.c_float), ("bn_eps", C.c_float), ("ln_eps", C.c_float),
                ("bn_momentum", C.c_float), ("act_dtype", C.c_int), ("features_only", C.c_int)]


class ClipCfg(C.Structure):
    _fields_ = [("hidden_size", C.c_int), ("intermediate_size", C.c_int), ("num_layers", C.c_int), ("num_heads", C.c_int),
                ("image_size", C.c_int), ("patch_size", C.c_int), ("ln_eps", C.c_float), ("act_dtype", C.c_int)]


STAGE_DONE_FN = C.CFUNCTYPE(None, C.c_int, C.c_void_p)      # GgStageDoneFn (host callback of gg_tinyvit_backward)

# every exported symbol of include/gg.h: name -> (restype, argtypes)
_P, _I, _L, _F = C.c_void_p, C.c_int, C.c_int64, C.c_float
SIGNATURES = {
    "gg_version": (_I, []),
    "gg_last_error": (C.c_char_p, []),
    "gg_gemm_nt": (_I, [C.POINTER(GemmArgs), _P]),
    "gg_gemm_colstats_rows": (_I, [_I]),
    "gg_stat_rows_capacity": (_I, [_I]),
    "gg_gemm_tn_splits": (_I, [_I, _I, _I]),
    "gg_gemm_tn": (_I, [_P, _L, _P, _L, _I, _I, _I, _P, _I, _P, _I, _P]),
    "gg_gemm_tn_bn": (_I, [_P, _P, _L, _P, _P, _L, _I, _I, _I, _P, _I, _P]),
    "gg_gemm_tn_bn_f32": (_I, [_P, _P, _L, _P, _P, _L, _I, _I, _I, _P, _I, _P]),
    "gg_splitk_reduce": (_I, [_P, _P, _L, _I, _I, _F, _P]),
    "gg_transpose_bf16": (_I, [_P, _L, _P, _L, _I, _I, _P, _I, _P]),
    "gg_cast_transpose_f32": (_I, [_P, _I, _I, _P, _L, _P, _L, _P]),
    "gg_cast_f32_to_bf16": (_I, [_P, _P, _L, _P]),
    "gg_cast_bf16_to_f32": (_I, [_P, _P, _L, _P]),
    "gg_colsum_scratch_floats": (_L, [_I, _I]),
    "gg_colsum_bf16": (_I, [_P, _L, _I, _I, _P, _I, _P, _P, _I, _P]),
    "gg_im2col_nchw3_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "gg_im2col_nhwc_bf16": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "gg_im2col_nhwc_bn_bf16": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "gg_col2im_nhwc_bf16": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "gg_col2im_nhwc_bnbwd_bf16": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "gg_col2im_nhwc_bnbwd_f32": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P]),
    "gg_dwconv_stat_rows": (_I, [_I, _I, _I, _I, _I]),
    "gg_dwconv_tiled_stat_rows": (_I, [_I, _I]),
    "gg_dwconv_fused_stat_rows": (_I, [_I, _I, _I, _I, _I]),
    "gg_dwconv_fwd_fused_stat_rows": (_I, [_I, _I, _I, _I, _I]),
    "gg_dwconv3x3_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "gg_dwconv3x3_fwd_fused": (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "gg_dwconv3x3_bwd_data_fused": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "gg_dwconv_s2_fused_stat_rows": (_I, [_I, _I, _I, _I]),
    "gg_dwconv3x3_s2_bwd_data_fused": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "gg_dwconv3x3_bwd_data": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "gg_dwconv_wgrad_scratch_floats": (_L, [_I, _I, _I, _I, _I]),
    "gg_dwconv3x3_bwd_weight": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "gg_bn_finalize": (_I, [_P, _I, _I, _L, _F, _F, _P, _P, _P, _P]),
    "gg_bn_eval_stat": (_I, [_P, _P, _I, _F, _P, _P]),
    "gg_bn_apply": (_I, [_P, _P, _P, _P, _L, _I, _I, _P, _P, _I, _P, _P]),
    "gg_bn_bwd_scratch_floats": (_L, [_L, _I]),
    "gg_bn_bwd_rows": (_I, [_L, _I]),
    "gg_bn_bwd_reduce": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _P, _P, _I, _P, _P, _P]),
    "gg_bn_bwd_finalize": (_I, [_P, _I, _I, _L, _P, _P, _P, _P, _P, _I, _P]),
    "gg_bn_bwd_apply": (_I, [_P, _P, _P, _L, _I, _P, _I, _P, _P]),
    "gg_bn_bwd_fold_weights": (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    "gg_bn_bwd": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P]),
    "gg_layernorm_fwd": (_I, [_P, _I, _P, _P, _L, _I, _F, _P, _I, _P, _P, _P]),
    "gg_layernorm_fwd_bn": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _P, _P, _P, _P]),
    "gg_gemm_f32_set_trace": (_I, [_P]),
    "gg_layernorm_fwd_bn_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _P, _P, _P, _P]),
    "gg_layernorm_bwd_scratch_floats": (_L, [_L, _I]),
    "gg_layernorm_bwd": (_I, [_P, _P, _I, _P, _P, _P, _L, _I, _P, _P, _P, _P, _P, _I, _P]),
    "gg_layernorm_bwd_colsum_rows": (_I, [_L]),
    "gg_layernorm_bwd_colsum": (_I, [_P, _P, _I, _P, _P, _P, _L, _I, _P, _P, _P, _P]),
    "gg_bn_bwd_coef_from_x": (_I, [_P, _I, _I, _L, _P, _P, _P, _P, _P]),
    "gg_token_mean_fwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "gg_token_mean_bwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "gg_view_mean_fwd": (_I, [_P, _P, _L, _I, _I, _I, _P]),
    "gg_view_mean_bwd": (_I, [_P, _L, _P, _I, _I, _I, _P]),
    "gg_attention_padded_tokens": (_I, [_I]),
    "gg_attention_expand_bias": (_I, [_P, _I, _I, _F, _P, _P]),
    "gg_attention_fwd": (_I, [C.POINTER(AttnArgs), _P]),
    "gg_attention_fwd_f16": (_I, [C.POINTER(AttnArgs), _P]),
    "gg_attention_bwd": (_I, [C.POINTER(AttnArgs), _P]),
    "gg_attention_flash_fwd": (_I, [C.POINTER(AttnArgs), _I, _P]),
    "gg_attention_flash_bwd": (_I, [C.POINTER(AttnArgs), _I, _P]),
    "gg_attention_flash_dbias_rows": (_L, [_I, _I]),
    "gg_attention_flash_ds_scratch_floats": (_L, [_I, _I, _I]),
    "gg_attention_flash_single_pass": (_I, [_I, _I, _I, _I]),
    "gg_split3_bf16": (_I, [_P, _L, _I, _L, _P, _P]),
    "gg_layernorm_fwd_split3": (_I, [_P, _P, _P, _L, _I, _F, _P, _P, _P, _P]),
    "gg_layernorm_fwd_bn_split3": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _P, _P, _P, _P]),
    "gg_gemm_nt_split3": (_I, [_P, _L, _P, _L, _P, _L, _I, _I, _I, _P, _P]),
    "gg_gemm_nt_split3_ex": (_I, [C.POINTER(Split3Args), _P]),
    "gg_gemm_nt_split3_af32": (_I, [C.POINTER(Split3Args), _P, _L, _L, _P]),
    "gg_gemm_nt_split3_af32_stats": (_I, [C.POINTER(Split3Args), _P, _L, _L, _P, _P]),
    "gg_gemm_nt_split3_af32_pro": (_I, [C.POINTER(Split3Args), _P, _L, _L, _P, _P, _P, _I, _P, _P]),
    "gg_gemm_tn_split3_splits": (_I, [_I, _I, _I]),
    "gg_gemm_tn_split3": (_I, [_P, _L, _P, _L, _I, _I, _I, _P, _I, _P, _I, _P]),
    "gg_gemm_nt_f32": (_I, [C.POINTER(GemmArgs), _P]),
    "gg_gemm_tn_f32_splits": (_I, [_I, _I, _I]),
    "gg_gemm_tn_f32": (_I, [_P, _L, _P, _L, _I, _I, _I, _P, _I, _P, _I, _P]),
    "gg_colsum_f32": (_I, [_P, _L, _I, _I, _P, _I, _P, _P, _I, _P]),
    "gg_im2col_nchw3_f32_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "gg_im2col_nhwc_f32": (_I, [_P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "gg_col2im_nhwc_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "gg_dwconv_f32_stat_rows": (_I, [_I, _I, _I, _I, _I]),
    "gg_dwconv3x3_fwd_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "gg_dwconv3x3_bwd_data_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "gg_dwconv_f32_wgrad_scratch_floats": (_L, [_I, _I, _I, _I, _I]),
    "gg_dwconv3x3_bwd_weight_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "gg_bn_apply_f32": (_I, [_P, _P, _P, _P, _L, _I, _I, _P, _P, _I, _P, _P]),
    "gg_bn_bwd_f32": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P]),
    "gg_bn_bwd_reduce_f32": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _P, _P, _I, _P, _P, _P]),
    "gg_bn_bwd_apply_f32": (_I, [_P, _P, _P, _L, _I, _P, _I, _P, _P]),
    "gg_dwconv3x3_fwd_fused_f32": (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "gg_dwconv3x3_bwd_data_fused_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "gg_dwconv_f32_s2_fused_stat_rows": (_I, [_I, _I, _I, _I]),
    "gg_dwconv3x3_s2_bwd_data_fused_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "gg_gemm_nt_f16": (_I, [C.POINTER(GemmArgs), _P]),
    "gg_layernorm_fwd_f16": (_I, [_P, _P, _P, _L, _I, _F, _P, _P]),
    "gg_token_mean_fwd_f16": (_I, [_P, _P, _I, _I, _I, _P]),
    "gg_cast_f32_to_f16": (_I, [_P, _P, _L, _P]),
    "gg_cast_f16_to_f32": (_I, [_P, _P, _L, _P]),
    "gg_token_mean_fwd_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "gg_token_mean_bwd_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "gg_view_mean_fwd_f32": (_I, [_P, _P, _L, _I, _I, _I, _P]),
    "gg_view_mean_bwd_f32": (_I, [_P, _L, _P, _I, _I, _I, _P]),
    "gg_geo_head": (_I, [C.POINTER(GeoHeadArgs), _P]),
    "gg_haversine_matrix": (_I, [_P, _P, _P, _I, _I, _P]),
    "gg_pe_add_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "gg_mha_q0_fwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "gg_mha_q0_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "gg_proto_refine": (_I, [C.POINTER(ProtoRefineArgs), _P]),
    "gg_geoguessr_score": (_I, [_P, _P, _I, _P, _P, _P]),      # (pred, truth, N, double* dist_km, int32* score, stream)
    "gg_geoguessr_score_f64": (_I, [_P, _P, _I, _P, _P, _P]),  # same with double* coordinates
    "gg_adamw_step": (_I, [_P, _P, _P, _P, _L, _I, _F, _F, _F, _F, _F, _F, _P]),
    "gg_fill_f32": (_I, [_P, _L, _F, _P]),
    "gg_comm_unique_id": (_I, [_P]),
    "gg_comm_create": (_I, [C.POINTER(_P), _P, _I, _I, _I]),
    "gg_comm_destroy": (_I, [_P]),
    "gg_comm_rank": (_I, [_P]),
    "gg_comm_world": (_I, [_P]),
    "gg_comm_allreduce_sum_f32": (_I, [_P, _P, _L, _P]),
    "gg_comm_broadcast": (_I, [_P, _P, _L, _I, _P]),
    "gg_comm_barrier": (_I, [_P, _P, _I]),
    "gg_prof_enable": (_I, [_I]),
    "gg_prof_reset": (_I, []),
    "gg_prof_read": (_I, [_I, C.POINTER(C.c_double), C.POINTER(_L), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gg_prof_count": (_I, []),
    "gg_prof_record": (_I, [_I, C.POINTER(_I), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gg_graph_set_mode": (_I, [_I]),
    "gg_graph_stats": (_I, [C.POINTER(_L), C.POINTER(_L), C.POINTER(_L)]),
    "gg_graph_clear": (_I, []),
    "gg_tinyvit_num_tensors": (_I, [C.POINTER(TinyVitCfg)]),
    "gg_tinyvit_tensor_info": (_I, [C.POINTER(TinyVitCfg), _I, C.c_char_p, _I, C.POINTER(_L), C.POINTER(_L), C.POINTER(_I),
                                    C.POINTER(_L), C.POINTER(_I)]),
    "gg_tinyvit_param_floats": (_L, [C.POINTER(TinyVitCfg)]),
    "gg_tinyvit_buffer_floats": (_L, [C.POINTER(TinyVitCfg)]),
    "gg_tinyvit_num_counters": (_I, [C.POINTER(TinyVitCfg)]),
    "gg_tinyvit_num_drop_slots": (_I, [C.POINTER(TinyVitCfg)]),
    "gg_drop_path_scales": (_I, [_P, _I, _I, C.c_uint64, C.c_uint64, _P, _P]),
    "gg_transpose_f32": (_I, [_P, _I, _I, _P, _L, _P]),
    "gg_tinyvit_wcache_bytes": (_L, [C.POINTER(TinyVitCfg)]),
    "gg_tinyvit_workspace_bytes": (_L, [C.POINTER(TinyVitCfg), _I, _I]),
    "gg_tinyvit_workspace_bytes_masked": (_L, [C.POINTER(TinyVitCfg), _I, _I, _P]),
    "gg_tinyvit_refresh_weights": (_I, [C.POINTER(TinyVitCfg), _P, _P, _P]),
    "gg_tinyvit_refresh_weights_masked": (_I, [C.POINTER(TinyVitCfg), _P, _P, C.c_char_p, _P]),
    "gg_preprocess_bilinear": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _P]),
    "gg_preprocess_pil_workspace_bytes": (C.c_int64, [_I, _I, _I, _I, _I, _I]),
    "gg_preprocess_pil": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _P, _P, _P]),
    "gg_segment_mean": (_I, [_P, _L, _P, _P, _I, _I, _P, _P]),
    "gg_tinyvit_forward": (_I, [C.POINTER(TinyVitCfg), _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, C.c_char_p, _P]),
    "gg_tinyvit_backward": (_I, [C.POINTER(TinyVitCfg), _I, _P, _P, _P, _P, _P, _P, C.c_char_p, _P, STAGE_DONE_FN, _P]),
    "gg_tinyvit_activation_info": (_I, [C.POINTER(TinyVitCfg), _I, C.c_char_p, C.POINTER(_L), C.POINTER(_L)]),
    "gg_tinyvit_activation_info_masked": (_I, [C.POINTER(TinyVitCfg), _I, C.c_char_p, _P, C.POINTER(_L), C.POINTER(_L)]),
    "gg_clip_num_tensors": (_I, [C.POINTER(ClipCfg)]),
    "gg_clip_tensor_info": (_I, [C.POINTER(ClipCfg), _I, C.c_char_p, _I, C.POINTER(_L), C.POINTER(_L), C.POINTER(_I),
                                 C.POINTER(_L)]),
    "gg_clip_param_floats": (_L, [C.POINTER(ClipCfg)]),
    "gg_clip_wcache_bytes": (_L, [C.POINTER(ClipCfg)]),
    "gg_clip_workspace_bytes": (_L, [C.POINTER(ClipCfg), _I, _I, C.c_char_p]),
    "gg_clip_first_trained_layer": (_I, [C.POINTER(ClipCfg), C.c_char_p]),
    "gg_clip_refresh_weights": (_I, [C.POINTER(ClipCfg), _P, _P, _P]),
    "gg_clip_forward": (_I, [C.POINTER(ClipCfg), _I, _I, _P, _P, _P, _P, _P, _P, C.c_char_p, _P]),
    "gg_clip_backward": (_I, [C.POINTER(ClipCfg), _I, _P, _P, _P, _P, _P, _P, C.c_char_p, _P]),
}
SYMBOLS = list(SIGNATURES)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GgError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)           # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise GgError(f"{what or 'libgg'} failed ({rc}): {lib().gg_last_error().decode(errors='replace')}")


def require_gpu() -> None:
    if not torch.cuda.is_available():
        raise GgError("no HIP device visible: the geoguessr_ai_amd hot path only runs on an MI355X (gfx950); "
                      "there is no CPU fallback")


def ptr(t: Optional[torch.Tensor], dtype: Optional[torch.dtype] = None, name: str = "tensor") -> Optional[int]:
    """Device pointer of a contiguous GPU tensor (None passes through as NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise GgError(f"{name} must live on the GPU (got device {t.device}); there is no CPU fallback")
    if not t.is_contiguous():
        raise GgError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise GgError(f"{name} must be {dtype}, got {t.dtype}")
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def f32(x: float) -> C.c_float:
    return C.c_float(float(x))


def source_hash() -> str:
    """sha256 over the sources libgg.so is built from (csrc/* and include/gg.h, in name order): identifies the build that a committed profile
    (profiles/*_hbm_traffic_pmc_*.json) was measured on -- bench.py marks its traffic figure stale when the running tree differs."""
    import hashlib
    root = os.path.dirname(os.path.abspath(__file__))
    files = sorted(os.path.join(root, "csrc", f) for f in os.listdir(os.path.join(root, "csrc")) if f.endswith((".hip", ".h", ".cpp", "Makefile")))
    files.append(os.path.join(os.path.dirname(root), "include", "gg.h"))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()
