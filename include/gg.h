/* libgg -- C-ABI of the MI355X-native (gfx950) hot path of CogitoNTNU/geoguessr-ai.
 *
 * The reference has no FFI layer: its boundary is a duck-typed Python nn.Module interface
 * (SURVEY.md 8b).  This header is what the drop-in Python shims under geoguessr-ai_amd/ bind with
 * ctypes (INTEGRATION.md shows the stubs).  Conventions for every entry point:
 *   - returns 0 on success, < 0 on error; gg_last_error() gives the thread-local message;
 *   - all pointers are CALLER-OWNED DEVICE pointers unless marked "host"; tensors, workspaces and optimizer state are never allocated or
 *     freed by the library.  Process-global state it does keep, all of it internal and none of it part of a result: the HIP-graph cache of
 *     gg_tinyvit_forward / _backward (captured graphs, one private capture stream per device, hit counters: gg_graph_stats reads them,
 *     gg_graph_clear() is the teardown and must run before the buffers a captured graph refers to are freed), the launch log of the profiling
 *     hooks (gg_prof_*: off unless enabled), and lazily allocated 16 MiB split-K slab buffers of gg_gemm_nt_f32: one per (device, stream) that issued the form, at
 *     most 8 (least recently used released), plus one per captured graph whose launches use the form; gg_graph_clear() releases them all.  Under a CALLER's own
 *     stream capture no slab can be handed out and the form runs unsplit (same product, different rounding than the eager call);
 *   - `stream` is a hipStream_t; work is only enqueued, nothing here synchronises;
 *   - bf16 tensors are passed as void*; matrices are row-major with an explicit leading dimension
 *     in ELEMENTS; activations are NHWC / [tokens, channels].
 * Reference citations are relative to the reference repository root.
 */
#ifndef GG_H
#define GG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GG_VERSION 1
int gg_version(void);
const char* gg_last_error(void);          /* host string, valid until the next failing call on this thread */

enum { GG_ACT_CODE_NONE = 0, GG_ACT_CODE_GELU = 1, GG_ACT_CODE_QUICK_GELU = 2 };

/* ---------------------------------------------------------------- dense contractions (MFMA)
 * C[M,N] = epilogue( A[M,K] . B[N,K]^T ).  Replaces every nn.Linear / 1x1 Conv2d / im2col'd 3x3 Conv2d the
 * reference reaches through timm (models/tinyvit.py:135), transformers CLIP (pretrain/clip_embedder.py:63)
 * and SuperGuessr.cell_layer (models/super_guessr.py:354), plus their autograd dgrad / wgrad.
 * epilogue order: +bias[n] -> (store preact) -> act -> *act'(dact_preact) -> *rowscale[m/rows_per_scale]
 *                 -> +residual[m,n] -> store (bf16 or f32).  K, lda, ldb multiples of 8. */
typedef struct GgGemmArgs {
    const void* A; int64_t lda;           /* bf16 [M,K] */
    const void* B; int64_t ldb;           /* bf16 [N,K] */
    void* C; int64_t ldc;                 /* bf16 or f32 [M,N]; split_k>1: f32 [split_k][M][ldc] partials */
    int M, N, K;
    const float* bias;                    /* f32 [N] or NULL */
    int act;                              /* GG_ACT_CODE_* */
    void* preact;                         /* bf16 [M,ldc] or NULL */
    const float* rowscale; int rows_per_scale;   /* DropPath per-sample scale, f32 [ceil(M/rows_per_scale)] or NULL */
    const void* residual; int64_t ldr;    /* bf16 [M,N] or NULL */
    const void* dact_preact; int dact;    /* bf16 [M,ldc]: multiply by act'(.) (backward through an activation) */
    float* colstats;                      /* f32 [gg_gemm_colstats_rows(M)][2][N] BatchNorm partials or NULL */
    int out_f32;
    int split_k;
    /* optional second A source: contraction columns k >= k_split come from A2[M, K - k_split] (same lda; k_split % 64 == 0).
     * With B = [diag(c0) W ; diag(c1) W] and bias = c2 . W this is the dgrad of a ConvNorm taken straight from
     * (dz, y): BatchNorm backward's  dy = c0*dz + c1*y + c2  never touches memory (gg_bn_bwd_fold_weights). */
    const void* A2; int k_split;
    /* BatchNorm-backward epilogue (conv dgrads feeding act(BN(y)), timm ConvNorm + GELU, models/tinyvit.py:135):
     * C = dz = acc * act'(gamma*xhat + beta), xhat = (bn_y - mean)*rstd with stat = [mean[N], rstd[N]], bn_y bf16 [M,ldc];
     * colstats <- per M-tile column sums of (dz, dz*xhat) for gg_bn_bwd_finalize. */
    const void* bn_y; const float* bn_stat; const float* bn_gamma; const float* bn_beta; int bn_act;
    /* BatchNorm prologue: A holds the PRE-BatchNorm output of the previous ConvNorm; act(gamma*(A-mean)*rstd+beta) with
     * a_bn_stat = [mean[K], rstd[K]] is applied while the A tile is staged (plain / colstats epilogue only, K <= 1024). */
    const float* a_bn_stat; const float* a_bn_gamma; const float* a_bn_beta; int a_bn_act;
} GgGemmArgs;
int gg_gemm_nt(const GgGemmArgs* args, void* stream);
int gg_gemm_colstats_rows(int M);
int gg_stat_rows_capacity(int rows);       /* rows a partial-statistics buffer must hold (valid rows + reduction scratch) */
/* weight-gradient form: partials[splits][N][K] (f32) = per m-split  dY[M,N]^T . X[M,K]  (no transposed operand copies) */
int gg_gemm_tn_splits(int M, int N, int K);
int gg_gemm_tn(const void* dY, int64_t ldy, const void* X, int64_t ldx, int M, int N, int K, const float* rowscale, int rows_per_scale,
               float* partials, int splits, void* stream);
/* same with dY := coef0*dz + coef1*y + coef2 per column (coef = gg_bn_bwd_finalize's [3][N]), formed while loading: the weight gradient of a
   ConvNorm whose dy feeds nothing else (patch_embed.conv1) without the BatchNorm-backward apply pass or the dy tensor */
int gg_gemm_tn_bn(const void* dz, const void* y, int64_t ldy, const float* coef, const void* X, int64_t ldx, int M, int N, int K, float* partials,
                  int splits, void* stream);
int gg_gemm_tn_bn_f32(const void* dz, const void* y, int64_t ldy, const float* coef, const void* X, int64_t ldx, int M, int N, int K, float* partials,
                      int splits /* gg_gemm_tn_f32_splits */, void* stream);      /* f32 storage twin (patch_embed.conv1 / conv2 in the fp32 mode) */
int gg_splitk_reduce(const float* partials, float* out, int64_t n, int splits, int accumulate, float scale, void* stream);
int gg_transpose_bf16(const void* in, int64_t ld, void* out, int64_t ldo, int R, int C, const float* rowscale,
                      int rows_per_scale, void* stream);
int gg_cast_transpose_f32(const float* in, int R, int C, void* out, int64_t ldo, void* outT, int64_t ldt, void* stream);
int gg_transpose_f32(const float* in, int R, int C, float* outT, int64_t ldt, void* stream);   /* f32 [R,C] -> f32 [C,ldt], ldt >= R (pad columns untouched) */
int gg_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream);
int gg_cast_bf16_to_f32(const void* in, float* out, int64_t n, void* stream);
int64_t gg_colsum_scratch_floats(int M, int C);
int gg_colsum_bf16(const void* x, int64_t ld, int M, int C, const float* rowscale, int rows_per_scale, float* scratch,
                   float* out, int accumulate, void* stream);

/* ---------------------------------------------------------------- 3x3 convolutions (timm ConvNorm convs)
 * Dense 3x3 (PatchEmbed) = im2col + gg_gemm_nt, k order (ky,kx,ci); depthwise 3x3 (MBConv.conv2,
 * PatchMerging.conv2, TinyVitBlock.local_conv) direct, taps f32 [9][C]. */
int gg_im2col_nchw3_f32(const float* x, void* col, int B, int H, int W, int stride, void* stream);   /* (B,3,H,W) f32 -> bf16 [B*Ho*Wo,32] */
int gg_im2col_nhwc_bf16(const void* x, void* col, int B, int H, int W, int C, int stride, void* stream);
/* same gather over act(BatchNorm(y)) of a saved pre-BatchNorm conv output y (stat = [mean | rstd][C]); the activation tensor is not stored */
int gg_im2col_nhwc_bn_bf16(const void* y, const float* stat, const float* gamma, const float* beta, int act, void* col, int B, int H, int W, int C,
                          int stride, void* stream);
int gg_col2im_nhwc_bf16(const void* dcol, void* dx, int B, int H, int W, int C, int stride, void* stream);
/* stride-2 col2im fused with the BatchNorm-backward reduce of the ConvNorm whose output was gathered: writes dz = da * act'(BN(y)) and
   nparts partial rows [2][C] (sum dz, sum dz*xhat) for gg_bn_bwd_finalize; da is never stored */
int gg_col2im_nhwc_bnbwd_bf16(const void* dcol, const void* y, const float* stat, const float* gamma, const float* beta, int act, void* dz,
                              float* part, int nparts, int B, int H, int W, int C, void* stream);
int gg_col2im_nhwc_bnbwd_f32(const float* dcol, const float* y, const float* stat, const float* gamma, const float* beta, int act, float* dz,
                             float* part, int nparts, int B, int H, int W, int C /* % 4 */, void* stream);
int gg_dwconv_stat_rows(int B, int Ho, int Wo, int C, int stride);   /* partial-statistics rows gg_dwconv3x3_fwd writes */
int gg_dwconv_tiled_stat_rows(int B, int Ho);                          /* ... the producer-fused forward variant (LDS-tiled kernel) */
int gg_dwconv_fused_stat_rows(int B, int H, int W, int C, int with_input_fusion);   /* ... the fused data gradient with ep_y */
int gg_dwconv_fwd_fused_stat_rows(int B, int H, int W, int C, int stride);   /* ... the fused forward; H, W = input size */
int gg_dwconv3x3_fwd(const void* x, const float* taps, void* y, int B, int H, int W, int C, int stride, float* colstats, void* stream);
int gg_dwconv3x3_fwd_fused(const void* x_prebn, const float* in_stat, const float* in_gamma, const float* in_beta, int in_act,
                           const float* taps, void* y, int B, int H, int W, int C, int stride, float* colstats, void* stream);
int gg_dwconv3x3_bwd_data_fused(const void* dz_in, const void* y_in, const float* in_coef, const float* taps, void* out, int B, int H, int W,
                                int C, const void* ep_y, const float* ep_stat, const float* ep_gamma, const float* ep_beta, int ep_act,
                                float* ep_partials, void* stream);
int gg_dwconv_s2_fused_stat_rows(int B, int H, int W, int C);            /* partial rows of the stride-2 fused data gradient */
int gg_dwconv3x3_s2_bwd_data_fused(const void* dz_in, const void* y_in, const float* in_coef, const float* taps, void* out, int B, int H,
                                   int W, int C, const void* ep_y, const float* ep_stat, const float* ep_gamma, const float* ep_beta,
                                   int ep_act, float* ep_partials, void* stream);   /* (H, W) = the conv INPUT map; stride 2 */
int gg_dwconv3x3_bwd_data(const void* dy, const float* taps, void* dx, int B, int H, int W, int C, int stride, void* stream);
int64_t gg_dwconv_wgrad_scratch_floats(int B, int H, int W, int C, int stride);
int gg_dwconv3x3_bwd_weight(const void* x, const void* dy, int B, int H, int W, int C, int stride, float* scratch, float* grad /* (C,1,3,3) */,
                            int accumulate, void* stream);

/* ---------------------------------------------------------------- BatchNorm2d (train-mode batch statistics)
 * stat = [2][C] (mean, rstd).  Partials come from gg_gemm_nt.colstats / gg_dwconv3x3_fwd.colstats. */
int gg_bn_finalize(float* partials /* capacity gg_stat_rows_capacity(nparts) rows */, int nparts, int C, int64_t count, float eps, float momentum, float* stat,
                   float* running_mean, float* running_var, void* stream);
int gg_bn_eval_stat(const float* running_mean, const float* running_var, int C, float eps, float* stat, void* stream);
int gg_bn_apply(const void* y, const float* stat, const float* gamma, const float* beta, int64_t M, int C, int act,
                const void* residual, const float* rowscale, int rows_per_scale, void* out, void* stream);
int64_t gg_bn_bwd_scratch_floats(int64_t M, int C);
int gg_bn_bwd_rows(int64_t M, int C);
/* the three passes of BatchNorm backward, separately (producers / consumers can absorb the outer two):
 *   reduce: dz = dout*act'(BN(y)) [+residual/DropPath form], partial rows (sum g, sum g*xhat);  finalize: rows -> coef [3][C]
 *   with dy = coef0*g + coef1*y + coef2, plus dgamma/dbeta;  apply: dy from (dz, y, coef) */
int gg_bn_bwd_reduce(const void* dout, const void* y, const float* stat, const float* gamma, const float* beta, int64_t M, int C, int act,
                     const void* residual, const float* rowscale, int rows_per_scale, void* dz, float* partials, void* stream);
int gg_bn_bwd_finalize(float* partials, int nparts, int C, int64_t count, const float* stat, const float* gamma, float* coef,
                       float* dgamma, float* dbeta, int accumulate, void* stream);
int gg_bn_bwd_apply(const void* dz, const void* y, const float* coef, int64_t M, int C, const float* rowscale, int rows_per_scale,
                    void* dy, void* stream);
/* W f32 [Cout][Cin] (1x1 conv), coef [3][Cout], stat [2][Cout] -> Bf bf16 [Cin][2*Cout], bias f32 [Cin] for the two-source
 * dgrad  dx = [dz | y] . Bf^T + bias  (GgGemmArgs.A2) */
int gg_bn_bwd_fold_weights(const float* W, const float* coef, const float* stat, int Cout, int Cin, void* Bf, float* bias, void* stream);
int gg_bn_bwd(const void* dout, const void* y, const float* stat, const float* gamma, const float* beta, int64_t M, int C, int act,
              const void* residual, const float* rowscale, int rows_per_scale, void* dz, void* dy, float* scratch,
              float* dgamma, float* dbeta, int accumulate, void* stream);

/* ---------------------------------------------------------------- LayerNorm / pooling */
int gg_layernorm_fwd(const void* x, int x_f32, const float* gamma, const float* beta, int64_t M, int C, float eps, void* out,
                     int out_f32, float* mean, float* rstd, void* stream);
/* LayerNorm of BatchNorm(y) for a saved pre-BatchNorm conv output y (TinyViT block: local_conv -> norm2): xout = bf16(BN(y)) is the
   residual stream, out = LN(xout); replaces gg_bn_apply + gg_layernorm_fwd (tiny_vit.py TinyVitBlock.forward) */
int gg_layernorm_fwd_bn(const void* y, const float* bn_stat, const float* bn_gamma, const float* bn_beta, void* xout, const float* gamma,
                        const float* beta, int64_t M, int C, float eps, void* out, float* mean, float* rstd, void* stream);
int64_t gg_layernorm_bwd_scratch_floats(int64_t M, int C);
int gg_layernorm_bwd(const void* dout, const void* x, int f32, const float* mean, const float* rstd, const float* gamma, int64_t M,
                     int C, const void* dres, void* dx, float* scratch, float* dgamma, float* dbeta, int accumulate, void* stream);
/* the same backward that also leaves per-block column sums [rows][2][C] = (sum dx*x, sum dx) of its result dx (LayerNorm backward + dres)
   against its raw input x.  When x = BatchNorm(y) with FROZEN BatchNorm parameters, gg_bn_bwd_coef_from_x turns them into the coefficients
   dy = c0*dx + c1*y + c2 of that BatchNorm's backward (consumed by the fused depthwise data gradient), so the stand-alone reduce pass over
   (dx, y) disappears (TinyViT block: local_conv's BatchNorm sits in front of norm2; timm tiny_vit.py TinyVitBlock.forward).
   part: (gg_layernorm_bwd_colsum_rows(M) + 64) rows; C <= 640; no LayerNorm parameter gradients in this form */
int gg_layernorm_bwd_colsum_rows(int64_t M);
int gg_layernorm_bwd_colsum(const void* dout, const void* x, int f32, const float* mean, const float* rstd, const float* gamma, int64_t M, int C,
                            const void* dres, void* dx, float* part, void* stream);
int gg_bn_bwd_coef_from_x(float* part, int nparts, int C, int64_t count, const float* stat /* [mean|rstd][C] */, const float* gamma,
                          const float* beta, float* coef /* [3][C] */, void* stream);
int gg_token_mean_fwd(const void* x, float* out, int B, int T, int C, void* stream);
int gg_token_mean_bwd(const float* dout, void* dx, int B, int T, int C, void* stream);
int gg_view_mean_fwd(const float* emb, void* out, int64_t ldo, int N, int V, int C, void* stream);   /* models/super_guessr.py:347 */
int gg_view_mean_bwd(const void* dmean, int64_t ld, float* demb, int N, int V, int C, void* stream);

/* ---------------------------------------------------------------- window attention (timm Attention.forward; CLIP MHSA) */
typedef struct GgAttnArgs {
    const void* qkv; int64_t ld;          /* bf16 [tokens, ld] */
    int q_off, k_off, v_off, head_stride; /* column of head h = off + h*head_stride */
    int head_dim;                         /* 32 (TinyViT) or 64 (CLIP) */
    int num_heads, num_windows, tokens_per_window;
    int window_size, map_h, map_w;        /* window_size > 0: ws x ws windows of an (map_h, map_w) NHWC token map; 0: linear */
    const void* bias;                     /* EXPANDED attention_biases / scale, bf16 [num_heads][Np][Np] (gg_attention_expand_bias with
                                             the same `scale`; Np = gg_attention_padded_tokens(tokens_per_window)) or NULL */
    float scale;
    void* out; int64_t ldo;               /* forward: bf16 [tokens, ldo], head h at column h*head_dim */
    const void* dout; int64_t lddo;       /* backward */
    void* dqkv;                           /* backward: same layout as qkv */
    float* dbias;                         /* backward: f32 [num_heads][ws*ws], ACCUMULATED, or NULL */
    float* dbias_scratch;                 /* optional f32 [(num_windows + 64) * num_heads * ws*ws]: per-window partials, reduced in a
                                             second deterministic stage; NULL: float atomics */
    float* lse;                           /* f32 [tokens][num_heads] row log-sum-exp: forward writes (may be NULL), backward reads;
                                             backward also reads `out` (the forward result) */
    const float* bias_table;              /* COMPACT attention_biases f32 [num_heads][ws*ws] (index |dy|*ws+|dx|, timm's first-seen order)
                                             or NULL: what the online-softmax and resident-window kernels read (gg_attention_flash_*;
                                             gg_attention_fwd/bwd beyond 256 tokens per window, where `bias` is ignored; gg_attention_bwd of
                                             12 x 12 / 14 x 14 windows, which runs the single-pass backward).  A biased gg_attention_fwd/bwd call
                                             therefore passes BOTH forms.  gg_attention_bwd (bf16 storage, both given) recomputes P with the value the
                                             expanded table holds, bf16(bias / scale), i.e. with its own forward's; the gg_attention_flash_fwd / _bwd pair
                                             reads only this compact table in both passes (`bias` is ignored there) */
    float* ds_scratch;                    /* optional (gg_attention_flash_bwd, windows beyond 256 tokens only -- see gg_attention_flash_single_pass): f32
                                             [gg_attention_flash_ds_scratch_floats(...)] = 4 * windows * heads * roundup16(tokens)^2 bytes (22 MB per image at
                                             CLIP ViT-L/14-336).  With it the dK/dV pass runs first and hands dS to the dQ pass, which then is ONE product
                                             (dQ = scale * dS K) instead of three plus a second round of exponentials; NULL: both passes recompute the scores */
} GgAttnArgs;
int gg_attention_padded_tokens(int tokens_per_window);
/* full[h][q][k] = bf16(table[h][|dy|*ws+|dx|] / scale) (-inf for padded keys): the kernels start their score accumulators from it,
 * softmax(scale * (q.k + bias/scale)) == softmax(scale*q.k + bias)  (timm TinyVit Attention, models/tinyvit.py:135) */
int gg_attention_expand_bias(const float* table /* [num_heads][ws*ws] */, int num_heads, int window_size, float scale, void* full /* bf16 */, void* stream);
int gg_attention_fwd(const GgAttnArgs* args, void* stream);
int gg_attention_bwd(const GgAttnArgs* args, void* stream);
/* fp16 forward (inference; fp16 qkv / out, v_mfma_f32_16x16x32_f16, f32 softmax): the CLIP tower's fp16 mode -- HF CLIPAttention as reached from
 * pretrain/clip_embedder.py:63-65, at the precision BASELINE config c4 names.  No bias; beyond 256 tokens it forwards to gg_attention_flash_fwd(dtype 2). */
int gg_attention_fwd_f16(const GgAttnArgs* args, void* stream);
/* Online-softmax (flash) form for ANY tokens_per_window (1024-token windows of the reference's default tiny_vit_21m_512, config.py:9;
 * 577 tokens of CLIP ViT-L/14-336, config.py:6) and for the reference-precision mode: dtype 0 = bf16, 1 = f32, 2 = fp16 (forward only) storage of
 * qkv / out / dout / dqkv; arithmetic is f32-accurate either way: f32 MFMA, or -- fp32 storage, head dim 32, 7 x 7 / 12 x 12 / 14 x 14 windows --
 * split products on the bf16 MFMA (x = x1 + x2 + x3 in bf16, six products per f32 product: DESIGN.md 5 has the error table); bf16 storage of those
 * window shapes runs the same single-pass backward with plain bf16 products.  window_size <= 32.  dbias_scratch (optional): f32
 * [gg_attention_flash_dbias_rows(num_windows, tokens_per_window)][num_heads][ws*ws]. */
int gg_attention_flash_fwd(const GgAttnArgs* args, int dtype, void* stream);
int gg_attention_flash_bwd(const GgAttnArgs* args, int dtype, void* stream);
int64_t gg_attention_flash_dbias_rows(int num_windows, int tokens_per_window);
int64_t gg_attention_flash_ds_scratch_floats(int num_windows, int num_heads, int tokens_per_window);
/* 1 when gg_attention_flash_bwd runs its single-pass kernel for this window (at most 256 tokens: Q, dO and the dQ accumulator of a whole window fit
 * one CU's LDS): ds_scratch is then neither needed nor read -- workspace planners skip it. */
int gg_attention_flash_single_pass(int tokens_per_window, int head_dim, int window_size, int with_dbias);

/* ---------------------------------------------------------------- experiment: fp32-accurate GEMM on the bf16 matrix pipe
 * Not part of the drop-in path (no reference interface behind it): measured by tools/bench_split3.py next to gg_gemm_nt_f32.
 * gg_split3_bf16: x (f32 [rows][ldx]) -> planes bf16 [3][rows][cols] with x = p1 + p2 + p3 to 24 bits.
 * gg_gemm_nt_split3: C[M,N] (f32) = A . B^T from the planes of both operands (six bf16 MFMA products, f32 accumulation); K, lda, ldb multiples of 8. */
int gg_split3_bf16(const float* x, int64_t rows, int cols, int64_t ldx, void* planes, void* stream);
/* the producer-side form of the split: f32 LayerNorm (timm LayerNorm in front of qkv / fc1) whose result leaves as the three planes [3][M][C] */
int gg_layernorm_fwd_split3(const float* x, const float* gamma, const float* beta, int64_t M, int C, float eps, void* planes, float* mean, float* rstd, void* stream);
int gg_layernorm_fwd_bn_split3(const float* y, const float* bn_stat, const float* bn_gamma, const float* bn_beta, float* xout, const float* gamma, const float* beta,
                               int64_t M, int C, float eps, void* planes, float* mean, float* rstd, void* stream);
int gg_gemm_nt_split3(const void* a_planes, int64_t lda, const void* b_planes, int64_t ldb, float* C, int64_t ldc, int M, int N, int K, const float* bias,
                      void* stream);
/* the same with the epilogue family of the model's Linears (what gg_gemm_nt_f32 offers for qkv / proj / fc1 / fc2 and their dgrads):
 * v = acc + bias; preact = v (optional copy); v = act(v)  |  v = acc * act'(dact_preact); v *= rowscale[m / rows_per_scale]; v += residual;
 * result as f32 C and / or as three bf16 planes c_planes [3][M][ldp] (the next split GEMM's A operand).  preact / dact_preact share ldc. */
typedef struct {
    const void* a_planes; int64_t lda;      /* bf16 [3][M][lda] */
    const void* b_planes; int64_t ldb;      /* bf16 [3][N][ldb] */
    int M, N, K;
    float* C; int64_t ldc;                  /* f32 [M][ldc] or NULL */
    void* c_planes; int64_t ldp;            /* bf16 [3][M][ldp] or NULL */
    const float* bias;                      /* [N] or NULL */
    int act;                                /* GG_ACT_* applied to acc + bias */
    float* preact;                          /* f32 [M][ldc] copy of acc + bias, or NULL */
    const float* rowscale; int rows_per_scale;
    const float* residual; int64_t ldr;
    const float* dact_preact; int dact;     /* v = acc * act'(dact_preact[m][n]) */
} GgSplit3Args;
int gg_gemm_nt_split3_ex(const GgSplit3Args* args, void* stream);
/* the same with the A operand as the f32 tensor itself ([M][lda], split into its three bf16 terms while the kernel stages it: no plane copy of an
 * activation in HBM; args->a_planes / lda are ignored) and the weight as cached planes: the form every Linear of the fp32 model can take. */
int gg_gemm_nt_split3_af32(const GgSplit3Args* args, const float* A, int64_t lda, int64_t b_plane_stride /* elements between the weight's planes (0: N * ldb) */,
                           void* stream);
/* ... with BatchNorm partials of the result (plain epilogue only): colstats [gg_gemm_colstats_rows(M)][2][N] as GgGemmArgs.colstats -- a ConvNorm's 1 x 1 convolution */
int gg_gemm_nt_split3_af32_stats(const GgSplit3Args* args, const float* A, int64_t lda, int64_t b_plane_stride, float* colstats, void* stream);
/* ... with A := act(BatchNorm(A)) formed in the loader, in front of the split (bn_stat = [mean | rstd][K] as gg_bn_finalize leaves it; act GG_ACT_NONE / GELU /
 * QUICK_GELU): the split twin of GgGemmArgs.a_bn_* -- MBConv.conv3 reading BatchNorm2 + GELU of the depthwise convolution's saved output without the activation
 * tensor ever being written (timm MBConv.forward, reached from models/tinyvit.py:135).  384 <= K <= 1024, plain epilogue, colstats optional. */
int gg_gemm_nt_split3_af32_pro(const GgSplit3Args* args, const float* A, int64_t lda, int64_t b_plane_stride, const float* bn_stat, const float* bn_gamma,
                               const float* bn_beta, int bn_act, float* colstats, void* stream);
/* weight gradient dW[N][K] = sum_m s_m dY[m][n] X[m][k] as split products (both f32 operands split in the kernel's loader): the arguments and the slab protocol
 * of gg_gemm_tn_f32 (partials [splits][N][K], reduced by gg_splitk_reduce); splits from gg_gemm_tn_split3_splits. */
int gg_gemm_tn_split3_splits(int M, int N, int K);
int gg_gemm_tn_split3(const float* dY, int64_t ldy, const float* X, int64_t ldx, int M, int N, int K, const float* rowscale, int rows_per_scale, float* partials,
                      int splits, void* stream);

/* ---------------------------------------------------------------- reference-precision (fp32) mode
 * The reference computes this whole path in fp32 (torch defaults).  These entry points are the f32-storage twins
 * of the kernels above: f32 activations [tokens, channels], f32 MFMA (v_mfma_f32_16x16x4_f32 -- exact f32 products, f32
 * accumulation), erf GELU through an fp32-accurate Phi (gg_phi_f32: 1.2 ulp of 1).  GgTinyVitCfg.act_dtype = 1 runs the whole encoder on them. */
int gg_gemm_nt_f32(const GgGemmArgs* args, void* stream);   /* all matrices f32; K, lda, ldb multiples of 4; args->split_k is ignored (the library splits a long contraction of a small launch itself, into slabs
                                                               * it owns per (device, stream); under a caller's own stream capture the launch runs unsplit, the library's graph cache gives every captured graph slabs of its own); the A2 (two-source) and BatchNorm-fused forms of GgGemmArgs are honoured (see below) */
int gg_gemm_tn_f32_splits(int M, int N, int K);
int gg_gemm_tn_f32(const void* dY, int64_t ldy, const void* X, int64_t ldx, int M, int N, int K, const float* rowscale, int rows_per_scale,
                   float* partials, int splits, void* stream);
int gg_colsum_f32(const float* x, int64_t ld, int M, int C, const float* rowscale, int rows_per_scale, float* scratch /* gg_colsum_scratch_floats */,
                  float* out, int accumulate, void* stream);
int gg_im2col_nchw3_f32_f32(const float* x, float* col, int B, int H, int W, int stride, void* stream);      /* -> f32 [B*Ho*Wo, 32] */
/* stat != NULL: x is a saved pre-BatchNorm conv output, act(BatchNorm(x)) is gathered (C <= 512) */
int gg_im2col_nhwc_f32(const float* x, const float* stat, const float* gamma, const float* beta, int act, float* col, int B, int H, int W, int C,
                       int stride, void* stream);
int gg_col2im_nhwc_f32(const float* dcol, float* dx, int B, int H, int W, int C, int stride, void* stream);
int gg_dwconv_f32_stat_rows(int B, int Ho, int Wo, int C, int stride);      /* partial rows written by gg_dwconv3x3_fwd_f32.colstats (Ho, Wo = output map) */
int gg_dwconv3x3_fwd_f32(const float* x, const float* taps, float* y, int B, int H, int W, int C, int stride, float* colstats, void* stream);
int gg_dwconv3x3_bwd_data_f32(const float* dy, const float* taps, float* dx, int B, int H, int W, int C, int stride, void* stream);
int64_t gg_dwconv_f32_wgrad_scratch_floats(int B, int H, int W, int C, int stride);
int gg_dwconv3x3_bwd_weight_f32(const float* x, const float* dy, int B, int H, int W, int C, int stride, float* scratch, float* grad, int accumulate,
                                void* stream);
int gg_bn_apply_f32(const float* y, const float* stat, const float* gamma, const float* beta, int64_t M, int C, int act, const float* residual,
                    const float* rowscale, int rows_per_scale, float* out, void* stream);
int gg_bn_bwd_f32(const float* dout, const float* y, const float* stat, const float* gamma, const float* beta, int64_t M, int C, int act,
                  const float* residual, const float* rowscale, int rows_per_scale, float* dz, float* dy, float* scratch /* gg_bn_bwd_scratch_floats */,
                  float* dgamma, float* dbeta, int accumulate, void* stream);
int gg_gemm_f32_set_trace(void* buf);   /* dev: per-workgroup timeline of gg_gemm_nt_f32's ring kernel, 8 x uint64 per tile; NULL = off */
int gg_layernorm_fwd_bn_f32(const float* y, const float* bn_stat, const float* bn_gamma, const float* bn_beta, float* xout, const float* gamma,
                            const float* beta, int64_t M, int C, float eps, float* out, float* mean, float* rstd, void* stream);   /* f32 twin of gg_layernorm_fwd_bn */
int gg_bn_bwd_reduce_f32(const float* dout, const float* y, const float* stat, const float* gamma, const float* beta, int64_t M, int C, int act,
                         const float* residual, const float* rowscale, int rows_per_scale, float* dz, float* partials, void* stream);
int gg_bn_bwd_apply_f32(const float* dz, const float* y, const float* coef, int64_t M, int C, const float* rowscale, int rows_per_scale, float* dy,
                        void* stream);
/* f32 twins of the fused depthwise kernels (BatchNorm passes of the ConvNorms either side ride on the loads / stores; stride 1 for the data
 * gradient).  gg_gemm_nt_f32 takes the matching GEMM-side fusions through GgGemmArgs: bn_y.. (BatchNorm-backward epilogue), a_bn_stat..
 * (A := act(BN(A)) while staging) and A2 + a_bn_stat = coef [3][K] (A := coef0*A + coef1*A2 + coef2: BatchNorm backward's apply step). */
int gg_dwconv3x3_fwd_fused_f32(const float* x_prebn, const float* in_stat, const float* in_gamma, const float* in_beta, int in_act, const float* taps,
                               float* y, int B, int H, int W, int C, int stride, float* colstats /* gg_dwconv_f32_stat_rows(B,Ho,Wo,C,stride) rows */, void* stream);
int gg_dwconv3x3_bwd_data_fused_f32(const float* dz_in, const float* y_in, const float* in_coef, const float* taps, float* out, int B, int H, int W, int C,
                                    const float* ep_y, const float* ep_stat, const float* ep_gamma, const float* ep_beta, int ep_act,
                                    float* ep_partials /* gg_dwconv_f32_stat_rows(B,H,W,C,1) rows */, void* stream);
int gg_dwconv_f32_s2_fused_stat_rows(int B, int H, int W, int C);         /* partial rows of the stride-2 fused data gradient (H, W = the conv INPUT map) */
int gg_dwconv3x3_s2_bwd_data_fused_f32(const float* dz_in, const float* y_in, const float* in_coef, const float* taps, float* out, int B, int H, int W,
                                       int C, const float* ep_y, const float* ep_stat, const float* ep_gamma, const float* ep_beta, int ep_act,
                                       float* ep_partials, void* stream);   /* f32 twin of gg_dwconv3x3_s2_bwd_data_fused (PatchMerging backward) */
/* fp16 storage (the CLIP tower's act_dtype 2 -- BASELINE config c4 "MFMA fp16"; inference only): gg_gemm_nt with fp16 A / B / C / residual on
 * v_mfma_f32_16x16x32_f16 (bias, QuickGELU / GELU, residual epilogues; no BatchNorm-fused / two-source / split-K forms), LayerNorm forward, the
 * token mean, an f32 -> fp16 cast; attention runs through gg_attention_flash_fwd with dtype 2 (fp16 storage, f32 arithmetic). */
int gg_gemm_nt_f16(const GgGemmArgs* args, void* stream);
int gg_layernorm_fwd_f16(const void* x, const float* gamma, const float* beta, int64_t M, int C, float eps, void* out, void* stream);
int gg_token_mean_fwd_f16(const void* x, float* out, int B, int T, int C, void* stream);
int gg_cast_f32_to_f16(const float* in, void* out, int64_t n, void* stream);
int gg_cast_f16_to_f32(const void* in, float* out, int64_t n, void* stream);
int gg_token_mean_fwd_f32(const float* x, float* out, int B, int T, int C, void* stream);
int gg_token_mean_bwd_f32(const float* dout, float* dx, int B, int T, int C, void* stream);
int gg_view_mean_fwd_f32(const float* emb, float* out, int64_t ldo, int N, int V, int C, void* stream);
int gg_view_mean_bwd_f32(const float* dmean, int64_t ld, float* demb, int N, int V, int C, void* stream);

/* ---------------------------------------------------------------- SuperGuessr head + loss (models/super_guessr.py:355-383,
 * models/utils.py:20-57, main_coordinator_idun_s3.py:390-391) fused per row over the (N,K) logits. */
typedef struct GgGeoHeadArgs {
    const float* logits; int64_t ldl;     /* f32 [N, ldl] */
    int N, K;
    const float* labels;                  /* f32 (N,2) lon,lat degrees or NULL */
    const float* centroids;               /* f32 (K,2) lon,lat degrees (geocell_centroid_coords) */
    const int64_t* labels_clf;            /* (N,) or NULL */
    int mode;                             /* 0 predictions only, 1 haversine-smoothed soft CE, 2 hard CE */
    float smoothing_km;                   /* LABEL_SMOOTHING_CONSTANT (config.py:52) = 65 */
    float grad_scale;                     /* dlogits = dloss_row/dlogits * grad_scale (pass upstream_grad / N) */
    float* loss_rows;                     /* f32 (N,) or NULL */
    float* loss;                          /* f32 scalar = mean(loss_rows) or NULL */
    void* dlogits; int64_t ldd;           /* bf16 (f32 if dlogits_f32) [N, ldd] (columns K..ldd zeroed) or NULL */
    int64_t* preds; float* llh;           /* argmax geocell (N,), its centroid (N,2) */
    float* topk_vals; int64_t* topk_idx; int num_candidates;   /* (N,num_candidates) softmax probabilities / indices */
    int64_t* nearest;                     /* (N,) argmin_k haversine(labels, centroids) or NULL */
    int dlogits_f32;                      /* reference-precision mode: dlogits is f32 */
} GgGeoHeadArgs;
/* mode 2 with labels_clf[n] outside [0,K): torch's CrossEntropyLoss raises (models/super_guessr.py:383); here loss_rows[n], the mean
 * loss and row n of dlogits come out NaN. */
int gg_geo_head(const GgGeoHeadArgs* args, void* stream);
int gg_haversine_matrix(const float* x, const float* centroids, float* out, int N, int K, void* stream);   /* models/utils.py:39 */

/* ---------------------------------------------------------------- hierarchical combine (models/super_guessr.py:89-99,340-345)
 * SuperGuessr(hierarchical=True): PositionalEncoder (position = batch index, models/layers/positional_encoder.py:44) ->
 * nn.MultiheadAttention(C, NUM_ATTENTION_HEADS = 16)(x, x, x)[0][:, 0].  The projections are gg_gemm_nt_f32; these are the rest.
 * fp32.  mask / pmask: optional dropout scales (keep / (1-p)); NULL in eval mode. */
int gg_pe_add_f32(const float* x /* (N,V,C) */, const float* pe /* (>=N, C) */, const float* mask /* (N,V,C) or NULL */, float* out, int N, int V, int C,
                  void* stream);
/* qkv f32 [N*V, 3C] = in_proj output [q | k | v]; o0 (N, C) = attention output of query token 0; probs (N, H, V) saved softmax row */
int gg_mha_q0_fwd(const float* qkv, const float* pmask /* (N,H,V) or NULL */, float* o0, float* probs, int N, int V, int C, int H, void* stream);
int gg_mha_q0_bwd(const float* qkv, const float* probs, const float* pmask, const float* do0 /* (N, C) */, float* dqkv /* [N*V, 3C], fully written */,
                  int N, int V, int C, int H, void* stream);

/* ---------------------------------------------------------------- ProtoRefiner.forward (models/proto_refiner.py:129-237) */
typedef struct GgProtoRefineArgs {
    const float* embedding; int B, V, D;  /* f32 (B,V,D); V views are averaged (:150-151); V=1 for (B,D) */
    const float* initial_preds;           /* f32 (B,2) lon,lat */
    const int64_t* candidate_cells;       /* (B,num_candidates) */
    const float* candidate_probs;         /* f32 (B,num_candidates) or NULL (=> one-hot on candidate 0, :154-156) */
    int num_candidates, topk;
    const int64_t* cell_ptr; int num_cells;   /* CSR (num_cells+1) */
    const float* proto_emb;               /* f32 (P,D) */
    const float* proto_lnglat;            /* f32 (P,2) */
    float max_refinement, temperature;
    float* out_llh; int64_t* out_cell; int64_t* out_idx;
    /* optional within-cluster refinement (models/proto_refiner.py:239-269): members of prototype (cluster) j are rows member_ptr[j]..member_ptr[j+1]
     * of member_emb (f32 (Nm,D), already averaged over views) / member_lnglat (f32 (Nm,2)).  A cluster with members answers with the coordinates of
     * the member at torch.argmax of the Euclidean distances (the reference's choice, :264-265); an empty cluster, or member_ptr == NULL, with its
     * centroid (:251-252). */
    const int64_t* member_ptr; const float* member_emb; const float* member_lnglat;
} GgProtoRefineArgs;
int gg_proto_refine(const GgProtoRefineArgs* args, void* stream);
/* run_benchmark.py:25-65: dist_km[i] = haversine_np (fp64, R = 6371 km; may be NULL), score[i] = geoguessr_score_from_distance =
 * int(round(clamp(5000*exp(-d/1492.7), 0, 5000))) with Python's round-half-to-even: integer output, bit-exact vs the reference */
int gg_geoguessr_score(const float* pred_llh, const float* true_llh, int N, double* dist_km, int32_t* score, void* stream);
 /* float64 coordinates (the dtype of the reference's arrays) are not narrowed; a is clamped to 1 before asin; a non-finite distance scores -1 */
int gg_geoguessr_score_f64(const double* pred_llh, const double* true_llh, int N, double* dist_km, int32_t* score, void* stream);

/* ---------------------------------------------------------------- optimizer (main_coordinator_idun_s3.py:286-291) */
int gg_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int step, float lr, float beta1,
                  float beta2, float eps, float weight_decay, float grad_scale, void* stream);
int gg_fill_f32(float* p, int64_t n, float value, void* stream);

/* ---------------------------------------------------------------- data-parallel exchange
 * One opaque communicator per process / GPU over RCCL (xGMI inside a node): the only collectives of the path.  Replaces Accelerate ->
 * DistributedDataParallel -> NCCL of training/train_eval_loop.py:184-187,200-202,234 (gradient all-reduce at :234, parameter broadcast
 * at construction, BatchNorm-buffer broadcast before each training forward).  The unique id (128 bytes, host) is created on one rank
 * and handed to the others by the launcher's own channel (the Python host uses torch.distributed's store).  Everything is enqueued
 * on `stream`.  The Python host runs the same three collectives through torch.distributed (backend "nccl" = this same RCCL) by
 * default and through these entry points with GG_NATIVE_COMM=1 (geoguessr_ai_amd/comm.py). */
typedef struct gg_comm gg_comm;
int gg_comm_unique_id(void* out128 /* host, 128 bytes */);
int gg_comm_create(gg_comm** out, const void* unique_id128 /* host */, int rank, int world, int device);
int gg_comm_destroy(gg_comm* comm);
int gg_comm_rank(const gg_comm* comm);
int gg_comm_world(const gg_comm* comm);
int gg_comm_allreduce_sum_f32(gg_comm* comm, float* buf /* in place */, int64_t n, void* stream);
int gg_comm_broadcast(gg_comm* comm, void* buf, int64_t bytes, int root, void* stream);
int gg_comm_barrier(gg_comm* comm, void* stream, int sync /* != 0: also wait on the host */);

/* ---------------------------------------------------------------- optional kernel timing (HIP events on the launch stream)
 * categories: 0 gemm, 1 attention, 2 dwconv, 3 norm/elementwise, 4 head-loss, 5 optimizer, 6 data movement */
int gg_prof_enable(int on);
int gg_prof_reset(void);
int gg_prof_read(int category, double* ms /* host */, int64_t* launches /* host */, double* flops /* host */, double* bytes /* host */);
/* per-launch records since the last reset, in launch order: count, then one record (category, duration, declared flops / bytes) */
int gg_prof_count(void);
int gg_prof_record(int index, int* category /* host */, double* ms /* host */, double* flops /* host */, double* bytes /* host */);

/* ---------------------------------------------------------------- captured HIP graphs for launch-bound calls
 * gg_tinyvit_forward / gg_tinyvit_backward (no stage callback) are fixed launch sequences of their arguments: the second call with
 * the same arguments (pointers included) is captured on a private stream, later ones replay the instantiated graph on the caller's stream.
 * mode -1 (default; env GG_GRAPH unset): only launch-bound sizes (<= 64 images of 224 x 224); 0 (GG_GRAPH=0): never; 1 (GG_GRAPH=1): always.  Off while
 * gg_prof_enable(1).  The reference has no counterpart (eager PyTorch: inference.py:162-170 issues the same few hundred launches per panorama). */
int gg_graph_set_mode(int mode);
/* returns the number of cached keys; counters since load */
int gg_graph_stats(int64_t* captures /* host */, int64_t* replays /* host */, int64_t* eager_calls /* host */);
int gg_graph_clear(void);

/* ---------------------------------------------------------------- TinyViT encoder (timm TinyVit as built by
 * models/tinyvit.py:48-53 with num_classes=0, global_pool="avg"): whole forward / backward in one call.
 * Parameters live in one flat f32 buffer, BN running stats in a second one; tensor i of the table has the timm
 * state-dict name returned by gg_tinyvit_tensor_info. */
typedef struct GgTinyVitCfg {
    int img_size, in_chans;
    int embed_dims[4], depths[4], num_heads[4], window_sizes[4];
    float mlp_ratio, mbconv_expand_ratio;
    float bn_eps, ln_eps, bn_momentum;
    int act_dtype;       /* 0: bf16 activations + bf16 MFMA operands (fp32 accumulate / statistics / master weights);
                            1: reference-precision mode -- f32 activations, f32 MFMA (v_mfma_f32_16x16x4_f32), erf GELU at fp32 accuracy (1.2 ulp): the
                               arithmetic of the reference's own torch fp32 forward / backward;
                            3: "fp32_split" (DESIGN.md 5): mode 1's storage, with the GEMMs whose weight operand has cached bf16 planes run as fp32-accurate
                               SPLIT products on the bf16 MFMA -- the f32 activation operand is split into three bf16 terms while the GEMM stages it
                               (gg_gemm_nt_split3_af32), the weight's terms are cached planes: the four Linears of every transformer block (qkv, proj, fc1,
                               fc2: forward and data gradients), the dense 1 x 1 / im2col convolutions of the ConvNorms (forward, with the BatchNorm
                               partials in the epilogue, and their plain data gradients); the block Linears' WEIGHT gradients run as gg_gemm_tn_split3
                               (both operands split in the loader).  Routing is by size: a Linear needs 128 output tiles, a weight gradient 1024 rows
                               (smaller calls stay on mode 1's f32-MFMA kernels, which fill the chip with 64 x 64 tiles / split-K).  The BatchNorm-fused
                               conv forms of stage 0 and the conv weight gradients as mode 1.  Error against fp64 at or below the f32 MFMA GEMM's */
    int features_only;   /* 1: models/tinyvit.py:38-46,139-143 (timm features_only=True): the output is the global-average-pooled
                               last feature map, head.norm is not applied (its parameters stay in the table, unused) */
} GgTinyVitCfg;
enum { GG_KIND_PARAM = 0, GG_KIND_BUFFER = 1, GG_KIND_COUNTER = 2 };
int gg_tinyvit_num_tensors(const GgTinyVitCfg* cfg);
int gg_tinyvit_tensor_info(const GgTinyVitCfg* cfg, int i, char* name /* host */, int name_cap, int64_t* offset, int64_t* numel,
                           int* ndim, int64_t* shape4 /* host[4] */, int* kind);
int64_t gg_tinyvit_param_floats(const GgTinyVitCfg* cfg);      /* flat f32 parameter buffer length (padded) */
int64_t gg_tinyvit_buffer_floats(const GgTinyVitCfg* cfg);     /* flat f32 running_mean/var buffer length */
int gg_tinyvit_num_counters(const GgTinyVitCfg* cfg);          /* num_batches_tracked entries (int64) */
int gg_tinyvit_num_drop_slots(const GgTinyVitCfg* cfg);        /* DropPath slots: 1 per MBConv, 2 per TinyVitBlock */
/* one step's DropPath rows for gg_tinyvit_forward's drop_scales: out[s][b] = Bernoulli(1 - rates[s]) / (1 - rates[s]) (timm DropPath with
 * scale_by_keep).  Counter-based: (seed, counter) fully determine the rows, the caller advances `counter` once per training forward. */
int gg_drop_path_scales(const float* rates, int slots, int batch, uint64_t seed, uint64_t counter, float* out, void* stream);
int64_t gg_tinyvit_wcache_bytes(const GgTinyVitCfg* cfg);      /* bf16 copies (W and W^T) of the GEMM weights */
int64_t gg_tinyvit_workspace_bytes(const GgTinyVitCfg* cfg, int batch, int training);      /* = ..._masked(cfg, batch, training, NULL) */
/* The training workspace depends on which tensors train (`trainable`: one byte per tensor of gg_tinyvit_tensor_info, NULL = all): the inputs of
 * frozen Linears / depthwise convs (ln1, x1, ln2, GELU(fc1) of a TinyVitBlock; act1 / act2 of MBConv and PatchMerging) are read once, right after
 * they are written, and share a two-slot ring instead of being kept for backward.  Under the reference's freeze_all_but_last_stage policy
 * (models/tinyvit.py:106-111) that is 7 of a frozen block's 19 C floats per token.  gg_tinyvit_forward / _backward lay the workspace out for the
 * mask they are CALLED with: pass the same mask to the size query, the forward and its backward. */
int64_t gg_tinyvit_workspace_bytes_masked(const GgTinyVitCfg* cfg, int batch, int training, const uint8_t* trainable);
int gg_tinyvit_refresh_weights(const GgTinyVitCfg* cfg, const float* params, void* wcache, void* stream);
/* the same for a subset: `only` (host, one byte per tensor of gg_tinyvit_tensor_info) marks the tensors that changed (after an optimizer step: the trainable ones) */
int gg_tinyvit_refresh_weights_masked(const GgTinyVitCfg* cfg, const float* params, void* wcache, const uint8_t* only, void* stream);
/* x: f32 NCHW (batch,in_chans,img,img).  drop_scales: f32 [num_drop_slots][batch] = keep/(1-p) or NULL.
 * out: f32 (batch, embed_dims[3]).  training: batch-stat BN + running-stat update + activations kept for backward. */
/* trainable: host uint8[num_tensors] or NULL -- the freeze policy the backward will run with; activations that only a frozen
 * weight's gradient would read are fused away (e.g. MBConv act2 goes through conv3's BatchNorm prologue), so
 * gg_tinyvit_backward must be given the SAME mask (NULL here = keep everything = any backward mask is fine). */
int gg_tinyvit_forward(const GgTinyVitCfg* cfg, int batch, int training, const float* params, float* buffers, int64_t* counters,
                       const void* wcache, const float* x, const float* drop_scales, void* workspace, float* out,
                       const uint8_t* trainable /* host */, void* stream);
/* d_out: f32 (batch, C).  grads: flat f32 like params, ACCUMULATED into.  trainable: host uint8[num_tensors]
 * (wgrad computed only where 1; dgrad always flows to patch_embed -- SURVEY.md C1). */
/* stage_done (may be NULL): HOST callback, called on the calling thread as soon as every kernel that writes the parameter gradients
 * of a stage has been enqueued on `stream` -- stage ids 3, 2, 1 (TinyVitStage incl. its PatchMerging; 3 also covers head.norm),
 * 0 (the MBConv stage), -1 (patch_embed), in that order.  The data-parallel host uses it to start the RCCL all-reduce of that
 * stage's gradient bucket behind an event on `stream` while the earlier stages are still running (the overlap the reference gets
 * from DistributedDataParallel's bucketed reducer, training/train_eval_loop.py:184-187,234). */
typedef void (*GgStageDoneFn)(int stage, void* user);
int gg_tinyvit_backward(const GgTinyVitCfg* cfg, int batch, const float* params, const void* wcache, const float* drop_scales,
                        void* workspace, const float* d_out, float* grads, const uint8_t* trainable /* host */, void* stream,
                        GgStageDoneFn stage_done, void* stage_user);
/* debug / parity: byte offset of a named saved activation inside the workspace (host) */
int gg_tinyvit_activation_info(const GgTinyVitCfg* cfg, int batch, const char* name, int64_t* offset, int64_t* bytes);
/* the same for the layout of a trainable mask; a tensor that is only a temporary under that mask is refused */
int gg_tinyvit_activation_info_masked(const GgTinyVitCfg* cfg, int batch, const char* name, const uint8_t* trainable, int64_t* offset, int64_t* bytes);

/* ---------------------------------------------------------------- either side of the encoder ("next" rows)
 * f3: the batch loop's input conditioning as one kernel (main_coordinator_idun_s3.py:337-381): bilinear resize
 * (F.interpolate align_corners=False) -> /255 when the source is uint8 -> (x - mean)/std.  src: NCHW f32 or u8 [N,3,Hs,Ws];
 * dst: NCHW f32 [N,3,Hd,Wd]; mean3/std3: HOST float[3] or both NULL (no normalisation). */
int gg_preprocess_bilinear(const void* src, int src_u8, int N, int Hs, int Ws, float* dst, int Hd, int Wd,
                           const float* mean3 /* host */, const float* std3 /* host */, void* stream);
/* The raw-image side of the embedders: what timm's eval transform (pretrain/tinyvit_embedder.py:51-53,67-69), transformers' CLIPProcessor
 * (pretrain/clip_embedder.py:25,51-55) and the torchvision Compose of inference.py:74-85 do to ONE RGB image -- Pillow resize of the whole image to
 * (Hr, Wr) (filter: 2 = PIL BILINEAR, 3 = PIL BICUBIC; support scaled with the reduction as Pillow does), crop window (crop_top, crop_left, Hc, Wc) of
 * the resized image, 1/255 (mul_rescale 0: x / 255 as torchvision's ToTensor, 1: x * (1/255) as transformers' rescale), (x - mean) / std.  The uint8
 * resize result is bit-identical to PIL.Image.resize (8-bit fixed-point resampling with an 8-bit intermediate image; tests/golden/preprocess_pil.npz).
 * src: device uint8 [Hs, Ws, 3] (HWC, RGB); dst_chw: device f32 [3, Hc, Wc]; dst_u8_hwc: optional device uint8 [Hc, Wc, 3] (the crop before 1/255);
 * mean3 / std3: HOST float[3] or both NULL; workspace: device, gg_preprocess_pil_workspace_bytes(...) bytes (-1: bad arguments).  The geometry rules of
 * the three pipelines (shortest edge, crop rounding) are host arithmetic: geoguessr-ai_amd/training/preprocess.py. */
int64_t gg_preprocess_pil_workspace_bytes(int Hs, int Ws, int filter, int Hr, int Wr, int Wc);
int gg_preprocess_pil(const void* src_hwc_u8, int Hs, int Ws, int filter, int Hr, int Wr, int crop_top, int crop_left, int Hc, int Wc, int mul_rescale,
                      const float* mean3 /* host */, const float* std3 /* host */, float* dst_chw, void* dst_u8_hwc, void* workspace, void* stream);
/* f2: prototype building (models/proto_refiner.py:461-517): per-segment mean of embedding rows, CSR segments ptr[K+1] over the
 * member row list, summed in list order in fp32 (the reference's running sum), zeros for empty segments. */
int gg_segment_mean(const float* emb, int64_t ld, const int64_t* ptr, const int64_t* member, int num_segments, int D, float* out,
                    void* stream);

/* ---------------------------------------------------------------- CLIP vision tower: inference and fine-tuning
 * transformers CLIPVisionModel as the reference uses it: the embedder's mean over all tokens of last_hidden_state (no post_layernorm;
 * pretrain/clip_embedder.py:51-66) and the trainable base model of SuperGuessr (models/super_guessr.py:134-150,323-325: the last
 * encoder layer is fine-tuned when the pretrained head exists, every layer otherwise; main_coordinator_idun_s3.py:183-203).
 * act_dtype 1 = fp32 (the reference's precision: f32 activations, f32 MFMA), 0 = bf16 and 2 = fp16 activations / MFMA operands with f32
 * accumulation (2: inference only -- BASELINE config c4 names fp16).
 * Parameters: one flat f32 buffer, HF state-dict names without the "vision_model." prefix (gg_clip_tensor_info).  `trainable` (host,
 * one byte per tensor, NULL = all) selects the tensors whose gradients gg_clip_backward accumulates; a training forward keeps the
 * activations of every layer from the first trainable one up (gg_clip_first_trained_layer), frozen layers below run in place. */
typedef struct GgClipCfg {
    int hidden_size, intermediate_size, num_layers, num_heads, image_size, patch_size;
    float ln_eps;
    int act_dtype;                         /* 0 bf16, 1 fp32, 2 fp16 (inference only) */
} GgClipCfg;
int gg_clip_num_tensors(const GgClipCfg* cfg);
int gg_clip_tensor_info(const GgClipCfg* cfg, int i, char* name, int name_cap, int64_t* offset, int64_t* numel, int* ndim, int64_t* shape4);
int64_t gg_clip_param_floats(const GgClipCfg* cfg);
int64_t gg_clip_wcache_bytes(const GgClipCfg* cfg);
int64_t gg_clip_workspace_bytes(const GgClipCfg* cfg, int batch, int training, const uint8_t* trainable);
int gg_clip_first_trained_layer(const GgClipCfg* cfg, const uint8_t* trainable);   /* 0 when an embedding-side tensor is trainable; num_layers: none */
int gg_clip_refresh_weights(const GgClipCfg* cfg, const float* params, void* wcache, void* stream);
int gg_clip_forward(const GgClipCfg* cfg, int batch, int training, const float* params, const void* wcache, const float* x, void* workspace,
                    float* out /* f32 (batch, hidden) */, float* last_hidden /* f32 (batch,T,hidden) or NULL */, const uint8_t* trainable, void* stream);
/* backward of the training forward that last wrote `workspace` (same cfg / batch / trainable).  d_out: gradient of `out` or NULL;
 * d_last_hidden: gradient of last_hidden or NULL (both: summed).  Gradients are ACCUMULATED into `grads` (flat, parameter offsets). */
int gg_clip_backward(const GgClipCfg* cfg, int batch, const float* params, const void* wcache, void* workspace, const float* d_out,
                     const float* d_last_hidden, float* grads, const uint8_t* trainable, void* stream);

#ifdef __cplusplus
}
#endif
#endif
