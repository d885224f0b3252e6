// CLIP vision tower (transformers CLIPVisionModel as the reference uses it): forward for the embedder
// (pretrain/clip_embedder.py:51-66: base_model(pixel_values).last_hidden_state.mean(dim=1), no post_layernorm) and forward + backward
// for SuperGuessr with a CLIP base (models/super_guessr.py:134-150,323-325: the last encoder layer is fine-tuned when the pretrained
// head exists, every layer otherwise; main_coordinator_idun_s3.py:183-203 builds that model for training).
//
// A static schedule of libgg launches on one stream, in one of three arithmetic modes (GgClipCfg.act_dtype):
//   1  fp32 -- the reference's precision: f32 activations, v_mfma_f32_16x16x4_f32 GEMMs (gg_gemm_nt_f32 / gg_gemm_tn_f32), f32 LayerNorm, f32
//      online-softmax attention (head dim 64);
//   0  bf16 -- bf16 activations / MFMA operands, f32 accumulation, f32 statistics;
//   2  fp16 -- the same with fp16 storage / v_mfma_f32_16x16x32_f16 (BASELINE config c4 names fp16), inference only.
// Patch embedding is a pure GEMM (stride == kernel); q/k/v projections are one [3D, D] GEMM; QuickGELU rides on fc1's epilogue (with the
// pre-activation copy the backward pass needs), residual adds on out_proj's and fc2's.  Training keeps, for every layer from the first
// trainable one up, the tensors its backward pass reads (layer input, both LayerNorm outputs + statistics, qkv, attention output + row
// log-sum-exp, fc1 pre-activation and activation): sized for HBM, nothing is recomputed.  Frozen layers below run in place on scratch.
#include <algorithm>
#include <string>
#include <vector>
#include <string.h>
#include <stdio.h>
#include "common.h"
#include "../../include/gg.h"

namespace {
struct TInfo { std::string name; int64_t offset, numel; int ndim; int64_t shape[4]; };
struct LayerP {
    int q_w, q_b, k_w, k_b, v_w, v_b, o_w, o_b, ln1_g, ln1_b, fc1_w, fc1_b, fc2_w, fc2_b, ln2_g, ln2_b;      // tensor ids
    int64_t wqkv, bqkv, wqkvT, wo, woT, w1, w1T, w2, w2T;                                                     // weight-cache offsets
};
struct CModel {
    GgClipCfg cfg;
    std::vector<TInfo> t;
    int64_t floats = 0, wc_bytes = 0;
    int cls, patch_w, pos, pre_g, pre_b, post_g, post_b;
    int64_t wpatch;
    std::vector<LayerP> layers;
    int T, G, Kpatch, Kraw;
    bool f32, f16; int es;             // activation / cached-weight element size: 4 (fp32 mode) or 2 (bf16 / fp16 modes)
};
static int addt(CModel& m, const std::string& n, std::initializer_list<int64_t> shape) {
    TInfo t; t.name = n; t.ndim = (int)shape.size(); t.numel = 1;
    for (int j = 0; j < 4; ++j) t.shape[j] = 1;
    int i = 0;
    for (auto s : shape) { t.shape[i++] = s; t.numel *= s; }
    t.offset = m.floats; m.floats += gg_align(t.numel, 8);
    m.t.push_back(t);
    return (int)m.t.size() - 1;
}
static int64_t wca(CModel& m, int64_t bytes) { int64_t o = m.wc_bytes; m.wc_bytes += gg_align(bytes, 256); return o; }

static int build(const GgClipCfg* c, CModel& m) {
    GG_CHECK(c, "clip: null config");
    m.cfg = *c;
    GG_CHECK(c->act_dtype >= 0 && c->act_dtype <= 2, "clip: act_dtype must be 0 (bf16), 1 (fp32) or 2 (fp16), got %d", c->act_dtype);
    m.f32 = c->act_dtype == 1; m.f16 = c->act_dtype == 2; m.es = m.f32 ? 4 : 2;
    const int D = c->hidden_size, I = c->intermediate_size, P = c->patch_size;
    GG_CHECK(D > 0 && D % 64 == 0 && D <= 1024 && c->num_heads > 0 && D / c->num_heads == 64, "clip: head_dim must be 64 and hidden <= 1024 (hidden %d, heads %d)", D, c->num_heads);
    GG_CHECK(P > 0 && c->image_size % P == 0 && I % 8 == 0 && c->num_layers > 0, "clip: bad patch/image/intermediate size or layer count");
    // patch-embedding contraction 3*P*P is padded to a multiple of 8 (ViT-L/14: 588 -> 592 zero columns); sequences beyond 256 tokens
    // (ViT-L/14-336: 577, the reference's CLIP_MODEL, config.py:6) run on the online-softmax attention kernels
    m.G = c->image_size / P; m.T = m.G * m.G + 1; m.Kraw = 3 * P * P; m.Kpatch = (int)gg_align(m.Kraw, 8);
    m.cls = addt(m, "embeddings.class_embedding", {D});
    m.patch_w = addt(m, "embeddings.patch_embedding.weight", {D, 3, P, P});
    m.pos = addt(m, "embeddings.position_embedding.weight", {m.T, D});
    m.pre_g = addt(m, "pre_layrnorm.weight", {D}); m.pre_b = addt(m, "pre_layrnorm.bias", {D});
    m.wpatch = wca(m, (int64_t)D * m.Kpatch * m.es);
    m.layers.resize(c->num_layers);
    for (int i = 0; i < c->num_layers; ++i) {
        LayerP& l = m.layers[i];
        const std::string p = "encoder.layers." + std::to_string(i);
        l.k_w = addt(m, p + ".self_attn.k_proj.weight", {D, D}); l.k_b = addt(m, p + ".self_attn.k_proj.bias", {D});
        l.v_w = addt(m, p + ".self_attn.v_proj.weight", {D, D}); l.v_b = addt(m, p + ".self_attn.v_proj.bias", {D});
        l.q_w = addt(m, p + ".self_attn.q_proj.weight", {D, D}); l.q_b = addt(m, p + ".self_attn.q_proj.bias", {D});
        l.o_w = addt(m, p + ".self_attn.out_proj.weight", {D, D}); l.o_b = addt(m, p + ".self_attn.out_proj.bias", {D});
        l.ln1_g = addt(m, p + ".layer_norm1.weight", {D}); l.ln1_b = addt(m, p + ".layer_norm1.bias", {D});
        l.fc1_w = addt(m, p + ".mlp.fc1.weight", {I, D}); l.fc1_b = addt(m, p + ".mlp.fc1.bias", {I});
        l.fc2_w = addt(m, p + ".mlp.fc2.weight", {D, I}); l.fc2_b = addt(m, p + ".mlp.fc2.bias", {D});
        l.ln2_g = addt(m, p + ".layer_norm2.weight", {D}); l.ln2_b = addt(m, p + ".layer_norm2.bias", {D});
        // W[N][K] for the forward GEMM and W^T[K][N] so that the data gradient dX = dY . W is the same NT GEMM
        l.wqkv = wca(m, (int64_t)3 * D * D * m.es); l.wqkvT = wca(m, (int64_t)3 * D * D * m.es);
        l.bqkv = wca(m, (int64_t)3 * D * 4);
        l.wo = wca(m, (int64_t)D * D * m.es); l.woT = wca(m, (int64_t)D * D * m.es);
        l.w1 = wca(m, (int64_t)I * D * m.es); l.w1T = wca(m, (int64_t)I * D * m.es);
        l.w2 = wca(m, (int64_t)D * I * m.es); l.w2T = wca(m, (int64_t)D * I * m.es);
    }
    m.post_g = addt(m, "post_layernorm.weight", {D}); m.post_b = addt(m, "post_layernorm.bias", {D});
    return 0;
}

// ---- element-type helpers (the storage types of the runtime) -------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return (bf16)v; }
template <> __device__ __forceinline__ f16 from_f<f16>(float v) { return (f16)v; }
// f32 [R][C] -> fp16 copy [R][ldo] and / or transpose [C][ldt] (weight cache of the fp16 mode; the bf16 twin is gg_cast_transpose_f32)
__global__ __launch_bounds__(256) void cast_transpose_f16_kernel(const float* __restrict__ in, int R, int C, f16* __restrict__ out, int64_t ldo,
                                                                f16* __restrict__ outT, int64_t ldt) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) {
            v = in[(int64_t)r * C + c];
            if (out) out[(int64_t)r * ldo + c] = (f16)v;
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    if (outT) {
        for (int i = ty; i < 64; i += 4) {
            const int c = c0 + i, r = r0 + tx;
            if (c < C && r < R) outT[(int64_t)c * ldt + r] = (f16)tile[tx][i];
        }
    }
}
template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (TO)(float)in[i];
}

// x f32 NCHW (B,3,S,S) -> col [B*G*G, K], k = (c, py, px)  (== Conv2d(kernel=stride=P) weight flatten); K = 3*P*P padded to 8 (zero columns)
// VEC (P % 8 == 0, S % 4 == 0, 16-byte aligned x): a thread's 8 consecutive k are 8 consecutive pixels of one image row -- two 16-byte loads, one 16-byte
// store, the (c, py, px) split done once per thread instead of once per element (the scalar form ran at 2.5 TB/s on the batch-1024 ViT-B/32 input)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, T* __restrict__ col, int B, int S, int P, int G, int K) {
    const int Kraw = 3 * P * P;
    const int64_t total = (int64_t)B * G * G * (K / 8);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int kc = (int)(i % (K / 8)) * 8;
        const int64_t p = i / (K / 8);
        const int gx = (int)(p % G), gy = (int)((p / G) % G), b = (int)(p / ((int64_t)G * G));
        T o[8];
        if (VEC) {
            if (kc < Kraw) {
                const int c = kc / (P * P), r = kc - c * P * P, py = r / P, px = r - py * P;
                const float* src = x + (((int64_t)b * 3 + c) * S + gy * P + py) * S + gx * P + px;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { o[j] = from_f<T>(v0[j]); o[4 + j] = from_f<T>(v1[j]); }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = from_f<T>(0.f);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = kc + j;
                const int c = k / (P * P), py = (k / P) % P, px = k % P;
                o[j] = from_f<T>(k < Kraw ? x[(((int64_t)b * 3 + c) * S + gy * P + py) * S + gx * P + px] : 0.f);
            }
        }
        T* dst = col + p * K + kc;
        if (VEC && sizeof(T) == 2) {
            typedef T t8 __attribute__((ext_vector_type(8)));
            t8 w;
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = o[j];
            *reinterpret_cast<t8*>(dst) = w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[j] = o[j];
        }
    }
}
// tokens[b,0,:] = cls + pos[0];  tokens[b,1+i,:] = patches[b,i,:] + pos[1+i]; a thread owns 4 consecutive channels (D % 4 == 0)
template <typename T>
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const T* __restrict__ patches, const float* __restrict__ cls, const float* __restrict__ pos,
                                                              T* __restrict__ tokens, int B, int Tn, int D) {
    const int D4 = D / 4;
    const int64_t total = (int64_t)B * Tn * D4;
    typedef T t4 __attribute__((ext_vector_type(4)));
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int dd = (int)(i % D4) * 4;
        const int64_t bt = i / D4;
        const int t = (int)(bt % Tn);
        const int64_t b = bt / Tn;
        const f32x4 ps = *reinterpret_cast<const f32x4*>(pos + (int64_t)t * D + dd);
        f32x4 v;
        if (t == 0) v = *reinterpret_cast<const f32x4*>(cls + dd);
        else {
            const t4 q = *reinterpret_cast<const t4*>(patches + (b * (Tn - 1) + (t - 1)) * D + dd);
            v = (f32x4){(float)q[0], (float)q[1], (float)q[2], (float)q[3]};
        }
        v += ps;
        t4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = from_f<T>(v[j]);
        *reinterpret_cast<t4*>(tokens + bt * D + dd) = w;
    }
}
// f32 [R][K] -> T [R][Kp] with zero padding columns
template <typename T>
__global__ void cast_pad_rows_kernel(const float* __restrict__ src, T* __restrict__ dst, int R, int K, int Kp) {
    const int64_t n = (int64_t)R * Kp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kp);
        dst[i] = from_f<T>(k < K ? src[(i / Kp) * K + k] : 0.f);
    }
}
// gradient of the token mean (+ an optional gradient on last_hidden_state itself): dx[b,t,:] = dpool[b,:] / T + dlast[b,t,:]
template <typename T>
__global__ void pool_bwd_kernel(const float* __restrict__ dpool, const float* __restrict__ dlast, T* __restrict__ dx, int B, int Tn, int D) {
    const int64_t total = (int64_t)B * Tn * D;
    const float inv = 1.0f / (float)Tn;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int dd = (int)(i % D);
        const int64_t b = i / ((int64_t)D * Tn);
        float v = dpool ? dpool[b * D + dd] * inv : 0.f;
        if (dlast) v += dlast[i];
        dx[i] = from_f<T>(v);
    }
}
// backward of assemble_tokens: dpos[t,:] += sum_b dtok[b,t,:], dcls += sum_b dtok[b,0,:], dpatch[b,i,:] = dtok[b,1+i,:] (contiguous rows for the
// patch-weight gradient GEMM).  One thread per (t, d) column walks the batch: the sums are deterministic.
template <typename T>
__global__ void embed_bwd_kernel(const T* __restrict__ dtok, float* __restrict__ dpos, float* __restrict__ dcls, T* __restrict__ dpatch, int B,
                                 int Tn, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Tn * D) return;
    const int dd = i % D, t = i / D;
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
        const T v = dtok[((int64_t)b * Tn + t) * D + dd];
        s += (float)v;
        if (t > 0 && dpatch) dpatch[((int64_t)b * (Tn - 1) + (t - 1)) * D + dd] = v;
    }
    if (dpos) dpos[i] += s;
    if (t == 0 && dcls) dcls[dd] += s;
}
// dW_patch (D, 3, P, P) += reduced partials [D][Kpatch] (drops the zero-padding columns)
__global__ void patch_wgrad_scatter_kernel(const float* __restrict__ src, int D, int Kp, int Kraw, float* __restrict__ grad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D * Kraw) return;
    grad[i] += src[(int64_t)(i / Kraw) * Kp + i % Kraw];
}
__global__ void copy_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static inline unsigned grid1d(int64_t n) { return (unsigned)std::min<int64_t>(gg_cdiv(n, 256), 32768); }

// ---- which layers keep their activations ----------------------------------------------------------------------------------------------------------
struct Train {
    bool embed;      // any embedding-side tensor trainable: the backward pass runs through every layer and the patch / position embeddings
    int l0;          // first layer whose activations are kept (num_layers: none)
};
static Train train_of(const CModel& m, int training, const uint8_t* mask) {
    Train tr{false, m.cfg.num_layers};
    if (!training) return tr;
    auto on = [&](int t) { return mask == nullptr || mask[t] != 0; };
    tr.embed = on(m.cls) || on(m.patch_w) || on(m.pos) || on(m.pre_g) || on(m.pre_b);
    if (tr.embed) { tr.l0 = 0; return tr; }
    for (int i = 0; i < m.cfg.num_layers; ++i) {
        const LayerP& l = m.layers[i];
        const int ids[] = {l.q_w, l.q_b, l.k_w, l.k_b, l.v_w, l.v_b, l.o_w, l.o_b, l.ln1_g, l.ln1_b, l.fc1_w, l.fc1_b, l.fc2_w, l.fc2_b, l.ln2_g, l.ln2_b};
        for (int t : ids) if (on(t)) { tr.l0 = i; return tr; }
    }
    return tr;
}

// ---- workspace plan ---------------------------------------------------------------------------------------------------------------------------
struct LayerA { int64_t xin, a1, qkv, o, lse, xmid, a2, pre, h, mean1, rstd1, mean2, rstd2; };
struct CPlan {
    int64_t col, patches, tok, mean0, rstd0;                  // embedding side (tok = tokens before pre_layrnorm)
    int64_t s_x, s_a, s_qkv, s_o, s_h;                        // scratch of the layers that keep nothing (in-place residual stream)
    std::vector<LayerA> la;
    int64_t xfinal;
    int64_t g_x0, g_x1, g_a, g_qkv, g_o, g_h, splitk, colsum, lnscr, lndump, attn_ds = -1;      // lndump: where a frozen LayerNorm tensor's half of a (gamma, beta) gradient pair goes      // backward scratch (attn_ds: GgAttnArgs.ds_scratch)
    int64_t total;
};
static void plan(const CModel& m, int B, const Train& tr, bool training, CPlan& L) {
    const int D = m.cfg.hidden_size, I = m.cfg.intermediate_size, nl = m.cfg.num_layers;
    const int64_t Mp = (int64_t)B * m.G * m.G, M = (int64_t)B * m.T, es = m.es;
    int64_t off = 0;
    auto al = [&](int64_t bytes) { int64_t o = off; off += gg_align(std::max<int64_t>(bytes, 1), 256); return o; };
    L.col = al(Mp * m.Kpatch * es); L.patches = al(Mp * D * es); L.tok = al(M * D * es); L.mean0 = al(M * 4); L.rstd0 = al(M * 4);
    L.s_x = al(M * D * es); L.s_a = al(M * D * es); L.s_qkv = al(M * 3 * D * es); L.s_o = al(M * D * es); L.s_h = al(M * I * es);
    L.la.assign(nl, LayerA{});
    L.xfinal = L.s_x;
    const bool bwd = training && tr.l0 < nl;
    if (bwd) {
        for (int i = tr.l0; i < nl; ++i) {
            LayerA& a = L.la[i];
            a.xin = al(M * D * es); a.a1 = al(M * D * es); a.qkv = al(M * 3 * D * es); a.o = al(M * D * es);
            a.lse = al(M * m.cfg.num_heads * 4); a.xmid = al(M * D * es); a.a2 = al(M * D * es); a.pre = al(M * I * es); a.h = al(M * I * es);
            a.mean1 = al(M * 4); a.rstd1 = al(M * 4); a.mean2 = al(M * 4); a.rstd2 = al(M * 4);
        }
        L.xfinal = al(M * D * es);
        L.g_x0 = al(M * D * es); L.g_x1 = al(M * D * es); L.g_a = al(M * D * es); L.g_qkv = al(M * 3 * D * es); L.g_o = al(M * D * es);
        L.g_h = al(M * I * es);
        auto splits = [&](int64_t Mm, int N, int K) { return (int64_t)(m.f32 ? gg_gemm_tn_f32_splits((int)Mm, N, K) : gg_gemm_tn_splits((int)Mm, N, K)) * N * K; };
        int64_t sk = std::max(std::max(splits(M, D, I), splits(M, I, D)), splits(M, D, D));
        if (tr.embed) sk = std::max(sk, splits(Mp, D, m.Kpatch));
        L.splitk = al(sk * 4);
        L.colsum = al(std::max(gg_colsum_scratch_floats((int)M, I), gg_colsum_scratch_floats((int)M, D)) * 4);
        L.lnscr = al(gg_layernorm_bwd_scratch_floats(M, D) * 4);
        L.lndump = al((int64_t)D * 4);
        // dS hand-off between the two passes of the attention backward: only towers beyond 256 tokens (ViT-L/14-336: 577 tokens, 22 MB per image) use it -- the
        // 50-token towers run the single-pass kernel --, and only while it stays at most 4 GB and 1/8 of the workspace planned so far
        const int64_t dsb = gg_attention_flash_ds_scratch_floats(B, m.cfg.num_heads, m.T) * 4;
        static const bool ds_off = gg_dev_env("GG_ATTN_NO_DS_SCRATCH") != nullptr;
        if (!ds_off && !gg_attention_flash_single_pass(m.T, D / m.cfg.num_heads, 0, 0) && dsb <= ((int64_t)4 << 30) && dsb <= off / 8) L.attn_ds = al(dsb);
    }
    L.total = off;
}

// ---- one executing call ---------------------------------------------------------------------------------------------------------------------------
struct Exec {
    const CModel* m; const CPlan* L; int B; hipStream_t st;
    const float* params; const char* wc; char* ws; float* grads; const uint8_t* mask;
    const float* P(int t) const { return params + m->t[t].offset; }
    float* Gd(int t) const { return grads + m->t[t].offset; }
    bool tr(int t) const { return mask == nullptr || mask[t] != 0; }
    void* A(int64_t o) const { return ws + o; }
    float* F(int64_t o) const { return reinterpret_cast<float*>(ws + o); }
    const void* W(int64_t o) const { return wc + o; }
    int gemm(const void* Am, int64_t lda, const void* Bm, int64_t ldb, void* C, int64_t ldc, int64_t Mm, int N, int K, const float* bias, int act = 0,
             void* preact = nullptr, const void* residual = nullptr, const void* dact_preact = nullptr, int dact = 0) const {
        GgGemmArgs g;
        memset(&g, 0, sizeof(g));
        g.A = Am; g.lda = lda; g.B = Bm; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = (int)Mm; g.N = N; g.K = K;
        g.bias = bias; g.act = act; g.preact = preact; g.residual = residual; g.ldr = ldc; g.dact_preact = dact_preact; g.dact = dact;
        return m->f32 ? gg_gemm_nt_f32(&g, st) : (m->f16 ? gg_gemm_nt_f16(&g, st) : gg_gemm_nt(&g, st));
    }
    int ln_fwd(const void* x, int tg, int tb, int64_t M, void* out, float* mean, float* rstd) const {
        if (m->f16) return gg_layernorm_fwd_f16(x, P(tg), P(tb), M, m->cfg.hidden_size, m->cfg.ln_eps, out, st);
        return gg_layernorm_fwd(x, m->f32, P(tg), P(tb), M, m->cfg.hidden_size, m->cfg.ln_eps, out, m->f32, mean, rstd, st);
    }
    // dx = LayerNorm backward of dout (+ dres), dgamma / dbeta accumulated when trainable
    int ln_bwd(const void* dout, const void* x, const float* mean, const float* rstd, int tg, int tb, int64_t M, const void* dres, void* dx) const {
        const bool t = tr(tg) || tr(tb);
        // the kernel forms dgamma and dbeta together: when only one of the two is trainable, the other's sums are added to a dump row nobody reads, never into the frozen
        // tensor's region of the flat gradient buffer (which the caller may be exchanging / reading as zeros)
        return gg_layernorm_bwd(dout, x, m->f32, mean, rstd, P(tg), M, m->cfg.hidden_size, dres, dx, F(L->lnscr), t ? (tr(tg) ? Gd(tg) : F(L->lndump)) : nullptr,
                                t ? (tr(tb) ? Gd(tb) : F(L->lndump)) : nullptr, 1, st);
    }
    // dW[N,K] += dY[M,N]^T . X[M,K]
    int wgrad(int tw, const void* dY, int64_t ldy, const void* X, int64_t ldx, int64_t M, int N, int K) const {
        if (!tr(tw)) return 0;
        const int sp = m->f32 ? gg_gemm_tn_f32_splits((int)M, N, K) : gg_gemm_tn_splits((int)M, N, K);
        if (m->f32) GG_TRY(gg_gemm_tn_f32(dY, ldy, X, ldx, (int)M, N, K, nullptr, 0, F(L->splitk), sp, st));
        else GG_TRY(gg_gemm_tn(dY, ldy, X, ldx, (int)M, N, K, nullptr, 0, F(L->splitk), sp, st));
        return gg_splitk_reduce(F(L->splitk), Gd(tw), (int64_t)N * K, sp, 1, 1.0f, st);
    }
    int bgrad(int tb, const void* dY, int64_t ld, int64_t M, int N) const {
        if (!tr(tb)) return 0;
        if (m->f32) return gg_colsum_f32((const float*)dY, ld, (int)M, N, nullptr, 0, F(L->colsum), Gd(tb), 1, st);
        return gg_colsum_bf16(dY, ld, (int)M, N, nullptr, 0, F(L->colsum), Gd(tb), 1, st);
    }
    void attn_args(GgAttnArgs& at, const void* qkv, void* out, float* lse) const {
        const int D = m->cfg.hidden_size;
        memset(&at, 0, sizeof(at));
        at.qkv = qkv; at.ld = 3 * D; at.q_off = 0; at.k_off = D; at.v_off = 2 * D; at.head_stride = 64; at.head_dim = 64;
        at.num_heads = m->cfg.num_heads; at.num_windows = B; at.tokens_per_window = m->T; at.window_size = 0;
        at.scale = 0.125f; at.out = out; at.ldo = D; at.lse = lse;
    }
};
template <typename T> static int embed_fwd(const Exec& e, const float* x, void* tok_out) {
    const CModel& m = *e.m; const CPlan& L = *e.L;
    const int D = m.cfg.hidden_size, B = e.B;
    const int64_t Mp = (int64_t)B * m.G * m.G, M = (int64_t)B * m.T;
    const bool pvec = (m.cfg.patch_size & 7) == 0 && (m.cfg.image_size & 3) == 0 && ((uintptr_t)x & 15) == 0;
    if (pvec) hipLaunchKernelGGL((patchify_kernel<T, true>), dim3(grid1d(Mp * (m.Kpatch / 8))), dim3(256), 0, e.st, x, (T*)e.A(L.col), B, m.cfg.image_size, m.cfg.patch_size, m.G, m.Kpatch);
    else hipLaunchKernelGGL((patchify_kernel<T, false>), dim3(grid1d(Mp * (m.Kpatch / 8))), dim3(256), 0, e.st, x, (T*)e.A(L.col), B, m.cfg.image_size, m.cfg.patch_size, m.G, m.Kpatch);
    GG_LAUNCH_CHECK();
    GG_TRY(e.gemm(e.A(L.col), m.Kpatch, e.W(m.wpatch), m.Kpatch, e.A(L.patches), D, Mp, D, m.Kpatch, nullptr));
    hipLaunchKernelGGL(assemble_tokens_kernel<T>, dim3(grid1d(M * D / 4)), dim3(256), 0, e.st, (const T*)e.A(L.patches), e.P(m.cls), e.P(m.pos), (T*)tok_out,
                       B, m.T, D);
    GG_LAUNCH_CHECK();
    return 0;
}
}  // namespace

extern "C" int gg_cast_f32_to_f16(const float* in, void* out, int64_t n, void* stream) {
    GG_CHECK(in && out && n > 0, "gg_cast_f32_to_f16: bad args");
    hipLaunchKernelGGL((cast_kernel<float, f16>), dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, in, (f16*)out, n);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_cast_f16_to_f32(const void* in, float* out, int64_t n, void* stream) {
    GG_CHECK(in && out && n > 0, "gg_cast_f16_to_f32: bad args");
    hipLaunchKernelGGL((cast_kernel<f16, float>), dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, (const f16*)in, out, n);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_clip_num_tensors(const GgClipCfg* cfg) { CModel m; return build(cfg, m) ? -1 : (int)m.t.size(); }
extern "C" int gg_clip_tensor_info(const GgClipCfg* cfg, int i, char* name, int cap, int64_t* offset, int64_t* numel, int* ndim, int64_t* shape4) {
    CModel m;
    GG_TRY(build(cfg, m));
    GG_CHECK(i >= 0 && i < (int)m.t.size(), "gg_clip_tensor_info: index out of range");
    if (name && cap > 0) snprintf(name, cap, "%s", m.t[i].name.c_str());
    if (offset) *offset = m.t[i].offset;
    if (numel) *numel = m.t[i].numel;
    if (ndim) *ndim = m.t[i].ndim;
    if (shape4) for (int j = 0; j < 4; ++j) shape4[j] = m.t[i].shape[j];
    return 0;
}
extern "C" int64_t gg_clip_param_floats(const GgClipCfg* cfg) { CModel m; return build(cfg, m) ? -1 : m.floats; }
extern "C" int64_t gg_clip_wcache_bytes(const GgClipCfg* cfg) { CModel m; return build(cfg, m) ? -1 : m.wc_bytes; }
extern "C" int64_t gg_clip_workspace_bytes(const GgClipCfg* cfg, int batch, int training, const uint8_t* trainable) {
    CModel m;
    if (build(cfg, m)) return -1;
    if (batch <= 0) { gg_set_error("gg_clip_workspace_bytes: batch must be > 0"); return -1; }
    CPlan L; plan(m, batch, train_of(m, training, trainable), training != 0, L);
    return L.total;
}
extern "C" int gg_clip_first_trained_layer(const GgClipCfg* cfg, const uint8_t* trainable) {
    CModel m;
    if (build(cfg, m)) return -1;
    return train_of(m, 1, trainable).l0;
}
extern "C" int gg_clip_refresh_weights(const GgClipCfg* cfg, const float* params, void* wcache, void* stream) {
    CModel m;
    GG_TRY(build(cfg, m));
    GG_CHECK(params && wcache, "gg_clip_refresh_weights: null pointer");
    char* wc = (char*)wcache;
    hipStream_t st = (hipStream_t)stream;
    const int D = m.cfg.hidden_size, I = m.cfg.intermediate_size;
    auto P = [&](int t) { return params + m.t[t].offset; };
    // W f32 [R][C] -> cache copy [R][C] at `n` (row offset r0 of a taller [.., C] image) and transpose [C][ldt] at `t` (column offset c0)
    auto put = [&](const float* W, int R, int C, int64_t n, int64_t r0, int64_t t, int64_t ldt, int64_t c0) -> int {
        if (m.f32) {
            GG_HIP(hipMemcpyAsync(wc + n + r0 * C * 4, W, (size_t)R * C * 4, hipMemcpyDeviceToDevice, st));
            return gg_transpose_f32(W, R, C, (float*)(wc + t) + c0, ldt, st);
        }
        if (m.f16) {
            hipLaunchKernelGGL(cast_transpose_f16_kernel, dim3((unsigned)gg_cdiv(C, 64), (unsigned)gg_cdiv(R, 64)), dim3(256), 0, st, W, R, C,
                               (f16*)(wc + n) + r0 * C, (int64_t)C, (f16*)(wc + t) + c0, ldt);
            GG_LAUNCH_CHECK();
            return 0;
        }
        return gg_cast_transpose_f32(W, R, C, (bf16*)(wc + n) + r0 * C, C, (bf16*)(wc + t) + c0, ldt, st);
    };
    if (m.f16) hipLaunchKernelGGL(cast_pad_rows_kernel<f16>, dim3(grid1d((int64_t)D * m.Kpatch)), dim3(256), 0, st, P(m.patch_w), (f16*)(wc + m.wpatch), D, m.Kraw, m.Kpatch);
    else if (m.f32) hipLaunchKernelGGL(cast_pad_rows_kernel<float>, dim3(grid1d((int64_t)D * m.Kpatch)), dim3(256), 0, st, P(m.patch_w), (float*)(wc + m.wpatch), D, m.Kraw, m.Kpatch);
    else hipLaunchKernelGGL(cast_pad_rows_kernel<bf16>, dim3(grid1d((int64_t)D * m.Kpatch)), dim3(256), 0, st, P(m.patch_w), (bf16*)(wc + m.wpatch), D, m.Kraw, m.Kpatch);
    GG_LAUNCH_CHECK();
    for (auto& l : m.layers) {
        GG_TRY(put(P(l.q_w), D, D, l.wqkv, 0, l.wqkvT, 3 * D, 0));
        GG_TRY(put(P(l.k_w), D, D, l.wqkv, D, l.wqkvT, 3 * D, D));
        GG_TRY(put(P(l.v_w), D, D, l.wqkv, 2 * D, l.wqkvT, 3 * D, 2 * D));
        float* bq = (float*)(wc + l.bqkv);
        hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)gg_cdiv(D, 256)), dim3(256), 0, st, P(l.q_b), bq, D);
        hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)gg_cdiv(D, 256)), dim3(256), 0, st, P(l.k_b), bq + D, D);
        hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)gg_cdiv(D, 256)), dim3(256), 0, st, P(l.v_b), bq + 2 * D, D);
        GG_TRY(put(P(l.o_w), D, D, l.wo, 0, l.woT, D, 0));
        GG_TRY(put(P(l.fc1_w), I, D, l.w1, 0, l.w1T, I, 0));
        GG_TRY(put(P(l.fc2_w), D, I, l.w2, 0, l.w2T, D, 0));
    }
    GG_LAUNCH_CHECK();
    return 0;
}

extern "C" int gg_clip_forward(const GgClipCfg* cfg, int batch, int training, const float* params, const void* wcache, const float* x, void* workspace,
                               float* out, float* last_hidden, const uint8_t* trainable, void* stream) {
    CModel m;
    GG_TRY(build(cfg, m));
    GG_CHECK(batch > 0 && params && wcache && x && workspace && out, "gg_clip_forward: null pointer / bad batch");
    GG_CHECK(((uintptr_t)workspace & 255) == 0 && ((uintptr_t)wcache & 255) == 0, "gg_clip_forward: workspace / wcache must be 256-byte aligned");
    const Train tr = train_of(m, training, trainable);
    GG_CHECK(!(m.f16 && training && tr.l0 < m.cfg.num_layers), "gg_clip_forward: the fp16 mode is inference-only (train in fp32 or bf16)");
    CPlan L; plan(m, batch, tr, training != 0, L);
    Exec e{&m, &L, batch, (hipStream_t)stream, params, (const char*)wcache, (char*)workspace, nullptr, trainable};
    const int D = m.cfg.hidden_size, I = m.cfg.intermediate_size, T = m.T, B = batch, nl = m.cfg.num_layers;
    const int64_t M = (int64_t)B * T;
    const bool keep = training && tr.l0 < nl;
    auto saved = [&](int i) { return keep && i >= tr.l0; };
    GG_TRY(m.f32 ? embed_fwd<float>(e, x, e.A(L.tok)) : (m.f16 ? embed_fwd<f16>(e, x, e.A(L.tok)) : embed_fwd<bf16>(e, x, e.A(L.tok))));
    int64_t cur = saved(0) ? L.la[0].xin : L.s_x;
    GG_TRY(e.ln_fwd(e.A(L.tok), m.pre_g, m.pre_b, M, e.A(cur), keep && tr.embed ? e.F(L.mean0) : nullptr, keep && tr.embed ? e.F(L.rstd0) : nullptr));
    for (int i = 0; i < nl; ++i) {
        const LayerP& l = m.layers[i];
        const bool sv = saved(i);
        const LayerA& a = L.la[i];
        const int64_t A1 = sv ? a.a1 : L.s_a, QKV = sv ? a.qkv : L.s_qkv, O = sv ? a.o : L.s_o, XMID = sv ? a.xmid : cur, A2 = sv ? a.a2 : L.s_a,
                      H = sv ? a.h : L.s_h;
        const int64_t next = (keep && i + 1 >= tr.l0) ? (i + 1 < nl ? L.la[i + 1].xin : L.xfinal) : cur;
        GG_TRY(e.ln_fwd(e.A(cur), l.ln1_g, l.ln1_b, M, e.A(A1), sv ? e.F(a.mean1) : nullptr, sv ? e.F(a.rstd1) : nullptr));
        GG_TRY(e.gemm(e.A(A1), D, e.W(l.wqkv), D, e.A(QKV), 3 * D, M, 3 * D, D, (const float*)e.W(l.bqkv)));
        GgAttnArgs at;
        e.attn_args(at, e.A(QKV), e.A(O), sv ? e.F(a.lse) : nullptr);
        if (m.f16) GG_TRY(gg_attention_fwd_f16(&at, e.st));      // fp16 MFMA for towers of at most 256 tokens (ViT-B/32: 50); beyond: fp16 storage, f32 arithmetic
        else if (m.f32 || sv || T > 256) GG_TRY(gg_attention_flash_fwd(&at, m.f32 ? 1 : 0, e.st));
        else GG_TRY(gg_attention_fwd(&at, e.st));
        // x_mid = x + out_proj(o)   (in place when nothing is kept: each element is read then written by the same lane)
        GG_TRY(e.gemm(e.A(O), D, e.W(l.wo), D, e.A(XMID), D, M, D, D, e.P(l.o_b), 0, nullptr, e.A(cur)));
        GG_TRY(e.ln_fwd(e.A(XMID), l.ln2_g, l.ln2_b, M, e.A(A2), sv ? e.F(a.mean2) : nullptr, sv ? e.F(a.rstd2) : nullptr));
        GG_TRY(e.gemm(e.A(A2), D, e.W(l.w1), D, e.A(H), I, M, I, D, e.P(l.fc1_b), GG_ACT_CODE_QUICK_GELU, sv ? e.A(a.pre) : nullptr));
        GG_TRY(e.gemm(e.A(H), I, e.W(l.w2), I, e.A(next), D, M, D, I, e.P(l.fc2_b), 0, nullptr, e.A(XMID)));
        cur = next;
    }
    if (m.f32) {
        GG_TRY(gg_token_mean_fwd_f32((const float*)e.A(cur), out, B, T, D, e.st));
        if (last_hidden) GG_HIP(hipMemcpyAsync(last_hidden, e.A(cur), (size_t)M * D * 4, hipMemcpyDeviceToDevice, e.st));
    } else if (m.f16) {
        GG_TRY(gg_token_mean_fwd_f16(e.A(cur), out, B, T, D, e.st));
        if (last_hidden) GG_TRY(gg_cast_f16_to_f32(e.A(cur), last_hidden, M * D, e.st));
    } else {
        GG_TRY(gg_token_mean_fwd(e.A(cur), out, B, T, D, e.st));
        if (last_hidden) GG_TRY(gg_cast_bf16_to_f32(e.A(cur), last_hidden, M * D, e.st));
    }
    return 0;
}

// Backward of the training forward that last wrote `workspace` (same cfg, batch and trainable mask).  d_out: gradient of the pooled mean
// (batch, hidden) or NULL; d_last_hidden: gradient of last_hidden_state (batch, T, hidden) or NULL (both given: summed).  Gradients of the
// trainable tensors are ACCUMULATED into `grads` (flat, same offsets as params); post_layernorm is not on the path (its gradient is zero).
extern "C" int gg_clip_backward(const GgClipCfg* cfg, int batch, const float* params, const void* wcache, void* workspace, const float* d_out,
                                const float* d_last_hidden, float* grads, const uint8_t* trainable, void* stream) {
    CModel m;
    GG_TRY(build(cfg, m));
    GG_CHECK(batch > 0 && params && wcache && workspace && grads && (d_out || d_last_hidden), "gg_clip_backward: null pointer / bad batch");
    const Train tr = train_of(m, 1, trainable);
    const int nl = m.cfg.num_layers;
    if (tr.l0 >= nl) return 0;                    // nothing in the tower is trainable
    GG_CHECK(!m.f16, "gg_clip_backward: the fp16 mode is inference-only");
    CPlan L; plan(m, batch, tr, true, L);
    Exec e{&m, &L, batch, (hipStream_t)stream, params, (const char*)wcache, (char*)workspace, grads, trainable};
    const int D = m.cfg.hidden_size, I = m.cfg.intermediate_size, T = m.T, B = batch;
    const int64_t M = (int64_t)B * T, Mp = (int64_t)B * m.G * m.G;
    if (m.f32) hipLaunchKernelGGL(pool_bwd_kernel<float>, dim3(grid1d(M * D)), dim3(256), 0, e.st, d_out, d_last_hidden, (float*)e.A(L.g_x0), B, T, D);
    else hipLaunchKernelGGL(pool_bwd_kernel<bf16>, dim3(grid1d(M * D)), dim3(256), 0, e.st, d_out, d_last_hidden, (bf16*)e.A(L.g_x0), B, T, D);
    GG_LAUNCH_CHECK();
    int64_t dx = L.g_x0, other = L.g_x1;
    for (int i = nl - 1; i >= tr.l0; --i) {
        const LayerP& l = m.layers[i];
        const LayerA& a = L.la[i];
        // ---- MLP: x_out = x_mid + fc2(quick_gelu(fc1(LN2(x_mid))))
        GG_TRY(e.wgrad(l.fc2_w, e.A(dx), D, e.A(a.h), I, M, D, I));
        GG_TRY(e.bgrad(l.fc2_b, e.A(dx), D, M, D));
        GG_TRY(e.gemm(e.A(dx), D, e.W(l.w2T), D, e.A(L.g_h), I, M, I, D, nullptr, 0, nullptr, nullptr, e.A(a.pre), GG_ACT_CODE_QUICK_GELU));   // d pre
        GG_TRY(e.wgrad(l.fc1_w, e.A(L.g_h), I, e.A(a.a2), D, M, I, D));
        GG_TRY(e.bgrad(l.fc1_b, e.A(L.g_h), I, M, I));
        GG_TRY(e.gemm(e.A(L.g_h), I, e.W(l.w1T), I, e.A(L.g_a), D, M, D, I, nullptr));                                                        // d LN2 out
        GG_TRY(e.ln_bwd(e.A(L.g_a), e.A(a.xmid), e.F(a.mean2), e.F(a.rstd2), l.ln2_g, l.ln2_b, M, e.A(dx), e.A(other)));                       // d x_mid
        std::swap(dx, other);
        // ---- attention: x_mid = x_in + out_proj(attn(qkv(LN1(x_in))))
        GG_TRY(e.wgrad(l.o_w, e.A(dx), D, e.A(a.o), D, M, D, D));
        GG_TRY(e.bgrad(l.o_b, e.A(dx), D, M, D));
        GG_TRY(e.gemm(e.A(dx), D, e.W(l.woT), D, e.A(L.g_o), D, M, D, D, nullptr));                                                            // d o
        GgAttnArgs at;
        e.attn_args(at, e.A(a.qkv), e.A(a.o), e.F(a.lse));
        at.dout = e.A(L.g_o); at.lddo = D; at.dqkv = e.A(L.g_qkv);
        if (L.attn_ds >= 0) at.ds_scratch = e.F(L.attn_ds);
        GG_TRY(gg_attention_flash_bwd(&at, m.f32 ? 1 : 0, e.st));
        const char* dq = (const char*)e.A(L.g_qkv);
        GG_TRY(e.wgrad(l.q_w, dq, 3 * D, e.A(a.a1), D, M, D, D));
        GG_TRY(e.wgrad(l.k_w, dq + (int64_t)D * m.es, 3 * D, e.A(a.a1), D, M, D, D));
        GG_TRY(e.wgrad(l.v_w, dq + (int64_t)2 * D * m.es, 3 * D, e.A(a.a1), D, M, D, D));
        GG_TRY(e.bgrad(l.q_b, dq, 3 * D, M, D));
        GG_TRY(e.bgrad(l.k_b, dq + (int64_t)D * m.es, 3 * D, M, D));
        GG_TRY(e.bgrad(l.v_b, dq + (int64_t)2 * D * m.es, 3 * D, M, D));
        GG_TRY(e.gemm(e.A(L.g_qkv), 3 * D, e.W(l.wqkvT), 3 * D, e.A(L.g_a), D, M, D, 3 * D, nullptr));                                          // d LN1 out
        GG_TRY(e.ln_bwd(e.A(L.g_a), e.A(a.xin), e.F(a.mean1), e.F(a.rstd1), l.ln1_g, l.ln1_b, M, e.A(dx), e.A(other)));                         // d x_in
        std::swap(dx, other);
    }
    if (!tr.embed) return 0;
    // ---- embeddings: x_0 = pre_layrnorm(tokens), tokens = [cls ; patches . W^T] + pos
    GG_TRY(e.ln_bwd(e.A(dx), e.A(L.tok), e.F(L.mean0), e.F(L.rstd0), m.pre_g, m.pre_b, M, nullptr, e.A(other)));
    const bool wp = e.tr(m.patch_w);
    float* dpos = e.tr(m.pos) ? e.Gd(m.pos) : nullptr;
    float* dcls = e.tr(m.cls) ? e.Gd(m.cls) : nullptr;
    if (m.f32) hipLaunchKernelGGL(embed_bwd_kernel<float>, dim3((unsigned)gg_cdiv((int64_t)T * D, 256)), dim3(256), 0, e.st, (const float*)e.A(other), dpos, dcls,
                                  wp ? (float*)e.A(L.g_a) : nullptr, B, T, D);
    else hipLaunchKernelGGL(embed_bwd_kernel<bf16>, dim3((unsigned)gg_cdiv((int64_t)T * D, 256)), dim3(256), 0, e.st, (const bf16*)e.A(other), dpos, dcls,
                            wp ? (bf16*)e.A(L.g_a) : nullptr, B, T, D);
    GG_LAUNCH_CHECK();
    if (wp) {
        const int sp = m.f32 ? gg_gemm_tn_f32_splits((int)Mp, D, m.Kpatch) : gg_gemm_tn_splits((int)Mp, D, m.Kpatch);
        if (m.f32) GG_TRY(gg_gemm_tn_f32(e.A(L.g_a), D, e.A(L.col), m.Kpatch, (int)Mp, D, m.Kpatch, nullptr, 0, e.F(L.splitk), sp, e.st));
        else GG_TRY(gg_gemm_tn(e.A(L.g_a), D, e.A(L.col), m.Kpatch, (int)Mp, D, m.Kpatch, nullptr, 0, e.F(L.splitk), sp, e.st));
        GG_TRY(gg_splitk_reduce(e.F(L.splitk), e.F(L.splitk), (int64_t)D * m.Kpatch, sp, 0, 1.0f, e.st));
        hipLaunchKernelGGL(patch_wgrad_scatter_kernel, dim3((unsigned)gg_cdiv((int64_t)D * m.Kraw, 256)), dim3(256), 0, e.st, e.F(L.splitk), D, m.Kpatch,
                           m.Kraw, e.Gd(m.patch_w));
        GG_LAUNCH_CHECK();
    }
    return 0;
}
