// CLIP vision tower, inference only (transformers CLIPVisionModel as used by pretrain/clip_embedder.py:63-65
// and models/super_guessr.py:323-325: mean over ALL tokens of last_hidden_state, no post_layernorm).
// Patch embedding is a pure GEMM (stride == kernel); q/k/v projections are fused into one [3D, D] GEMM; the
// attention kernel is the same MFMA window-attention kernel as TinyViT with head_dim 64 and no bias table.
#include <string>
#include <vector>
#include <string.h>
#include <stdio.h>
#include "common.h"
#include "../../include/gg.h"

namespace {
struct TInfo { std::string name; int64_t offset, numel; int ndim; int64_t shape[4]; };
struct LayerP { int q_w, q_b, k_w, k_b, v_w, v_b, o_w, o_b, ln1_g, ln1_b, fc1_w, fc1_b, fc2_w, fc2_b, ln2_g, ln2_b;
                int64_t wqkv, bqkv, wo, w1, w2; };
struct CModel {
    GgClipCfg cfg;
    std::vector<TInfo> t;
    int64_t floats = 0, wc_bytes = 0;
    int cls, patch_w, pos, pre_g, pre_b, post_g, post_b;
    int64_t wpatch;
    std::vector<LayerP> layers;
    int T, G, Kpatch, Kraw;
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
    const int D = c->hidden_size, I = c->intermediate_size, P = c->patch_size;
    GG_CHECK(D > 0 && D % 64 == 0 && c->num_heads > 0 && D / c->num_heads == 64, "clip: head_dim must be 64 (hidden %d, heads %d)", D, c->num_heads);
    GG_CHECK(P > 0 && c->image_size % P == 0 && I % 8 == 0, "clip: bad patch/image/intermediate size");
    // patch-embedding contraction 3*P*P is padded to a multiple of 8 (ViT-L/14: 588 -> 592 zero columns); sequences beyond 256 tokens
    // (ViT-L/14-336: 577, the reference's CLIP_MODEL, config.py:6) run on the online-softmax attention kernels
    m.G = c->image_size / P; m.T = m.G * m.G + 1; m.Kraw = 3 * P * P; m.Kpatch = (int)gg_align(m.Kraw, 8);
    m.cls = addt(m, "embeddings.class_embedding", {D});
    m.patch_w = addt(m, "embeddings.patch_embedding.weight", {D, 3, P, P});
    m.pos = addt(m, "embeddings.position_embedding.weight", {m.T, D});
    m.pre_g = addt(m, "pre_layrnorm.weight", {D}); m.pre_b = addt(m, "pre_layrnorm.bias", {D});
    m.wpatch = wca(m, (int64_t)D * m.Kpatch * 2);
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
        l.wqkv = wca(m, (int64_t)3 * D * D * 2);
        l.bqkv = wca(m, (int64_t)3 * D * 4);
        l.wo = wca(m, (int64_t)D * D * 2);
        l.w1 = wca(m, (int64_t)I * D * 2);
        l.w2 = wca(m, (int64_t)D * I * 2);
    }
    m.post_g = addt(m, "post_layernorm.weight", {D}); m.post_b = addt(m, "post_layernorm.bias", {D});
    return 0;
}

// x f32 NCHW (B,3,S,S) -> col bf16 [B*G*G, 3*P*P], k = (c, py, px)  (== Conv2d(kernel=stride=P) weight flatten)
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, bf16* __restrict__ col, int B, int S, int P, int G, int K) {
    const int Kraw = 3 * P * P;         // K = Kraw padded to a multiple of 8 (zero columns)
    const int64_t total = (int64_t)B * G * G * (K / 8);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int kc = (int)(i % (K / 8)) * 8;
        const int64_t p = i / (K / 8);
        const int gx = (int)(p % G), gy = (int)((p / G) % G), b = (int)(p / ((int64_t)G * G));
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kc + j;
            const int c = k / (P * P), py = (k / P) % P, px = k % P;
            o[j] = k < Kraw ? (bf16)x[(((int64_t)b * 3 + c) * S + gy * P + py) * S + gx * P + px] : (bf16)0.f;
        }
        *reinterpret_cast<bf16x8*>(col + p * K + kc) = o;
    }
}
// tokens[b,0,:] = cls + pos[0];  tokens[b,1+i,:] = patches[b,i,:] + pos[1+i]
__global__ void assemble_tokens_kernel(const bf16* __restrict__ patches, const float* __restrict__ cls, const float* __restrict__ pos,
                                       bf16* __restrict__ tokens, int B, int T, int D) {
    const int64_t total = (int64_t)B * T * D;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int dd = (int)(i % D);
        const int t = (int)((i / D) % T);
        const int64_t b = i / ((int64_t)D * T);
        const float v = t == 0 ? cls[dd] : (float)patches[(b * (T - 1) + (t - 1)) * D + dd];
        tokens[i] = (bf16)(v + pos[(int64_t)t * D + dd]);
    }
}
// f32 [R][K] -> bf16 [R][Kp] with zero padding columns
__global__ void cast_pad_rows_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int R, int K, int Kp) {
    const int64_t n = (int64_t)R * Kp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kp);
        dst[i] = k < K ? (bf16)src[(i / Kp) * K + k] : (bf16)0.f;
    }
}
__global__ void cast_rows_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (bf16)src[i];
}
__global__ void copy_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static int cast_w(const float* src, bf16* dst, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(cast_rows_kernel, dim3((unsigned)std::min<int64_t>(gg_cdiv(n, 256), 8192)), dim3(256), 0, st, src, dst, n);
    GG_LAUNCH_CHECK();
    return 0;
}
struct CLayout { int64_t col, patches, x, a, qkv, o, h, total; };
static void layout(const CModel& m, int B, CLayout& L) {
    const int D = m.cfg.hidden_size, I = m.cfg.intermediate_size;
    const int64_t Mp = (int64_t)B * m.G * m.G, M = (int64_t)B * m.T;
    int64_t off = 0;
    auto al = [&](int64_t bytes) { int64_t o = off; off += gg_align(bytes, 256); return o; };
    L.col = al(Mp * m.Kpatch * 2); L.patches = al(Mp * D * 2); L.x = al(M * D * 2); L.a = al(M * D * 2);
    L.qkv = al(M * 3 * D * 2); L.o = al(M * D * 2); L.h = al(M * I * 2);
    L.total = off;
}
}  // namespace

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
extern "C" int64_t gg_clip_workspace_bytes(const GgClipCfg* cfg, int batch) {
    CModel m;
    if (build(cfg, m) || batch <= 0) return -1;
    CLayout L; layout(m, batch, L);
    return L.total;
}
extern "C" int gg_clip_refresh_weights(const GgClipCfg* cfg, const float* params, void* wcache, void* stream) {
    CModel m;
    GG_TRY(build(cfg, m));
    GG_CHECK(params && wcache, "gg_clip_refresh_weights: null pointer");
    char* wc = (char*)wcache;
    hipStream_t st = (hipStream_t)stream;
    const int64_t D = m.cfg.hidden_size, I = m.cfg.intermediate_size;
    auto P = [&](int t) { return params + m.t[t].offset; };
    hipLaunchKernelGGL(cast_pad_rows_kernel, dim3((unsigned)std::min<int64_t>(gg_cdiv(D * m.Kpatch, 256), 8192)), dim3(256), 0, st, P(m.patch_w),
                       (bf16*)(wc + m.wpatch), (int)D, m.Kraw, m.Kpatch);
    for (auto& l : m.layers) {
        bf16* wq = (bf16*)(wc + l.wqkv);
        GG_TRY(cast_w(P(l.q_w), wq, D * D, st));
        GG_TRY(cast_w(P(l.k_w), wq + D * D, D * D, st));
        GG_TRY(cast_w(P(l.v_w), wq + 2 * D * D, D * D, st));
        float* bq = (float*)(wc + l.bqkv);
        hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)gg_cdiv(D, 256)), dim3(256), 0, st, P(l.q_b), bq, D);
        hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)gg_cdiv(D, 256)), dim3(256), 0, st, P(l.k_b), bq + D, D);
        hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)gg_cdiv(D, 256)), dim3(256), 0, st, P(l.v_b), bq + 2 * D, D);
        GG_TRY(cast_w(P(l.o_w), (bf16*)(wc + l.wo), D * D, st));
        GG_TRY(cast_w(P(l.fc1_w), (bf16*)(wc + l.w1), I * D, st));
        GG_TRY(cast_w(P(l.fc2_w), (bf16*)(wc + l.w2), D * I, st));
    }
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_clip_forward(const GgClipCfg* cfg, int batch, const float* params, const void* wcache, const float* x, void* workspace,
                               float* out, float* last_hidden, void* stream) {
    CModel m;
    GG_TRY(build(cfg, m));
    GG_CHECK(batch > 0 && params && wcache && x && workspace && out, "gg_clip_forward: null pointer / bad batch");
    CLayout L; layout(m, batch, L);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace; const char* wc = (const char*)wcache;
    const int D = m.cfg.hidden_size, I = m.cfg.intermediate_size, T = m.T, B = batch;
    const int64_t Mp = (int64_t)B * m.G * m.G, M = (int64_t)B * T;
    auto P = [&](int t) { return params + m.t[t].offset; };
    auto A = [&](int64_t o) { return reinterpret_cast<bf16*>(ws + o); };
    auto gemm = [&](const bf16* Am, int64_t lda, const bf16* Bm, int64_t ldb, bf16* C, int64_t ldc, int64_t Mm, int N, int K,
                    const float* bias, int act, const bf16* residual) {
        GgGemmArgs g;
        memset(&g, 0, sizeof(g));
        g.A = Am; g.lda = lda; g.B = Bm; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = (int)Mm; g.N = N; g.K = K;
        g.bias = bias; g.act = act; g.residual = residual; g.ldr = ldc;
        return gg_gemm_nt(&g, st);
    };
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)std::min<int64_t>(gg_cdiv(Mp * (m.Kpatch / 8), 256), 32768)), dim3(256), 0, st, x,
                       A(L.col), B, m.cfg.image_size, m.cfg.patch_size, m.G, m.Kpatch);
    GG_TRY(gemm(A(L.col), m.Kpatch, (const bf16*)(wc + m.wpatch), m.Kpatch, A(L.patches), D, Mp, D, m.Kpatch, nullptr, 0, nullptr));
    hipLaunchKernelGGL(assemble_tokens_kernel, dim3((unsigned)std::min<int64_t>(gg_cdiv(M * D, 256), 32768)), dim3(256), 0, st, A(L.patches),
                       P(m.cls), P(m.pos), A(L.a), B, T, D);
    GG_LAUNCH_CHECK();
    GG_TRY(gg_layernorm_fwd(A(L.a), 0, P(m.pre_g), P(m.pre_b), M, D, m.cfg.ln_eps, A(L.x), 0, nullptr, nullptr, st));
    bf16* xcur = A(L.x);
    for (auto& l : m.layers) {
        GG_TRY(gg_layernorm_fwd(xcur, 0, P(l.ln1_g), P(l.ln1_b), M, D, m.cfg.ln_eps, A(L.a), 0, nullptr, nullptr, st));
        GG_TRY(gemm(A(L.a), D, (const bf16*)(wc + l.wqkv), D, A(L.qkv), 3 * D, M, 3 * D, D, (const float*)(wc + l.bqkv), 0, nullptr));
        GgAttnArgs at;
        memset(&at, 0, sizeof(at));
        at.qkv = A(L.qkv); at.ld = 3 * D; at.q_off = 0; at.k_off = D; at.v_off = 2 * D; at.head_stride = 64; at.head_dim = 64;
        at.num_heads = m.cfg.num_heads; at.num_windows = B; at.tokens_per_window = T; at.window_size = 0;
        at.scale = 0.125f; at.out = A(L.o); at.ldo = D;
        GG_TRY(gg_attention_fwd(&at, st));
        // x = x + out_proj(o)   (in place: each element is read then written by the same lane)
        GG_TRY(gemm(A(L.o), D, (const bf16*)(wc + l.wo), D, xcur, D, M, D, D, P(l.o_b), 0, xcur));
        GG_TRY(gg_layernorm_fwd(xcur, 0, P(l.ln2_g), P(l.ln2_b), M, D, m.cfg.ln_eps, A(L.a), 0, nullptr, nullptr, st));
        GG_TRY(gemm(A(L.a), D, (const bf16*)(wc + l.w1), D, A(L.h), I, M, I, D, P(l.fc1_b), GG_ACT_QUICK_GELU, nullptr));
        GG_TRY(gemm(A(L.h), I, (const bf16*)(wc + l.w2), I, xcur, D, M, D, I, P(l.fc2_b), 0, xcur));
    }
    GG_TRY(gg_token_mean_fwd(xcur, out, B, T, D, st));
    if (last_hidden) GG_TRY(gg_cast_bf16_to_f32(xcur, last_hidden, M * D, st));
    return 0;
}
