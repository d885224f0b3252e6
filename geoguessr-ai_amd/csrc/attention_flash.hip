// Flash (online-softmax) attention, forward and backward, for ANY window / sequence length, with f32 arithmetic on
// v_mfma_f32_16x16x4_f32 (gfx950).  Two users:
//   * the reference-precision (fp32) mode: every attention of TinyViT (timm Attention.forward reached through models/tinyvit.py:135)
//     on f32 activations;
//   * the reference's own default shapes, which the register-resident kernels of attention.hip (<= 256 tokens) cannot hold:
//     tiny_vit_21m_512 (32x32 = 1024-token windows, config.py:9), tiny_vit_21m_384 (24x24), CLIP ViT-L/14-336 (577 tokens,
//     config.py:6) -- on bf16 activations (operands are widened to f32 on their way into LDS).
//
// Work decomposition: one 256-thread workgroup per (window, head, 64-query tile) [forward, dQ pass] or per (window, head, 64-key
// tile) [dK/dV pass]; a wave owns 16 of those rows.  K/V (resp. Q/dO) tiles of 64 tokens are staged in LDS as f32 rows of D + 4
// floats.  Scores are computed "swapped" (S^T = K Q^T in the forward / dQ pass, S = Q K^T in the dK/dV pass) so that the row a lane
// owns -- its query resp. its key -- is the MFMA column index (lane & 15): running max / sum / log-sum-exp / delta are lane-local,
// a tile's softmax reduction is two shuffles, and the exponentiated tile is directly the B operand of the next product.
// MFMA operands are one f32 per lane: row-contiguous fragments come out of LDS as ds_read_b128 (4 consecutive k feed 4 successive
// MFMA steps, the k permutation is the same on both operands), "k-major" fragments as ds_read_b32 of 16 consecutive floats per row.
// The relative-position bias is looked up in the compact f32 table attention_biases[h][|dy|*ws+|dx|] (LDS copy), its gradient is
// summed per workgroup in LDS.  The backward pass runs in two passes (dQ; dK,dV): no atomics on dq/dk/dv, deterministic.
#include "common.h"
#include "../../include/gg.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

struct FlashParams {
    const void* qkv; int64_t ld;
    int q_off, k_off, v_off, head_stride;
    void* out; int64_t ldo;
    const float* bias_table;          // [nh][ws*ws] f32 or null
    int ws, nWx, nWy, H, W;
    int N, nh;
    float scale;
    const void* dout; int64_t lddo;
    void* dqkv;
    float* dbias; float* dbias_part;  // [nh][ws*ws] accumulated into (atomics) / per-workgroup partial rows [rows][nh][ws*ws]
    float* lse;                       // [tokens][nh]
    float* ds_scratch;                // [windows][nh][npad][npad] f32 or null: dS handed from the dK/dV pass to the dQ pass
    int ntile;                        // ceil(N / 64)
    int npad, nbpad;                  // tokens rounded up to 16 (rows of a resident LDS image); floats reserved for a bias table
};

template <typename T> struct Ld4;
template <> struct Ld4<float> {
    static __device__ __forceinline__ f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void store(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Ld4<bf16> {
    static __device__ __forceinline__ f32x4 load(const bf16* p) {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
        return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
    static __device__ __forceinline__ void store(bf16* p, f32x4 v) {
        *reinterpret_cast<bf16x4*>(p) = (bf16x4){(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    }
};

template <> struct Ld4<f16> {
    static __device__ __forceinline__ f32x4 load(const f16* p) {
        const f16x4 v = *reinterpret_cast<const f16x4*>(p);
        return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
    static __device__ __forceinline__ void store(f16* p, f32x4 v) {
        *reinterpret_cast<f16x4*>(p) = (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
    }
};

__device__ __forceinline__ int64_t fl_origin(const FlashParams& p, int w) {
    if (p.ws == 0) return (int64_t)w * p.N;
    const int per_img = p.nWx * p.nWy;
    const int b = w / per_img, r = w % per_img;
    const int wy = r / p.nWx, wx = r % p.nWx;
    return ((int64_t)b * p.H + wy * p.ws) * p.W + wx * p.ws;
}
__device__ __forceinline__ int64_t fl_token(const FlashParams& p, int64_t origin, int t) {      // t < N
    if (p.ws == 0) return origin + t;
    const int i = t / p.ws, j = t - i * p.ws;
    return origin + (int64_t)i * p.W + j;
}

// stage rows [t0, t0+64) x D of one head's column block into X[.][D+4] (f32; zero rows beyond N; rows >= rlim are not written)
template <typename T, int D>
__device__ __forceinline__ void fl_stage(const FlashParams& p, const T* base, int64_t ld, int col, int64_t origin, int t0, float* X, int rlim = 64) {
    constexpr int CH = D / 4, RS = D + 4;
#pragma unroll
    for (int i = 0; i < 64 * CH / 256; ++i) {
        const int id = threadIdx.x + i * 256;
        const int row = id / CH, ch = id % CH;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t0 + row < p.N) v = Ld4<T>::load(base + fl_token(p, origin, t0 + row) * ld + col + ch * 4);
        if (row < rlim) *reinterpret_cast<f32x4*>(X + row * RS + ch * 4) = v;
    }
}
// resident form: all p.npad rows of one head's column block (zero rows beyond N), any block size
template <typename T, int D>
__device__ __forceinline__ void fl_stage_all(const FlashParams& p, const T* base, int64_t ld, int col, int64_t origin, float* X) {
    constexpr int CH = D / 4, RS = D + 4;
    for (int id = threadIdx.x; id < p.npad * CH; id += blockDim.x) {
        const int row = id / CH, ch = id % CH;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < p.N) v = Ld4<T>::load(base + fl_token(p, origin, row) * ld + col + ch * 4);
        *reinterpret_cast<f32x4*>(X + row * RS + ch * 4) = v;
    }
}
// Relative-position bias without per-element index arithmetic: the compact table attention_biases[h][|dy|*ws+|dx|] is expanded in LDS to
// E[(dy + ws-1) * (2ws-1) + (dx + ws-1)] (pre-multiplied by log2 e: the softmax runs in the exp2 domain), so that the entry for a
// (query, key) pair sits at byte offset qlin4 - klin4[key] with  lin4(t) = 4 * (cy(t) * (2ws-1) + cx(t)),  qlin4 = lin4(q) + 4*(ws-1)*2ws:
// one subtraction and one LDS read per score (it was two coordinate reads, two abs-differences, a multiply-add and the table read).
__device__ __forceinline__ void fl_stage_bias(const FlashParams& p, int h, float* E) {
    const int w2 = 2 * p.ws - 1;
    for (int i = threadIdx.x; i < w2 * w2; i += blockDim.x) {
        const int dy = i / w2 - (p.ws - 1), dx = i % w2 - (p.ws - 1);
        E[i] = p.bias_table[h * p.ws * p.ws + abs(dy) * p.ws + abs(dx)] * 1.4426950408889634f;
    }
}
__device__ __forceinline__ int fl_lin4(const FlashParams& p, int t) {       // t < N
    const int r = t / p.ws;
    return 4 * (r * (2 * p.ws - 1) + (t - r * p.ws));
}
// klin4 of tokens [t0, t0+n)
__device__ __forceinline__ void fl_stage_coords(const FlashParams& p, int t0, int* klin, int n = 64) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) klin[i] = fl_lin4(p, min(t0 + i, p.N - 1));
}

// RES (resident) variants: windows whose K/V (resp. Q/dO) fit in LDS next to a second workgroup (p.npad rows = tokens rounded up to 16, e.g. the
// 14x14 windows of TinyViT stage 2) are staged ONCE by one workgroup per (window, head); its four waves then walk the 16-row strips
// (strip = wave, wave + 4, ...) with no barrier inside.  The streaming form re-staged every K/V tile for each 64-query tile (4x for 196
// tokens) behind two barriers per tile.  Same arithmetic in the same order per row: results are bit-identical between the two forms.
// LDS carve-up (dynamic): [R][RS] x 2 operand images, the bias table (p.nbpad floats), coordinates and per-row scalars; R = RES ? npad : 64.

// ------------------------------------------------------------------------------------------- forward
template <typename T, int D, bool RES>
__global__ __launch_bounds__(RES ? 1024 : 256) void flash_fwd_kernel(FlashParams p) {
    constexpr int RS = D + 4, DC = D / 16;
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    const int R = RES ? p.npad : 64;
    float* Ks = fsm;
    float* Vs = Ks + R * RS;
    float* btab = Vs + R * RS;
    int* klin = reinterpret_cast<int*>(btab + p.nbpad);            // byte-scaled linear window coordinate of every staged key
    const int qt = RES ? 0 : blockIdx.x % p.ntile;
    const int wh = RES ? blockIdx.x : blockIdx.x / p.ntile;
    const int h = wh % p.nh, w = wh / p.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int64_t origin = fl_origin(p, w);
    const T* qkv = reinterpret_cast<const T*>(p.qkv);
    const int hc = h * p.head_stride;
    const bool has_bias = p.bias_table != nullptr;
    if (has_bias) fl_stage_bias(p, h, btab);
    if (RES) {
        fl_stage_all<T, D>(p, qkv, p.ld, p.k_off + hc, origin, Ks);
        fl_stage_all<T, D>(p, qkv, p.ld, p.v_off + hc, origin, Vs);
        if (has_bias) fl_stage_coords(p, 0, klin, p.npad);
        __syncthreads();
    }
    const int nstrips = (p.N + 15) >> 4;
    for (int strip = RES ? wave : qt * 4 + wave; !RES || strip < nstrips; strip += (int)(blockDim.x >> 6)) {
    const int qi = strip * 16 + lr;
    const bool qok = qi < p.N;
    const bool wave_on = strip * 16 < p.N;                          // wave-uniform
    const int64_t qtok = fl_token(p, origin, min(qi, p.N - 1));
    f32x4 qf[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) {
        qf[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (qok) qf[c] = Ld4<T>::load(qkv + qtok * p.ld + p.q_off + hc + 16 * c + 4 * lg);
    }
    const int qq = min(qi, p.N - 1);
    const int qlin = has_bias ? fl_lin4(p, qq) + 4 * (p.ws - 1) * 2 * p.ws : 0;       // + the table's centre (dy = dx = 0)
    float m = -1e30f, l = 0.f;
    const float sc2 = p.scale * 1.4426950408889634f;
    f32x4 oacc[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) oacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kt0 = 0; kt0 < p.ntile; ++kt0) {
        const int t0 = kt0 * 64;
        if (!RES) {
            __syncthreads();
            fl_stage<T, D>(p, qkv, p.ld, p.k_off + hc, origin, t0, Ks);
            fl_stage<T, D>(p, qkv, p.ld, p.v_off + hc, origin, t0, Vs);
            if (has_bias) fl_stage_coords(p, t0, klin);
            __syncthreads();
            if (!wave_on) continue;
        }
        const float* Kt = Ks + (RES ? t0 * RS : 0);
        const float* Vt = Vs + (RES ? t0 * RS : 0);
        const int* klt = klin + (RES ? t0 : 0);
        const int nsub = min(4, (p.N - t0 + 15) / 16);
        f32x4 st[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            // padded keys are masked through the accumulators' INITIAL value (-inf stays -inf under the products and the bias): no compare + select
            // per score (fp32 MFMA and VALU share the SIMD's issue).  Wave-uniform cases: only the window's last sub-tile holds padded keys
            st[kt] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (kt < nsub) {
                st[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (t0 + 16 * kt + 15 >= p.N) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[kt][r] = (t0 + 16 * kt + 4 * lg + r < p.N) ? 0.f : -INFINITY;
                }
#pragma unroll
                for (int c = 0; c < DC; ++c) {
                    const f32x4 kf = *reinterpret_cast<const f32x4*>(Kt + (16 * kt + lr) * RS + 16 * c + 4 * lg);
#pragma unroll
                    for (int s = 0; s < 4; ++s) st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s], qf[c][s], st[kt], 0, 0, 0);
                }
            }
        }
        // lane holds S^T[key = t0 + 16kt + 4lg + r][q = lr]
        float tmax = -1e30f;
        f32x4 bia[4];                               // the 16 bias entries of this lane, all LDS reads in flight together
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            bia[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (has_bias && kt < nsub) {
                const i32x4 kl4 = *reinterpret_cast<const i32x4*>(klt + 16 * kt + 4 * lg);
#pragma unroll
                for (int r = 0; r < 4; ++r) bia[kt][r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(btab) + (qlin - kl4[r]));
            }
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {          // scores in the exp2 domain: s * scale * log2 e + bias * log2 e
                const float s = fmaf(st[kt][r], sc2, bia[kt][r]);
                st[kt][r] = s;
                tmax = fmaxf(tmax, s);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mn = fmaxf(m, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        float ls = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(st[kt][r] - mn); st[kt][r] = e; ls += e; }
        ls += __shfl_xor(ls, 16, 64);
        ls += __shfl_xor(ls, 32, 64);
        l = l * alpha + ls;
        m = mn;
#pragma unroll
        for (int c = 0; c < DC; ++c) oacc[c] *= alpha;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt < nsub) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int c = 0; c < DC; ++c) {
                        const float vf = Vt[(16 * kt + 4 * lg + r) * RS + 16 * c + lr];
                        oacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf, st[kt][r], oacc[c], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (qok) {
        const float inv = 1.0f / l;
        T* out = reinterpret_cast<T*>(p.out);
#pragma unroll
        for (int c = 0; c < DC; ++c) Ld4<T>::store(out + qtok * p.ldo + h * D + 16 * c + 4 * lg, oacc[c] * inv);
        if (p.lse && lg == 0) p.lse[qtok * p.nh + h] = m * 0.6931471805599453f + __logf(l);       // natural-log lse (m is a base-2 exponent)
    }
    if (!RES) break;
    }
}

// ------------------------------------------------------------------------------------------- backward, pass A: dQ
// QS = query strips a wave works on at once: every K / V fragment read from LDS then feeds QS independent score / dP / dQ chains (QS = 2
// in the resident form: half the LDS reads per MFMA and twice the independent MFMA work per wave).
template <typename T, int D, bool RES, int QS = 1>
__global__ __launch_bounds__(RES ? 1024 : 256) void flash_bwd_dq_kernel(FlashParams p) {
    constexpr int RS = D + 4, DC = D / 16;
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    const int R = RES ? p.npad : 64;
    float* Ks = fsm;
    float* Vs = Ks + R * RS;
    float* btab = Vs + R * RS;
    int* klin = reinterpret_cast<int*>(btab + p.nbpad);            // byte-scaled linear window coordinate of every staged key
    const int qt = RES ? 0 : blockIdx.x % p.ntile;
    const int wh = RES ? blockIdx.x : blockIdx.x / p.ntile;
    const int h = wh % p.nh, w = wh / p.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int64_t origin = fl_origin(p, w);
    const T* qkv = reinterpret_cast<const T*>(p.qkv);
    const T* dout = reinterpret_cast<const T*>(p.dout);
    const T* outp = reinterpret_cast<const T*>(p.out);
    const int hc = h * p.head_stride;
    const bool has_bias = p.bias_table != nullptr;
    if (has_bias) fl_stage_bias(p, h, btab);
    if (RES) {
        fl_stage_all<T, D>(p, qkv, p.ld, p.k_off + hc, origin, Ks);
        fl_stage_all<T, D>(p, qkv, p.ld, p.v_off + hc, origin, Vs);
        if (has_bias) fl_stage_coords(p, 0, klin, p.npad);
        __syncthreads();
    }
    const int nstrips = (p.N + 15) >> 4;
    const int nwaves = (int)(blockDim.x >> 6);
    const float sc2 = p.scale * 1.4426950408889634f;
    for (int strip0 = RES ? wave * QS : qt * 4 + wave; !RES || strip0 < nstrips; strip0 += nwaves * QS) {
    bool qok[QS];
    int64_t qtok[QS];
    f32x4 qf[QS][DC], dof[QS][DC], dq[QS][DC];
    float delta[QS], lse2[QS];
    int qlin[QS];
#pragma unroll
    for (int u = 0; u < QS; ++u) {
        const int qi = (strip0 + u) * 16 + lr;
        qok[u] = qi < p.N;
        qtok[u] = fl_token(p, origin, min(qi, p.N - 1));
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            qf[u][c] = dof[u][c] = dq[u][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (qok[u]) {
                qf[u][c] = Ld4<T>::load(qkv + qtok[u] * p.ld + p.q_off + hc + 16 * c + 4 * lg);
                dof[u][c] = Ld4<T>::load(dout + qtok[u] * p.lddo + h * D + 16 * c + 4 * lg);
                const f32x4 o = Ld4<T>::load(outp + qtok[u] * p.ldo + h * D + 16 * c + 4 * lg);
#pragma unroll
                for (int s = 0; s < 4; ++s) d = fmaf(dof[u][c][s], o[s], d);
            }
        }
        d += __shfl_xor(d, 16, 64);
        d += __shfl_xor(d, 32, 64);
        delta[u] = d;
        lse2[u] = -(qok[u] ? p.lse[qtok[u] * p.nh + h] : 0.f) / p.scale;          // initial value of the S accumulators (see fl_stage_rowstats)
        qlin[u] = has_bias ? fl_lin4(p, min(qi, p.N - 1)) + 4 * (p.ws - 1) * 2 * p.ws : 0;       // + the table's centre (dy = dx = 0)
    }
    const bool wave_on = strip0 * 16 < p.N;

    for (int kt0 = 0; kt0 < p.ntile; ++kt0) {
        const int t0 = kt0 * 64;
        if (!RES) {
            __syncthreads();
            fl_stage<T, D>(p, qkv, p.ld, p.k_off + hc, origin, t0, Ks);
            fl_stage<T, D>(p, qkv, p.ld, p.v_off + hc, origin, t0, Vs);
            if (has_bias) fl_stage_coords(p, t0, klin);
            __syncthreads();
            if (!wave_on) continue;
        }
        const float* Kt = Ks + (RES ? t0 * RS : 0);
        const float* Vt = Vs + (RES ? t0 * RS : 0);
        const int* klt = klin + (RES ? t0 : 0);
        const int nsub = min(4, (p.N - t0 + 15) / 16);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt >= nsub) continue;
            f32x4 st[QS], dp[QS];
#pragma unroll
            for (int u = 0; u < QS; ++u) { st[u] = (f32x4){lse2[u], lse2[u], lse2[u], lse2[u]}; dp[u] = (f32x4){-delta[u], -delta[u], -delta[u], -delta[u]}; }
            if (t0 + 16 * kt + 15 >= p.N) {          // wave-uniform: only the window's last sub-tile holds padded keys; -inf there -> P = 0
#pragma unroll
                for (int u = 0; u < QS; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (t0 + 16 * kt + 4 * lg + r >= p.N) st[u][r] = -INFINITY;
            }
#pragma unroll
            for (int c = 0; c < DC; ++c) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(Kt + (16 * kt + lr) * RS + 16 * c + 4 * lg);
                const f32x4 vf = *reinterpret_cast<const f32x4*>(Vt + (16 * kt + lr) * RS + 16 * c + 4 * lg);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int u = 0; u < QS; ++u) {
                        st[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s], qf[u][c][s], st[u], 0, 0, 0);
                        dp[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[s], dof[u][c][s], dp[u], 0, 0, 0);
                    }
            }
            // lane holds S^T / dP^T [key = t0 + 16kt + 4lg + r][q = lr] of each strip
            f32x4 bia[QS];
#pragma unroll
            for (int u = 0; u < QS; ++u) bia[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (has_bias) {
                const i32x4 kl4 = *reinterpret_cast<const i32x4*>(klt + 16 * kt + 4 * lg);
#pragma unroll
                for (int u = 0; u < QS; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) bia[u][r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(btab) + (qlin[u] - kl4[r]));
            }
#pragma unroll
            for (int u = 0; u < QS; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = __builtin_amdgcn_exp2f(fmaf(st[u][r], sc2, bia[u][r]));      // rows beyond N are never stored: no query mask needed in this pass
                    st[u][r] = pr * dp[u][r];                            // dS^T; the softmax scale is applied once, to dQ
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < DC; ++c) {
                    const float kfs = Kt[(16 * kt + 4 * lg + r) * RS + 16 * c + lr];
#pragma unroll
                    for (int u = 0; u < QS; ++u) dq[u][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(kfs, st[u][r], dq[u][c], 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int u = 0; u < QS; ++u)
        if (qok[u]) {
            T* dqkv = reinterpret_cast<T*>(p.dqkv);
#pragma unroll
            for (int c = 0; c < DC; ++c) Ld4<T>::store(dqkv + qtok[u] * p.ld + p.q_off + hc + 16 * c + 4 * lg, dq[u][c] * p.scale);
        }
    if (!RES) break;
    }
}

// ------------------------------------------------------------------------------------------- backward, pass A from stored dS
// With a dS scratch (GgAttnArgs.ds_scratch) the dK/dV pass runs FIRST and leaves dS = P o (dP - delta) of every (query, key) pair as f32
// [window][head][key strip][query][16 keys] (npad^2 floats per head); this pass then is one product, dQ = scale * dS K: no Q K^T, no dO V^T, no exponentials, no bias lookups -- 1 of the
// two-pass scheme's 7 products instead of 3 (the scratch costs 2 x npad^2 x 4 bytes of HBM traffic per (window, head): 4.2 GB per 14 x 14 layer).
// Lane (lr, lg) of a strip reads its query row lr, keys 4 lg .. 4 lg + 3 of a 16-key sub-tile: exactly the B operand of the next 4 MFMA steps.
// dQ rows of QS query strips from the stored dS and a RESIDENT K image (all p.npad rows in LDS): shared by the stand-alone pass below and by the
// second phase of the resident dK/dV kernel
template <typename T, int D, int QS>
__device__ __forceinline__ void fl_dq_from_ds_resident(const FlashParams& p, const float* Ks, const float* dsb, int strip0, int64_t origin, int hc, int lr, int lg) {
    constexpr int RS = D + 4, DC = D / 16;
    f32x4 dq[QS][DC];
    const float* dsrow[QS];
#pragma unroll
    for (int u = 0; u < QS; ++u) {
#pragma unroll
        for (int c = 0; c < DC; ++c) dq[u][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dsrow[u] = dsb + (int64_t)min((strip0 + u) * 16 + lr, p.npad - 1) * 16 + 4 * lg;          // + key strip * npad * 16
    }
    for (int kt0 = 0; kt0 < p.ntile; ++kt0) {
        const int t0 = kt0 * 64;
        const float* Kt = Ks + t0 * RS;
        const int nsub = min(4, (p.N - t0 + 15) / 16);
        f32x4 st[4][QS];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int u = 0; u < QS; ++u) st[kt][u] = kt < nsub ? *reinterpret_cast<const f32x4*>(dsrow[u] + (int64_t)(4 * kt0 + kt) * p.npad * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt >= nsub) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < DC; ++c) {
                    const float kfs = Kt[(16 * kt + 4 * lg + r) * RS + 16 * c + lr];
#pragma unroll
                    for (int u = 0; u < QS; ++u) dq[u][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(kfs, st[kt][u][r], dq[u][c], 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int u = 0; u < QS; ++u) {
        const int qi = (strip0 + u) * 16 + lr;
        if (qi < p.N) {
            T* dqkv = reinterpret_cast<T*>(p.dqkv);
            const int64_t qtok = fl_token(p, origin, qi);
#pragma unroll
            for (int c = 0; c < DC; ++c) Ld4<T>::store(dqkv + qtok * p.ld + p.q_off + hc + 16 * c + 4 * lg, dq[u][c] * p.scale);
        }
    }
}
template <typename T, int D, bool RES, int QS>
__global__ __launch_bounds__(RES ? 1024 : 256) void flash_bwd_dq_ds_kernel(FlashParams p) {
    constexpr int RS = D + 4, DC = D / 16;
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    float* Ks = fsm;
    const int qt = RES ? 0 : blockIdx.x % p.ntile;
    const int wh = RES ? blockIdx.x : blockIdx.x / p.ntile;
    const int h = wh % p.nh, w = wh / p.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int64_t origin = fl_origin(p, w);
    const T* qkv = reinterpret_cast<const T*>(p.qkv);
    const int hc = h * p.head_stride;
    if (RES) {
        fl_stage_all<T, D>(p, qkv, p.ld, p.k_off + hc, origin, Ks);
        __syncthreads();
    }
    const float* dsb = p.ds_scratch + (int64_t)wh * p.npad * p.npad;
    const int nstrips = (p.N + 15) >> 4;
    const int nwaves = (int)(blockDim.x >> 6);
    for (int strip0 = RES ? wave * QS : qt * 4 + wave; !RES || strip0 < nstrips; strip0 += nwaves * QS) {
    f32x4 dq[QS][DC];
    const float* dsrow[QS];
#pragma unroll
    for (int u = 0; u < QS; ++u) {
#pragma unroll
        for (int c = 0; c < DC; ++c) dq[u][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dsrow[u] = dsb + (int64_t)min((strip0 + u) * 16 + lr, p.npad - 1) * 16 + 4 * lg;          // + key strip * npad * 16
    }
    const bool wave_on = strip0 * 16 < p.N;
    for (int kt0 = 0; kt0 < p.ntile; ++kt0) {
        const int t0 = kt0 * 64;
        if (!RES) {
            __syncthreads();
            fl_stage<T, D>(p, qkv, p.ld, p.k_off + hc, origin, t0, Ks);
            __syncthreads();
            if (!wave_on) continue;
        }
        const float* Kt = Ks + (RES ? t0 * RS : 0);
        const int nsub = min(4, (p.N - t0 + 15) / 16);
        // all dS fragments of the 64-key tile in flight before its first MFMA (the pass is bound by these loads: 2 x npad^2 floats per (window, head))
        f32x4 st[4][QS];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int u = 0; u < QS; ++u) st[kt][u] = kt < nsub ? *reinterpret_cast<const f32x4*>(dsrow[u] + (int64_t)(4 * kt0 + kt) * p.npad * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kt >= nsub) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < DC; ++c) {
                    const float kfs = Kt[(16 * kt + 4 * lg + r) * RS + 16 * c + lr];
#pragma unroll
                    for (int u = 0; u < QS; ++u) dq[u][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(kfs, st[kt][u][r], dq[u][c], 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int u = 0; u < QS; ++u) {
        const int qi = (strip0 + u) * 16 + lr;
        if (qi < p.N) {
            T* dqkv = reinterpret_cast<T*>(p.dqkv);
            const int64_t qtok = fl_token(p, origin, qi);
#pragma unroll
            for (int c = 0; c < DC; ++c) Ld4<T>::store(dqkv + qtok * p.ld + p.q_off + hc + 16 * c + 4 * lg, dq[u][c] * p.scale);
        }
    }
    if (!RES) break;
    }
}

// ------------------------------------------------------------------------------------------- backward, pass B: dK, dV (+ dbias)
// -delta[q] = -sum_d dO[q][d] O[q][d] and -lse[q] / scale of rows [t0, t0+n) into LDS: 4 lanes per row
template <typename T, int D>
__device__ __forceinline__ void fl_stage_rowstats(const FlashParams& p, const T* dout, const T* outp, int64_t origin, int h, int t0, int n,
                                                  float* lse_s, float* del_s) {
    for (int base = 0; base < n; base += blockDim.x >> 2) {
        const int row = base + (threadIdx.x >> 2), part = threadIdx.x & 3;
        float dsum = 0.f;
        const bool ok = t0 + row < p.N;
        const int64_t tok = fl_token(p, origin, min(t0 + row, p.N - 1));
        if (ok) {
#pragma unroll
            for (int c = 0; c < D / 16; ++c) {
                const f32x4 a = Ld4<T>::load(dout + tok * p.lddo + h * D + part * (D / 4) + 4 * c);
                const f32x4 b = Ld4<T>::load(outp + tok * p.ldo + h * D + part * (D / 4) + 4 * c);
#pragma unroll
                for (int s = 0; s < 4; ++s) dsum = fmaf(a[s], b[s], dsum);
            }
        }
        dsum += __shfl_xor(dsum, 1, 64);
        dsum += __shfl_xor(dsum, 2, 64);
        // stored negated and in the units of the raw products: they become the INITIAL VALUES of the S / dP accumulators, so that "- lse" and
        // "- delta" cost no VALU instruction per score (fp32 MFMA and VALU share the SIMD's issue: tools/mfma_shadow.hip).  -inf for padded queries: P = 0
        if (part == 0 && row < n) { del_s[row] = -dsum; lse_s[row] = ok ? -p.lse[tok * p.nh + h] / p.scale : -INFINITY; }
    }
}
template <typename T, int D, bool DBIAS, bool RES, bool STORE_DS = false>
__global__ __launch_bounds__(RES ? 1024 : 256) void flash_bwd_dkv_kernel(FlashParams p) {
    constexpr int RS = D + 4, DC = D / 16;
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    const int R = RES ? p.npad : 64;
    float* Qs = fsm;
    float* Os = Qs + R * RS;                                       // dO image
    float* btab = Os + R * RS;
    float* dbt = btab + p.nbpad;
    float* lse_s = dbt + (DBIAS ? p.nbpad : 0);
    float* del_s = lse_s + R;
    int* qlin = reinterpret_cast<int*>(del_s + R);                 // per staged query: byte-scaled linear coordinate + the table's centre
    const int kvt = RES ? 0 : blockIdx.x % p.ntile;
    const int wh = RES ? blockIdx.x : blockIdx.x / p.ntile;
    const int h = wh % p.nh, w = wh / p.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int64_t origin = fl_origin(p, w);
    const T* qkv = reinterpret_cast<const T*>(p.qkv);
    const T* dout = reinterpret_cast<const T*>(p.dout);
    const T* outp = reinterpret_cast<const T*>(p.out);
    const int hc = h * p.head_stride;
    const bool has_bias = p.bias_table != nullptr;
    const int nb = p.ws * p.ws;
    const int w2 = 2 * p.ws - 1;
    if (has_bias) fl_stage_bias(p, h, btab);
    if (DBIAS) for (int i = threadIdx.x; i < w2 * w2; i += blockDim.x) dbt[i] = 0.f;      // bins in the expanded (signed-offset) index space
    if (RES) {
        fl_stage_all<T, D>(p, qkv, p.ld, p.q_off + hc, origin, Qs);
        fl_stage_all<T, D>(p, dout, p.lddo, h * D, origin, Os);
        if (has_bias) fl_stage_coords(p, 0, qlin, p.npad);
        fl_stage_rowstats<T, D>(p, dout, outp, origin, h, 0, p.npad, lse_s, del_s);
        __syncthreads();
    }
    const int nstrips = (p.N + 15) >> 4;
    for (int strip = RES ? wave : kvt * 4 + wave; !RES || strip < nstrips; strip += (int)(blockDim.x >> 6)) {
    const int ki = strip * 16 + lr;
    const bool kok = ki < p.N;
    const bool wave_on = strip * 16 < p.N;
    const int64_t ktok = fl_token(p, origin, min(ki, p.N - 1));
    f32x4 kf[DC], vf[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) {
        kf[c] = vf[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (kok) {
            kf[c] = Ld4<T>::load(qkv + ktok * p.ld + p.k_off + hc + 16 * c + 4 * lg);
            vf[c] = Ld4<T>::load(qkv + ktok * p.ld + p.v_off + hc + 16 * c + 4 * lg);
        }
    }
    const int kk = min(ki, p.N - 1);
    const int klin = has_bias ? fl_lin4(p, kk) - 4 * (p.ws - 1) * 2 * p.ws : 0;       // minus the table's centre: offset = qlin[q] - klin
    const float sc2 = p.scale * 1.4426950408889634f;
    f32x4 dk[DC], dv[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) dk[c] = dv[c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int qt0 = 0; qt0 < p.ntile; ++qt0) {
        const int t0 = qt0 * 64;
        if (!RES) {
            __syncthreads();
            fl_stage<T, D>(p, qkv, p.ld, p.q_off + hc, origin, t0, Qs);
            fl_stage<T, D>(p, dout, p.lddo, h * D, origin, t0, Os);
            if (has_bias) fl_stage_coords(p, t0, qlin);
            fl_stage_rowstats<T, D>(p, dout, outp, origin, h, t0, 64, lse_s, del_s);
            __syncthreads();
            if (!wave_on) continue;
        }
        const float* Qt = Qs + (RES ? t0 * RS : 0);
        const float* Ot = Os + (RES ? t0 * RS : 0);
        const int ro = RES ? t0 : 0;
        const int nsub = min(4, (p.N - t0 + 15) / 16);
#pragma unroll
        for (int qs = 0; qs < 4; ++qs) {
            if (qs >= nsub) continue;
            const int q4 = ro + 16 * qs + 4 * lg;                              // this lane's 4 consecutive queries: one 16-byte read per array
            f32x4 st = *reinterpret_cast<const f32x4*>(lse_s + q4), dp = *reinterpret_cast<const f32x4*>(del_s + q4);      // S - lse / scale, dP - delta
#pragma unroll
            for (int c = 0; c < DC; ++c) {
                const f32x4 qa = *reinterpret_cast<const f32x4*>(Qt + (16 * qs + lr) * RS + 16 * c + 4 * lg);
                const f32x4 oa = *reinterpret_cast<const f32x4*>(Ot + (16 * qs + lr) * RS + 16 * c + 4 * lg);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    st = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s], kf[c][s], st, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[s], vf[c][s], dp, 0, 0, 0);
                }
            }
            // lane holds S / dP [q = t0 + 16qs + 4lg + r][key = lr]
            f32x4 pr, ds;
            i32x4 boff4 = {0, 0, 0, 0};
            f32x4 bia = {0.f, 0.f, 0.f, 0.f};
            if (has_bias) {
                boff4 = *reinterpret_cast<const i32x4*>(qlin + q4) - klin;
#pragma unroll
                for (int r = 0; r < 4; ++r) bia[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(btab) + boff4[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql = 16 * qs + 4 * lg + r;
                const int boff = boff4[r];
                const float e = __builtin_amdgcn_exp2f(fmaf(st[r], sc2, bia[r]));     // -inf for padded queries -> 0; a padded key's column is never stored
                const float g = e * dp[r];
                pr[r] = e;
                ds[r] = g;                                                     // the softmax scale is applied once, to dK
                // (padded query rows hold exact zeros -- P = 0 --, padded key columns finite values that meet zero K rows in the dQ pass)
                if (STORE_DS) p.ds_scratch[(((int64_t)wh * (p.npad >> 4) + strip) * p.npad + (t0 + ql)) * 16 + lr] = g;      // [key strip][query][16 keys]: a 16 x 16 tile is 1 KB contiguous
                if (DBIAS && kok && t0 + ql < p.N) atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(dbt) + boff), g);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < DC; ++c) {
                    const float of = Ot[(16 * qs + 4 * lg + r) * RS + 16 * c + lr];
                    const float qf = Qt[(16 * qs + 4 * lg + r) * RS + 16 * c + lr];
                    dv[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(of, pr[r], dv[c], 0, 0, 0);
                    dk[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf, ds[r], dk[c], 0, 0, 0);
                }
        }
    }
    if (kok) {
        T* dqkv = reinterpret_cast<T*>(p.dqkv);
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            Ld4<T>::store(dqkv + ktok * p.ld + p.k_off + hc + 16 * c + 4 * lg, dk[c] * p.scale);
            Ld4<T>::store(dqkv + ktok * p.ld + p.v_off + hc + 16 * c + 4 * lg, dv[c]);
        }
    }
    if (!RES) break;
    }
    if (DBIAS) {
        __syncthreads();
        const int prow = RES ? w : w * p.ntile + kvt;
        for (int i = threadIdx.x; i < nb; i += blockDim.x) {      // fold the signed offsets back onto attention_biases[|dy| * ws + |dx|]
            const int ady = i / p.ws, adx = i - ady * p.ws, c0 = (p.ws - 1) * w2 + (p.ws - 1);
            float t = dbt[c0 + ady * w2 + adx];
            if (ady) t += dbt[c0 - ady * w2 + adx];
            if (adx) t += dbt[c0 + ady * w2 - adx];
            if (ady && adx) t += dbt[c0 - ady * w2 - adx];
            if (p.dbias_part) p.dbias_part[((int64_t)prow * p.nh + h) * nb + i] = t;
            else atomicAdd(&p.dbias[h * nb + i], t);
        }
    }
    if (STORE_DS && RES) {
        // second phase of the resident form: this workgroup wrote the whole dS of its (window, head); once every wave's stores are visible the same
        // workgroup turns it into dQ (one strip per wave) -- the dS it reads back is still in L2 / Infinity Cache, and the separate dQ launch is gone
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        float* Ks = Qs;                                             // the Q image is dead: K takes its place
        fl_stage_all<T, D>(p, qkv, p.ld, p.k_off + hc, origin, Ks);
        __syncthreads();
        const float* dsb = p.ds_scratch + (int64_t)wh * p.npad * p.npad;
        for (int strip = wave; strip < nstrips; strip += (int)(blockDim.x >> 6))
            fl_dq_from_ds_resident<T, D, 1>(p, Ks, dsb, strip, origin, hc, lr, lg);
    }
}

__global__ void flash_dbias_final_kernel(const float* __restrict__ rows, int nrows, int W, float* __restrict__ dbias) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W) return;
    double s = 0.0;
    for (int r = 0; r < nrows; ++r) s += (double)rows[(int64_t)r * W + i];
    dbias[i] += (float)s;
}

int flash_fill(FlashParams& p, const GgAttnArgs* a, int dtype, const char* who) {
    GG_CHECK(a && a->qkv, "%s: null qkv", who);
    GG_CHECK(dtype == 0 || dtype == 1 || dtype == 2, "%s: dtype must be 0 (bf16), 1 (f32) or 2 (fp16, forward only)", who);
    GG_CHECK(a->head_dim == 32 || a->head_dim == 64, "%s: head_dim must be 32 or 64 (got %d)", who, a->head_dim);
    GG_CHECK(a->tokens_per_window > 0 && a->num_windows > 0 && a->num_heads > 0, "%s: bad window/head/token count", who);
    GG_CHECK((a->ld & 3) == 0 && (a->q_off & 3) == 0 && (a->k_off & 3) == 0 && (a->v_off & 3) == 0 && (a->head_stride & 3) == 0,
             "%s: qkv offsets/strides must be multiples of 4 elements", who);
    GG_CHECK(((uintptr_t)a->qkv & 15) == 0, "%s: qkv must be 16-byte aligned", who);
    if (a->window_size > 0) {
        GG_CHECK(a->window_size * a->window_size == a->tokens_per_window, "%s: window_size^2 != tokens_per_window", who);
        GG_CHECK(a->map_h % a->window_size == 0 && a->map_w % a->window_size == 0, "%s: map not divisible by window", who);
        GG_CHECK(a->num_windows % ((a->map_h / a->window_size) * (a->map_w / a->window_size)) == 0, "%s: window count", who);
        GG_CHECK(a->window_size <= 32, "%s: window_size > 32 unsupported (bias table of at most 1024 entries)", who);
    }
    if (a->bias_table || a->dbias) GG_CHECK(a->window_size > 0, "%s: the relative-position bias needs a window geometry", who);
    GG_CHECK(!a->dbias || a->bias_table, "%s: dbias without bias_table", who);
    GG_CHECK((int64_t)a->num_windows * a->num_heads * gg_cdiv(a->tokens_per_window, 64) < ((int64_t)1 << 31), "%s: grid too large", who);
    p.qkv = a->qkv; p.ld = a->ld; p.q_off = a->q_off; p.k_off = a->k_off; p.v_off = a->v_off; p.head_stride = a->head_stride;
    p.out = a->out; p.ldo = a->ldo; p.bias_table = a->bias_table;
    p.ws = a->window_size; p.H = a->map_h; p.W = a->map_w;
    p.nWx = a->window_size ? a->map_w / a->window_size : 1;
    p.nWy = a->window_size ? a->map_h / a->window_size : 1;
    p.N = a->tokens_per_window; p.nh = a->num_heads; p.scale = a->scale;
    p.ds_scratch = a->ds_scratch;
    p.dout = a->dout; p.lddo = a->lddo; p.dqkv = a->dqkv; p.dbias = a->dbias; p.dbias_part = a->dbias ? a->dbias_scratch : nullptr; p.lse = a->lse;
    p.ntile = (int)gg_cdiv(a->tokens_per_window, 64);
    p.npad = (int)gg_align(a->tokens_per_window, 16);
    p.nbpad = (int)gg_align(std::max(4, (2 * a->window_size - 1) * (2 * a->window_size - 1)), 4);      // expanded (signed-offset) bias table
    return 0;
}
// dynamic LDS of the forward / dQ kernels (two operand images, bias table, coordinates) and of the dK/dV kernel (+ bias-gradient bins,
// lse, delta) for R staged rows
size_t flash_lds_fwd(const FlashParams& p, int D, int R) { return ((size_t)2 * R * (D + 4) + p.nbpad + R) * 4; }
size_t flash_lds_dkv(const FlashParams& p, int D, int R, bool dbias) { return ((size_t)2 * R * (D + 4) + p.nbpad * (dbias ? 2 : 1) + 3 * R) * 4; }
// resident form: more than one 64-row tile (a single tile is already staged once) and two workgroups still fit a CU's 160 KB
bool flash_resident(const FlashParams& p, int D, bool dbias) {
    static const bool off = gg_dev_env("GG_ATTN_FLASH_NO_RES") != nullptr;
    return !off && p.ntile > 1 && flash_lds_dkv(p, D, p.npad, dbias) <= 64 * 1024;
}

}  // namespace

extern "C" int gg_attention_flash_fwd(const GgAttnArgs* a, int dtype, void* stream) {
    FlashParams p;
    GG_TRY(flash_fill(p, a, dtype, "gg_attention_flash_fwd"));
    GG_CHECK(a->out && (a->ldo & 3) == 0 && ((uintptr_t)a->out & 15) == 0, "gg_attention_flash_fwd: bad out");
    const bool res = flash_resident(p, a->head_dim, false);
    const dim3 grid((unsigned)(a->num_windows * a->num_heads * (res ? 1 : p.ntile))), block(res ? 64 * std::min(16, p.npad / 16) : 256);
    const size_t lds = flash_lds_fwd(p, a->head_dim, res ? p.npad : 64);
    hipStream_t s = (hipStream_t)stream;
    const double es = dtype == 1 ? 4.0 : 2.0;
    GG_PROF(GG_CAT_ATTN, 4.0 * a->num_windows * a->num_heads * (double)p.N * p.N * a->head_dim,
            4.0 * es * a->num_windows * a->num_heads * (double)p.N * a->head_dim, stream);
#define GG_FL_FWD(T_, D_)                                                                                     \
    do {                                                                                                      \
        if (res) hipLaunchKernelGGL((flash_fwd_kernel<T_, D_, true>), grid, block, lds, s, p);                \
        else hipLaunchKernelGGL((flash_fwd_kernel<T_, D_, false>), grid, block, lds, s, p);                   \
    } while (0)
    if (dtype == 1) { if (a->head_dim == 32) GG_FL_FWD(float, 32); else GG_FL_FWD(float, 64); }
    else if (dtype == 2) { if (a->head_dim == 32) GG_FL_FWD(f16, 32); else GG_FL_FWD(f16, 64); }
    else { if (a->head_dim == 32) GG_FL_FWD(bf16, 32); else GG_FL_FWD(bf16, 64); }
#undef GG_FL_FWD
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int64_t gg_attention_flash_ds_scratch_floats(int num_windows, int num_heads, int tokens_per_window) {
    const int64_t npad = gg_align(tokens_per_window, 16);
    return (int64_t)num_windows * num_heads * npad * npad;
}
extern "C" int64_t gg_attention_flash_dbias_rows(int num_windows, int tokens_per_window) {
    return (int64_t)num_windows * gg_cdiv(tokens_per_window, 64) + GG_REDUCE_SLICES;
}
extern "C" int gg_attention_flash_bwd(const GgAttnArgs* a, int dtype, void* stream) {
    FlashParams p;
    GG_CHECK(dtype != 2, "gg_attention_flash_bwd: fp16 storage is inference-only");
    GG_TRY(flash_fill(p, a, dtype, "gg_attention_flash_bwd"));
    GG_CHECK(a->dout && a->dqkv && (a->lddo & 3) == 0 && ((uintptr_t)a->dout & 15) == 0 && ((uintptr_t)a->dqkv & 15) == 0,
             "gg_attention_flash_bwd: bad dout/dqkv");
    GG_CHECK(a->lse && a->out && (a->ldo & 3) == 0, "gg_attention_flash_bwd: needs the forward's lse and out");
    // resident form also for single-tile windows (7 x 7) when a dS scratch is there: the dQ phase then runs inside the dK/dV kernel
    const bool res = flash_resident(p, a->head_dim, p.dbias != nullptr) || (p.ds_scratch != nullptr && p.ntile == 1);
    const dim3 grid((unsigned)(a->num_windows * a->num_heads * (res ? 1 : p.ntile))), block(res ? 64 * std::min(16, p.npad / 16) : 256);
    const int R = res ? p.npad : 64;
    const size_t lds_q = flash_lds_fwd(p, a->head_dim, R), lds_kv = flash_lds_dkv(p, a->head_dim, R, p.dbias != nullptr);
    hipStream_t s = (hipStream_t)stream;
    const double es = dtype ? 4.0 : 2.0;
    // algorithmic: 5 products (S, dP, dV, dK, dQ) = 10 N^2 D, whatever the pass structure recomputes (attention.hip declares the same)
    GG_PROF(GG_CAT_ATTN, 10.0 * a->num_windows * a->num_heads * (double)p.N * p.N * a->head_dim,
            8.0 * es * a->num_windows * a->num_heads * (double)p.N * a->head_dim, stream);
    // with a dS scratch: dK/dV pass first (stores dS), then the one-product dQ pass
    const size_t lds_k = ((size_t)R * (a->head_dim + 4)) * 4;
#define GG_FL_BWD_DS(T_, D_, R_)                                                                              \
    do {                                                                                                      \
        if (p.dbias) hipLaunchKernelGGL((flash_bwd_dkv_kernel<T_, D_, true, R_, true>), grid, block, lds_kv, s, p); \
        else hipLaunchKernelGGL((flash_bwd_dkv_kernel<T_, D_, false, R_, true>), grid, block, lds_kv, s, p);  \
        if (!R_) hipLaunchKernelGGL((flash_bwd_dq_ds_kernel<T_, D_, false, 1>), grid, block, lds_k, s, p);    /* (resident form: second phase of the kernel above) */ \
    } while (0)
#define GG_FL_BWD2(T_, D_, R_)                                                                                \
    do {                                                                                                      \
        if (p.ds_scratch) { GG_FL_BWD_DS(T_, D_, R_); break; }                                                \
        if (R_ && D_ == 32 && !gg_dev_env("GG_ATTN_DQ_QS1")) hipLaunchKernelGGL((flash_bwd_dq_kernel<T_, D_, R_, 2>), grid, dim3(64 * ((p.npad / 16 + 1) / 2)), lds_q, s, p); \
        else hipLaunchKernelGGL((flash_bwd_dq_kernel<T_, D_, R_>), grid, block, lds_q, s, p);                 \
        if (p.dbias) hipLaunchKernelGGL((flash_bwd_dkv_kernel<T_, D_, true, R_>), grid, block, lds_kv, s, p); \
        else hipLaunchKernelGGL((flash_bwd_dkv_kernel<T_, D_, false, R_>), grid, block, lds_kv, s, p);        \
    } while (0)
#define GG_FL_BWD(T_, D_) do { if (res) GG_FL_BWD2(T_, D_, true); else GG_FL_BWD2(T_, D_, false); } while (0)
    if (dtype == 1) { if (a->head_dim == 32) GG_FL_BWD(float, 32); else GG_FL_BWD(float, 64); }
    else { if (a->head_dim == 32) GG_FL_BWD(bf16, 32); else GG_FL_BWD(bf16, 64); }
#undef GG_FL_BWD
#undef GG_FL_BWD2
#undef GG_FL_BWD_DS
    if (p.dbias && p.dbias_part) {
        const int Wd = p.nh * p.ws * p.ws;
        const float* rows; int nrows;
        gg_reduce_rows(p.dbias_part, a->num_windows * (res ? 1 : p.ntile), Wd, s, &rows, &nrows);
        hipLaunchKernelGGL(flash_dbias_final_kernel, dim3((unsigned)gg_cdiv(Wd, 256)), dim3(256), 0, s, rows, nrows, Wd, p.dbias);
    }
    GG_LAUNCH_CHECK();
    return 0;
}
