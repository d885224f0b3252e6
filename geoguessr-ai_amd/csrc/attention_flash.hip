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
    uint32_t rcp_ws, rcp_w2;          // ceil(2^20 / ws), ceil(2^20 / (2 ws - 1)): t / ws == (t * rcp_ws) >> 20 for every index these kernels form (checked on the host)
    uint32_t rcp_img, rcp_nwx;        // ceil(2^32 / windows per image), ceil(2^32 / windows per row) (0: divisor 1): window -> (image, row, column) in scalar arithmetic
    float neg_inv_scale;              // -1 / scale
    int round_bias;                   // bf16 storage, expanded bias table also given: the forward was attention.hip's (bias added as bf16(bias / scale)); the backward recomputes P with the same value
};
// integer division by a launch constant without the ~30-instruction software divide (fp32 MFMA and VALU share the SIMD's issue: the divides of the
// staging loops were most of the vector instructions of a 7 x 7 window's workgroup)
__device__ __forceinline__ int fl_div(int t, uint32_t rcp) { return (int)(((uint32_t)t * rcp) >> 20); }

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
    const int b = p.rcp_img ? (int)__umulhi((unsigned)w, p.rcp_img) : w, r = w - b * per_img;
    const int wy = p.rcp_nwx ? (int)__umulhi((unsigned)r, p.rcp_nwx) : r, wx = r - wy * p.nWx;
    return ((int64_t)b * p.H + wy * p.ws) * p.W + wx * p.ws;
}
__device__ __forceinline__ int64_t fl_token(const FlashParams& p, int64_t origin, int t) {      // t < N
    if (p.ws == 0) return origin + t;
    const int i = fl_div(t, p.rcp_ws), j = t - i * p.ws;
    return origin + (i * p.W + j);
}

// ---- window-relative addressing through buffer descriptors (resident kernels) ----------------------------------------------------------------
// A window's tokens lie within (ws - 1) * W + ws rows of its first one, so every tensor is addressed as descriptor(base + origin * ld) + a 32-bit
// byte offset: no 64-bit address arithmetic per element (2-4 vector instructions each; fp32 MFMA and VALU share the SIMD's issue), rows beyond the
// window come back as zeros from the range check (offset FL_OOB) and stores to them are dropped -- no exec-mask branches.
#define FL_OOB 0x7FFFFFF0
typedef unsigned int fl_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int fl_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int fl_tokrel(const FlashParams& p, int t) {       // t < N: element-row index relative to the window's first token
    if (p.ws == 0) return t;
    const int i = fl_div(t, p.rcp_ws);
    return i * p.W + (t - i * p.ws);
}
__device__ __forceinline__ int fl_span(const FlashParams& p) { return p.ws == 0 ? p.N : (p.ws - 1) * p.W + p.ws; }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fl_rsrc(const void* ptr, int bytes) {
    const unsigned long long a = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
template <typename T> struct Bld;          // 4 consecutive elements at a byte offset of a descriptor, widened to f32
template <> struct Bld<float> {
    static __device__ __forceinline__ f32x4 load(__amdgpu_buffer_rsrc_t rs, int off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0)); }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t rs, int off, f32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fl_u32x4, v), rs, off, 0, 0); }
};
template <> struct Bld<bf16> {
    static __device__ __forceinline__ f32x4 load(__amdgpu_buffer_rsrc_t rs, int off) {
        const bf16x4 v = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
        return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t rs, int off, f32x4 v) {
        const bf16x4 b = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fl_u32x2, b), rs, off, 0, 0);
    }
};
template <> struct Bld<f16> {
    static __device__ __forceinline__ f32x4 load(__amdgpu_buffer_rsrc_t rs, int off) {
        const f16x4 v = __builtin_bit_cast(f16x4, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
        return (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
    static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t rs, int off, f32x4 v) {
        const f16x4 b = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fl_u32x2, b), rs, off, 0, 0);
    }
};
// all R rows (16-row tiles) of one head's column block into X[.][D+4] (f32; zero rows beyond N); colb = byte offset of the block in a row, ldb = row bytes
template <typename T, int D>
__device__ __forceinline__ void fl_stage_all_b(const FlashParams& p, __amdgpu_buffer_rsrc_t rs, int ldb, int colb, float* X, int R) {
    constexpr int CH = D / 4, RS = D + 4;
    for (int id = threadIdx.x; id < R * CH; id += blockDim.x) {
        const int row = id / CH, ch = id % CH;
        const int off = row < p.N ? fl_tokrel(p, row) * ldb + colb + ch * 4 * (int)sizeof(T) : FL_OOB;
        *reinterpret_cast<f32x4*>(X + row * RS + ch * 4) = Bld<T>::load(rs, off);
    }
}
// -delta[q] and -lse[q] / scale of all R rows (see fl_stage_rowstats), 4 lanes per row, through descriptors
template <typename T, int D>
__device__ __forceinline__ void fl_stage_rowstats_b(const FlashParams& p, __amdgpu_buffer_rsrc_t rsdo, int lddob, __amdgpu_buffer_rsrc_t rso, int ldob, __amdgpu_buffer_rsrc_t rslse,
                                                    int h, int R, float* lse_s, float* del_s) {
    for (int base = 0; base < R; base += blockDim.x >> 2) {
        const int row = base + (threadIdx.x >> 2), part = threadIdx.x & 3;
        const bool ok = row < p.N;
        const int rel = ok ? fl_tokrel(p, row) : 0;
        const int cb = (h * D + part * (D / 4)) * (int)sizeof(T);
        float dsum = 0.f;
#pragma unroll
        for (int c = 0; c < D / 16; ++c) {
            const f32x4 a = Bld<T>::load(rsdo, ok ? rel * lddob + cb + c * 4 * (int)sizeof(T) : FL_OOB);
            const f32x4 b = Bld<T>::load(rso, ok ? rel * ldob + cb + c * 4 * (int)sizeof(T) : FL_OOB);
#pragma unroll
            for (int s = 0; s < 4; ++s) dsum = fmaf(a[s], b[s], dsum);
        }
        dsum += __shfl_xor(dsum, 1, 64);
        dsum += __shfl_xor(dsum, 2, 64);
        if (part == 0 && row < R) {
            const float l = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rslse, ok ? (rel * p.nh + h) * 4 : FL_OOB, 0, 0));
            del_s[row] = -dsum;
            lse_s[row] = ok ? -1.4426950408889634f * l : -INFINITY;      // -lse in the exp2 domain: added to the exponent with the bias (a score accumulator seeded
                                                                         // with -lse / scale, magnitude ~170, rounds every partial product to the ulp of 170: twice the error)
        }
    }
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
        const int iy = fl_div(i, p.rcp_w2);
        const int dy = iy - (p.ws - 1), dx = i - iy * w2 - (p.ws - 1);
        E[i] = p.bias_table[h * p.ws * p.ws + abs(dy) * p.ws + abs(dx)] * 1.4426950408889634f;
    }
}
__device__ __forceinline__ int fl_lin4(const FlashParams& p, int t) {       // t < N
    const int r = fl_div(t, p.rcp_ws);
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
    // resident form: descriptors based at the window's first token, 32-bit offsets (see fl_rsrc)
    constexpr int ES = (int)sizeof(T);
    const int ldb = (int)p.ld * ES, ldob = (int)p.ldo * ES;
    __amdgpu_buffer_rsrc_t rsQKV, rsOUT, rsLSE;
    if (RES) {
        const int span = fl_span(p);
        rsQKV = fl_rsrc(qkv + origin * p.ld, span * ldb);
        rsOUT = fl_rsrc(reinterpret_cast<T*>(p.out) + origin * p.ldo, span * ldob);
        rsLSE = fl_rsrc(p.lse ? p.lse + origin * p.nh : nullptr, p.lse ? span * p.nh * 4 : 0);
        fl_stage_all_b<T, D>(p, rsQKV, ldb, (p.k_off + hc) * ES, Ks, p.npad);
        fl_stage_all_b<T, D>(p, rsQKV, ldb, (p.v_off + hc) * ES, Vs, p.npad);
        if (has_bias) fl_stage_coords(p, 0, klin, p.npad);
        __syncthreads();
    }
    const int nstrips = (p.N + 15) >> 4;
    // SIMD balance of the resident form: 13 strips (14 x 14 windows) on 13 waves put a fourth wave on one SIMD.  Launched with one wave fewer
    // than strips (TAIL), the owner loop covers strips < waves and the last strip is split by KEY tiles over all waves afterwards (below)
    const int nwaves = (int)(blockDim.x >> 6);
    const bool tail = RES && nwaves < nstrips;
    const int nown = tail ? nwaves : nstrips;
    for (int strip = RES ? wave : qt * 4 + wave; !RES || strip < nown; strip += nwaves) {
    const int qi = strip * 16 + lr;
    const bool qok = qi < p.N;
    const bool wave_on = strip * 16 < p.N;                          // wave-uniform
    const int64_t qtok = RES ? 0 : fl_token(p, origin, min(qi, p.N - 1));
    const int qrel = (RES && qok) ? fl_tokrel(p, qi) : 0;
    f32x4 qf[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) {
        if (RES) qf[c] = Bld<T>::load(rsQKV, qok ? qrel * ldb + (p.q_off + hc + 16 * c + 4 * lg) * ES : FL_OOB);
        else {
            qf[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (qok) qf[c] = Ld4<T>::load(qkv + qtok * p.ld + p.q_off + hc + 16 * c + 4 * lg);
        }
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
    if (RES) {
        const float inv = 1.0f / l;
#pragma unroll
        for (int c = 0; c < DC; ++c) Bld<T>::store(rsOUT, qok ? qrel * ldob + (h * D + 16 * c + 4 * lg) * ES : FL_OOB, oacc[c] * inv);
        if (p.lse) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m * 0.6931471805599453f + __logf(l)), rsLSE,
                                                         (qok && lg == 0) ? (qrel * p.nh + h) * 4 : FL_OOB, 0, 0);
    } else if (qok) {
        const float inv = 1.0f / l;
        T* out = reinterpret_cast<T*>(p.out);
#pragma unroll
        for (int c = 0; c < DC; ++c) Ld4<T>::store(out + qtok * p.ldo + h * D + 16 * c + 4 * lg, oacc[c] * inv);
        if (p.lse && lg == 0) p.lse[qtok * p.nh + h] = m * 0.6931471805599453f + __logf(l);       // natural-log lse (m is a base-2 exponent)
    }
    if (!RES) break;
    }
    if (RES && tail) {
        // cooperative tail strip: wave w takes key sub-tile w (+ waves) of the strip's queries, leaves (max, sum, unnormalised O) of its keys in LDS for
        // the live queries; wave 0 merges the partials in sub-tile order (flash-decoding style: fixed order, deterministic)
        const int ts = nstrips - 1, nlive = p.N - 16 * ts;
        float* part = reinterpret_cast<float*>(klin + p.npad);                  // [sub][q < nlive][lg][DC][4], then m[sub][nlive], l[sub][nlive]
        float* pm = part + nstrips * nlive * 16 * DC;
        float* pl = pm + nstrips * nlive;
        const int qi = ts * 16 + lr;
        const bool qok = qi < p.N;
        const int qrel = qok ? fl_tokrel(p, qi) : 0;
        f32x4 qf[DC];
#pragma unroll
        for (int c = 0; c < DC; ++c) qf[c] = Bld<T>::load(rsQKV, qok ? qrel * ldb + (p.q_off + hc + 16 * c + 4 * lg) * ES : FL_OOB);
        const int qlin = has_bias ? fl_lin4(p, min(qi, p.N - 1)) + 4 * (p.ws - 1) * 2 * p.ws : 0;
        const float sc2 = p.scale * 1.4426950408889634f;
        for (int sub = wave; sub < nstrips; sub += nwaves) {
            const float* Kt = Ks + sub * 16 * RS;
            const float* Vt = Vs + sub * 16 * RS;
            f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) if (16 * sub + 4 * lg + r >= p.N) st[r] = -INFINITY;
#pragma unroll
            for (int c = 0; c < DC; ++c) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(Kt + lr * RS + 16 * c + 4 * lg);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s2], qf[c][s2], st, 0, 0, 0);
            }
            f32x4 bia = {0.f, 0.f, 0.f, 0.f};
            if (has_bias) {
                const i32x4 kl4 = *reinterpret_cast<const i32x4*>(klin + 16 * sub + 4 * lg);
#pragma unroll
                for (int r = 0; r < 4; ++r) bia[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(btab) + (qlin - kl4[r]));
            }
            float tmax = -1e30f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { st[r] = fmaf(st[r], sc2, bia[r]); tmax = fmaxf(tmax, st[r]); }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            float ls = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) { st[r] = __builtin_amdgcn_exp2f(st[r] - tmax); ls += st[r]; }
            ls += __shfl_xor(ls, 16, 64);
            ls += __shfl_xor(ls, 32, 64);
            f32x4 oacc[DC];
#pragma unroll
            for (int c = 0; c < DC; ++c) oacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < DC; ++c) oacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vt[(4 * lg + r) * RS + 16 * c + lr], st[r], oacc[c], 0, 0, 0);
            if (lr < nlive) {
#pragma unroll
                for (int c = 0; c < DC; ++c) *reinterpret_cast<f32x4*>(part + (((sub * nlive + lr) * 4 + lg) * DC + c) * 4) = oacc[c];
                if (lg == 0) { pm[sub * nlive + lr] = tmax; pl[sub * nlive + lr] = ls; }
            }
        }
        __syncthreads();
        if (wave == 0) {
            float m = -1e30f;
            if (lr < nlive) for (int sub = 0; sub < nstrips; ++sub) m = fmaxf(m, pm[sub * nlive + lr]);
            float l = 0.f;
            f32x4 oacc[DC];
#pragma unroll
            for (int c = 0; c < DC; ++c) oacc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (lr < nlive)
                for (int sub = 0; sub < nstrips; ++sub) {
                    const float wgt = __builtin_amdgcn_exp2f(pm[sub * nlive + lr] - m);
                    l = fmaf(pl[sub * nlive + lr], wgt, l);
#pragma unroll
                    for (int c = 0; c < DC; ++c) oacc[c] += wgt * *reinterpret_cast<const f32x4*>(part + (((sub * nlive + lr) * 4 + lg) * DC + c) * 4);
                }
            const float inv = 1.0f / l;
#pragma unroll
            for (int c = 0; c < DC; ++c) Bld<T>::store(rsOUT, qok ? qrel * ldob + (h * D + 16 * c + 4 * lg) * ES : FL_OOB, oacc[c] * inv);
            if (p.lse) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, m * 0.6931471805599453f + __logf(l)), rsLSE,
                                                             (qok && lg == 0) ? (qrel * p.nh + h) * 4 : FL_OOB, 0, 0);
        }
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
        lse2[u] = -1.4426950408889634f * (qok[u] ? p.lse[qtok[u] * p.nh + h] : 0.f);          // -lse in the exp2 domain: joins the bias in the exponent (see fl_stage_rowstats_b)
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
            for (int u = 0; u < QS; ++u) { st[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[u] = (f32x4){-delta[u], -delta[u], -delta[u], -delta[u]}; }
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
                    const float pr = __builtin_amdgcn_exp2f(fmaf(st[u][r], sc2, bia[u][r] + lse2[u]));      // rows beyond N are never stored: no query mask needed in this pass
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
        if (part == 0 && row < n) { del_s[row] = -dsum; lse_s[row] = ok ? -1.4426950408889634f * p.lse[tok * p.nh + h] : -INFINITY; }      // (-lse, exp2 domain: see fl_stage_rowstats_b)
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
            const f32x4 nl4 = *reinterpret_cast<const f32x4*>(lse_s + q4);      // -lse (exp2 domain; -inf for padded queries)
            f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = *reinterpret_cast<const f32x4*>(del_s + q4);      // S, dP - delta
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
                const float e = __builtin_amdgcn_exp2f(fmaf(st[r], sc2, bia[r] + nl4[r]));     // -inf for padded queries -> 0; a padded key's column is never stored
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
            const int ady = fl_div(i, p.rcp_ws), adx = i - ady * p.ws, c0 = (p.ws - 1) * w2 + (p.ws - 1);
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

// ------------------------------------------------------------------------------------------- backward, single pass (resident windows)
// One workgroup per (window, head), one wave per 16-key strip j: its K / V rows stay in registers (row-contiguous as the B operand of S / dP, and K
// once more k-major as the A operand of the dQ product), Q and dO of the whole window sit in LDS.  For every 16-query tile i the wave forms
//   S = Q_i K_j^T, dP = dO_i V_j^T   (accumulators start at -lse / scale and -delta)        P = exp2(.), dS = P o dP
//   dV_j += P^T dO_i,  dK_j += dS^T Q_i                                                      (registers, as in the two-pass kernel)
//   dQ_i += dS K_j                                                                           (NEW: no dS ever leaves the CU)
// The dQ product contracts over the lane's COLUMN index of dS (C layout: row = query 4 lg + r, column = key lr), so the tile crosses a 1.25 KB LDS
// slot ([query][20 floats]: 4 conflict-free ds_write_b32, 1 ds_read_b128 = dS[q = lr][keys 4 lg ..]: the B operand of 4 MFMA steps), and the
// 16 x D partial is summed into an LDS image of dQ by making that image the MFMA's accumulator: 2 ds_read_b128 -> 8 MFMAs -> 2 ds_write_b128.
// The waves walk the query tiles STAGGERED (wave j is on tile (j + t) mod tiles at step t), so at any step every tile has at most one writer; one
// barrier per step orders step t's writes before step t + 1's reads: no atomics, the summation order of every dQ element is fixed.
// 5 products and one exponential per score; HBM traffic = q, k, v, dO, O in and dq, dk, dv out.
//
// SIMD balance.  A workgroup this size is alone on its CU, and 13 strips (14 x 14 windows: 196 = 12 x 16 + 4) on 13 waves put four waves on one
// SIMD and three on the others: every step then lasts four blocks.  With TAIL the workgroup has strips - 1 waves (12: three per SIMD); the last
// key strip is worked off after the main loop by all of them together: wave w forms P / dS of (tile w, last strip) and the dQ product, leaves both
// tiles in LDS, and after a barrier 2 D / 16 waves (one per accumulator tile, one per SIMD) contract them with dO / Q into dV / dK of the strip.
// LDS: 3 window images + bias table (+ gradient bins) + row scalars + two 1.25 KB slots per tile = 128 KB for 14 x 14 windows, 38 KB for 7 x 7.
template <typename T, int D, bool DBIAS, int NT>
__global__ __launch_bounds__(NT ? (NT < 4 ? 256 : 64 * NT) : 1024) void flash_bwd_fused_kernel(FlashParams p) {
    constexpr int RS = D + 4, DC = D / 16, TS = 20, ES = (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) float fsm[];
    const int R = NT ? 16 * NT : p.npad;                           // NT = tiles of the window as a compile-time constant (0: any): LDS offsets become immediates
    float* Qs = fsm;
    float* Os = Qs + R * RS;                                       // dO image
    float* dQs = Os + R * RS;                                      // dQ accumulator image (unscaled)
    float* btab = dQs + R * RS;
    float* dbt = btab + p.nbpad;
    float* lse_s = dbt + (DBIAS ? p.nbpad : 0);
    float* del_s = lse_s + R;
    int* qlin = reinterpret_cast<int*>(del_s + R);
    float* Gt = reinterpret_cast<float*>(qlin + R);                // dS slots [tiles][16][TS] (main loop: slot of wave w = tile slot w)
    float* Pt = Gt + (R >> 4) * 16 * TS;                           // P slots (tail phase only)
    const int wh = blockIdx.x;
    const int h = wh % p.nh, w = wh / p.nh;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lg = lane >> 4;
    const int64_t origin = fl_origin(p, w);
    const int hc = h * p.head_stride;
    const bool has_bias = p.bias_table != nullptr;
    const int nb = p.ws * p.ws;
    const int w2 = 2 * p.ws - 1;
    const int ntiles = R >> 4;
    const int NW = (int)(blockDim.x >> 6);                         // owner waves: ntiles, or ntiles - 1 with a cooperative tail strip
    // descriptors based at the window's first token: 32-bit offsets from here on
    const int span = fl_span(p);
    const int ldb = (int)p.ld * ES, lddob = (int)p.lddo * ES, ldob = (int)p.ldo * ES;
    const __amdgpu_buffer_rsrc_t rsQKV = fl_rsrc(reinterpret_cast<const T*>(p.qkv) + origin * p.ld, span * ldb);
    const __amdgpu_buffer_rsrc_t rsDO = fl_rsrc(reinterpret_cast<const T*>(p.dout) + origin * p.lddo, span * lddob);
    const __amdgpu_buffer_rsrc_t rsO = fl_rsrc(reinterpret_cast<const T*>(p.out) + origin * p.ldo, span * ldob);
    const __amdgpu_buffer_rsrc_t rsLSE = fl_rsrc(p.lse + origin * p.nh, span * p.nh * 4);
    const __amdgpu_buffer_rsrc_t rsDQKV = fl_rsrc(reinterpret_cast<T*>(p.dqkv) + origin * p.ld, span * ldb);
    if (has_bias) fl_stage_bias(p, h, btab);
    if (DBIAS) for (int i = threadIdx.x; i < w2 * w2; i += blockDim.x) dbt[i] = 0.f;
    fl_stage_all_b<T, D>(p, rsQKV, ldb, (p.q_off + hc) * ES, Qs, R);
    fl_stage_all_b<T, D>(p, rsDO, lddob, h * D * ES, Os, R);
    for (int i = threadIdx.x; i < R * RS / 4; i += blockDim.x) reinterpret_cast<f32x4*>(dQs)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (has_bias) fl_stage_coords(p, 0, qlin, R);
    fl_stage_rowstats_b<T, D>(p, rsDO, lddob, rsO, ldob, rsLSE, h, R, lse_s, del_s);
    const float sc2 = p.scale * 1.4426950408889634f;
    float* slot = Gt + wave * (16 * TS);

    f32x4 kf[DC], vf[DC];
    float kT[DC][4];
    int klin, krow;                                                // krow: byte offset of this lane's key row in the window (FL_OOB beyond N: loads 0, stores dropped)
    bool kok;
    // K / V rows of key strip `strip` (B operands of S / dP) and K once more k-major (A operand of the dQ product; transposed through the wave's slot)
    auto load_strip = [&](int strip) {
        const int ki = strip * 16 + lr;
        kok = ki < p.N;
        krow = kok ? fl_tokrel(p, ki) * ldb : FL_OOB;
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            kf[c] = Bld<T>::load(rsQKV, kok ? krow + (p.k_off + hc + 16 * c + 4 * lg) * ES : FL_OOB);
            vf[c] = Bld<T>::load(rsQKV, kok ? krow + (p.v_off + hc + 16 * c + 4 * lg) * ES : FL_OOB);
        }
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            *reinterpret_cast<f32x4*>(slot + lr * TS + 4 * lg) = kf[c];          // [key = lr][d = 16 c + 4 lg ..]
#pragma unroll
            for (int r = 0; r < 4; ++r) kT[c][r] = slot[(4 * lg + r) * TS + lr];   // [key = 4 lg + r][d = 16 c + lr]
        }
        klin = has_bias ? fl_lin4(p, min(ki, p.N - 1)) - 4 * (p.ws - 1) * 2 * p.ws : 0;
    };
    // S, dP, P, dS of (query tile, loaded key strip); lane holds [q = 16 tile + 4 lg + r][key = lr].  dS goes to `gslot` ([q][TS]), P to `pslot` if given
    auto scores = [&](int tile, f32x4& pr, f32x4& ds, float* gslot, float* pslot) {
        const float* Qt = Qs + tile * 16 * RS;
        const float* Ot = Os + tile * 16 * RS;
        const int q4 = tile * 16 + 4 * lg;
        const f32x4 nl4 = *reinterpret_cast<const f32x4*>(lse_s + q4);      // -lse (exp2 domain; -inf for padded queries)
        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = *reinterpret_cast<const f32x4*>(del_s + q4);      // S, dP - delta
#pragma unroll
        for (int c = 0; c < DC; ++c) {
            const f32x4 qa = *reinterpret_cast<const f32x4*>(Qt + lr * RS + 16 * c + 4 * lg);
            const f32x4 oa = *reinterpret_cast<const f32x4*>(Ot + lr * RS + 16 * c + 4 * lg);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                st = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[s], kf[c][s], st, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[s], vf[c][s], dp, 0, 0, 0);
            }
        }
        i32x4 boff4 = {0, 0, 0, 0};
        f32x4 bia = {0.f, 0.f, 0.f, 0.f};
        if (has_bias) {
            boff4 = *reinterpret_cast<const i32x4*>(qlin + q4) - klin;
#pragma unroll
            for (int r = 0; r < 4; ++r) bia[r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(btab) + boff4[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(st[r], sc2, bia[r] + nl4[r]));     // -inf for padded queries -> 0
            const float g = e * dp[r];
            pr[r] = e;
            ds[r] = g;                                                         // the softmax scale is applied once, to dK and dQ
            gslot[(4 * lg + r) * TS + lr] = g;                                 // (a padded key's column is finite and meets zero K rows in the dQ product)
            if (pslot) pslot[(4 * lg + r) * TS + lr] = e;
            if (DBIAS && kok && q4 + r < p.N) atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(dbt) + boff4[r]), g);
        }
    };
    // dQ^T[d][q] += sum_key K^T[d][key] dS^T[key][q]: B operand = dS[q = lr][key = 4 lg + r'] read back transposed from the slot; the LDS image is the accumulator
    auto dq_update = [&](int tile, const float* gslot) {
        const f32x4 dst = *reinterpret_cast<const f32x4*>(gslot + lr * TS + 4 * lg);
        float* dQt = dQs + (tile * 16 + lr) * RS + 4 * lg;
        f32x4 dq[DC];
#pragma unroll
        for (int c = 0; c < DC; ++c) dq[c] = *reinterpret_cast<const f32x4*>(dQt + 16 * c);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < DC; ++c) dq[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(kT[c][r], dst[r], dq[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < DC; ++c) *reinterpret_cast<f32x4*>(dQt + 16 * c) = dq[c];
    };

    load_strip(wave);
    f32x4 dk[DC], dv[DC];
#pragma unroll
    for (int c = 0; c < DC; ++c) dk[c] = dv[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    int tile = wave;
    for (int t = 0; t < ntiles; ++t) {
        f32x4 pr, ds;
        scores(tile, pr, ds, slot, nullptr);
        const float* Qt = Qs + tile * 16 * RS;
        const float* Ot = Os + tile * 16 * RS;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < DC; ++c) {
                const float of = Ot[(4 * lg + r) * RS + 16 * c + lr];
                const float qf = Qt[(4 * lg + r) * RS + 16 * c + lr];
                dv[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(of, pr[r], dv[c], 0, 0, 0);
                dk[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf, ds[r], dk[c], 0, 0, 0);
            }
        dq_update(tile, slot);
        __syncthreads();                                                       // the tile's next writer (the wave one strip down) reads after this
        tile = tile + 1 == ntiles ? 0 : tile + 1;
    }
#pragma unroll
    for (int c = 0; c < DC; ++c) {              // (rows beyond N carry FL_OOB: dropped by the range check)
        Bld<T>::store(rsDQKV, kok ? krow + (p.k_off + hc + 16 * c + 4 * lg) * ES : FL_OOB, dk[c] * p.scale);
        Bld<T>::store(rsDQKV, kok ? krow + (p.v_off + hc + 16 * c + 4 * lg) * ES : FL_OOB, dv[c]);
    }
    if (NW < ntiles) {
        // cooperative tail: key strip ntiles - 1.  Phase 1: wave w forms P / dS of (tile w [+ NW], tail strip), leaves them in the tile's slots and
        // adds its dQ product (tile w is this wave's alone in this phase)
        const int ts = ntiles - 1;
        load_strip(ts);
        for (int tl = wave; tl < ntiles; tl += NW) {
            f32x4 pr, ds;
            scores(tl, pr, ds, Gt + tl * (16 * TS), Pt + tl * (16 * TS));
            dq_update(tl, Gt + tl * (16 * TS));
        }
        __syncthreads();
        // Phase 2: accumulator tile u of the strip (dV column blocks, then dK column blocks) on wave u: contraction over every query of the window
        if (wave < 2 * DC) {
            const bool is_k = wave >= DC;
            const int c = is_k ? wave - DC : wave;
            const float* X = is_k ? Qs : Os;
            const float* Y = is_k ? Gt : Pt;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int tl = 0; tl < ntiles; ++tl) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(X[(tl * 16 + 4 * lg + r) * RS + 16 * c + lr], Y[tl * (16 * TS) + (4 * lg + r) * TS + lr], acc, 0, 0, 0);
            }
            Bld<T>::store(rsDQKV, kok ? krow + ((is_k ? p.k_off : p.v_off) + hc + 16 * c + 4 * lg) * ES : FL_OOB, is_k ? acc * p.scale : acc);
        }
    }
    constexpr int CH = D / 4;
    for (int id = threadIdx.x; id < p.N * CH; id += blockDim.x) {
        const int row = id / CH, ch = id % CH;
        const f32x4 v = *reinterpret_cast<const f32x4*>(dQs + row * RS + ch * 4);
        Bld<T>::store(rsDQKV, fl_tokrel(p, row) * ldb + (p.q_off + hc + ch * 4) * ES, v * p.scale);
    }
    if (DBIAS) {
        for (int i = threadIdx.x; i < nb; i += blockDim.x) {      // fold the signed offsets back onto attention_biases[|dy| * ws + |dx|]
            const int ady = fl_div(i, p.rcp_ws), adx = i - ady * p.ws, c0 = (p.ws - 1) * w2 + (p.ws - 1);
            float tt = dbt[c0 + ady * w2 + adx];
            if (ady) tt += dbt[c0 - ady * w2 + adx];
            if (adx) tt += dbt[c0 + ady * w2 - adx];
            if (ady && adx) tt += dbt[c0 - ady * w2 - adx];
            if (p.dbias_part) p.dbias_part[((int64_t)w * p.nh + h) * nb + i] = tt;
            else atomicAdd(&p.dbias[h * nb + i], tt);
        }
    }
}

#include "attention_split.h"

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
    p.rcp_ws = p.rcp_w2 = p.rcp_img = p.rcp_nwx = 0;
    p.neg_inv_scale = -1.0f / p.scale;
    p.round_bias = 0;                                     // (set by gg_attention_flash_bwd_impl when the forward was attention.hip's)
    {
        auto magic = [](int d) { return d <= 1 ? 0u : (uint32_t)((((uint64_t)1 << 32) + d - 1) / d); };      // exact for numerators < 2^32 / d
        p.rcp_img = magic(p.nWx * p.nWy); p.rcp_nwx = magic(p.nWx);
        GG_CHECK((int64_t)a->num_windows * (p.nWx * p.nWy) < ((int64_t)1 << 31), "%s: too many windows", who);
    }
    if (p.ws > 0) {
        const int w2 = 2 * p.ws - 1;
        p.rcp_ws = (uint32_t)(((1u << 20) + p.ws - 1) / p.ws);
        p.rcp_w2 = (uint32_t)(((1u << 20) + w2 - 1) / w2);
        for (int t = 0; t < p.npad + 64; ++t) GG_CHECK((int)(((uint32_t)t * p.rcp_ws) >> 20) == t / p.ws, "%s: reciprocal division fails at %d / %d", who, t, p.ws);
        for (int t = 0; t < w2 * w2; ++t) GG_CHECK((int)(((uint32_t)t * p.rcp_w2) >> 20) == t / w2, "%s: reciprocal division fails at %d / %d", who, t, w2);
    }
    return 0;
}
// dynamic LDS of the forward / dQ kernels (two operand images, bias table, coordinates) and of the dK/dV kernel (+ bias-gradient bins,
// lse, delta) for R staged rows
size_t flash_lds_fwd(const FlashParams& p, int D, int R) { return ((size_t)2 * R * (D + 4) + p.nbpad + R) * 4; }
// resident forward with a cooperative tail strip (strips = 4 n + 1: one wave fewer than strips): partial (O, max, sum) of every key sub-tile for the strip's live queries
bool flash_fwd_tail(const FlashParams& p) {
    static const bool notail = gg_dev_env("GG_ATTN_FWD_NO_TAIL") != nullptr;
    const int strips = p.npad / 16;
    return !notail && strips > 4 && strips <= 17 && strips % 4 == 1;
}
size_t flash_lds_fwd_tail(const FlashParams& p, int D) { const int strips = p.npad / 16, nlive = p.N - 16 * (strips - 1); return (size_t)strips * nlive * (D + 2) * 4; }
size_t flash_lds_dkv(const FlashParams& p, int D, int R, bool dbias) { return ((size_t)2 * R * (D + 4) + p.nbpad * (dbias ? 2 : 1) + 3 * R) * 4; }
// single-pass backward: three window images, bias table (+ bins), row scalars, one 16 x 20 slot per wave
size_t flash_lds_fused(const FlashParams& p, int D, bool dbias) { return ((size_t)3 * p.npad * (D + 4) + p.nbpad * (dbias ? 2 : 1) + 3 * p.npad + 2 * (p.npad / 16) * 320) * 4; }
// owner waves of the single-pass kernel: one per key strip, or -- when that would leave one SIMD with an extra wave (strips = 4 n + 1, e.g. the 13 strips
// of a 14 x 14 window) -- one fewer, the last strip then being the cooperative tail
int flash_fused_waves(const FlashParams& p, int D) {
    static const bool notail = gg_dev_env("GG_ATTN_FUSED_NO_TAIL") != nullptr;
    const int strips = p.npad / 16;
    return (!notail && strips > 4 && strips % 4 == 1 && strips - 1 >= 2 * (D / 16)) ? strips - 1 : strips;
}
bool flash_fused_ok(const FlashParams& p, int D, bool dbias) {
    static const bool off = gg_dev_env("GG_ATTN_NO_FUSED_BWD") != nullptr;
    return !off && p.npad <= 256 && flash_lds_fused(p, D, dbias) <= 160 * 1024;
}
// resident form: more than one 64-row tile (a single tile is already staged once) and two workgroups still fit a CU's 160 KB
bool flash_resident(const FlashParams& p, int D, bool dbias) {
    static const bool off = gg_dev_env("GG_ATTN_FLASH_NO_RES") != nullptr;
    return !off && flash_lds_dkv(p, D, p.npad, dbias) <= 64 * 1024;      // (single-tile windows too: same staging, descriptor addressing)
}

}  // namespace

extern "C" int gg_attention_flash_fwd(const GgAttnArgs* a, int dtype, void* stream) {
    FlashParams p;
    GG_TRY(flash_fill(p, a, dtype, "gg_attention_flash_fwd"));
    GG_CHECK(a->out && (a->ldo & 3) == 0 && ((uintptr_t)a->out & 15) == 0, "gg_attention_flash_fwd: bad out");
    // fp32 storage, head dim 32, windows of 4 / 9 / 13 tiles (7 x 7, 12 x 12, 14 x 14): the products run as split-bf16 MFMAs (attention_split.h).  (The same
    // kernel instantiated for bf16 storage -- one plane -- measured 403 us per 14 x 14 layer with a wave per strip and 497 us with two strips per wave, against
    // 301 us of attention.hip's attn_fwd_kernel: the bf16 forward stays there.)
    static const bool nosplit = gg_dev_env("GG_ATTN_NO_SPLIT") != nullptr;
    const int nt16 = p.npad / 16;
    if (dtype == 1 && a->head_dim == 32 && !nosplit && (nt16 == 4 || nt16 == 9 || nt16 == 13)) {
        const size_t lds = sp_lds_fwd(p, 3);
        GG_PROF(GG_CAT_ATTN, 4.0 * a->num_windows * a->num_heads * (double)p.N * p.N * a->head_dim, 16.0 * a->num_windows * a->num_heads * (double)p.N * a->head_dim, stream);
        void (*kern)(FlashParams) = nt16 == 13 ? flash_fwd_split_kernel<float, 3, 13> : nt16 == 9 ? flash_fwd_split_kernel<float, 3, 9> : flash_fwd_split_kernel<float, 3, 4>;
        if (lds > 64 * 1024) {
            static bool raised[3] = {false, false, false};
            const int ri = nt16 == 13 ? 0 : nt16 == 9 ? 1 : 2;
            if (!raised[ri]) {
                GG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess,
                         "gg_attention_flash_fwd: cannot raise the dynamic LDS limit of the split kernel");
                raised[ri] = true;
            }
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)(a->num_windows * a->num_heads)), dim3(64 * nt16), lds, (hipStream_t)stream, p);
        GG_LAUNCH_CHECK();
        return 0;
    }
    const bool res = flash_resident(p, a->head_dim, false);
    const bool ftail = res && flash_fwd_tail(p);
    const dim3 grid((unsigned)(a->num_windows * a->num_heads * (res ? 1 : p.ntile))), block(res ? 64 * (ftail ? p.npad / 16 - 1 : std::min(16, p.npad / 16)) : 256);
    const size_t lds = flash_lds_fwd(p, a->head_dim, res ? p.npad : 64) + (ftail ? flash_lds_fwd_tail(p, a->head_dim) : 0);
    hipStream_t s = (hipStream_t)stream;
    const double es = dtype == 1 ? 4.0 : 2.0;
    GG_PROF(GG_CAT_ATTN, 4.0 * a->num_windows * a->num_heads * (double)p.N * p.N * a->head_dim,
            4.0 * es * a->num_windows * a->num_heads * (double)p.N * a->head_dim, stream);
#define GG_FL_FWD(T_, D_)                                                                                     \
    do {                                                                                                      \
        if (res && lds > 64 * 1024) {                                                                         \
            static bool raised = false;                                                                       \
            if (!raised) {                                                                                    \
                GG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(flash_fwd_kernel<T_, D_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess, \
                         "gg_attention_flash_fwd: cannot raise the dynamic LDS limit");                       \
                raised = true;                                                                                \
            }                                                                                                 \
        }                                                                                                     \
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
extern "C" int gg_attention_flash_single_pass(int tokens_per_window, int head_dim, int window_size, int with_dbias) {
    FlashParams p{};
    p.N = tokens_per_window; p.ws = window_size; p.npad = (int)gg_align(tokens_per_window, 16);
    p.nbpad = (int)gg_align(std::max(4, (2 * window_size - 1) * (2 * window_size - 1)), 4);
    return (head_dim == 32 || head_dim == 64) && flash_fused_ok(p, head_dim, with_dbias != 0) ? 1 : 0;
}
extern "C" int64_t gg_attention_flash_ds_scratch_floats(int num_windows, int num_heads, int tokens_per_window) {
    const int64_t npad = gg_align(tokens_per_window, 16);
    return (int64_t)num_windows * num_heads * npad * npad;
}
extern "C" int64_t gg_attention_flash_dbias_rows(int num_windows, int tokens_per_window) {
    return (int64_t)num_windows * gg_cdiv(tokens_per_window, 64) + GG_REDUCE_SLICES;
}
// forward_rounded_bias: the forward that produced `lse` was gg_attention_fwd's bf16 kernel, which adds the bias as bf16(bias / scale) from the expanded table: P is
// recomputed with that value.  Only gg_attention_bwd passes 1 (attention.hip); the public entry point pairs with gg_attention_flash_fwd, which reads the compact
// f32 table -- a caller who fills both `bias` and `bias_table` and calls the flash pair directly gets the same bias in both passes
int gg_attention_flash_bwd_impl(const GgAttnArgs* a, int dtype, int forward_rounded_bias, void* stream);
extern "C" int gg_attention_flash_bwd(const GgAttnArgs* a, int dtype, void* stream) { return gg_attention_flash_bwd_impl(a, dtype, 0, stream); }
int gg_attention_flash_bwd_impl(const GgAttnArgs* a, int dtype, int forward_rounded_bias, void* stream) {
    FlashParams p;
    GG_CHECK(dtype != 2, "gg_attention_flash_bwd: fp16 storage is inference-only");
    GG_TRY(flash_fill(p, a, dtype, "gg_attention_flash_bwd"));
    p.round_bias = (forward_rounded_bias && dtype == 0 && a->bias != nullptr) ? 1 : 0;
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
    {
        // head dim 32: the single-pass backward of attention_split.h (fp32 storage: split-bf16 products; bf16 storage: plain bf16 products)
        static const bool nosplit = gg_dev_env("GG_ATTN_NO_SPLIT") != nullptr;
        const int nt16 = p.npad / 16;
        const int npl = dtype == 1 ? 3 : 1;
        if ((dtype == 1 || dtype == 0) && a->head_dim == 32 && !nosplit && (nt16 == 4 || nt16 == 9 || nt16 == 13) && sp_lds_bwd(p, p.dbias != nullptr, npl) <= 160 * 1024) {
            GG_CHECK(dtype == 1 || ((a->ld & 7) == 0 && (a->ldo & 7) == 0 && (a->lddo & 7) == 0 && ((a->q_off | a->k_off | a->v_off | a->head_stride) & 7) == 0),
                     "gg_attention_flash_bwd: bf16 rows must be 16-byte aligned");
            GG_PROF(GG_CAT_ATTN, 10.0 * a->num_windows * a->num_heads * (double)p.N * p.N * a->head_dim,
                    8.0 * es * a->num_windows * a->num_heads * (double)p.N * a->head_dim, stream);
            const size_t lds = sp_lds_bwd(p, p.dbias != nullptr, npl);
            void (*kern)(FlashParams);
#define GG_SP_BWD(T_, N_)                                                                                                                     \
    do {                                                                                                                                      \
        if (p.dbias) kern = nt16 == 13 ? flash_bwd_split_kernel<T_, N_, true, 13> : nt16 == 9 ? flash_bwd_split_kernel<T_, N_, true, 9> : flash_bwd_split_kernel<T_, N_, true, 4>;      \
        else kern = nt16 == 13 ? flash_bwd_split_kernel<T_, N_, false, 13> : nt16 == 9 ? flash_bwd_split_kernel<T_, N_, false, 9> : flash_bwd_split_kernel<T_, N_, false, 4>;        \
    } while (0)
            if (dtype == 1) GG_SP_BWD(float, 3); else GG_SP_BWD(bf16, 1);
#undef GG_SP_BWD
            if (lds > 64 * 1024) {
                static bool raised[12] = {};
                const int ri = (dtype == 1 ? 0 : 6) + (p.dbias != nullptr) * 3 + (nt16 == 13 ? 0 : nt16 == 9 ? 1 : 2);
                if (!raised[ri]) {
                    GG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess,
                             "gg_attention_flash_bwd: cannot raise the dynamic LDS limit of the split kernel");
                    raised[ri] = true;
                }
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)(a->num_windows * a->num_heads)), dim3(64 * ((nt16 + 1) / 2)), lds, s, p);
            if (p.dbias && p.dbias_part) {
                const int Wd = p.nh * p.ws * p.ws;
                const float* rows; int nrows;
                gg_reduce_rows(p.dbias_part, a->num_windows, Wd, s, &rows, &nrows);
                hipLaunchKernelGGL(flash_dbias_final_kernel, dim3((unsigned)gg_cdiv(Wd, 256)), dim3(256), 0, s, rows, nrows, Wd, p.dbias);
            }
            GG_LAUNCH_CHECK();
            return 0;
        }
    }
    if (flash_fused_ok(p, a->head_dim, p.dbias != nullptr)) {
        // windows of at most 256 tokens: the single-pass kernel (no dS scratch, no second phase)
        GG_PROF(GG_CAT_ATTN, 10.0 * a->num_windows * a->num_heads * (double)p.N * p.N * a->head_dim,
                8.0 * es * a->num_windows * a->num_heads * (double)p.N * a->head_dim, stream);
        const size_t lds = flash_lds_fused(p, a->head_dim, p.dbias != nullptr);
        const dim3 fgrid((unsigned)(a->num_windows * a->num_heads)), fblock(64 * flash_fused_waves(p, a->head_dim));
#define GG_FL_FUSED(T_, D_)                                                                                   \
    do {                                                                                                      \
        void (*kern)(FlashParams);                                                                            \
        const int nt_ = p.npad / 16;                                                                          \
        if (p.dbias) kern = nt_ == 13 ? flash_bwd_fused_kernel<T_, D_, true, 13> : nt_ == 4 ? flash_bwd_fused_kernel<T_, D_, true, 4> : flash_bwd_fused_kernel<T_, D_, true, 0>; \
        else kern = nt_ == 13 ? flash_bwd_fused_kernel<T_, D_, false, 13> : nt_ == 4 ? flash_bwd_fused_kernel<T_, D_, false, 4> : flash_bwd_fused_kernel<T_, D_, false, 0>; \
        if (lds > 64 * 1024) {                                                                                \
            static bool raised[6] = {false, false, false, false, false, false};                               \
            const int ri_ = (p.dbias != nullptr) * 3 + (nt_ == 13 ? 1 : nt_ == 4 ? 2 : 0);                     \
            if (!raised[ri_]) {                                                                               \
                GG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess, \
                         "gg_attention_flash_bwd: cannot raise the dynamic LDS limit of the single-pass kernel");  \
                raised[ri_] = true;                                                                           \
            }                                                                                                 \
        }                                                                                                     \
        hipLaunchKernelGGL(kern, fgrid, fblock, lds, s, p);                                                   \
    } while (0)
        if (dtype == 1) { if (a->head_dim == 32) GG_FL_FUSED(float, 32); else GG_FL_FUSED(float, 64); }
        else { if (a->head_dim == 32) GG_FL_FUSED(bf16, 32); else GG_FL_FUSED(bf16, 64); }
#undef GG_FL_FUSED
        if (p.dbias && p.dbias_part) {
            const int Wd = p.nh * p.ws * p.ws;
            const float* rows; int nrows;
            gg_reduce_rows(p.dbias_part, a->num_windows, Wd, s, &rows, &nrows);
            hipLaunchKernelGGL(flash_dbias_final_kernel, dim3((unsigned)gg_cdiv(Wd, 256)), dim3(256), 0, s, rows, nrows, Wd, p.dbias);
        }
        GG_LAUNCH_CHECK();
        return 0;
    }
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
