// Window attention (TinyViT Attention / CLIP MHSA) forward and backward on MFMA (gfx950).
//
// One 256-thread workgroup per (window, head).  A window has N <= 256 tokens (49 / 196 for TinyViT @224,
// 50 for CLIP ViT-B/32), so a full score row lives in one wave's registers and softmax needs no online
// rescaling.  Scores are computed "swapped" (S^T = K Q^T) with v_mfma_f32_16x16x32_bf16 so that a lane owns one
// query column and its keys sit in registers: row max / sum are register reductions + two shuffles, and the
// exponentiated tile is directly the B operand of the P.V product (no LDS round trip; the MFMA k index is
// permuted identically on both operands).  Only operands that must be read "k-major" (V^T forward; K^T, Q^T,
// dO^T backward) are staged transposed in LDS.
//
// Layout: qkv is a row-major [tokens, ld] bf16 matrix; head h has q at column q_off + h*head_stride, k at
// k_off + h*head_stride, v at v_off + h*head_stride (TinyViT: per-head interleaved 3*D blocks -> offsets
// 0/D/2D, stride 3D; CLIP: [q|k|v] blocks of width nh*D -> offsets 0/C/2C, stride D).
#include "common.h"
#include <stdlib.h>
#include "../../include/gg.h"
int gg_attention_flash_bwd_impl(const GgAttnArgs* a, int dtype, int forward_rounded_bias, void* stream);      // attention_flash.hip

struct AttnParams {
    const bf16* qkv; int64_t ld;
    int q_off, k_off, v_off, head_stride;
    bf16* out; int64_t ldo;           // forward output [tokens, ldo], head h at column h*D
    const bf16* bias; int nbias;      // EXPANDED bias / scale [nh][Np][Np] bf16 (gg_attention_expand_bias; padded keys = -inf) or null;
                                      // nbias = ws*ws = size of the table the bias gradient is reduced into
    int ws, nWx, nWy, H, W;           // ws > 0: windows of ws x ws tokens inside an H x W map; ws == 0: linear
    int ws_inv;                       // ceil(65536 / ws)
    int N;                            // tokens per window
    int nh;
    float scale;
    // backward
    const bf16* dout; int64_t lddo;   // [tokens, lddo]
    bf16* dqkv;                       // same layout as qkv
    float* dbias;                     // [nh][nbias] accumulated into, or null
    float* dbias_part;                // [num_windows + 64][nh][nbias] per-window partials (two-stage deterministic sum) or null: atomics
    float* lse;                       // [tokens][nh] log-sum-exp of the scaled+biased scores (fwd writes, bwd reads)
};

// token index of window-local position t: the window origin is block-uniform (computed once), the in-window split
// t -> (t / ws, t % ws) uses a 16-bit reciprocal (exact for t < 256, ws <= 16) -- no integer division per token
__device__ __forceinline__ int attn_origin(const AttnParams& p, int w) {
    if (p.ws == 0) return w * p.N;
    const int per_img = p.nWx * p.nWy;
    const int b = w / per_img, r = w % per_img;
    const int wy = r / p.nWx, wx = r % p.nWx;
    return (b * p.H + wy * p.ws) * p.W + wx * p.ws;
}
__device__ __forceinline__ int attn_token(const AttnParams& p, int origin, int t) {
    if (t >= p.N) return -1;
    if (p.ws == 0) return origin + t;
    const int i = (t * p.ws_inv) >> 16, j = t - i * p.ws;
    return origin + i * p.W + j;
}
// 4 consecutive entries of the expanded bf16 bias table as an MFMA accumulator tile (bf16 -> f32 is a 16-bit shift)
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 attn_bias_cvt(u32x2 raw) {
    return (f32x4){__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u), __uint_as_float(raw.y << 16),
                   __uint_as_float(raw.y & 0xffff0000u)};
}
__device__ __forceinline__ f32x4 attn_bias4(const bf16* ptr) { return attn_bias_cvt(*reinterpret_cast<const u32x2*>(ptr)); }
// the same 4 entries for the bias-free case: 0 for valid positions, -inf (bf16 0xFF80) beyond N
__device__ __forceinline__ u32x2 attn_mask_raw(int pos0, int N) {
    u32x2 r;
    r.x = (pos0 < N ? 0u : 0xFF80u) | (pos0 + 1 < N ? 0u : 0xFF800000u);
    r.y = (pos0 + 2 < N ? 0u : 0xFF80u) | (pos0 + 3 < N ? 0u : 0xFF800000u);
    return r;
}
// 8 consecutive bf16 of row `tok` at column `col` (zeros for padded rows)
__device__ __forceinline__ bf16x8 attn_row_frag(const bf16* base, int64_t ld, int tok, int col) {
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (tok >= 0) v = *reinterpret_cast<const bf16x8*>(base + (int64_t)tok * ld + col);
    return v;
}
// 4+4 bf16 of a transposed LDS image T[d][key]: keys k0..k0+3 and k0+16..k0+19
__device__ __forceinline__ bf16x8 attn_tr_frag(const bf16* T, int stride, int d, int k0) {
    const bf16x4 a = *reinterpret_cast<const bf16x4*>(T + d * stride + k0);
    const bf16x4 b = *reinterpret_cast<const bf16x4*>(T + d * stride + k0 + 16);
    bf16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return v;
}
__device__ __forceinline__ bf16x8 attn_pack(const f32x4& a, const f32x4& b) {
    bf16x8 v = {(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
    return v;
}
// row-major staging of a [N, D] column block of `src` into X[Np][RS] (zero rows for padded tokens), split into the load
// half and the LDS-write half so that a kernel can put every operand load in flight before the first write waits.
template <int D, int Np>
__device__ __forceinline__ void attn_load_rows(bf16x8 (&v)[(Np * (D / 8) + 255) / 256], const bf16* src, int64_t ld, int col,
                                               const int (&tokv)[(Np * (D / 8) + 255) / 256]) {
    constexpr int CH = D / 8;
    constexpr int IT = (Np * CH + 255) / 256;
#pragma unroll
    for (int i = 0; i < IT; ++i) v[i] = attn_row_frag(src, ld, tokv[i], col + ((threadIdx.x + i * 256) % CH) * 8);
}
template <int D, int Np>
__device__ __forceinline__ void attn_store_rows(bf16* X, int RS, const bf16x8 (&v)[(Np * (D / 8) + 255) / 256]) {
    constexpr int CH = D / 8;
    constexpr int IT = (Np * CH + 255) / 256;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int idx = threadIdx.x + i * 256;
        if (idx < Np * CH) *reinterpret_cast<bf16x8*>(X + (idx / CH) * RS + (idx % CH) * 8) = v[i];
    }
}
// token ids of the rows this thread stages (chunk idx = tid + 256 i -> row idx / CH): computed arithmetically, so the
// operand loads do not wait for an LDS token table + barrier
template <int D, int Np>
__device__ __forceinline__ void attn_stage_tokens(const AttnParams& p, int origin, int (&tokv)[(Np * (D / 8) + 255) / 256]) {
    constexpr int CH = D / 8;
    constexpr int IT = (Np * CH + 255) / 256;
#pragma unroll
    for (int i = 0; i < IT; ++i) tokv[i] = attn_token(p, origin, min((int)(threadIdx.x + i * 256) / CH, Np - 1));
}
// MFMA operand fragment of row `row`: 8 consecutive bf16 at column 8*lg (+32*ks)
__device__ __forceinline__ bf16x8 attn_lds_row_frag(const bf16* X, int RS, int row, int col) {
    return *reinterpret_cast<const bf16x8*>(X + row * RS + col);
}
// "k-major" fragment straight from the row-major image with the transposing LDS read (ds_read_b64_tr_b16): lane (lr, lg)
// receives X[kbase + 4lg + q][d0 + lr] (q = 0..3) and X[kbase + 16 + 4lg + q][d0 + lr]; within each 16-lane group lane
// 4q+p supplies the address of row q, columns 4p..4p+3.  EXEC must be all ones (all call sites are wave-uniform).
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 attn_lds_tr_frag(const bf16* X, int RS, int d0, int kbase, int lr, int lg) {
    const bf16* p0 = X + (kbase + 4 * lg + (lr >> 2)) * RS + d0 + 4 * (lr & 3);
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 16 * RS));
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = a; u.s[1] = b;
    return u.v;
}
// ---- element type of the FORWARD kernel: bf16 (training / inference of the bf16 mode) or fp16 (inference of the CLIP tower's fp16 mode, BASELINE
// config c4: "MFMA fp16").  Same 16-bit layout, same tiling; what differs is the MFMA instruction and the f32 -> 16-bit conversion.
template <typename E> struct AE;
template <> struct AE<bf16> {
    typedef bf16x8 x8; typedef bf16x4 x4;
    static __device__ __forceinline__ f32x4 mfma(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct AE<f16> {
    typedef f16x8 x8; typedef f16x4 x4;
    static __device__ __forceinline__ f32x4 mfma(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <typename E> __device__ __forceinline__ typename AE<E>::x8 attn_row_frag_e(const E* base, int64_t ld, int tok, int col) {
    typename AE<E>::x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (tok >= 0) v = *reinterpret_cast<const typename AE<E>::x8*>(base + (int64_t)tok * ld + col);
    return v;
}
template <typename E> __device__ __forceinline__ typename AE<E>::x8 attn_pack_e(const f32x4& a, const f32x4& b) {
    typename AE<E>::x8 v = {(E)a[0], (E)a[1], (E)a[2], (E)a[3], (E)b[0], (E)b[1], (E)b[2], (E)b[3]};
    return v;
}
template <typename E, int D, int Np>
__device__ __forceinline__ void attn_load_rows_e(typename AE<E>::x8 (&v)[(Np * (D / 8) + 255) / 256], const E* src, int64_t ld, int col,
                                                 const int (&tokv)[(Np * (D / 8) + 255) / 256]) {
    constexpr int CH = D / 8;
    constexpr int IT = (Np * CH + 255) / 256;
#pragma unroll
    for (int i = 0; i < IT; ++i) v[i] = attn_row_frag_e<E>(src, ld, tokv[i], col + ((threadIdx.x + i * 256) % CH) * 8);
}
template <typename E, int D, int Np>
__device__ __forceinline__ void attn_store_rows_e(E* X, int RS, const typename AE<E>::x8 (&v)[(Np * (D / 8) + 255) / 256]) {
    constexpr int CH = D / 8;
    constexpr int IT = (Np * CH + 255) / 256;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int idx = threadIdx.x + i * 256;
        if (idx < Np * CH) *reinterpret_cast<typename AE<E>::x8*>(X + (idx / CH) * RS + (idx % CH) * 8) = v[i];
    }
}
template <typename E> __device__ __forceinline__ typename AE<E>::x8 attn_lds_row_frag_e(const E* X, int RS, int row, int col) {
    return *reinterpret_cast<const typename AE<E>::x8*>(X + row * RS + col);
}
template <typename E> __device__ __forceinline__ typename AE<E>::x8 attn_lds_tr_frag_e(const E* X, int RS, int d0, int kbase, int lr, int lg) {
    const E* p0 = X + (kbase + 4 * lg + (lr >> 2)) * RS + d0 + 4 * (lr & 3);
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 16 * RS));
    union { s16x4 s[2]; typename AE<E>::x8 v; } u;
    u.s[0] = a; u.s[1] = b;
    return u.v;
}
// attention_biases[h][|di|*ws+|dj|] -> full[h][q][k] (Np x Np, row-major): the kernels then fetch 4 consecutive keys of a
// query row with one 16-byte load instead of 4 x (index arithmetic + LDS gather).  Padded keys carry -inf so no mask
// select is needed; the matrix is symmetric, which the backward's [query][key] orientation uses.
__global__ void attn_expand_bias_kernel(const float* __restrict__ table, int nh, int ws, int N, int Np, float inv_scale,
                                        bf16* __restrict__ full) {
    const int64_t total = (int64_t)nh * Np * Np;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % Np), q = (int)((i / Np) % Np), h = (int)(i / ((int64_t)Np * Np));
        float v = -INFINITY;
        if (k < N) {
            const int qq = min(q, N - 1);
            v = table[h * ws * ws + abs(qq / ws - k / ws) * ws + abs(qq % ws - k % ws)] * inv_scale;
        }
        full[i] = (bf16)v;
    }
}

// One chunk of KN key tiles (starting at tile K0) of one query tile.  The score accumulators START as the relative-position
// bias (the expanded table holds bias / scale, -inf on padded keys), so the bias costs no VALU work and the chunk's tile loads
// are in flight together before the first MFMA.
template <typename E, int D, int NKT, int K0, int KN>
__device__ __forceinline__ void attn_fwd_chunk(const AttnParams& p, const E* Ks, const E* Vs, const bf16* bias_h,
                                               const typename AE<E>::x8 (&qf)[D / 32], int qi, int lr, int lg, float c2, float& m, float& l,
                                               f32x4 (&o)[D / 16]) {
    constexpr int Np = NKT * 16, RS = D + 8, KS = D / 32, DT = D / 16;
    static_assert((K0 & 1) == 0 && (KN & 1) == 0, "key tiles are consumed in pairs");
    // keep the scheduler from hoisting this chunk's loads above the previous chunk's tail: it trades the occupancy the chunking
    // buys (registers) for latency hiding that the other resident waves already provide
    __builtin_amdgcn_sched_barrier(0);
    f32x4 s[KN];
    if (bias_h) {
#pragma unroll
        for (int kt = 0; kt < KN; ++kt) s[kt] = attn_bias4(bias_h + (int64_t)qi * Np + (K0 + kt) * 16 + lg * 4);
    } else {
#pragma unroll
        for (int kt = 0; kt < KN; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][r] = ((K0 + kt) * 16 + lg * 4 + r < p.N) ? 0.f : -INFINITY;
    }
    float mx = m;
#pragma unroll
    for (int kt = 0; kt < KN; ++kt) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const typename AE<E>::x8 kf = attn_lds_row_frag_e<E>(Ks, RS, (K0 + kt) * 16 + lr, ks * 32 + lg * 8);
            s[kt] = AE<E>::mfma(kf, qf[ks], s[kt]);   // D[i=key][j=query]
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (K0 > 0) {       // carry rescale (the first chunk starts from m = -inf, l = 0, o = 0)
        const float alpha = __builtin_amdgcn_exp2f((m - mx) * c2);
        l *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
    }
    m = mx;
    const float mb = mx * c2;
#pragma unroll
    for (int kt = 0; kt < KN; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(s[kt][r], c2, -mb));
            s[kt][r] = e;
            l += e;
        }
#pragma unroll
    for (int kp = 0; kp < KN / 2; ++kp) {
        const typename AE<E>::x8 pf = attn_pack_e<E>(s[2 * kp], s[2 * kp + 1]);   // k = key 32kp + 16(jj>>2) + 4lg + (jj&3)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const typename AE<E>::x8 vf = attn_lds_tr_frag_e<E>(Vs, RS, dt * 16, (K0 / 2 + kp) * 32, lr, lg);
            o[dt] = AE<E>::mfma(vf, pf, o[dt]);   // D[i=d][j=query]
        }
    }
}

#ifndef GG_ATTN_CK
#define GG_ATTN_CK 8
#endif
template <typename E, int D, int NKT, int K0>
__device__ __forceinline__ void attn_fwd_chunks(const AttnParams& p, const E* Ks, const E* Vs, const bf16* bias_h,
                                                const typename AE<E>::x8 (&qf)[D / 32], int qi, int lr, int lg, float c2, float& m, float& l,
                                                f32x4 (&o)[D / 16]) {
    if constexpr (K0 < NKT) {
        constexpr int KN = (NKT - K0) < GG_ATTN_CK ? (NKT - K0) : GG_ATTN_CK;
        attn_fwd_chunk<E, D, NKT, K0, KN>(p, Ks, Vs, bias_h, qf, qi, lr, lg, c2, m, l, o);
        attn_fwd_chunks<E, D, NKT, K0 + KN>(p, Ks, Vs, bias_h, qf, qi, lr, lg, c2, m, l, o);
    }
}

// ------------------------------------------------------------------------------------------- forward
template <int D, int NKT, typename E = bf16>
__global__ __launch_bounds__(256, (D == 32 && NKT <= 14) ? 3 : 1) void attn_fwd_kernel(AttnParams p) {
    constexpr int Np = NKT * 16;
    constexpr int RS = D + 8;     // 80 / 144-byte rows: 16-B aligned fragments, 8-B aligned transposing reads
    constexpr int KS = D / 32;    // MFMA k-steps over the head dim
    constexpr int DT = D / 16;    // output d tiles
    __shared__ __attribute__((aligned(16))) E Ks[Np * RS];
    __shared__ __attribute__((aligned(16))) E Vs[Np * RS];
    const E* qkv = reinterpret_cast<const E*>(p.qkv);       // (AttnParams carries the 16-bit tensors as bf16*: same layout for fp16)
    E* outp = reinterpret_cast<E*>(p.out);

    // all heads of a window run on one XCD (logical ids are laid out XCD by XCD): their 192-byte slices of a token row share
    // 128-byte lines, which round-robin dispatch made every XCD fetch separately
    const int bid = gg_xcd_remap(blockIdx.x, gridDim.x);
    const int w = bid / p.nh, h = bid % p.nh;
    const int origin = attn_origin(p, w);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const bf16* bias_h = p.bias ? p.bias + (int64_t)h * Np * Np : nullptr;
    const int nqt = (p.N + 15) / 16;

    // every global operand of the first query tile is requested before the first wait: K / V rows, the Q fragment, the bias
    int tokv[(Np * (D / 8) + 255) / 256];
    attn_stage_tokens<D, Np>(p, origin, tokv);
    int qtok = wave < nqt ? attn_token(p, origin, wave * 16 + lr) : -1;
    typename AE<E>::x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = attn_row_frag_e<E>(qkv, p.ld, qtok, p.q_off + h * p.head_stride + ks * 32 + lg * 8);
    {
        typename AE<E>::x8 kr[(Np * (D / 8) + 255) / 256], vr[(Np * (D / 8) + 255) / 256];
        attn_load_rows_e<E, D, Np>(kr, qkv, p.ld, p.k_off + h * p.head_stride, tokv);
        attn_load_rows_e<E, D, Np>(vr, qkv, p.ld, p.v_off + h * p.head_stride, tokv);
        attn_store_rows_e<E, D, Np>(Ks, RS, kr);
        attn_store_rows_e<E, D, Np>(Vs, RS, vr);
    }
    __syncthreads();

    for (int qt = wave; qt < nqt; qt += 4) {
        const int qi = qt * 16 + lr;
        if (qt != wave) {
            qtok = attn_token(p, origin, qi);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[ks] = attn_row_frag_e<E>(qkv, p.ld, qtok, p.q_off + h * p.head_stride + ks * 32 + lg * 8);
        }
        // Keys are visited in chunks of <= 8 tiles with an online-softmax carry (running max m, partial sum l, output o):
        // a chunk's scores are the only big live register array (32 instead of 4*NKT), which is what sets occupancy.
        float m = -INFINITY, l = 0.f;
        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float c2 = p.scale * 1.4426950408889634f;       // exp(scale*(s - m)) = 2^(s*c2 - m*c2)
        attn_fwd_chunks<E, D, NKT, 0>(p, Ks, Vs, bias_h, qf, qi, lr, lg, c2, m, l, o);
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float mx = m;
        if (qtok >= 0) {
            const float inv = 1.f / l;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                typename AE<E>::x4 ov = {(E)(o[dt][0] * inv), (E)(o[dt][1] * inv), (E)(o[dt][2] * inv), (E)(o[dt][3] * inv)};
                *reinterpret_cast<typename AE<E>::x4*>(outp + (int64_t)qtok * p.ldo + h * D + dt * 16 + lg * 4) = ov;
            }
            if (p.lse && lg == 0) p.lse[(int64_t)qtok * p.nh + h] = mx * p.scale + __logf(l);
        }
    }
}

// Small windows (N <= 64 tokens: the 7x7 stages, 4 key tiles = one query tile per wave): a workgroup walks GW consecutive windows
// of one head.  The query tile's bias tiles stay in registers across the windows (they depend on the head and the tile only: the
// per-window re-read was a third of this kernel's time), K / V of window g+1 travel through registers while window g is computed
// (two LDS images), and launch + prologue cost is paid once per GW windows.
template <int D, int GW>
__global__ __launch_bounds__(256) void attn_fwd_small_kernel(AttnParams p, int num_windows) {
    constexpr int NKT = 4, Np = 64, RS = D + 8, KS = D / 32, DT = D / 16;
    constexpr int IT = (Np * (D / 8) + 255) / 256;
    __shared__ __attribute__((aligned(16))) bf16 Ks[2][Np * RS];
    __shared__ __attribute__((aligned(16))) bf16 Vs[2][Np * RS];
    const int bid = gg_xcd_remap(blockIdx.x, gridDim.x);
    const int wb = bid / p.nh, h = bid % p.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int nqt = (p.N + 15) / 16;
    const int qi = wave * 16 + lr;
    const int kcol = p.k_off + h * p.head_stride, vcol = p.v_off + h * p.head_stride, qcol = p.q_off + h * p.head_stride;
    // bias tiles of this wave's query tile (raw bf16 pairs), or the padding mask
    u32x2 braw[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
        braw[kt] = p.bias ? *reinterpret_cast<const u32x2*>(p.bias + ((int64_t)h * Np + qi) * Np + kt * 16 + lg * 4) : attn_mask_raw(kt * 16 + lg * 4, p.N);
    const float c2 = p.scale * 1.4426950408889634f;
    const int w0 = wb * GW, w1 = min(num_windows, w0 + GW);
    bf16x8 kr[IT], vr[IT], qf[KS];
    int qtok = -1;
    auto fetch = [&](int w) {          // K / V rows and this wave's Q fragment of window w -> registers
        const int origin = attn_origin(p, w);
        int tokv[IT];
        attn_stage_tokens<D, Np>(p, origin, tokv);
        attn_load_rows<D, Np>(kr, p.qkv, p.ld, kcol, tokv);
        attn_load_rows<D, Np>(vr, p.qkv, p.ld, vcol, tokv);
        qtok = wave < nqt ? attn_token(p, origin, qi) : -1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = attn_row_frag(p.qkv, p.ld, qtok, qcol + ks * 32 + lg * 8);
    };
    if (w0 < w1) fetch(w0);
    for (int w = w0; w < w1; ++w) {
        const int sel = (w - w0) & 1;
        attn_store_rows<D, Np>(Ks[sel], RS, kr);
        attn_store_rows<D, Np>(Vs[sel], RS, vr);
        bf16x8 qcur[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qcur[ks] = qf[ks];
        const int qtok_cur = qtok;
        __syncthreads();               // image `sel` complete; image `sel^1` was last read before the previous iteration's barrier
        if (w + 1 < w1) fetch(w + 1);
        if (wave < nqt) {
            f32x4 sc[NKT];
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                sc[kt] = attn_bias_cvt(braw[kt]);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 kf = attn_lds_row_frag(Ks[sel], RS, kt * 16 + lr, ks * 32 + lg * 8);
                    sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qcur[ks], sc[kt], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[kt][r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mb = mx * c2;
            float l = 0.f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], c2, -mb));
                    sc[kt][r] = e;
                    l += e;
                }
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            f32x4 o[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kp = 0; kp < NKT / 2; ++kp) {
                const bf16x8 pf = attn_pack(sc[2 * kp], sc[2 * kp + 1]);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 vf = attn_lds_tr_frag(Vs[sel], RS, dt * 16, kp * 32, lr, lg);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
                }
            }
            if (qtok_cur >= 0) {
                const float inv = 1.f / l;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    bf16x4 ov = {(bf16)(o[dt][0] * inv), (bf16)(o[dt][1] * inv), (bf16)(o[dt][2] * inv), (bf16)(o[dt][3] * inv)};
                    *reinterpret_cast<bf16x4*>(p.out + (int64_t)qtok_cur * p.ldo + h * D + dt * 16 + lg * 4) = ov;
                }
                if (p.lse && lg == 0) p.lse[(int64_t)qtok_cur * p.nh + h] = mx * p.scale + __logf(l);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------- backward
// Flash-style: P is recomputed from the forward's per-row log-sum-exp, delta_q = sum_d dO[q][d] O[q][d] comes from the
// saved forward output, so no score row has to stay in registers (3+ waves per SIMD instead of 1).
template <int D, int NKT, bool DBIAS>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnParams p) {
    constexpr int Np = NKT * 16;
    constexpr int RS = D + 8;
    constexpr int KS = D / 32;
    constexpr int DT = D / 16;
    // Two operand images at a time: phase 1 (dQ) contracts against K and V, phase 2 (dK, dV) against Q and dO, and each
    // wave's own tile rows come straight from global memory -- half the LDS of keeping all four resident, i.e. 4 instead of
    // 2 workgroups per CU for 14x14 windows.
    // Small windows (7x7) are not LDS-limited and keep all four images resident (no reload, no extra barriers).
    constexpr bool TWO_PHASE = NKT > 4;
    __shared__ __attribute__((aligned(16))) bf16 buf0[Np * RS];
    __shared__ __attribute__((aligned(16))) bf16 buf1[Np * RS];
    __shared__ __attribute__((aligned(16))) bf16 buf2[TWO_PHASE ? 8 : Np * RS];
    __shared__ __attribute__((aligned(16))) bf16 buf3[TWO_PHASE ? 8 : Np * RS];
    bf16* const Ks = buf0; bf16* const Vs = buf1;                                       // phase 1
    bf16* const Qs = TWO_PHASE ? buf0 : buf2; bf16* const dOs = TWO_PHASE ? buf1 : buf3;   // phase 2
    __shared__ __attribute__((aligned(16))) float row_lse[Np], row_delta[Np];
    // bias-gradient bins, privatised 8 ways by query lane so that the LDS atomics of one instruction rarely collide (a single
    // 196-bin table took 64-way conflicts and tripled the kernel time); compiled in only for trainable attention_biases
    constexpr int DBC = 8;
    __shared__ float dbias_s[DBIAS ? DBC * 256 : 1];
    __shared__ __attribute__((aligned(4))) unsigned char ci[Np], cj[Np];   // window coordinates: bias-gradient binning only

    // all heads of a window run on one XCD (logical ids are laid out XCD by XCD): their 192-byte slices of a token row share
    // 128-byte lines, which round-robin dispatch made every XCD fetch separately
    const int bid = gg_xcd_remap(blockIdx.x, gridDim.x);
    const int w = bid / p.nh, h = bid % p.nh;
    const int origin = attn_origin(p, w);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int qcol = p.q_off + h * p.head_stride, kcol = p.k_off + h * p.head_stride, vcol = p.v_off + h * p.head_stride;
    const int ocol = h * D;

    const bf16* bias_h = p.bias ? p.bias + (int64_t)h * Np * Np : nullptr;
    {
        // all operand rows (K, V, dO, O) are requested before anything waits; token ids are computed, not looked up
        constexpr int CH = D / 8;
        constexpr int IT = (Np * CH + 255) / 256;
        int tokv[IT];
        attn_stage_tokens<D, Np>(p, origin, tokv);
        bf16x8 kr[IT], vr[IT], dr[IT], orr[IT];
        attn_load_rows<D, Np>(kr, p.qkv, p.ld, kcol, tokv);
        attn_load_rows<D, Np>(vr, p.qkv, p.ld, vcol, tokv);
        attn_load_rows<D, Np>(dr, p.dout, p.lddo, ocol, tokv);
        attn_load_rows<D, Np>(orr, p.out, p.ldo, ocol, tokv);
        for (int t = threadIdx.x; t < Np; t += blockDim.x) {
            const int tk = attn_token(p, origin, t);
            const int tt = min(t, p.N - 1);
            const int ti = (tt * p.ws_inv) >> 16;
            ci[t] = p.ws ? (unsigned char)ti : 0;
            cj[t] = p.ws ? (unsigned char)(tt - ti * p.ws) : 0;
            row_lse[t] = tk >= 0 ? -1.4426950408889634f * p.lse[(int64_t)tk * p.nh + h] : 0.f;      // stored as -lse*log2(e): P = 2^(s*c2 + this)
        }
        if (DBIAS) { for (int t = threadIdx.x; t < DBC * 256; t += blockDim.x) dbias_s[t] = 0.f; }
        attn_store_rows<D, Np>(Ks, RS, kr);
        attn_store_rows<D, Np>(Vs, RS, vr);
        if constexpr (!TWO_PHASE) {
            bf16x8 qr[IT];
            attn_load_rows<D, Np>(qr, p.qkv, p.ld, qcol, tokv);
            attn_store_rows<D, Np>(dOs, RS, dr);
            attn_store_rows<D, Np>(Qs, RS, qr);
        }
        // delta = rowsum(dO * O): the CH chunks of a row sit in adjacent lanes (Np*CH is a multiple of 64: full waves)
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = threadIdx.x + i * 256;
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) sm += (float)dr[i][j] * (float)orr[i][j];
#pragma unroll
            for (int o = 1; o < CH; o <<= 1) sm += __shfl_xor(sm, o, 64);
            if (idx < Np * CH && (idx % CH) == 0) row_delta[idx / CH] = -sm;      // stored negated: it is the dP accumulator's initial value
        }
    }
    __syncthreads();

    const int nt = (p.N + 15) / 16;
    // ---- phase 1: a wave owns a query tile -> dQ (and dbias) ----
    for (int qt = wave; qt < nt; qt += 4) {
        const int qi = qt * 16 + lr;
        const int qtok = attn_token(p, origin, qi);
        bf16x8 qf[KS], dof[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if constexpr (TWO_PHASE) {
                qf[ks] = attn_row_frag(p.qkv, p.ld, qtok, qcol + ks * 32 + lg * 8);
                dof[ks] = attn_row_frag(p.dout, p.lddo, qtok, ocol + ks * 32 + lg * 8);
            } else {
                qf[ks] = attn_lds_row_frag(Qs, RS, qi, ks * 32 + lg * 8);
                dof[ks] = attn_lds_row_frag(dOs, RS, qi, ks * 32 + lg * 8);
            }
        }
        const int qci = ci[qi], qcj = cj[qi];
        const float nlse_q = row_lse[qi], ndelta_q = row_delta[qi];        // -lse*log2(e), -delta
        f32x4 dq[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float c2 = p.scale * 1.4426950408889634f;
        // score accumulators start as bias / scale (see forward); the next pair of tiles is fetched one iteration ahead
        // (kept raw: converting at the prefetch point would wait for the load right there)
        auto bias_tiles = [&](int kp, u32x2 (&b)[2]) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int key0 = (kp * 2 + u) * 16 + lg * 4;
                b[u] = bias_h ? *reinterpret_cast<const u32x2*>(bias_h + (int64_t)qi * Np + key0) : attn_mask_raw(key0, p.N);
            }
        };
        u32x2 bnext[2];
        bias_tiles(0, bnext);
#pragma unroll 1
        for (int kp = 0; kp < NKT / 2; ++kp) {
            f32x4 dst[2];
            const f32x4 bcur[2] = {attn_bias_cvt(bnext[0]), attn_bias_cvt(bnext[1])};
            if (kp + 1 < NKT / 2) bias_tiles(kp + 1, bnext);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int kt = kp * 2 + u;
                f32x4 acc = bcur[u], acc2 = {ndelta_q, ndelta_q, ndelta_q, ndelta_q};      // dP - delta comes straight out of the MFMA
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 kf = attn_lds_row_frag(Ks, RS, kt * 16 + lr, ks * 32 + lg * 8);
                    const bf16x8 vf = attn_lds_row_frag(Vs, RS, kt * 16 + lr, ks * 32 + lg * 8);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], acc, 0, 0, 0);      // S^T  [key][query]
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[ks], acc2, 0, 0, 0);   // dP^T [key][query]
                }
                const int key0 = kt * 16 + lg * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = __builtin_amdgcn_exp2f(fmaf(acc[r], c2, nlse_q));     // padded keys: 2^-inf = 0
                    acc2[r] *= pr;
                }
                if (DBIAS && qtok >= 0) {       // bias gradient, binned by |di|,|dj|
                    const uchar4 kci = *reinterpret_cast<const uchar4*>(&ci[key0]);
                    const uchar4 kcj = *reinterpret_cast<const uchar4*>(&cj[key0]);
                    const int kcis[4] = {kci.x, kci.y, kci.z, kci.w}, kcjs[4] = {kcj.x, kcj.y, kcj.z, kcj.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (key0 + r < p.N) atomicAdd(&dbias_s[(lr & (DBC - 1)) * 256 + abs(qci - kcis[r]) * p.ws + abs(qcj - kcjs[r])], acc2[r]);
                }
                dst[u] = acc2;
            }
            const bf16x8 df = attn_pack(dst[0], dst[1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 kf = attn_lds_tr_frag(Ks, RS, dt * 16, kp * 32, lr, lg);
                dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, df, dq[dt], 0, 0, 0);   // dQ^T [d][query]
            }
        }
        if (qtok >= 0) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                bf16x4 ov = {(bf16)(dq[dt][0] * p.scale), (bf16)(dq[dt][1] * p.scale), (bf16)(dq[dt][2] * p.scale),
                             (bf16)(dq[dt][3] * p.scale)};
                *reinterpret_cast<bf16x4*>(p.dqkv + (int64_t)qtok * p.ld + qcol + dt * 16 + lg * 4) = ov;
            }
        }
    }
    // ---- phase switch: K, V images -> Q, dO images ----
    if constexpr (TWO_PHASE) {
        constexpr int IT = (Np * (D / 8) + 255) / 256;
        int tokv[IT];
        attn_stage_tokens<D, Np>(p, origin, tokv);
        bf16x8 qr[IT], dr[IT];
        attn_load_rows<D, Np>(qr, p.qkv, p.ld, qcol, tokv);
        attn_load_rows<D, Np>(dr, p.dout, p.lddo, ocol, tokv);
        __syncthreads();                       // every wave is done reading K / V
        attn_store_rows<D, Np>(Qs, RS, qr);
        attn_store_rows<D, Np>(dOs, RS, dr);
        __syncthreads();
    }
    // ---- phase 2: a wave owns a key tile -> dK, dV ----
    for (int kt = wave; kt < nt; kt += 4) {
        const int ki = kt * 16 + lr;
        const int ktok = attn_token(p, origin, ki);
        bf16x8 kf[KS], vf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if constexpr (TWO_PHASE) {
                kf[ks] = attn_row_frag(p.qkv, p.ld, ktok, kcol + ks * 32 + lg * 8);
                vf[ks] = attn_row_frag(p.qkv, p.ld, ktok, vcol + ks * 32 + lg * 8);
            } else {
                kf[ks] = attn_lds_row_frag(Ks, RS, ki, ks * 32 + lg * 8);
                vf[ks] = attn_lds_row_frag(Vs, RS, ki, ks * 32 + lg * 8);
            }
        }
        f32x4 dk[DT], dv[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        const float c2 = p.scale * 1.4426950408889634f;
        const int64_t brow = (int64_t)min(ki, p.N - 1) * Np;      // symmetric table: bias[q][k] == bias[k][q]; columns >= N hold -inf
        auto bias_tiles = [&](int qp, u32x2 (&b)[2]) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q0 = (qp * 2 + u) * 16 + lg * 4;
                b[u] = bias_h ? *reinterpret_cast<const u32x2*>(bias_h + brow + q0) : attn_mask_raw(q0, p.N);
            }
        };
        u32x2 bnext[2];
        bias_tiles(0, bnext);
#pragma unroll 1
        for (int qp = 0; qp < NKT / 2; ++qp) {
            f32x4 pt[2], dst[2];
            const f32x4 bcur[2] = {attn_bias_cvt(bnext[0]), attn_bias_cvt(bnext[1])};
            if (qp + 1 < NKT / 2) bias_tiles(qp + 1, bnext);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int qt = qp * 2 + u;
                const int q0 = qt * 16 + lg * 4;
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(&row_lse[q0]);          // -lse*log2(e) of the 4 queries
                f32x4 acc = bcur[u], acc2 = *reinterpret_cast<const f32x4*>(&row_delta[q0]);   // -delta: dP - delta out of the MFMA
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 qf = attn_lds_row_frag(Qs, RS, qt * 16 + lr, ks * 32 + lg * 8);
                    const bf16x8 dof = attn_lds_row_frag(dOs, RS, qt * 16 + lr, ks * 32 + lg * 8);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf[ks], acc, 0, 0, 0);     // S  [query][key]
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof, vf[ks], acc2, 0, 0, 0);  // dP [query][key]
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // padded queries: -inf score -> P = 0.  Padded keys need no mask: a lane's key column only feeds its own dK / dV
                    // columns, and those are never stored.
                    const float pr = __builtin_amdgcn_exp2f(fmaf(acc[r], c2, l4[r]));
                    acc[r] = pr;
                    acc2[r] *= pr;
                }
                pt[u] = acc;
                dst[u] = acc2;
            }
            const bf16x8 pf = attn_pack(pt[0], pt[1]);      // k = query 32qp + 16(jj>>2) + 4lg + (jj&3)
            const bf16x8 df = attn_pack(dst[0], dst[1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 qtf = attn_lds_tr_frag(Qs, RS, dt * 16, qp * 32, lr, lg);
                const bf16x8 dotf = attn_lds_tr_frag(dOs, RS, dt * 16, qp * 32, lr, lg);
                dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, df, dk[dt], 0, 0, 0);    // dK^T [d][key]
                dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dotf, pf, dv[dt], 0, 0, 0);   // dV^T [d][key]
            }
        }
        if (ktok >= 0) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                bf16x4 a = {(bf16)(dk[dt][0] * p.scale), (bf16)(dk[dt][1] * p.scale), (bf16)(dk[dt][2] * p.scale), (bf16)(dk[dt][3] * p.scale)};
                bf16x4 b = {(bf16)dv[dt][0], (bf16)dv[dt][1], (bf16)dv[dt][2], (bf16)dv[dt][3]};
                *reinterpret_cast<bf16x4*>(p.dqkv + (int64_t)ktok * p.ld + kcol + dt * 16 + lg * 4) = a;
                *reinterpret_cast<bf16x4*>(p.dqkv + (int64_t)ktok * p.ld + vcol + dt * 16 + lg * 4) = b;
            }
        }
    }
    if (DBIAS) {
        __syncthreads();
        for (int t = threadIdx.x; t < p.nbias; t += blockDim.x) {
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < DBC; ++c) sum += dbias_s[c * 256 + t];
            if (p.dbias_part) p.dbias_part[((int64_t)w * p.nh + h) * p.nbias + t] = sum;       // deterministic: reduced over windows afterwards
            else atomicAdd(&p.dbias[h * p.nbias + t], sum);
        }
    }
}
// Backward for small windows (N <= 64 tokens), GW consecutive windows of one head per workgroup: Q, K, V, dO of window g+1 are in
// flight while window g is computed (two sets of LDS images), the bias tiles of the wave's query tile (phase 1) and key tile
// (phase 2) stay in registers for all GW windows, and the bias-gradient bins are flushed once per workgroup.
template <int D, int GW, bool DBIAS>
__global__ __launch_bounds__(256) void attn_bwd_small_kernel(AttnParams p, int num_windows) {
    constexpr int NKT = 4, Np = 64, RS = D + 8, KS = D / 32, DT = D / 16, CH = D / 8;
    constexpr int IT = (Np * CH + 255) / 256;
    constexpr int DBC = 8;
    __shared__ __attribute__((aligned(16))) bf16 img[2][4][Np * RS];          // [buffer][Q, K, V, dO]
    __shared__ __attribute__((aligned(16))) float row_lse[2][Np], row_delta[2][Np];
    __shared__ float dbias_s[DBIAS ? DBC * 256 : 1];
    __shared__ __attribute__((aligned(4))) unsigned char ci[Np], cj[Np];
    const int bid = gg_xcd_remap(blockIdx.x, gridDim.x);
    const int wb = bid / p.nh, h = bid % p.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int qcol = p.q_off + h * p.head_stride, kcol = p.k_off + h * p.head_stride, vcol = p.v_off + h * p.head_stride;
    const int ocol = h * D;
    const int nt = (p.N + 15) / 16;
    const int ti = wave * 16 + lr;                 // this lane's query row (phase 1) / key row (phase 2)
    for (int t = threadIdx.x; t < Np; t += blockDim.x) {
        const int tt = min(t, p.N - 1), i = (tt * p.ws_inv) >> 16;
        ci[t] = p.ws ? (unsigned char)i : 0;
        cj[t] = p.ws ? (unsigned char)(tt - i * p.ws) : 0;
    }
    if (DBIAS) { for (int t = threadIdx.x; t < DBC * 256; t += blockDim.x) dbias_s[t] = 0.f; }
    // bias tiles: phase 1 reads row ti (query) x key tiles; phase 2 reads row min(ti, N-1) (key; symmetric table) x query tiles
    u32x2 bq[NKT], bk[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const int c0 = kt * 16 + lg * 4;
        bq[kt] = p.bias ? *reinterpret_cast<const u32x2*>(p.bias + ((int64_t)h * Np + ti) * Np + c0) : attn_mask_raw(c0, p.N);
        bk[kt] = p.bias ? *reinterpret_cast<const u32x2*>(p.bias + ((int64_t)h * Np + min(ti, p.N - 1)) * Np + c0) : attn_mask_raw(c0, p.N);
    }
    const float c2 = p.scale * 1.4426950408889634f;
    const int w0 = wb * GW, w1 = min(num_windows, w0 + GW);
    bf16x8 qr[IT], kr[IT], vr[IT], dr[IT], orr[IT];
    float lse_r = 0.f;
    auto fetch = [&](int w) {
        const int origin = attn_origin(p, w);
        int tokv[IT];
        attn_stage_tokens<D, Np>(p, origin, tokv);
        attn_load_rows<D, Np>(qr, p.qkv, p.ld, qcol, tokv);
        attn_load_rows<D, Np>(kr, p.qkv, p.ld, kcol, tokv);
        attn_load_rows<D, Np>(vr, p.qkv, p.ld, vcol, tokv);
        attn_load_rows<D, Np>(dr, p.dout, p.lddo, ocol, tokv);
        attn_load_rows<D, Np>(orr, p.out, p.ldo, ocol, tokv);
        if (threadIdx.x < Np) {
            const int tk = attn_token(p, origin, threadIdx.x);
            lse_r = tk >= 0 ? -1.4426950408889634f * p.lse[(int64_t)tk * p.nh + h] : 0.f;
        }
    };
    f32x4 dsum[DBIAS ? NKT : 1];           // dS of this lane's (query row, 4 keys x NKT tiles) summed over the workgroup's windows
#pragma unroll
    for (int kt = 0; kt < (DBIAS ? NKT : 1); ++kt) dsum[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (w0 < w1) fetch(w0);
    for (int w = w0; w < w1; ++w) {
        const int sel = (w - w0) & 1;
        bf16* Qs = img[sel][0]; bf16* Ks = img[sel][1]; bf16* Vs = img[sel][2]; bf16* dOs = img[sel][3];
        attn_store_rows<D, Np>(Qs, RS, qr);
        attn_store_rows<D, Np>(Ks, RS, kr);
        attn_store_rows<D, Np>(Vs, RS, vr);
        attn_store_rows<D, Np>(dOs, RS, dr);
        if (threadIdx.x < Np) row_lse[sel][threadIdx.x] = lse_r;
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int idx = threadIdx.x + i * 256;
            float sm = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) sm += (float)dr[i][j] * (float)orr[i][j];
#pragma unroll
            for (int o = 1; o < CH; o <<= 1) sm += __shfl_xor(sm, o, 64);
            if (idx < Np * CH && (idx % CH) == 0) row_delta[sel][idx / CH] = -sm;
        }
        const int origin = attn_origin(p, w);
        const int tok_own = wave < nt ? attn_token(p, origin, ti) : -1;
        __syncthreads();
        if (w + 1 < w1) fetch(w + 1);
        if (wave < nt) {
            // ---- phase 1: this wave's query tile -> dQ (and dbias) ----
            {
                bf16x8 qf[KS], dof[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    qf[ks] = attn_lds_row_frag(Qs, RS, ti, ks * 32 + lg * 8);
                    dof[ks] = attn_lds_row_frag(dOs, RS, ti, ks * 32 + lg * 8);
                }
                const float nlse_q = row_lse[sel][ti], ndelta_q = row_delta[sel][ti];
                f32x4 dq[DT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kp = 0; kp < NKT / 2; ++kp) {
                    f32x4 dst[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int kt = kp * 2 + u;
                        f32x4 acc = attn_bias_cvt(bq[kt]), acc2 = {ndelta_q, ndelta_q, ndelta_q, ndelta_q};
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            const bf16x8 kf = attn_lds_row_frag(Ks, RS, kt * 16 + lr, ks * 32 + lg * 8);
                            const bf16x8 vf = attn_lds_row_frag(Vs, RS, kt * 16 + lr, ks * 32 + lg * 8);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], acc, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[ks], acc2, 0, 0, 0);
                        }
                        const int key0 = kt * 16 + lg * 4;
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc2[r] *= __builtin_amdgcn_exp2f(fmaf(acc[r], c2, nlse_q));
                        if (DBIAS) dsum[kt] += acc2;        // same (query, key) slot in every window: scattered to the bias bins once per workgroup
                        dst[u] = acc2;
                    }
                    const bf16x8 df = attn_pack(dst[0], dst[1]);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const bf16x8 kf = attn_lds_tr_frag(Ks, RS, dt * 16, kp * 32, lr, lg);
                        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, df, dq[dt], 0, 0, 0);
                    }
                }
                if (tok_own >= 0) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        bf16x4 ov = {(bf16)(dq[dt][0] * p.scale), (bf16)(dq[dt][1] * p.scale), (bf16)(dq[dt][2] * p.scale), (bf16)(dq[dt][3] * p.scale)};
                        *reinterpret_cast<bf16x4*>(p.dqkv + (int64_t)tok_own * p.ld + qcol + dt * 16 + lg * 4) = ov;
                    }
                }
            }
            // ---- phase 2: this wave's key tile -> dK, dV ----
            {
                bf16x8 kf[KS], vf[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    kf[ks] = attn_lds_row_frag(Ks, RS, ti, ks * 32 + lg * 8);
                    vf[ks] = attn_lds_row_frag(Vs, RS, ti, ks * 32 + lg * 8);
                }
                f32x4 dk[DT], dv[DT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int qp = 0; qp < NKT / 2; ++qp) {
                    f32x4 pt[2], dst[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int qt = qp * 2 + u;
                        const int q0 = qt * 16 + lg * 4;
                        const f32x4 l4 = *reinterpret_cast<const f32x4*>(&row_lse[sel][q0]);
                        f32x4 acc = attn_bias_cvt(bk[qt]), acc2 = *reinterpret_cast<const f32x4*>(&row_delta[sel][q0]);
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            const bf16x8 qf = attn_lds_row_frag(Qs, RS, qt * 16 + lr, ks * 32 + lg * 8);
                            const bf16x8 dof = attn_lds_row_frag(dOs, RS, qt * 16 + lr, ks * 32 + lg * 8);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf[ks], acc, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof, vf[ks], acc2, 0, 0, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pr = __builtin_amdgcn_exp2f(fmaf(acc[r], c2, l4[r]));
                            acc[r] = pr;
                            acc2[r] *= pr;
                        }
                        pt[u] = acc;
                        dst[u] = acc2;
                    }
                    const bf16x8 pf = attn_pack(pt[0], pt[1]);
                    const bf16x8 df = attn_pack(dst[0], dst[1]);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const bf16x8 qtf = attn_lds_tr_frag(Qs, RS, dt * 16, qp * 32, lr, lg);
                        const bf16x8 dotf = attn_lds_tr_frag(dOs, RS, dt * 16, qp * 32, lr, lg);
                        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, df, dk[dt], 0, 0, 0);
                        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dotf, pf, dv[dt], 0, 0, 0);
                    }
                }
                if (tok_own >= 0) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        bf16x4 a = {(bf16)(dk[dt][0] * p.scale), (bf16)(dk[dt][1] * p.scale), (bf16)(dk[dt][2] * p.scale), (bf16)(dk[dt][3] * p.scale)};
                        bf16x4 b = {(bf16)dv[dt][0], (bf16)dv[dt][1], (bf16)dv[dt][2], (bf16)dv[dt][3]};
                        *reinterpret_cast<bf16x4*>(p.dqkv + (int64_t)tok_own * p.ld + kcol + dt * 16 + lg * 4) = a;
                        *reinterpret_cast<bf16x4*>(p.dqkv + (int64_t)tok_own * p.ld + vcol + dt * 16 + lg * 4) = b;
                    }
                }
            }
        }
    }
    if (DBIAS) {
        if (wave < nt && ti < p.N) {           // LDS float atomics retire about one lane per clock: once per workgroup, not once per window
            const int qci = ci[ti], qcj = cj[ti];
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const int key0 = kt * 16 + lg * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (key0 + r < p.N) atomicAdd(&dbias_s[(lr & (DBC - 1)) * 256 + abs(qci - (int)ci[key0 + r]) * p.ws + abs(qcj - (int)cj[key0 + r])], dsum[kt][r]);
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < p.nbias; t += blockDim.x) {
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < DBC; ++c) sum += dbias_s[c * 256 + t];
            if (p.dbias_part) p.dbias_part[((int64_t)wb * p.nh + h) * p.nbias + t] = sum;      // one row per (window group, head)
            else atomicAdd(&p.dbias[h * p.nbias + t], sum);
        }
    }
}

// dbias[h][t] += sum over the (folded) window rows of part[row][h][t]
__global__ void attn_dbias_final_kernel(const float* __restrict__ rows, int nrows, int W, float* __restrict__ dbias) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W) return;
    double s = 0.0;
    for (int r = 0; r < nrows; ++r) s += (double)rows[(int64_t)r * W + i];
    dbias[i] += (float)s;
}

// ------------------------------------------------------------------------------------------- host
static int attn_fill(AttnParams& p, const GgAttnArgs* a, const char* who) {
    GG_CHECK(a && a->qkv, "%s: null qkv", who);
    GG_CHECK(a->head_dim == 32 || a->head_dim == 64, "%s: head_dim must be 32 or 64 (got %d)", who, a->head_dim);
    GG_CHECK(a->tokens_per_window > 0 && a->tokens_per_window <= 256,
             "%s: %d tokens per window unsupported (max 256: full-row softmax in registers)", who, a->tokens_per_window);
    GG_CHECK(a->num_windows > 0 && a->num_heads > 0, "%s: bad window/head count", who);
    GG_CHECK((a->ld & 7) == 0 && (a->q_off & 7) == 0 && (a->k_off & 7) == 0 && (a->v_off & 7) == 0 && (a->head_stride & 7) == 0,
             "%s: qkv offsets/strides must be multiples of 8 elements", who);
    if (a->window_size > 0) {
        GG_CHECK(a->window_size * a->window_size == a->tokens_per_window, "%s: window_size^2 != tokens_per_window", who);
        GG_CHECK(a->map_h % a->window_size == 0 && a->map_w % a->window_size == 0, "%s: map not divisible by window", who);
        GG_CHECK(a->num_windows % ((a->map_h / a->window_size) * (a->map_w / a->window_size)) == 0, "%s: window count", who);
        GG_CHECK(a->window_size <= 16, "%s: window_size > 16 unsupported", who);
    }
    if (a->bias || a->dbias) GG_CHECK(a->window_size > 0 && a->tokens_per_window <= 256, "%s: bias needs a window geometry", who);
    if (a->bias) GG_CHECK(((uintptr_t)a->bias & 7) == 0, "%s: expanded bias must be 8-byte aligned", who);
    p.qkv = (const bf16*)a->qkv; p.ld = a->ld;
    p.q_off = a->q_off; p.k_off = a->k_off; p.v_off = a->v_off; p.head_stride = a->head_stride;
    p.out = (bf16*)a->out; p.ldo = a->ldo;
    p.bias = (const bf16*)a->bias; p.nbias = a->window_size * a->window_size;
    p.ws = a->window_size;
    p.ws_inv = a->window_size ? (65536 + a->window_size - 1) / a->window_size : 0;
    p.H = a->map_h; p.W = a->map_w;
    p.nWx = a->window_size ? a->map_w / a->window_size : 1;
    p.nWy = a->window_size ? a->map_h / a->window_size : 1;
    p.N = a->tokens_per_window; p.nh = a->num_heads; p.scale = a->scale;
    p.dout = (const bf16*)a->dout; p.lddo = a->lddo; p.dqkv = (bf16*)a->dqkv; p.dbias = a->dbias; p.dbias_part = a->dbias ? a->dbias_scratch : nullptr; p.lse = a->lse;
    return 0;
}
static int attn_nkt(int N) { return N <= 64 ? 4 : (N <= 160 ? 10 : (N <= 224 ? 14 : 16)); }
extern "C" int gg_attention_padded_tokens(int tokens_per_window) { return 16 * attn_nkt(tokens_per_window); }
extern "C" int gg_attention_expand_bias(const float* table, int num_heads, int window_size, float scale, void* full, void* stream) {
    GG_CHECK(table && full && num_heads > 0 && window_size > 0 && window_size <= 16 && scale > 0.f, "gg_attention_expand_bias: bad args");
    const int N = window_size * window_size, Np = 16 * attn_nkt(N);
    const int64_t total = (int64_t)num_heads * Np * Np;
    hipLaunchKernelGGL(attn_expand_bias_kernel, dim3((unsigned)std::min<int64_t>(gg_cdiv(total, 256), 4096)), dim3(256), 0, (hipStream_t)stream,
                       table, num_heads, window_size, N, Np, 1.0f / scale, (bf16*)full);
    GG_LAUNCH_CHECK();
    return 0;
}

// beyond 256 tokens per window (tiny_vit_21m_384 / _512 stage 2, CLIP ViT-L/14-336) a score row no longer fits a wave's registers:
// those shapes run on the online-softmax kernels of attention_flash.hip (bias from the compact f32 table)
static bool attn_use_flash(const GgAttnArgs* a) { return a && (a->tokens_per_window > 256 || a->window_size > 16); }
// The BACKWARD of 12 x 12 / 14 x 14 windows of head dim 32 also leaves this file: attention_split.h's single-pass kernel (bf16 storage, one plane) reads q, k, v,
// dO, O once and forms five products instead of the seven of attn_bwd_kernel's two phases, and gathers the bias from the compact f32 table in LDS instead of
// reading the expanded [heads][Np][Np] table from memory twice per workgroup (590 against 756 us per 14 x 14 layer; the expanded table's re-reads were the
// 1.49x HBM traffic of round 4's bf16 attention class).  7 x 7 windows stay on the grouped kernels below, every forward stays here.
static bool attn_use_split_bwd(const GgAttnArgs* a) {
    static const bool nosplit = gg_dev_env("GG_ATTN_NO_SPLIT") != nullptr;
    if (!a || nosplit || a->head_dim != 32 || a->window_size <= 0) return false;
    const int nt16 = (a->tokens_per_window + 15) / 16;
    return (nt16 == 13 || nt16 == 9) && (!a->bias || a->bias_table);
}
extern "C" int gg_attention_fwd(const GgAttnArgs* a, void* stream) {
    if (attn_use_flash(a)) {
        GG_CHECK(!a->bias || a->bias_table, "gg_attention_fwd: this window shape takes the bias as bias_table (compact f32)");
        return gg_attention_flash_fwd(a, 0, stream);
    }
    AttnParams p;
    GG_TRY(attn_fill(p, a, "gg_attention_fwd"));
    GG_CHECK(a->out && (a->ldo & 3) == 0, "gg_attention_fwd: bad out");
    dim3 grid((unsigned)(a->num_windows * a->num_heads)), block(256);
    hipStream_t s = (hipStream_t)stream;
    GG_PROF(GG_CAT_ATTN, 4.0 * a->num_windows * a->num_heads * (double)p.N * p.N * a->head_dim, 8.0 * a->num_windows * a->num_heads * (double)p.N * a->head_dim, stream);
    const int nkt = attn_nkt(p.N);
#define GG_FWD(D_, K_) hipLaunchKernelGGL((attn_fwd_kernel<D_, K_>), grid, block, 0, s, p)
    static const char* small_env = gg_dev_env("GG_ATTN_SMALL");
    if (a->head_dim == 32 && nkt == 4 && !(small_env && small_env[0] == '0') && a->num_windows >= 64) {
        constexpr int GW = 8;          // windows per workgroup
        const dim3 g2((unsigned)(gg_cdiv(a->num_windows, GW) * a->num_heads));
        hipLaunchKernelGGL((attn_fwd_small_kernel<32, GW>), g2, block, 0, s, p, a->num_windows);
        GG_LAUNCH_CHECK();
        return 0;
    }
    if (a->head_dim == 32) {
        if (nkt == 4) GG_FWD(32, 4); else if (nkt == 10) GG_FWD(32, 10); else if (nkt == 14) GG_FWD(32, 14); else GG_FWD(32, 16);
    } else {
        if (nkt == 4) GG_FWD(64, 4); else if (nkt == 10) GG_FWD(64, 10); else if (nkt == 14) GG_FWD(64, 14); else GG_FWD(64, 16);
    }
#undef GG_FWD
    GG_LAUNCH_CHECK();
    return 0;
}
// fp16 twin of the forward (inference of the CLIP tower's fp16 mode: pretrain/clip_embedder.py:63-65 at BASELINE config c4's precision): fp16 storage,
// QK^T and PV on v_mfma_f32_16x16x32_f16, f32 accumulation and softmax.  No relative-position bias (the CLIP tower has none); sequences beyond 256
// tokens go to the online-softmax kernel (fp16 storage, f32 arithmetic).
extern "C" int gg_attention_fwd_f16(const GgAttnArgs* a, void* stream) {
    GG_CHECK(a && !a->bias && !a->bias_table, "gg_attention_fwd_f16: the fp16 forward takes no bias");
    if (attn_use_flash(a)) return gg_attention_flash_fwd(a, 2, stream);
    AttnParams p;
    GG_TRY(attn_fill(p, a, "gg_attention_fwd_f16"));
    GG_CHECK(a->out && (a->ldo & 3) == 0, "gg_attention_fwd_f16: bad out");
    dim3 grid((unsigned)(a->num_windows * a->num_heads)), block(256);
    hipStream_t s = (hipStream_t)stream;
    GG_PROF(GG_CAT_ATTN, 4.0 * a->num_windows * a->num_heads * (double)p.N * p.N * a->head_dim, 8.0 * a->num_windows * a->num_heads * (double)p.N * a->head_dim, stream);
    const int nkt = attn_nkt(p.N);
#define GG_FWD16(D_, K_) hipLaunchKernelGGL((attn_fwd_kernel<D_, K_, f16>), grid, block, 0, s, p)
    if (a->head_dim == 32) {
        if (nkt == 4) GG_FWD16(32, 4); else if (nkt == 10) GG_FWD16(32, 10); else if (nkt == 14) GG_FWD16(32, 14); else GG_FWD16(32, 16);
    } else {
        if (nkt == 4) GG_FWD16(64, 4); else if (nkt == 10) GG_FWD16(64, 10); else if (nkt == 14) GG_FWD16(64, 14); else GG_FWD16(64, 16);
    }
#undef GG_FWD16
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_attention_bwd(const GgAttnArgs* a, void* stream) {
    if (attn_use_flash(a) || attn_use_split_bwd(a)) {
        GG_CHECK(!a->bias || a->bias_table, "gg_attention_bwd: this window shape takes the bias as bias_table (compact f32)");
        return gg_attention_flash_bwd_impl(a, 0, 1, stream);      // (this forward added the bias as bf16(bias / scale): the backward recomputes P with that value)
    }
    AttnParams p;
    GG_TRY(attn_fill(p, a, "gg_attention_bwd"));
    GG_CHECK(a->dout && a->dqkv && (a->lddo & 7) == 0, "gg_attention_bwd: bad dout/dqkv");
    GG_CHECK(a->lse && a->out && (a->ldo & 7) == 0, "gg_attention_bwd: needs the forward's lse and out");
    GG_CHECK(a->head_dim == 32, "gg_attention_bwd: only head_dim 32 (TinyViT) is built");
    dim3 grid((unsigned)(a->num_windows * a->num_heads)), block(256);
    hipStream_t s = (hipStream_t)stream;
    GG_PROF(GG_CAT_ATTN, 10.0 * a->num_windows * a->num_heads * (double)p.N * p.N * a->head_dim, 16.0 * a->num_windows * a->num_heads * (double)p.N * a->head_dim, stream);
    const int nkt = attn_nkt(p.N);
#define GG_BWD(K_)                                                                                       \
    do {                                                                                                 \
        if (p.dbias) hipLaunchKernelGGL((attn_bwd_kernel<32, K_, true>), grid, block, 0, s, p);          \
        else hipLaunchKernelGGL((attn_bwd_kernel<32, K_, false>), grid, block, 0, s, p);                 \
    } while (0)
    static const char* small_env = gg_dev_env("GG_ATTN_SMALL");
    int part_rows = a->num_windows;
    if (nkt == 4 && !(small_env && small_env[0] == '0') && a->num_windows >= 64) {
        constexpr int GW = 8;
        part_rows = (int)gg_cdiv(a->num_windows, GW);
        const dim3 g2((unsigned)(part_rows * a->num_heads));
        if (p.dbias) hipLaunchKernelGGL((attn_bwd_small_kernel<32, GW, true>), g2, block, 0, s, p, a->num_windows);
        else hipLaunchKernelGGL((attn_bwd_small_kernel<32, GW, false>), g2, block, 0, s, p, a->num_windows);
    } else if (nkt == 4) GG_BWD(4); else if (nkt == 10) GG_BWD(10); else if (nkt == 14) GG_BWD(14); else GG_BWD(16);
#undef GG_BWD
    if (p.dbias && p.dbias_part) {
        const int Wd = p.nh * p.nbias;
        const float* rows; int nrows;
        gg_reduce_rows(p.dbias_part, part_rows, Wd, s, &rows, &nrows);
        hipLaunchKernelGGL(attn_dbias_final_kernel, dim3((unsigned)gg_cdiv(Wd, 256)), dim3(256), 0, s, rows, nrows, Wd, p.dbias);
    }
    GG_LAUNCH_CHECK();
    return 0;
}
