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
#include "../../include/gg.h"

struct AttnParams {
    const bf16* qkv; int64_t ld;
    int q_off, k_off, v_off, head_stride;
    bf16* out; int64_t ldo;           // forward output [tokens, ldo], head h at column h*D
    const float* bias; int nbias;     // EXPANDED bias / scale [nh][Np][Np] f32 (gg_attention_expand_bias; padded keys = -inf) or null;
                                      // nbias = ws*ws = size of the table the bias gradient is reduced into
    int ws, nWx, nWy, H, W;           // ws > 0: windows of ws x ws tokens inside an H x W map; ws == 0: linear
    int N;                            // tokens per window
    int nh;
    float scale;
    // backward
    const bf16* dout; int64_t lddo;   // [tokens, lddo]
    bf16* dqkv;                       // same layout as qkv
    float* dbias;                     // [nh][nbias] accumulated with atomics, or null
    float* lse;                       // [tokens][nh] log-sum-exp of the scaled+biased scores (fwd writes, bwd reads)
};

__device__ __forceinline__ int attn_token(const AttnParams& p, int w, int t) {
    if (t >= p.N) return -1;
    if (p.ws == 0) return w * p.N + t;
    const int per_img = p.nWx * p.nWy;
    const int b = w / per_img, r = w % per_img;
    const int wy = r / p.nWx, wx = r % p.nWx;
    const int i = t / p.ws, j = t % p.ws;
    return (b * p.H + wy * p.ws + i) * p.W + wx * p.ws + j;
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
// row-major staging of a [N, D] column block of `src` into X[Np][RS] (zero rows for padded tokens).  All of a thread's
// 16-byte loads are issued before the first LDS write (one memory round trip instead of one per chunk).
template <int D, int Np>
__device__ __forceinline__ void attn_stage_rows(bf16* X, int RS, const bf16* src, int64_t ld, int col, const int* tok) {
    constexpr int CH = D / 8;
    constexpr int IT = (Np * CH + 255) / 256;
    bf16x8 v[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int idx = threadIdx.x + i * 256;
        const int key = min(idx / CH, Np - 1), dc = idx % CH;
        v[i] = attn_row_frag(src, ld, tok[key], col + dc * 8);
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int idx = threadIdx.x + i * 256;
        if (idx < Np * CH) *reinterpret_cast<bf16x8*>(X + (idx / CH) * RS + (idx % CH) * 8) = v[i];
    }
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
// transposed staging of a [N, D] column block of `src` into T[D][stride] (zero for padded keys)
template <int D>
__device__ __forceinline__ void attn_stage_transposed(bf16* T, int stride, const bf16* src, int64_t ld, int col, const int* tok,
                                                      int Np) {
    constexpr int CH = D / 8;
    for (int idx = threadIdx.x; idx < Np * CH; idx += blockDim.x) {
        const int key = idx / CH, dc = idx % CH;
        const bf16x8 v = attn_row_frag(src, ld, tok[key], col + dc * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) T[(dc * 8 + j) * stride + key] = v[j];
    }
}

// attention_biases[h][|di|*ws+|dj|] -> full[h][q][k] (Np x Np, row-major): the kernels then fetch 4 consecutive keys of a
// query row with one 16-byte load instead of 4 x (index arithmetic + LDS gather).  Padded keys carry -inf so no mask
// select is needed; the matrix is symmetric, which the backward's [query][key] orientation uses.
__global__ void attn_expand_bias_kernel(const float* __restrict__ table, int nh, int ws, int N, int Np, float inv_scale,
                                        float* __restrict__ full) {
    const int64_t total = (int64_t)nh * Np * Np;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % Np), q = (int)((i / Np) % Np), h = (int)(i / ((int64_t)Np * Np));
        float v = -INFINITY;
        if (k < N) {
            const int qq = min(q, N - 1);
            v = table[h * ws * ws + abs(qq / ws - k / ws) * ws + abs(qq % ws - k % ws)] * inv_scale;
        }
        full[i] = v;
    }
}

// ------------------------------------------------------------------------------------------- forward
template <int D, int NKT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
    constexpr int Np = NKT * 16;
    constexpr int RS = D + 8;     // 80 / 144-byte rows: 16-B aligned fragments, 8-B aligned transposing reads
    constexpr int KS = D / 32;    // MFMA k-steps over the head dim
    constexpr int DT = D / 16;    // output d tiles
    __shared__ __attribute__((aligned(16))) bf16 Ks[Np * RS];
    __shared__ __attribute__((aligned(16))) bf16 Vs[Np * RS];
    __shared__ int tok[Np];

    const int w = blockIdx.x / p.nh, h = blockIdx.x % p.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const float* bias_h = p.bias ? p.bias + (int64_t)h * Np * Np : nullptr;

    for (int t = threadIdx.x; t < Np; t += blockDim.x) tok[t] = attn_token(p, w, t);
    __syncthreads();
    attn_stage_rows<D, Np>(Ks, RS, p.qkv, p.ld, p.k_off + h * p.head_stride, tok);
    attn_stage_rows<D, Np>(Vs, RS, p.qkv, p.ld, p.v_off + h * p.head_stride, tok);
    __syncthreads();

    const int nqt = (p.N + 15) / 16;
    for (int qt = wave; qt < nqt; qt += 4) {
        const int qi = qt * 16 + lr;
        const int qtok = tok[qi];
        bf16x8 qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = attn_row_frag(p.qkv, p.ld, qtok, p.q_off + h * p.head_stride + ks * 32 + lg * 8);
        // The score accumulators START as the relative-position bias (the expanded table holds bias / scale, -inf on padded
        // keys), so the bias costs no VALU work and all NKT tile loads are in flight together before the first MFMA.
        f32x4 s[NKT];
        if (bias_h) {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) s[kt] = *reinterpret_cast<const f32x4*>(bias_h + (int64_t)qi * Np + kt * 16 + lg * 4);
        } else {
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kt][r] = (kt * 16 + lg * 4 + r < p.N) ? 0.f : -INFINITY;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kf = attn_lds_row_frag(Ks, RS, kt * 16 + lr, ks * 32 + lg * 8);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[kt], 0, 0, 0);   // D[i=key][j=query]
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float c2 = p.scale * 1.4426950408889634f, mb = mx * c2;       // exp(scale*(s - mx)) = 2^(s*c2 - mx*c2)
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(s[kt][r], c2, -mb));
                s[kt][r] = e;
                l += e;
            }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);

        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < NKT / 2; ++kp) {
            const bf16x8 pf = attn_pack(s[2 * kp], s[2 * kp + 1]);   // k = key 32kp + 16(jj>>2) + 4lg + (jj&3)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16x8 vf = attn_lds_tr_frag(Vs, RS, dt * 16, kp * 32, lr, lg);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);   // D[i=d][j=query]
            }
        }
        if (qtok >= 0) {
            const float inv = 1.f / l;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                bf16x4 ov = {(bf16)(o[dt][0] * inv), (bf16)(o[dt][1] * inv), (bf16)(o[dt][2] * inv), (bf16)(o[dt][3] * inv)};
                *reinterpret_cast<bf16x4*>(p.out + (int64_t)qtok * p.ldo + h * D + dt * 16 + lg * 4) = ov;
            }
            if (p.lse && lg == 0) p.lse[(int64_t)qtok * p.nh + h] = mx * p.scale + __logf(l);
        }
    }
}

// ------------------------------------------------------------------------------------------- backward
// Flash-style: P is recomputed from the forward's per-row log-sum-exp, delta_q = sum_d dO[q][d] O[q][d] comes from the
// saved forward output, so no score row has to stay in registers (3+ waves per SIMD instead of 1).
template <int D, int NKT>
__global__ __launch_bounds__(256) void attn_bwd_kernel(AttnParams p) {
    constexpr int Np = NKT * 16;
    constexpr int RS = D + 8;
    constexpr int KS = D / 32;
    constexpr int DT = D / 16;
    __shared__ __attribute__((aligned(16))) bf16 Qs[Np * RS];
    __shared__ __attribute__((aligned(16))) bf16 Ks[Np * RS];
    __shared__ __attribute__((aligned(16))) bf16 Vs[Np * RS];
    __shared__ __attribute__((aligned(16))) bf16 dOs[Np * RS];
    __shared__ __attribute__((aligned(16))) float row_lse[Np], row_delta[Np];
    __shared__ int tok[Np];
    __shared__ float dbias_s[256];
    __shared__ __attribute__((aligned(4))) unsigned char ci[Np], cj[Np];   // window coordinates: bias-gradient binning only

    const int w = blockIdx.x / p.nh, h = blockIdx.x % p.nh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    const int qcol = p.q_off + h * p.head_stride, kcol = p.k_off + h * p.head_stride, vcol = p.v_off + h * p.head_stride;
    const int ocol = h * D;

    for (int t = threadIdx.x; t < Np; t += blockDim.x) {
        const int tk = attn_token(p, w, t);
        tok[t] = tk;
        const int tt = min(t, p.N - 1);
        ci[t] = p.ws ? (unsigned char)(tt / p.ws) : 0;
        cj[t] = p.ws ? (unsigned char)(tt % p.ws) : 0;
        row_lse[t] = tk >= 0 ? p.lse[(int64_t)tk * p.nh + h] : 0.f;
    }
    for (int t = threadIdx.x; t < 256; t += blockDim.x) dbias_s[t] = 0.f;
    const float* bias_h = p.bias ? p.bias + (int64_t)h * Np * Np : nullptr;
    __syncthreads();
    attn_stage_rows<D, Np>(Qs, RS, p.qkv, p.ld, qcol, tok);
    attn_stage_rows<D, Np>(Ks, RS, p.qkv, p.ld, kcol, tok);
    attn_stage_rows<D, Np>(Vs, RS, p.qkv, p.ld, vcol, tok);
    {   // dO rows + delta = rowsum(dO * O)
        constexpr int CH = D / 8;
        for (int idx = threadIdx.x; idx < Np * CH; idx += blockDim.x) {       // Np*CH is a multiple of 64: full waves
            const int row = idx / CH, dc = idx % CH;
            const bf16x8 dv = attn_row_frag(p.dout, p.lddo, tok[row], ocol + dc * 8);
            const bf16x8 ov = attn_row_frag(p.out, p.ldo, tok[row], ocol + dc * 8);
            *reinterpret_cast<bf16x8*>(dOs + row * RS + dc * 8) = dv;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (float)dv[j] * (float)ov[j];
#pragma unroll
            for (int o = 1; o < CH; o <<= 1) s += __shfl_xor(s, o, 64);
            if (dc == 0) row_delta[row] = s;
        }
    }
    __syncthreads();

    const int nt = (p.N + 15) / 16;
    // ---- phase 1: a wave owns a query tile -> dQ (and dbias) ----
    for (int qt = wave; qt < nt; qt += 4) {
        const int qi = qt * 16 + lr;
        const int qtok = tok[qi];
        bf16x8 qf[KS], dof[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[ks] = attn_lds_row_frag(Qs, RS, qi, ks * 32 + lg * 8);
            dof[ks] = attn_lds_row_frag(dOs, RS, qi, ks * 32 + lg * 8);
        }
        const int qci = ci[qi], qcj = cj[qi];
        const float lse_q = row_lse[qi], delta_q = row_delta[qi];
        f32x4 dq[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float c2 = p.scale * 1.4426950408889634f, lq2 = lse_q * 1.4426950408889634f;
        // score accumulators start as bias / scale (see forward); the next pair of tiles is fetched one iteration ahead
        auto bias_tiles = [&](int kp, f32x4 (&b)[2]) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int key0 = (kp * 2 + u) * 16 + lg * 4;
                if (bias_h) b[u] = *reinterpret_cast<const f32x4*>(bias_h + (int64_t)qi * Np + key0);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) b[u][r] = (key0 + r < p.N) ? 0.f : -INFINITY;
                }
            }
        };
        f32x4 bnext[2];
        bias_tiles(0, bnext);
#pragma unroll 1
        for (int kp = 0; kp < NKT / 2; ++kp) {
            f32x4 dst[2];
            f32x4 bcur[2] = {bnext[0], bnext[1]};
            if (kp + 1 < NKT / 2) bias_tiles(kp + 1, bnext);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int kt = kp * 2 + u;
                f32x4 acc = bcur[u], acc2 = {0.f, 0.f, 0.f, 0.f};
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
                    const float pr = __builtin_amdgcn_exp2f(fmaf(acc[r], c2, -lq2));      // padded keys: 2^-inf = 0
                    acc2[r] = pr * (acc2[r] - delta_q);
                }
                if (p.dbias && qtok >= 0) {     // bias gradient, binned by |di|,|dj| (only the trainable last stage takes this path)
                    const uchar4 kci = *reinterpret_cast<const uchar4*>(&ci[key0]);
                    const uchar4 kcj = *reinterpret_cast<const uchar4*>(&cj[key0]);
                    const int kcis[4] = {kci.x, kci.y, kci.z, kci.w}, kcjs[4] = {kcj.x, kcj.y, kcj.z, kcj.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (key0 + r < p.N) atomicAdd(&dbias_s[abs(qci - kcis[r]) * p.ws + abs(qcj - kcjs[r])], acc2[r]);
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
    // ---- phase 2: a wave owns a key tile -> dK, dV ----
    for (int kt = wave; kt < nt; kt += 4) {
        const int ki = kt * 16 + lr;
        const int ktok = tok[ki];
        bf16x8 kf[KS], vf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kf[ks] = attn_lds_row_frag(Ks, RS, ki, ks * 32 + lg * 8);
            vf[ks] = attn_lds_row_frag(Vs, RS, ki, ks * 32 + lg * 8);
        }
        const bool kvalid = ki < p.N;
        f32x4 dk[DT], dv[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        const float c2 = p.scale * 1.4426950408889634f;
        const int64_t brow = (int64_t)min(ki, p.N - 1) * Np;      // symmetric table: bias[q][k] == bias[k][q]; columns >= N hold -inf
        auto bias_tiles = [&](int qp, f32x4 (&b)[2]) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q0 = (qp * 2 + u) * 16 + lg * 4;
                if (bias_h) b[u] = *reinterpret_cast<const f32x4*>(bias_h + brow + q0);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) b[u][r] = (q0 + r < p.N) ? 0.f : -INFINITY;
                }
            }
        };
        f32x4 bnext[2];
        bias_tiles(0, bnext);
#pragma unroll 1
        for (int qp = 0; qp < NKT / 2; ++qp) {
            f32x4 pt[2], dst[2];
            f32x4 bcur[2] = {bnext[0], bnext[1]};
            if (qp + 1 < NKT / 2) bias_tiles(qp + 1, bnext);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int qt = qp * 2 + u;
                f32x4 acc = bcur[u], acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 qf = attn_lds_row_frag(Qs, RS, qt * 16 + lr, ks * 32 + lg * 8);
                    const bf16x8 dof = attn_lds_row_frag(dOs, RS, qt * 16 + lr, ks * 32 + lg * 8);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf[ks], acc, 0, 0, 0);     // S  [query][key]
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof, vf[ks], acc2, 0, 0, 0);  // dP [query][key]
                }
                const int q0 = qt * 16 + lg * 4;
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(&row_lse[q0]);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(&row_delta[q0]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // padded queries: -inf score -> 0;  padded keys (kvalid false) produce rows that are never stored
                    const float pr = kvalid ? __builtin_amdgcn_exp2f(fmaf(acc[r], c2, -l4[r] * 1.4426950408889634f)) : 0.f;
                    acc[r] = pr;
                    acc2[r] = pr * (acc2[r] - d4[r]);      // rows >= N: lse = delta = 0 (initialised), pr = 0
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
    if (p.dbias) {
        __syncthreads();
        for (int t = threadIdx.x; t < p.nbias; t += blockDim.x) atomicAdd(&p.dbias[h * p.nbias + t], dbias_s[t]);
    }
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
    if (a->bias) GG_CHECK(((uintptr_t)a->bias & 15) == 0, "%s: expanded bias must be 16-byte aligned", who);
    p.qkv = (const bf16*)a->qkv; p.ld = a->ld;
    p.q_off = a->q_off; p.k_off = a->k_off; p.v_off = a->v_off; p.head_stride = a->head_stride;
    p.out = (bf16*)a->out; p.ldo = a->ldo;
    p.bias = a->bias; p.nbias = a->window_size * a->window_size;
    p.ws = a->window_size;
    p.H = a->map_h; p.W = a->map_w;
    p.nWx = a->window_size ? a->map_w / a->window_size : 1;
    p.nWy = a->window_size ? a->map_h / a->window_size : 1;
    p.N = a->tokens_per_window; p.nh = a->num_heads; p.scale = a->scale;
    p.dout = (const bf16*)a->dout; p.lddo = a->lddo; p.dqkv = (bf16*)a->dqkv; p.dbias = a->dbias; p.lse = a->lse;
    return 0;
}
static int attn_nkt(int N) { return N <= 64 ? 4 : (N <= 160 ? 10 : (N <= 224 ? 14 : 16)); }
extern "C" int gg_attention_padded_tokens(int tokens_per_window) { return 16 * attn_nkt(tokens_per_window); }
extern "C" int gg_attention_expand_bias(const float* table, int num_heads, int window_size, float scale, float* full, void* stream) {
    GG_CHECK(table && full && num_heads > 0 && window_size > 0 && window_size <= 16 && scale > 0.f, "gg_attention_expand_bias: bad args");
    const int N = window_size * window_size, Np = 16 * attn_nkt(N);
    const int64_t total = (int64_t)num_heads * Np * Np;
    hipLaunchKernelGGL(attn_expand_bias_kernel, dim3((unsigned)std::min<int64_t>(gg_cdiv(total, 256), 4096)), dim3(256), 0, (hipStream_t)stream,
                       table, num_heads, window_size, N, Np, 1.0f / scale, full);
    GG_LAUNCH_CHECK();
    return 0;
}

extern "C" int gg_attention_fwd(const GgAttnArgs* a, void* stream) {
    AttnParams p;
    GG_TRY(attn_fill(p, a, "gg_attention_fwd"));
    GG_CHECK(a->out && (a->ldo & 3) == 0, "gg_attention_fwd: bad out");
    dim3 grid((unsigned)(a->num_windows * a->num_heads)), block(256);
    hipStream_t s = (hipStream_t)stream;
    GG_PROF(GG_CAT_ATTN, 4.0 * a->num_windows * a->num_heads * (double)p.N * p.N * a->head_dim, 8.0 * a->num_windows * a->num_heads * (double)p.N * a->head_dim, stream);
    const int nkt = attn_nkt(p.N);
#define GG_FWD(D_, K_) hipLaunchKernelGGL((attn_fwd_kernel<D_, K_>), grid, block, 0, s, p)
    if (a->head_dim == 32) {
        if (nkt == 4) GG_FWD(32, 4); else if (nkt == 10) GG_FWD(32, 10); else if (nkt == 14) GG_FWD(32, 14); else GG_FWD(32, 16);
    } else {
        if (nkt == 4) GG_FWD(64, 4); else if (nkt == 10) GG_FWD(64, 10); else if (nkt == 14) GG_FWD(64, 14); else GG_FWD(64, 16);
    }
#undef GG_FWD
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_attention_bwd(const GgAttnArgs* a, void* stream) {
    AttnParams p;
    GG_TRY(attn_fill(p, a, "gg_attention_bwd"));
    GG_CHECK(a->dout && a->dqkv && (a->lddo & 7) == 0, "gg_attention_bwd: bad dout/dqkv");
    GG_CHECK(a->lse && a->out && (a->ldo & 7) == 0, "gg_attention_bwd: needs the forward's lse and out");
    GG_CHECK(a->head_dim == 32, "gg_attention_bwd: only head_dim 32 (TinyViT) is built");
    dim3 grid((unsigned)(a->num_windows * a->num_heads)), block(256);
    hipStream_t s = (hipStream_t)stream;
    GG_PROF(GG_CAT_ATTN, 10.0 * a->num_windows * a->num_heads * (double)p.N * p.N * a->head_dim, 16.0 * a->num_windows * a->num_heads * (double)p.N * a->head_dim, stream);
    const int nkt = attn_nkt(p.N);
#define GG_BWD(K_) hipLaunchKernelGGL((attn_bwd_kernel<32, K_>), grid, block, 0, s, p)
    if (nkt == 4) GG_BWD(4); else if (nkt == 10) GG_BWD(10); else if (nkt == 14) GG_BWD(14); else GG_BWD(16);
#undef GG_BWD
    GG_LAUNCH_CHECK();
    return 0;
}
