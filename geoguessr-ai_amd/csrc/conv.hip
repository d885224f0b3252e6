// Spatial (3x3) kernels of PatchEmbed / MBConv / PatchMerging / local_conv on NHWC bf16 activations.
// All of them are HBM-bound byte movers: 16-byte (8 x bf16) accesses along the channel axis, one
// thread per (pixel, 8-channel group).  Dense 3x3 convs become im2col + the MFMA GEMM; depthwise
// convs are computed here directly with fp32 taps.
#include "common.h"
#include "../../include/gg.h"
#include <string.h>

// ---------------------------------------------------------------- im2col (dense 3x3, pad 1)
// x f32 NCHW (B,3,H,W) -> col bf16 [B*Ho*Wo, 32]; k = (ky*3+kx)*3 + ci for k < 27, zeros above.
__global__ __launch_bounds__(256) void im2col_nchw3_kernel(const float* __restrict__ x, bf16* __restrict__ col, int B, int H,
                                                           int W, int Ho, int Wo, int stride) {
    // thread = (output pixel, 16-byte quarter of its 64-byte col row): a wave's store is 1 KiB contiguous
    const int64_t total = (int64_t)B * Ho * Wo * 4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned pu = (unsigned)(i >> 2);
        const int q = (int)(i & 3);
        const int ox = (int)(pu % (unsigned)Wo);
        const int oy = (int)((pu / (unsigned)Wo) % (unsigned)Ho);
        const int b = (int)(pu / ((unsigned)Wo * (unsigned)Ho));
        const float* xb = x + (int64_t)b * 3 * H * W;
        bf16x8 t;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = q * 8 + j;
            const int tap = (k * 11) >> 5, ci = k - tap * 3;          // k / 3, k % 3 for k < 32
            const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
            const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
            const bool ok = k < 27 && iy >= 0 && iy < H && ix >= 0 && ix < W;
            t[j] = ok ? (bf16)xb[((int64_t)ci * H + iy) * W + ix] : (bf16)0.f;
        }
        *reinterpret_cast<bf16x8*>(col + (int64_t)pu * 32 + q * 8) = t;
    }
}

// Same result through LDS: block = (image, band of RB output rows); the 2*RB+1 input rows of the 3 planes are read once as
// full-width 16-byte loads (the per-pixel gather above issues 27 strided 4-byte loads per output pixel and is TA-bound at
// 2.4 TB/s), then every (pixel, 16-byte quarter) picks its 8 values from LDS.  stride 2, W % 4 == 0.
template <int RB>
__global__ __launch_bounds__(256) void im2col_nchw3_s2_lds_kernel(const float* __restrict__ x, bf16* __restrict__ col, int B, int H,
                                                                  int W, int Ho, int Wo, int nbands) {
    extern __shared__ float im_lds[];                       // [3][2*RB+1][W]
    constexpr int NR = 2 * RB + 1;
    const int b = blockIdx.x / nbands, band = blockIdx.x % nbands;
    const int oy0 = band * RB, iy0 = 2 * oy0 - 1;
    const int w4 = W >> 2;
    for (int i = threadIdx.x; i < 3 * NR * w4; i += blockDim.x) {
        const int c4 = i % w4, r = (i / w4) % NR, ci = i / (w4 * NR);
        const int iy = iy0 + r;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < H) v = *reinterpret_cast<const f32x4*>(x + (((int64_t)b * 3 + ci) * H + iy) * W + c4 * 4);
        *reinterpret_cast<f32x4*>(im_lds + (ci * NR + r) * W + c4 * 4) = v;
    }
    __syncthreads();
    const int rows = min(RB, Ho - oy0);
    for (int i = threadIdx.x; i < rows * Wo * 4; i += blockDim.x) {
        const int q = i & 3, ox = (i >> 2) % Wo, ry = (i >> 2) / Wo;
        bf16x8 t;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = q * 8 + j;
            const int tap = (k * 11) >> 5, ci = k - tap * 3;          // k / 3, k % 3 for k < 32
            const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
            const int ix = ox * 2 + kx - 1;
            const bool ok = k < 27 && ix >= 0 && ix < W;              // rows outside the image were staged as zeros
            t[j] = ok ? (bf16)im_lds[(ci * NR + 2 * ry + ky) * W + ix] : (bf16)0.f;
        }
        *reinterpret_cast<bf16x8*>(col + (((int64_t)b * Ho + oy0 + ry) * Wo + ox) * 32 + q * 8) = t;
    }
}

// x bf16 NHWC (B,H,W,C) -> col bf16 [B*Ho*Wo, 9*C]; k = (ky*3+kx)*C + c.
__global__ __launch_bounds__(256) void im2col_nhwc_kernel(const bf16* __restrict__ x, bf16* __restrict__ col, int B, int H, int W,
                                                          int C, int Ho, int Wo, int stride) {
    const int cg = C >> 3;
    const int per_pix = 9 * cg;
    const int64_t total = (int64_t)B * Ho * Wo * per_pix;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)((unsigned)i % (unsigned)per_pix);
        const int64_t p = (unsigned)i / (unsigned)per_pix;
        const int tap = ch / cg, g = ch % cg;
        const int ky = tap / 3, kx = tap % 3;
        const unsigned pu = (unsigned)p;
        const int ox = (int)(pu % (unsigned)Wo);
        const int oy = (int)((pu / (unsigned)Wo) % (unsigned)Ho);
        const int b = (int)(pu / ((unsigned)Wo * (unsigned)Ho));
        const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W)
            v = *reinterpret_cast<const bf16x8*>(x + (((int64_t)b * H + iy) * W + ix) * C + g * 8);
        *reinterpret_cast<bf16x8*>(col + p * (9 * C) + tap * C + g * 8) = v;
    }
}

// im2col of act(BatchNorm(y)) without materialising that tensor: y is the producer ConvNorm's saved pre-BatchNorm output, the
// affine + activation runs on each gathered 16-byte chunk (bf16-rounded exactly like the stored activation would be); padding
// taps stay zero.  scale / shift per channel come from an LDS table (C <= 512).
__global__ __launch_bounds__(256) void im2col_nhwc_bn_kernel(const bf16* __restrict__ x, const float* __restrict__ stat,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                             bf16* __restrict__ col, int B, int H, int W, int C, int Ho, int Wo, int stride) {
    __shared__ __attribute__((aligned(16))) float tab[2 * 512];
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float sc = stat[C + c] * gamma[c];
        tab[c] = sc;
        tab[C + c] = beta[c] - stat[c] * sc;
    }
    __syncthreads();
    const int cg = C >> 3;
    const int per_pix = 9 * cg;
    const int64_t total = (int64_t)B * Ho * Wo * per_pix;
    const bool gelu = act == GG_ACT_GELU;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)((unsigned)i % (unsigned)per_pix);
        const int64_t p = (unsigned)i / (unsigned)per_pix;
        const int tap = ch / cg, g = ch % cg;
        const int ky = tap / 3, kx = tap % 3;
        const unsigned pu = (unsigned)p;
        const int ox = (int)(pu % (unsigned)Wo);
        const int oy = (int)((pu / (unsigned)Wo) % (unsigned)Ho);
        const int b = (int)(pu / ((unsigned)Wo * (unsigned)Ho));
        const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const bf16x8 r = *reinterpret_cast<const bf16x8*>(x + (((int64_t)b * H + iy) * W + ix) * C + g * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 sc = *reinterpret_cast<const f32x2*>(tab + g * 8 + 2 * q), sh = *reinterpret_cast<const f32x2*>(tab + C + g * 8 + 2 * q);
                const f32x2 z = gg_act_v2((f32x2){(float)r[2 * q], (float)r[2 * q + 1]} * sc + sh, gelu);
                v[2 * q] = (bf16)z.x; v[2 * q + 1] = (bf16)z.y;
            }
        }
        *reinterpret_cast<bf16x8*>(col + p * (9 * C) + tap * C + g * 8) = v;
    }
}

// transpose of im2col_nhwc: dcol bf16 [B*Ho*Wo, 9*C] -> dx bf16 NHWC (gather form, no atomics)
__global__ __launch_bounds__(256) void col2im_nhwc_kernel(const bf16* __restrict__ dcol, bf16* __restrict__ dx, int B, int H, int W,
                                                          int C, int Ho, int Wo, int stride) {
    const int cg = C >> 3;
    const int64_t total = (int64_t)B * H * W * cg;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)((unsigned)i % (unsigned)cg);
        const int64_t p = (unsigned)i / (unsigned)cg;
        const unsigned pu = (unsigned)p;
        const int ix = (int)(pu % (unsigned)W);
        const int iy = (int)((pu / (unsigned)W) % (unsigned)H);
        const int b = (int)(pu / ((unsigned)W * (unsigned)H));
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + 1 - ky;
            if (ty < 0 || (ty % stride) != 0) continue;
            const int oy = ty / stride;
            if (oy >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + 1 - kx;
                if (tx < 0 || (tx % stride) != 0) continue;
                const int ox = tx / stride;
                if (ox >= Wo) continue;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(dcol + (((int64_t)b * Ho + oy) * Wo + ox) * (9 * C) +
                                                                  (ky * 3 + kx) * C + g * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
            }
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)acc[j];
        *reinterpret_cast<bf16x8*>(dx + p * C + g * 8) = o;
    }
}

// ---------------------------------------------------------------- column-walking depthwise 3x3 (stride 1)
// Thread = (8 channels, one output column); the block covers PX adjacent columns x all C channels and walks down the whole
// image.  Lanes run over channels first, then columns, so every wave-level access is one contiguous run of the NHWC row
// (16 bytes per lane, full cache lines) -- the LDS-tiled kernel's 128-byte channel chunks at a C*2-byte stride reached only
// 2.8 TB/s.  A 3x3 input window lives in registers as fp32; each step loads the three columns of one new input row (the
// left / right neighbours come out of L1: HBM sees every input byte once), rotates the window by renaming (the loop is
// unrolled by 3) and emits one output row.  Rows / columns outside the image are zero through the buffer range check.
// colstats: one row per block, [gridDim.x][2][C] = per-channel sum / sum of squares of the stored (bf16-rounded) result.
typedef unsigned int dw_u32x4 __attribute__((ext_vector_type(4)));
#define DW_COL_OOB 0x80000000u
#define DW_ROW_OOB 0x40000000u
// NQ channel pairs per thread: 4 (16-byte accesses; the plain kernel) or 2 (8-byte accesses; the fused backward kernels, whose
// per-channel state does not fit the register file at 8 channels per thread -- measured 2.8 ms vs 1.9 ms at 56x56x384)
template <int NQ> struct DwRaw;
template <> struct DwRaw<4> {
    typedef unsigned int T __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ T load(__amdgpu_buffer_rsrc_t rs, int vo) { return __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0); }
};
template <> struct DwRaw<2> {
    typedef unsigned int T __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ T load(__amdgpu_buffer_rsrc_t rs, int vo) { return __builtin_amdgcn_raw_buffer_load_b64(rs, vo, 0, 0); }
};
__device__ __forceinline__ f32x2 dw_unpack2(unsigned r) { return (f32x2){__uint_as_float(r << 16), __uint_as_float(r & 0xffff0000u)}; }
__device__ __forceinline__ unsigned dw_pack2(f32x2 v) {
    const bf16 lo = (bf16)v.x, hi = (bf16)v.y;
    return (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
}
// Backward fusions (stride-1 data gradient, both optional, compile-time):
//   IN2 : the conv input is  coef0*x + coef1*in2 + coef2  = BatchNorm backward's "apply" of the ConvNorm BEHIND this conv, formed
//         from (dz, y) while loading -- the separate apply pass and the dy tensor disappear;
//   EPI : the stored value is  acc * act'(BN(ep_y))  = gradient w.r.t. the pre-activation of the ConvNorm IN FRONT, and colstats
//         becomes (sum dz, sum dz*xhat): that ConvNorm's "reduce" pass disappears.
// The per-channel coefficient tables sit in LDS (they are touched once per row; registers go to the window and the taps).
// Window element: fp32 pairs, or (IN2) the packed bf16 pair the unfused path would have stored in dy, unpacked at use.
template <bool PACKED> struct DwWin;
template <> struct DwWin<false> {
    typedef f32x2 E;
    static __device__ __forceinline__ E zero() { return (f32x2)(0.f); }
    static __device__ __forceinline__ f32x2 get(E e) { return e; }
};
template <> struct DwWin<true> {
    typedef unsigned E;
    static __device__ __forceinline__ E zero() { return 0u; }
    static __device__ __forceinline__ f32x2 get(E e) { return dw_unpack2(e); }
};
struct DwWalkFuse {
    const float* in_stat; const float* in_gamma; const float* in_beta; int in_act;  // IN1
    const bf16* in2; const float* in_coef;                                         // IN2
    const bf16* ep_y; const float* ep_stat; const float* ep_gamma; const float* ep_beta; int ep_act;   // EPI
};
template <int IN, bool EPI, int NQ>
__global__ __launch_bounds__(256, NQ == 4 ? 2 : 3) void dwconv3x3_walk_kernel(const bf16* __restrict__ x, const float* __restrict__ wt,
                                                                              bf16* __restrict__ y, int H, int W, int C, int CG, int PX, int nbx,
                                                                              int flip, float* __restrict__ colstats, DwWalkFuse f) {
    constexpr bool IN2 = IN == 2;              // backward: BatchNorm-backward apply of (x, in2) while loading
    constexpr bool IN1 = IN == 1;              // forward: act(BatchNorm(x)) of the producer ConvNorm while loading (ctab rows 0 / 1 = scale / shift)
    typedef typename DwRaw<NQ>::T Raw;
    typedef typename DwWin<IN == 2>::E WinE;
    constexpr int NC = 2 * NQ;                 // channels per thread
    extern __shared__ float dw_red[];          // [PX][2][C] statistics scratch; then (fusions) 16 coefficient rows [C]
    float* ctab = dw_red + PX * 2 * C;         // [0..2] in coef a,b,c   [3] ep scale  [4] ep shift  [5] ep rstd  [6] ep -mean*rstd  [7..15] taps
    constexpr bool TAPS_LDS = IN != 0 || EPI;      // the fused variants have no registers left for the tap values
    const int cg = threadIdx.x % CG, px = threadIdx.x / CG;
    // the strips of one image are neighbours in the logical block order, and that order is laid out XCD by XCD: the halo columns a
    // strip shares with the next one are then served by the same L2 (round-robin dispatch put them on different XCDs and every
    // strip fetched its halo from HBM again: measured 1.85x the algorithmic read traffic)
    const int bid = gg_xcd_remap(blockIdx.x, gridDim.x);
    const int bx = bid % nbx, b = bid / nbx;
    const int xo = bx * PX + px;
    const int c0 = cg * NC;
    if (IN != 0 || EPI) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            if (IN2) { ctab[c] = f.in_coef[c]; ctab[C + c] = f.in_coef[C + c]; ctab[2 * C + c] = f.in_coef[2 * C + c]; }
            if (IN1) { const float sc = f.in_stat[C + c] * f.in_gamma[c]; ctab[c] = sc; ctab[C + c] = f.in_beta[c] - f.in_stat[c] * sc; }
            if (EPI) {
                const float mu = f.ep_stat[c], rstd = f.ep_stat[C + c], sc = rstd * f.ep_gamma[c];
                ctab[3 * C + c] = sc; ctab[4 * C + c] = f.ep_beta[c] - mu * sc; ctab[5 * C + c] = rstd; ctab[6 * C + c] = -mu * rstd;
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) ctab[(7 + t) * C + c] = wt[(flip ? 8 - t : t) * C + c];
        }
        __syncthreads();
    }
    // `cofs` is laundered through an empty asm once per row step: the table reads are loop-invariant, and hoisting them would
    // put all 16 rows back into registers
    int cofs = c0;
    auto ctab2 = [&](int row, int q) { return *reinterpret_cast<const f32x2*>(ctab + row * C + cofs + 2 * q); };
    f32x2 tap[TAPS_LDS ? 1 : 9][NQ];
    if (!TAPS_LDS) {
#pragma unroll
        for (int t = 0; t < (TAPS_LDS ? 1 : 9); ++t)
#pragma unroll
            for (int q = 0; q < NQ; ++q) tap[t][q] = *reinterpret_cast<const f32x2*>(wt + (flip ? 8 - t : t) * C + c0 + 2 * q);
    }
    const int64_t img = (int64_t)b * H * W * C;
    // (the image bases are block-uniform; saying so keeps the descriptors in SGPRs -- otherwise every buffer load is wrapped in a
    // waterfall loop over "divergent" descriptors)
    auto uniform_ptr = [](const bf16* ptr) {
        const unsigned long long a = (unsigned long long)ptr;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return (void*)(((unsigned long long)hi << 32) | lo);
    };
    const int img_bytes = __builtin_amdgcn_readfirstlane(H * W * C * 2);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(x + img), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((IN2 ? f.in2 : x) + img), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rse = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((EPI ? f.ep_y : x) + img), 0, img_bytes, 0x00020000);
    // byte offsets of the three columns inside a row
    unsigned colo[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int ix = xo + k - 1;
        colo[k] = (ix >= 0 && ix < W && xo < W) ? (unsigned)(ix * C + c0) * 2u : DW_COL_OOB;
    }
    const unsigned rowb = (unsigned)W * C * 2u;
    // branch-free addressing: an invalid column carries bit 31, an invalid row adds bit 30; either pushes the offset past the
    // (< 1 GiB) descriptor, and no combination wraps back into range.  (Branches around the loads made the compiler wait for
    // ALL outstanding loads at the join, which serialised the row prefetch.)
    auto load_row = [&](int iy, Raw (&raw)[3], Raw (&raw2)[3]) {
        const unsigned ro = (iy >= 0 && iy < H) ? (unsigned)iy * rowb : DW_ROW_OOB;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int vo = (int)(colo[k] + ro);
            raw[k] = DwRaw<NQ>::load(rs, vo);
            if (IN2) raw2[k] = DwRaw<NQ>::load(rs2, vo);
        }
    };
    // raw row -> window slot; IN2: the BatchNorm-backward apply on the way (positions outside the image stay exactly 0)
    auto fill = [&](int iy, const Raw (&raw)[3], const Raw (&raw2)[3], WinE (&slot)[3][NQ]) {
        if constexpr (IN == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int q = 0; q < NQ; ++q) slot[k][q] = dw_unpack2(raw[k][q]);
        } else if constexpr (IN1) {
            const bool rok = iy >= 0 && iy < H;
            const bool in_gelu = f.in_act == GG_ACT_GELU;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const f32x2 sc = ctab2(0, q), sh = ctab2(1, q);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    // the bf16 rounding of the activation tensor the unfused path stores is kept (same numerics)
                    const f32x2 r = gg_act_v2(dw_unpack2(raw[k][q]) * sc + sh, in_gelu);
                    slot[k][q] = (rok && colo[k] != DW_COL_OOB) ? dw_unpack2(dw_pack2(r)) : (f32x2)(0.f);
                }
            }
        } else {
            const bool rok = iy >= 0 && iy < H;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const f32x2 ca = ctab2(0, q), cb = ctab2(1, q), cc = ctab2(2, q);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const f32x2 r = ca * dw_unpack2(raw[k][q]) + (cb * dw_unpack2(raw2[k][q]) + cc);
                    slot[k][q] = (rok && colo[k] != DW_COL_OOB) ? dw_pack2(r) : 0u;
                }
            }
        }
    };
    WinE win[3][3][NQ];           // [row slot][column][channel pair]
    Raw raw[3], raw2[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int q = 0; q < NQ; ++q) win[0][k][q] = DwWin<IN == 2>::zero();          // row -1
    load_row(0, raw, raw2);
    fill(0, raw, raw2, win[1]);
    load_row(1, raw, raw2);
    f32x2 s2[NQ], q2[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) s2[q] = q2[q] = (f32x2)(0.f);
    bf16* yb = y + img + (int64_t)xo * C + c0;
    const bool store_ok = xo < W;
    const unsigned ctr = store_ok ? (unsigned)(xo * C + c0) * 2u : DW_COL_OOB;
    const bool ep_gelu = f.ep_act == GG_ACT_GELU;
    Raw eraw;
    if (EPI) eraw = DwRaw<NQ>::load(rse, (int)ctr);
    // one output row: slot RC receives input row yy+1 (already in flight), rows yy-1 / yy sit in slots RA / RB
#define GG_DW_STEP(RA, RB, RC, yy)                                                                                         \
    {                                                                                                                      \
        if (IN != 0 || EPI) asm volatile("" : "+v"(cofs));                                                                     \
        fill((yy) + 1, raw, raw2, win[RC]);                                                                                \
        load_row((yy) + 2, raw, raw2);                                                                                     \
        const float live = (yy) < H ? 1.f : 0.f;                                                                           \
        f32x2 acc[NQ];                                                                                                     \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                                   \
            f32x2 a = (f32x2)(0.f);                                                                                        \
            _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                                                \
                const int rsl = t < 3 ? RA : (t < 6 ? RB : RC);                                                            \
                const f32x2 tp = TAPS_LDS ? ctab2(7 + t, q) : tap[TAPS_LDS ? 0 : t][q];                                    \
                a = DwWin<IN == 2>::get(win[rsl][t % 3][q]) * tp + a;                                                          \
            }                                                                                                              \
            acc[q] = a;                                                                                                    \
        }                                                                                                                  \
        Raw o;                                                                                                             \
        if (EPI) {                                                                                                         \
            const Raw ycur = eraw;                                                                                         \
            eraw = DwRaw<NQ>::load(rse, (int)(ctr + ((yy) + 1 < H ? (unsigned)((yy) + 1) * rowb : DW_ROW_OOB)));          \
            _Pragma("unroll") for (int q = 0; q < NQ; ++q) {       /* one channel pair at a time: short live ranges */      \
                const f32x2 yv = dw_unpack2(ycur[q]);                                                                      \
                const f32x2 dz = acc[q] * gg_act_grad_v2(yv * ctab2(3, q) + ctab2(4, q), ep_gelu);                         \
                o[q] = dw_pack2(dz);                                                                                       \
                const f32x2 r = dw_unpack2(o[q]) * live;                                                                   \
                s2[q] += r; q2[q] += r * (yv * ctab2(5, q) + ctab2(6, q));                                                 \
            }                                                                                                              \
        } else {                                                                                                           \
            _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                               \
                o[q] = dw_pack2(acc[q]);                                                                                   \
                const f32x2 r = dw_unpack2(o[q]) * live;                                                                   \
                s2[q] += r; q2[q] += r * r;                                                                                \
            }                                                                                                              \
        }                                                                                                                  \
        if (store_ok && (yy) < H) *reinterpret_cast<Raw*>(yb + (int64_t)(yy) * W * C) = o;                                 \
    }
    // (the tail of the last trip may run 1-2 rows past the image: its loads are out of range, its store and statistics masked)
    for (int y0 = 0; y0 < H; y0 += 3) {
        GG_DW_STEP(0, 1, 2, y0)
        GG_DW_STEP(1, 2, 0, y0 + 1)
        GG_DW_STEP(2, 0, 1, y0 + 2)
    }
#undef GG_DW_STEP
    if (colstats) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            // columns beyond the image produced zeros (all inputs out of range): they add nothing
            dw_red[(px * 2 + 0) * C + c0 + 2 * q] = s2[q].x; dw_red[(px * 2 + 0) * C + c0 + 2 * q + 1] = s2[q].y;
            dw_red[(px * 2 + 1) * C + c0 + 2 * q] = q2[q].x; dw_red[(px * 2 + 1) * C + c0 + 2 * q + 1] = q2[q].y;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
            float t = 0.f;
            for (int k = 0; k < PX; ++k) t += dw_red[k * 2 * C + i];
            colstats[(int64_t)blockIdx.x * 2 * C + i] = t;
        }
    }
}

// ---------------------------------------------------------------- stride-1 walk, NCOL output columns per thread, producer BatchNorm + activation on load
// MBConv.conv2 over GELU(BN1(y1)): with one column per thread every input element is transformed by the three threads whose windows
// contain it (the IN = 1 variant of the kernel above: 2.6 ms against 1.24 + 1.0 ms for plain conv + apply pass).  Here a thread owns
// NCOL adjacent columns x 4 channels: NCOL + 2 loads and transforms per row step for NCOL results -- 1.5 evaluations per element at
// NCOL = 4 -- and the apply pass with its 2 x [M, C] of traffic disappears.  Forward only; statistics as in the other walkers.
// IN = 0 plain / 1 producer BatchNorm + activation (forward) / 2 BatchNorm-backward apply of (x, in2) (data gradient); EPI: the stored value
// is acc * act'(BN(ep_y)) and the statistics are (sum dz, sum dz*xhat) -- the same fusions as dwconv3x3_walk_kernel, whose stride-1 uses
// this kernel takes over: every per-element transform on the way in runs 1.5 instead of 3 times, and a result costs 1.5 instead of 3 loads.
// `flip` reverses the taps (data gradient).  Per-channel coefficients live in registers (4 channels per thread).
template <int NCOL, int IN, bool EPI>
__global__ __launch_bounds__(256, 2) void dwconv3x3_s1_multi_kernel(const bf16* __restrict__ x, const float* __restrict__ wt, bf16* __restrict__ y,
                                                                    int H, int W, int C, int CG, int PX, int nbx, int flip,
                                                                    float* __restrict__ colstats, DwWalkFuse f) {
    constexpr bool IN1 = IN == 1, IN2 = IN == 2;
    constexpr int NQ = 2, NW = NCOL + 2;
    typedef DwRaw<NQ>::T Raw;
    typedef typename DwWin<IN2>::E WinE;        // IN2: the packed bf16 pair the unfused path would have stored in dy
    extern __shared__ float dwm_red[];          // [PX][2][C] statistics scratch
    const int cg = threadIdx.x % CG, px = threadIdx.x / CG;
    const int bid = gg_xcd_remap(blockIdx.x, gridDim.x);
    const int bx = bid % nbx, b = bid / nbx;
    const int x0 = (bx * PX + px) * NCOL;       // first output column of this thread
    const int c0 = cg * 4;
    f32x2 tap[9][NQ];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < NQ; ++q) tap[t][q] = *reinterpret_cast<const f32x2*>(wt + (flip ? 8 - t : t) * C + c0 + 2 * q);
    f32x2 ia[NQ], ib[NQ], ic[NQ];               // IN1: scale, shift;  IN2: coef a, b, c
    f32x2 esc[NQ], esh[NQ], ers[NQ], emr[NQ];   // EPI: scale, shift, rstd, -mean*rstd
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int c = c0 + 2 * q;
        if (IN1) {
            const f32x2 rs_ = *reinterpret_cast<const f32x2*>(f.in_stat + C + c), mu = *reinterpret_cast<const f32x2*>(f.in_stat + c);
            ia[q] = rs_ * *reinterpret_cast<const f32x2*>(f.in_gamma + c);
            ib[q] = *reinterpret_cast<const f32x2*>(f.in_beta + c) - mu * ia[q];
        }
        if (IN2) {
            ia[q] = *reinterpret_cast<const f32x2*>(f.in_coef + c); ib[q] = *reinterpret_cast<const f32x2*>(f.in_coef + C + c);
            ic[q] = *reinterpret_cast<const f32x2*>(f.in_coef + 2 * C + c);
        }
        if (EPI) {
            const f32x2 mu = *reinterpret_cast<const f32x2*>(f.ep_stat + c), rstd = *reinterpret_cast<const f32x2*>(f.ep_stat + C + c);
            esc[q] = rstd * *reinterpret_cast<const f32x2*>(f.ep_gamma + c);
            esh[q] = *reinterpret_cast<const f32x2*>(f.ep_beta + c) - mu * esc[q];
            ers[q] = rstd; emr[q] = (f32x2)(0.f) - mu * rstd;
        }
    }
    const bool in_gelu = f.in_act == GG_ACT_GELU, ep_gelu = f.ep_act == GG_ACT_GELU;
    const int64_t img = (int64_t)b * H * W * C;
    auto uniform_ptr = [](const bf16* ptr) {    // block-uniform bases: descriptors in SGPRs
        const unsigned long long a = (unsigned long long)ptr;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return (void*)(((unsigned long long)hi << 32) | lo);
    };
    const int img_bytes = __builtin_amdgcn_readfirstlane(H * W * C * 2);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(x + img), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((IN2 ? f.in2 : x) + img), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rse = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((EPI ? f.ep_y : x) + img), 0, img_bytes, 0x00020000);
    unsigned colo[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int ix = x0 + k - 1;
        colo[k] = (ix >= 0 && ix < W) ? (unsigned)(ix * C + c0) * 2u : DW_COL_OOB;
    }
    const unsigned rowb = (unsigned)W * C * 2u;
    auto load_row = [&](int iy, Raw (&raw)[NW], Raw (&raw2)[NW]) {
        const unsigned ro = (iy >= 0 && iy < H) ? (unsigned)iy * rowb : DW_ROW_OOB;
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            raw[k] = DwRaw<NQ>::load(rs, (int)(colo[k] + ro));
            if (IN2) raw2[k] = DwRaw<NQ>::load(rs2, (int)(colo[k] + ro));
        }
    };
    auto load_ep = [&](int iy, Raw (&er)[NCOL]) {
        const unsigned ro = (iy >= 0 && iy < H) ? (unsigned)iy * rowb : DW_ROW_OOB;
#pragma unroll
        for (int j = 0; j < NCOL; ++j) er[j] = DwRaw<NQ>::load(rse, (int)(colo[j + 1] + ro));
    };
    // raw row -> window slot; the transform runs once per loaded element; positions outside the image stay exactly 0
    auto fill = [&](int iy, const Raw (&raw)[NW], const Raw (&raw2)[NW], WinE (&slot)[NW][NQ]) {
        const bool rok = iy >= 0 && iy < H;
#pragma unroll
        for (int k = 0; k < NW; ++k)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if constexpr (IN1) {
                    const f32x2 a = dw_unpack2(dw_pack2(gg_act_v2(dw_unpack2(raw[k][q]) * ia[q] + ib[q], in_gelu)));
                    slot[k][q] = (rok && colo[k] != DW_COL_OOB) ? a : (f32x2)(0.f);
                } else if constexpr (IN2) {
                    const f32x2 r = ia[q] * dw_unpack2(raw[k][q]) + (ib[q] * dw_unpack2(raw2[k][q]) + ic[q]);
                    slot[k][q] = (rok && colo[k] != DW_COL_OOB) ? dw_pack2(r) : 0u;
                } else {
                    slot[k][q] = dw_unpack2(raw[k][q]);
                }
            }
    };
    WinE win[3][NW][NQ];
    Raw raw[NW], raw2[NW], eraw[NCOL];
#pragma unroll
    for (int k = 0; k < NW; ++k)
#pragma unroll
        for (int q = 0; q < NQ; ++q) win[0][k][q] = DwWin<IN2>::zero();      // row -1
    load_row(0, raw, raw2);
    fill(0, raw, raw2, win[1]);
    load_row(1, raw, raw2);
    if (EPI) load_ep(0, eraw);
    f32x2 s2[NQ], q2[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) s2[q] = q2[q] = (f32x2)(0.f);
    bf16* yb = y + img + (int64_t)x0 * C + c0;
#define GG_DWM_STEP(RA, RB, RC, yy)                                                                                        \
    {                                                                                                                      \
        fill((yy) + 1, raw, raw2, win[RC]);                                                                                \
        load_row((yy) + 2, raw, raw2);                                                                                     \
        Raw ecur[NCOL];                                                                                                    \
        if (EPI) {                                                                                                         \
            _Pragma("unroll") for (int j = 0; j < NCOL; ++j) ecur[j] = eraw[j];                                            \
            load_ep((yy) + 1, eraw);                                                                                       \
        }                                                                                                                  \
        _Pragma("unroll") for (int j = 0; j < NCOL; ++j) {                                                                 \
            const bool live = (yy) < H && x0 + j < W;                                                                      \
            Raw o;                                                                                                         \
            _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                               \
                f32x2 a = (f32x2)(0.f);                                                                                    \
                _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                                            \
                    const int rsl = t < 3 ? RA : (t < 6 ? RB : RC);                                                        \
                    a = DwWin<IN2>::get(win[rsl][j + t % 3][q]) * tap[t][q] + a;                                           \
                }                                                                                                          \
                if (EPI) {                                                                                                 \
                    const f32x2 yv = dw_unpack2(ecur[j][q]);                                                               \
                    o[q] = dw_pack2(a * gg_act_grad_v2(yv * esc[q] + esh[q], ep_gelu));                                    \
                    const f32x2 r = live ? dw_unpack2(o[q]) : (f32x2)(0.f);                                                \
                    s2[q] += r; q2[q] += r * (yv * ers[q] + emr[q]);                                                       \
                } else {                                                                                                   \
                    o[q] = dw_pack2(a);                                                                                    \
                    const f32x2 r = live ? dw_unpack2(o[q]) : (f32x2)(0.f);                                                \
                    s2[q] += r; q2[q] += r * r;                                                                            \
                }                                                                                                          \
            }                                                                                                              \
            if (live) *reinterpret_cast<Raw*>(yb + ((int64_t)(yy) * W + j) * C) = o;                                       \
        }                                                                                                                  \
    }
    for (int y0 = 0; y0 < H; y0 += 3) {
        GG_DWM_STEP(0, 1, 2, y0)
        GG_DWM_STEP(1, 2, 0, y0 + 1)
        GG_DWM_STEP(2, 0, 1, y0 + 2)
    }
#undef GG_DWM_STEP
    if (colstats) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            dwm_red[(px * 2 + 0) * C + c0 + 2 * q] = s2[q].x; dwm_red[(px * 2 + 0) * C + c0 + 2 * q + 1] = s2[q].y;
            dwm_red[(px * 2 + 1) * C + c0 + 2 * q] = q2[q].x; dwm_red[(px * 2 + 1) * C + c0 + 2 * q + 1] = q2[q].y;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
            float t = 0.f;
            for (int k = 0; k < PX; ++k) t += dwm_red[k * 2 * C + i];
            colstats[(int64_t)blockIdx.x * 2 * C + i] = t;
        }
    }
}

// ---------------------------------------------------------------- stride-2 column-walking depthwise 3x3 (PatchMerging.conv2 forward)
// thread = (8 channels, one OUTPUT column), walking down the output rows: output row oy needs input rows 2oy-1 .. 2oy+1, of which
// 2oy-1 was the last row of the previous step -> two new input rows x three columns per step (6 x 16-byte loads for a 16-byte
// result; the LDS-tiled kernel staged 17x17 inputs per 8x8 outputs through LDS and ran at 3.6 TB/s).  IN1: the input is the saved
// pre-BatchNorm output of the ConvNorm in front and act(BatchNorm(x)) is applied to every loaded element (bf16-rounded like the
// stored activation; 1.5 evaluations per input element instead of a 2 x [M, C] apply pass); padding stays exactly zero.
// BatchNorm partial statistics of the bf16-rounded result: one row per block in colstats [gridDim.x][2][C].
template <bool IN1>
__global__ __launch_bounds__(256, 2) void dwconv3x3_s2_walk_kernel(const bf16* __restrict__ x, const float* __restrict__ wt, bf16* __restrict__ y,
                                                                   int H, int W, int C, int Ho, int Wo, int CG, int PX, int nbx,
                                                                   float* __restrict__ colstats, DwWalkFuse f) {
    constexpr int NQ = 4;
    typedef DwRaw<NQ>::T Raw;
    extern __shared__ float dw2_red[];          // [PX][2][C] statistics scratch; IN1: scale, shift rows [2][C]
    float* ctab = dw2_red + PX * 2 * C;
    const int cg = threadIdx.x % CG, px = threadIdx.x / CG;
    const int bid = gg_xcd_remap(blockIdx.x, gridDim.x);
    const int bx = bid % nbx, b = bid / nbx;
    const int ox = bx * PX + px;
    const int c0 = cg * 8;
    if (IN1) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            const float sc = f.in_stat[C + c] * f.in_gamma[c];
            ctab[c] = sc; ctab[C + c] = f.in_beta[c] - f.in_stat[c] * sc;
        }
        __syncthreads();
    }
    f32x2 tap[9][NQ], isc[NQ], ish[NQ];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < NQ; ++q) tap[t][q] = *reinterpret_cast<const f32x2*>(wt + t * C + c0 + 2 * q);
    if (IN1) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) { isc[q] = *reinterpret_cast<const f32x2*>(ctab + c0 + 2 * q); ish[q] = *reinterpret_cast<const f32x2*>(ctab + C + c0 + 2 * q); }
    }
    const bool in_gelu = f.in_act == GG_ACT_GELU;
    const int64_t img = (int64_t)b * H * W * C;
    const unsigned long long xa = (unsigned long long)(x + img);
    const unsigned xlo = __builtin_amdgcn_readfirstlane((unsigned)xa), xhi = __builtin_amdgcn_readfirstlane((unsigned)(xa >> 32));   // block-uniform: descriptor in SGPRs
    const void* xb = (const void*)(((unsigned long long)xhi << 32) | xlo);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, __builtin_amdgcn_readfirstlane(H * W * C * 2), 0x00020000);
    unsigned colo[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int ix = 2 * ox + k - 1;
        colo[k] = (ix >= 0 && ix < W && ox < Wo) ? (unsigned)(ix * C + c0) * 2u : DW_COL_OOB;
    }
    const unsigned rowb = (unsigned)W * C * 2u;
    auto load_row = [&](int iy, Raw (&raw)[3]) {
        const unsigned ro = (iy >= 0 && iy < H) ? (unsigned)iy * rowb : DW_ROW_OOB;
#pragma unroll
        for (int k = 0; k < 3; ++k) raw[k] = DwRaw<NQ>::load(rs, (int)(colo[k] + ro));
    };
    auto fill = [&](int iy, const Raw (&raw)[3], f32x2 (&slot)[3][NQ]) {
        const bool rok = iy >= 0 && iy < H;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (IN1) {
                    const f32x2 a = dw_unpack2(dw_pack2(gg_act_v2(dw_unpack2(raw[k][q]) * isc[q] + ish[q], in_gelu)));
                    slot[k][q] = (rok && colo[k] != DW_COL_OOB) ? a : (f32x2)(0.f);
                } else {
                    slot[k][q] = dw_unpack2(raw[k][q]);
                }
            }
    };
    f32x2 win[3][3][NQ];          // [row slot][column][channel pair]; slot 0 = input row 2oy-1, 1 = 2oy, 2 = 2oy+1
    Raw ra[3], rb[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int q = 0; q < NQ; ++q) win[0][k][q] = (f32x2)(0.f);      // row -1
    load_row(0, ra);
    load_row(1, rb);
    f32x2 s2[NQ], q2[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) s2[q] = q2[q] = (f32x2)(0.f);
    const bool store_ok = ox < Wo;
    bf16* yb = y + (int64_t)b * Ho * Wo * C + (int64_t)ox * C + c0;
    // one output row; TOP names the slot holding input row 2oy-1, the other two receive rows 2oy and 2oy+1; the slot that held 2oy+1
    // is the next step's TOP, so two steps with the roles of slots 0 and 2 swapped make one trip (no register copies)
#define GG_S2W_STEP(TOP, BOT, oy)                                                                                          \
    {                                                                                                                      \
        fill(2 * (oy), ra, win[1]);                                                                                        \
        fill(2 * (oy) + 1, rb, win[BOT]);                                                                                  \
        load_row(2 * (oy) + 2, ra);                                                                                        \
        load_row(2 * (oy) + 3, rb);                                                                                        \
        Raw o;                                                                                                             \
        const float live = (oy) < Ho ? 1.f : 0.f;                                                                          \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                                   \
            f32x2 a = (f32x2)(0.f);                                                                                        \
            _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                                                \
                const int rsl = t < 3 ? TOP : (t < 6 ? 1 : BOT);                                                           \
                a = win[rsl][t % 3][q] * tap[t][q] + a;                                                                    \
            }                                                                                                              \
            o[q] = dw_pack2(a);                                                                                            \
            const f32x2 r = dw_unpack2(o[q]) * live;                                                                       \
            s2[q] += r; q2[q] += r * r;                                                                                    \
        }                                                                                                                  \
        if (store_ok && (oy) < Ho) *reinterpret_cast<Raw*>(yb + (int64_t)(oy) * Wo * C) = o;                               \
    }
    for (int oy = 0; oy < Ho; oy += 2) {
        GG_S2W_STEP(0, 2, oy)
        GG_S2W_STEP(2, 0, oy + 1)
    }
#undef GG_S2W_STEP
    if (colstats) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            dw2_red[(px * 2 + 0) * C + c0 + 2 * q] = s2[q].x; dw2_red[(px * 2 + 0) * C + c0 + 2 * q + 1] = s2[q].y;
            dw2_red[(px * 2 + 1) * C + c0 + 2 * q] = q2[q].x; dw2_red[(px * 2 + 1) * C + c0 + 2 * q + 1] = q2[q].y;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
            float t = 0.f;
            for (int k = 0; k < PX; ++k) t += dw2_red[k * 2 * C + i];
            colstats[(int64_t)blockIdx.x * 2 * C + i] = t;
        }
    }
}

// ---------------------------------------------------------------- LDS-tiled depthwise 3x3
// Block = (image, band of 8 output rows, chunk of 64 channels); it walks the band in 8-pixel-wide tiles.  The input
// tile (+halo) is staged once in LDS (next tile prefetched through registers), so HBM sees every input byte once and
// each output costs 9 LDS reads instead of 9 global loads.  Optional fused producer: the staged values are
// act(gamma*(x-mean)*rstd+beta) of the previous ConvNorm (BatchNorm + GELU applied on the fly; zero padding stays
// zero), which removes the separate bn_apply pass and its 2 x [M,C] of traffic.  `flip` reverses the taps (= the
// data gradient of a stride-1 depthwise conv).  Per-channel sum / sum-of-squares of the result (BatchNorm partials) are
// written as one row per (image, band): colstats[row][2][C].
// MODE 0: plain (forward / flipped-tap data gradient); 1: fused producer BN+act while staging; 2: backward fusions
// (2-input staging and/or act'(BN) epilogue).  Compile-time so that the plain kernel keeps its register budget.
template <int S, int MODE>
__global__ __launch_bounds__(256) void dwconv3x3_tiled_kernel(const bf16* __restrict__ x, const float* __restrict__ wt,
                                                              bf16* __restrict__ y, int B, int H, int W, int C, int Ho, int Wo,
                                                              int nbands, int nchunks, int flip, const float* __restrict__ in_stat,
                                                              const float* __restrict__ in_gamma, const float* __restrict__ in_beta,
                                                              int in_act, float* __restrict__ colstats,
                                                              const bf16* __restrict__ in2, const float* __restrict__ in_coef,
                                                              const bf16* __restrict__ ep_y, const float* __restrict__ ep_stat,
                                                              const float* __restrict__ ep_gamma, const float* __restrict__ ep_beta,
                                                              int ep_act) {
    // backward-pass fusions (both optional):
    //  in2/in_coef : the staged value is  coef0*x + coef1*in2 + coef2  = BatchNorm-backward "apply" of the ConvNorm BEHIND this
    //                conv, formed on the fly from (dz, y) -- the separate apply pass and the dy tensor disappear;
    //  ep_*        : the stored value is  acc * act'(BN(ep_y))  = gradient w.r.t. the pre-activation of the ConvNorm IN FRONT,
    //                and colstats becomes (sum dz, sum dz*xhat): that ConvNorm's BatchNorm-backward "reduce" pass disappears.
    // thread = (channel quad g of 16, pixel slot ps of 16): 4 channels (8-byte accesses) keep the 9x4 taps, the
    // accumulators and the prefetched chunks within ~100 VGPRs
    constexpr int TH = 8, TW = 8, CT = 64, Q = 4, NG = CT / Q;
    constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    constexpr int NCH = IH * IW * NG;
    constexpr int LPT = (NCH + 255) / 256;
    __shared__ __attribute__((aligned(16))) bf16 tile[IH * IW * CT];
    const int cc = blockIdx.x % nchunks;
    const int band = (blockIdx.x / nchunks) % nbands;
    const int b = blockIdx.x / (nchunks * nbands);
    const int c0 = cc * CT;
    const int g = threadIdx.x & (NG - 1), ps = threadIdx.x / NG;
    const int cg0 = c0 + g * Q;
    const bool cok = cg0 < C;                       // C % 8 == 0; the last chunk may be partial (C % 64 != 0)
    float wreg[9][Q];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < Q; ++j) wreg[t][j] = cok ? wt[(flip ? 8 - t : t) * C + cg0 + j] : 0.f;
    float sc[Q], sh[Q], s2[Q];
    const bool fused = MODE == 1 && in_stat != nullptr, fused2 = MODE == 2 && in_coef != nullptr;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        sc[j] = (fused && cok) ? in_stat[C + cg0 + j] * in_gamma[cg0 + j] : 1.f;
        sh[j] = (fused && cok) ? in_beta[cg0 + j] - in_stat[cg0 + j] * sc[j] : 0.f;
        s2[j] = 0.f;
        if (fused2 && cok) { sc[j] = in_coef[cg0 + j]; s2[j] = in_coef[C + cg0 + j]; sh[j] = in_coef[2 * C + cg0 + j]; }
    }
    float esc[Q], esh[Q], ers[Q], emr[Q];
    const bool epi = MODE == 2 && ep_y != nullptr;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        ers[j] = (epi && cok) ? ep_stat[C + cg0 + j] : 0.f;
        emr[j] = (epi && cok) ? ep_stat[cg0 + j] * ers[j] : 0.f;
        esc[j] = (epi && cok) ? ers[j] * ep_gamma[cg0 + j] : 0.f;
        esh[j] = (epi && cok) ? ep_beta[cg0 + j] - ep_stat[cg0 + j] * esc[j] : 0.f;
    }
    const bf16* x2b = in2 ? in2 + (int64_t)b * H * W * C : nullptr;
    const int oy0 = band * TH;
    const int iy0 = oy0 * S - 1;
    const bf16* xb = x + (int64_t)b * H * W * C;
    bf16x4 pre[LPT];
    const bf16x4 zero4 = {0, 0, 0, 0};
    auto load_tile = [&](int ox0) {
        const int ix0 = ox0 * S - 1;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int ch = threadIdx.x + i * 256;            // chunk = (pixel, channel quad): quad == g for every i
            const int pix = ch / NG;
            const int py = pix / IW, px = pix - py * IW;
            const int iy = iy0 + py, ix = ix0 + px;
            bf16x4 v = zero4;
            if (ch < NCH && cok && iy >= 0 && iy < H && ix >= 0 && ix < W) {
                v = *reinterpret_cast<const bf16x4*>(xb + ((int64_t)iy * W + ix) * C + cg0);
                if (fused2) {
                    const bf16x4 w2 = *reinterpret_cast<const bf16x4*>(x2b + ((int64_t)iy * W + ix) * C + cg0);
#pragma unroll
                    for (int j = 0; j < Q; ++j) v[j] = (bf16)fmaf(sc[j], (float)v[j], fmaf(s2[j], (float)w2[j], sh[j]));
                } else if (fused) {
#pragma unroll
                    for (int j = 0; j < Q; ++j) v[j] = (bf16)gg_act((float)v[j] * sc[j] + sh[j], in_act);
                }
            }
            pre[i] = v;
        }
    };
    float s[Q], q[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) s[j] = q[j] = 0.f;
    const int ntx = (Wo + TW - 1) / TW;
    load_tile(0);
    for (int tx = 0; tx < ntx; ++tx) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int ch = threadIdx.x + i * 256;
            if (ch < NCH) *reinterpret_cast<bf16x4*>(tile + ch * Q) = pre[i];
        }
        __syncthreads();
        if (tx + 1 < ntx) load_tile((tx + 1) * TW);
        // fused epilogue: y of the NEXT output pixel is fetched while the current one is computed
        auto ep_load = [&](int pi) {
            const int p = ps + 16 * pi;
            const int oy = oy0 + (p >> 3), ox = tx * TW + (p & 7);
            return (oy < Ho && ox < Wo && cok) ? *reinterpret_cast<const bf16x4*>(ep_y + (((int64_t)b * Ho + oy) * Wo + ox) * C + cg0) : zero4;
        };
        bf16x4 ep_next = zero4;
        if (epi) ep_next = ep_load(0);
#pragma unroll 1
        for (int pi = 0; pi < (TH * TW) / 16; ++pi) {
            const bf16x4 ep_cur = ep_next;
            if (epi && pi + 1 < (TH * TW) / 16) ep_next = ep_load(pi + 1);
            const int p = ps + 16 * pi;
            const int oyl = p >> 3, oxl = p & 7;
            const int oy = oy0 + oyl, ox = tx * TW + oxl;
            float acc[Q] = {0, 0, 0, 0};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const bf16x4 v = *reinterpret_cast<const bf16x4*>(tile + ((oyl * S + ky) * IW + oxl * S + kx) * CT + g * Q);
#pragma unroll
                    for (int j = 0; j < Q; ++j) acc[j] += (float)v[j] * wreg[ky * 3 + kx][j];
                }
            if (oy < Ho && ox < Wo && cok) {
                bf16x4 o;
                const int64_t oidx = (((int64_t)b * Ho + oy) * Wo + ox) * C + cg0;
                if (epi) {
                    const bf16x4 yv = ep_cur;
#pragma unroll
                    for (int j = 0; j < Q; ++j) {
                        const float yy = (float)yv[j];
                        const float dzv = acc[j] * gg_act_grad(fmaf(esc[j], yy, esh[j]), ep_act);
                        o[j] = (bf16)dzv;
                        s[j] += dzv;
                        q[j] += dzv * fmaf(ers[j], yy, -emr[j]);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < Q; ++j) {
                        o[j] = (bf16)acc[j];
                        const float r = (float)o[j];       // statistics of the stored value
                        s[j] += r;
                        q[j] += r * r;
                    }
                }
                *reinterpret_cast<bf16x4*>(y + oidx) = o;
            }
        }
        __syncthreads();
    }
    if (colstats) {
        float* red = reinterpret_cast<float*>(tile);        // [16 slots][2][64]  (8 KiB <= tile)
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            red[(ps * 2 + 0) * CT + g * Q + j] = s[j];
            red[(ps * 2 + 1) * CT + g * Q + j] = q[j];
        }
        __syncthreads();
        if (threadIdx.x < 2 * CT) {
            const int which = threadIdx.x / CT, col = threadIdx.x % CT;
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[(k * 2 + which) * CT + col];
            if (c0 + col < C) colstats[((int64_t)(b * nbands + band) * 2 + which) * C + c0 + col] = t;
        }
    }
}

// stride-2 data gradient (PatchMerging.conv2): thread = (8-channel group g fixed, pixel lane); the taps of the thread's channels
// sit in LDS; an input pixel receives only the taps whose parity matches -- ky = 1 for even rows, ky in {0, 2} for odd rows, same
// in x -- i.e. 1, 2 or 4 of the 9, found with bit tests instead of the generic kernel's 9 x (modulo, divide, branch).
// Optional BatchNorm-backward fusions as in the stride-1 kernel: IN2 forms dy = c0*dz + c1*y + c2 at every tap from (dz, y_in);
// EPI stores acc * act'(BN(ep_y)) and leaves one (sum dz, sum dz*xhat) row per block in `part`.
struct DwS2Fuse {
    const bf16* y_in; const float* in_coef;
    const bf16* ep_y; const float* ep_stat; const float* ep_gamma; const float* ep_beta; int ep_act; float* part;
};
template <bool IN2, bool EPI>
__global__ __launch_bounds__(256) void dwconv3x3_s2_bwd_data_kernel(const bf16* __restrict__ dy, const float* __restrict__ wt,
                                                                    bf16* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo,
                                                                    int CG, int PP, DwS2Fuse f) {
    // One trip = the 2x2 input quad (2a..2a+1, 2c..2c+1): its four pixels draw on the same 2x2 neighbourhood of dy, {a, a+1} x {c, c+1},
    // with 1 + 2 + 2 + 4 = all 9 taps between them -- 4 (IN2: 8) loads per quad instead of 9 (18), every load issued up front and
    // range-checked by the buffer descriptor instead of branched around, and the IN2 affine applied once per dy element.
    extern __shared__ float s2_lds[];       // taps [9][C]; IN2: coef [3][C]; EPI: scale, shift, rstd, -mean*rstd [4][C]; EPI: red [PP][2][C]
    float* taps = s2_lds;
    float* ctab = s2_lds + 9 * C;
    float* red = s2_lds + 16 * C;
    for (int i = threadIdx.x; i < 9 * C; i += blockDim.x) taps[i] = wt[i];
    if (IN2 || EPI) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            if (IN2) { ctab[c] = f.in_coef[c]; ctab[C + c] = f.in_coef[C + c]; ctab[2 * C + c] = f.in_coef[2 * C + c]; }
            if (EPI) {
                const float mu = f.ep_stat[c], rstd = f.ep_stat[C + c], sc = rstd * f.ep_gamma[c];
                ctab[3 * C + c] = sc; ctab[4 * C + c] = f.ep_beta[c] - mu * sc; ctab[5 * C + c] = rstd; ctab[6 * C + c] = -mu * rstd;
            }
        }
    }
    __syncthreads();
    const int g = threadIdx.x % CG, pp = threadIdx.x / CG;
    int gofs = g * 8;                       // laundered once per trip: keeps the table reads inside the loop (9 tap rows = 72 VGPRs if hoisted)
    auto row2 = [&](const float* base, f32x2 (&o)[4]) {
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(base + gofs), t1 = *reinterpret_cast<const f32x4*>(base + gofs + 4);
        o[0] = (f32x2){t0[0], t0[1]}; o[1] = (f32x2){t0[2], t0[3]}; o[2] = (f32x2){t1[0], t1[1]}; o[3] = (f32x2){t1[2], t1[3]};
    };
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((unsigned)B * Ho * Wo * C * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)(IN2 ? f.y_in : dy), 0, (int)((unsigned)B * Ho * Wo * C * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI ? f.ep_y : dy), 0, EPI ? (int)((unsigned)B * H * W * C * 2u) : 0, 0x00020000);
    f32x2 s2[4], q2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) s2[q] = q2[q] = (f32x2)(0.f);
    const bool ep_gelu = f.ep_act == GG_ACT_GELU;
    const int64_t nquad = (int64_t)B * Ho * Wo;
    const unsigned HWo = (unsigned)Ho * (unsigned)Wo;
    const unsigned rowO = (unsigned)Wo * C * 2u, rowI = (unsigned)W * C * 2u, pixB = (unsigned)C * 2u;
    for (int64_t p = (int64_t)blockIdx.x * PP + pp; p < nquad; p += (int64_t)gridDim.x * PP) {
        asm volatile("" : "+v"(gofs));
        const unsigned pu = (unsigned)p;
        const unsigned b = pu / HWo, rem = pu - b * HWo;
        const int a = (int)(rem / (unsigned)Wo), c = (int)(rem - (unsigned)a * (unsigned)Wo);
        const bool a1 = a + 1 < Ho, c1 = c + 1 < Wo;                 // neighbours inside dy
        const bool r1 = 2 * a + 1 < H, x1 = 2 * c + 1 < W;           // odd row / column of the quad inside dx
        const unsigned o00 = pu * pixB + g * 16u;
        const unsigned ofs[4] = {o00, c1 ? o00 + pixB : DW_COL_OOB, a1 ? o00 + rowO : DW_COL_OOB, (a1 && c1) ? o00 + rowO + pixB : DW_COL_OOB};
        dw_u32x4 rd[4], ry[4], re[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            rd[k] = __builtin_amdgcn_raw_buffer_load_b128(rsD, (int)ofs[k], 0, 0);
            if (IN2) ry[k] = __builtin_amdgcn_raw_buffer_load_b128(rsY, (int)ofs[k], 0, 0);
        }
        const unsigned i00 = ((b * (unsigned)H + 2u * a) * (unsigned)W + 2u * c) * pixB + g * 16u;
        const unsigned iofs[4] = {i00, x1 ? i00 + pixB : DW_COL_OOB, r1 ? i00 + rowI : DW_COL_OOB, (r1 && x1) ? i00 + rowI + pixB : DW_COL_OOB};
        if (EPI) {
#pragma unroll
            for (int k = 0; k < 4; ++k) re[k] = __builtin_amdgcn_raw_buffer_load_b128(rsE, (int)iofs[k], 0, 0);
        }
        // v[k] = dy element k of the neighbourhood (00, 01, 10, 11); IN2: as the unfused path stores it (bf16-rounded), 0 outside dy
        f32x2 v[4][4];
        if (IN2) {
            f32x2 ca[4], cb[4], cc[4];
            row2(ctab, ca); row2(ctab + C, cb); row2(ctab + 2 * C, cc);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool ok = ofs[k] != DW_COL_OOB;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 t = dw_unpack2(dw_pack2(ca[q] * dw_unpack2(rd[k][q]) + (cb[q] * dw_unpack2(ry[k][q]) + cc[q])));
                    v[k][q] = ok ? t : (f32x2)(0.f);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[k][q] = dw_unpack2(rd[k][q]);
        }
        // taps [ky][kx]; pixel (even,even): t11 v00 | (even,odd): t10 v01 + t12 v00 | (odd,even): t01 v10 + t21 v00 | (odd,odd): t00 v11 + t02 v10 + t20 v01 + t22 v00
        f32x2 acc[4][4], tp[4];
        row2(taps + 4 * C, tp);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[0][q] = v[0][q] * tp[q];
        row2(taps + 3 * C, tp);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[1][q] = v[1][q] * tp[q];
        row2(taps + 5 * C, tp);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[1][q] = v[0][q] * tp[q] + acc[1][q];
        row2(taps + 1 * C, tp);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[2][q] = v[2][q] * tp[q];
        row2(taps + 7 * C, tp);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[2][q] = v[0][q] * tp[q] + acc[2][q];
        row2(taps + 0 * C, tp);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[3][q] = v[3][q] * tp[q];
        row2(taps + 2 * C, tp);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[3][q] = v[2][q] * tp[q] + acc[3][q];
        row2(taps + 6 * C, tp);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[3][q] = v[1][q] * tp[q] + acc[3][q];
        row2(taps + 8 * C, tp);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[3][q] = v[0][q] * tp[q] + acc[3][q];
        f32x2 esc[4], esh[4], ers[4], emr[4];
        if (EPI) { row2(ctab + 3 * C, esc); row2(ctab + 4 * C, esh); row2(ctab + 5 * C, ers); row2(ctab + 6 * C, emr); }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool live = iofs[k] != DW_COL_OOB;
            dw_u32x4 o;
            if (EPI) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 yv = dw_unpack2(re[k][q]);
                    o[q] = dw_pack2(acc[k][q] * gg_act_grad_v2(yv * esc[q] + esh[q], ep_gelu));
                    const f32x2 r = live ? dw_unpack2(o[q]) : (f32x2)(0.f);
                    s2[q] += r; q2[q] += r * (yv * ers[q] + emr[q]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = dw_pack2(acc[k][q]);
            }
            if (live) *reinterpret_cast<dw_u32x4*>(reinterpret_cast<char*>(dx) + iofs[k]) = o;
        }
    }
    if (EPI) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            red[(pp * 2 + 0) * C + g * 8 + 2 * q] = s2[q].x; red[(pp * 2 + 0) * C + g * 8 + 2 * q + 1] = s2[q].y;
            red[(pp * 2 + 1) * C + g * 8 + 2 * q] = q2[q].x; red[(pp * 2 + 1) * C + g * 8 + 2 * q + 1] = q2[q].y;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
            float t = 0.f;
            for (int k = 0; k < PP; ++k) t += red[k * 2 * C + i];
            f.part[(int64_t)blockIdx.x * 2 * C + i] = t;
        }
    }
}

// dx[b,iy,ix,c] = sum_{ky,kx} w[ky][kx][c] * dy[b,oy,ox,c],  oy*stride + ky - 1 == iy
__global__ __launch_bounds__(256) void dwconv3x3_bwd_data_kernel(const bf16* __restrict__ dy, const float* __restrict__ wt,
                                                                 bf16* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo,
                                                                 int stride) {
    const int cg = C >> 3;
    const int64_t total = (int64_t)B * H * W * cg;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)((unsigned)i % (unsigned)cg);
        const int64_t p = (unsigned)i / (unsigned)cg;
        const unsigned pu = (unsigned)p;
        const int ix = (int)(pu % (unsigned)W);
        const int iy = (int)((pu / (unsigned)W) % (unsigned)H);
        const int b = (int)(pu / ((unsigned)W * (unsigned)H));
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + 1 - ky;
            if (ty < 0 || (ty % stride) != 0) continue;
            const int oy = ty / stride;
            if (oy >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + 1 - kx;
                if (tx < 0 || (tx % stride) != 0) continue;
                const int ox = tx / stride;
                if (ox >= Wo) continue;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(dy + (((int64_t)b * Ho + oy) * Wo + ox) * C + g * 8);
                const float* wp = wt + (ky * 3 + kx) * C + g * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)v[j] * wp[j];
            }
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)acc[j];
        *reinterpret_cast<bf16x8*>(dx + p * C + g * 8) = o;
    }
}

// Weight gradient of a stride-1 depthwise 3x3 in the column-walking geometry: dW[c][ky][kx] = sum_{b,y,x} x[b,y+ky-1,x+kx-1,c] dy[b,y,x,c].
// Thread = (NQ channel pairs, one column), block = PX columns x all channels, walking down every image b = blockIdx.y,
// blockIdx.y + gridDim.y, ...; the 3x3 window of x sits in registers (as in the forward), 9 x NQ fp32 pair accumulators.
// One partial row [9][C] per block -> the usual two-stage reduction.  (The gather kernel it replaces issued 9 loads per output
// and ran at 0.6 TB/s.)
template <int NQ>
__global__ __launch_bounds__(256, NQ == 4 ? 2 : 3) void dwconv3x3_wgrad_walk_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                                                                    int B, int H, int W, int C, int CG, int PX, int nbx,
                                                                                    float* __restrict__ part) {
    typedef typename DwRaw<NQ>::T Raw;
    constexpr int NC = 2 * NQ;
    extern __shared__ float wg_red[];          // [PX][9][C]
    const int cg = threadIdx.x % CG, px = threadIdx.x / CG;
    const int bx = blockIdx.x;
    const int xo = bx * PX + px;
    const int c0 = cg * NC;
    auto uniform_ptr = [](const bf16* ptr) {
        const unsigned long long a = (unsigned long long)ptr;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return (void*)(((unsigned long long)hi << 32) | lo);
    };
    const int img_bytes = __builtin_amdgcn_readfirstlane(H * W * C * 2);
    unsigned colo[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int ix = xo + k - 1;
        colo[k] = (ix >= 0 && ix < W && xo < W) ? (unsigned)(ix * C + c0) * 2u : DW_COL_OOB;
    }
    const unsigned rowb = (unsigned)W * C * 2u;
    f32x2 acc[9][NQ];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[t][q] = (f32x2)(0.f);
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        const int64_t img = (int64_t)b * H * W * C;
        const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(x + img), 0, img_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(dy + img), 0, img_bytes, 0x00020000);
        auto load_row = [&](int iy, Raw (&raw)[3]) {
            const unsigned ro = (iy >= 0 && iy < H) ? (unsigned)iy * rowb : DW_ROW_OOB;
#pragma unroll
            for (int k = 0; k < 3; ++k) raw[k] = DwRaw<NQ>::load(rsx, (int)(colo[k] + ro));
        };
        auto load_dy = [&](int iy) { return DwRaw<NQ>::load(rsd, (int)(colo[1] + ((iy >= 0 && iy < H) ? (unsigned)iy * rowb : DW_ROW_OOB))); };
        f32x2 win[3][3][NQ];
        Raw raw[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int q = 0; q < NQ; ++q) win[0][k][q] = (f32x2)(0.f);
        load_row(0, raw);
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int q = 0; q < NQ; ++q) win[1][k][q] = dw_unpack2(raw[k][q]);
        load_row(1, raw);
        Raw draw = load_dy(0);
#define GG_WG_STEP(RA, RB, RC, yy)                                                                                           \
        {                                                                                                                    \
            _Pragma("unroll") for (int k = 0; k < 3; ++k)                                                                    \
                _Pragma("unroll") for (int q = 0; q < NQ; ++q) win[RC][k][q] = dw_unpack2(raw[k][q]);                        \
            load_row((yy) + 2, raw);                                                                                         \
            f32x2 d[NQ];                                                                                                     \
            _Pragma("unroll") for (int q = 0; q < NQ; ++q) d[q] = dw_unpack2(draw[q]);                                       \
            draw = load_dy((yy) + 1);                                                                                        \
            _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                                                  \
                const int rsl = t < 3 ? RA : (t < 6 ? RB : RC);                                                              \
                _Pragma("unroll") for (int q = 0; q < NQ; ++q) acc[t][q] = win[rsl][t % 3][q] * d[q] + acc[t][q];            \
            }                                                                                                                \
        }
        for (int y0 = 0; y0 < H; y0 += 3) {      // rows past the image read zeros from dy: no contribution
            GG_WG_STEP(0, 1, 2, y0)
            GG_WG_STEP(1, 2, 0, y0 + 1)
            GG_WG_STEP(2, 0, 1, y0 + 2)
        }
#undef GG_WG_STEP
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            wg_red[(px * 9 + t) * C + c0 + 2 * q] = acc[t][q].x;
            wg_red[(px * 9 + t) * C + c0 + 2 * q + 1] = acc[t][q].y;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * C; i += blockDim.x) {
        float t = 0.f;
        for (int k = 0; k < PX; ++k) t += wg_red[k * 9 * C + i];
        part[((int64_t)blockIdx.y * nbx + bx) * 9 * C + i] = t;
    }
}

// dw partials [gridDim.x][9][C]: sum over this block's output pixels of dy * x_tap
__global__ void dwconv3x3_bwd_weight_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, int B, int H, int W, int C,
                                            int Ho, int Wo, int stride, int CG, int PP, int pix_per_block,
                                            float* __restrict__ part) {
    extern __shared__ float sred[];   // [PP][9][C]
    const int g = threadIdx.x % CG, pp = threadIdx.x / CG;
    const int64_t total = (int64_t)B * Ho * Wo;
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = min(total, p0 + pix_per_block);
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    for (int64_t p = p0 + pp; p < p1; p += PP) {
        const unsigned pu = (unsigned)p;
        const int ox = (int)(pu % (unsigned)Wo);
        const int oy = (int)((pu / (unsigned)Wo) % (unsigned)Ho);
        const int b = (int)(pu / ((unsigned)Wo * (unsigned)Ho));
        const bf16x8 d = *reinterpret_cast<const bf16x8*>(dy + p * C + g * 8);
        float df[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) df[j] = (float)d[j];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * stride + ky - 1;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * stride + kx - 1;
                if (ix < 0 || ix >= W) continue;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (((int64_t)b * H + iy) * W + ix) * C + g * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[ky * 3 + kx][j] += df[j] * (float)v[j];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) sred[(pp * 9 + t) * C + g * 8 + j] = acc[t][j];
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * C; i += blockDim.x) {
        float t = 0.f;
        for (int k = 0; k < PP; ++k) t += sred[k * 9 * C + i];
        part[(int64_t)blockIdx.x * 9 * C + i] = t;
    }
}
// part [nparts][9][C] -> grad of conv.weight (C,1,3,3): grad[c*9 + tap] (+)= sum
__global__ void dwconv_wgrad_final_kernel(const float* __restrict__ part, int nparts, int C, float* __restrict__ grad, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // index into [9][C]
    if (i >= 9 * C) return;
    double s = 0.0;
    for (int k = 0; k < nparts; ++k) s += (double)part[(int64_t)k * 9 * C + i];
    const int tap = i / C, c = i % C;
    float* o = grad + c * 9 + tap;
    *o = accumulate ? *o + (float)s : (float)s;
}

// ------------------------------------------------------------------------------------------- host
static int grid_for(int64_t n, int cap = 16384) { return (int)std::min<int64_t>(gg_cdiv(n, 256), cap); }

extern "C" int gg_im2col_nchw3_f32(const float* x, void* col, int B, int H, int W, int stride, void* stream) {
    GG_CHECK(x && col && B > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2), "gg_im2col_nchw3_f32: bad args");
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    GG_PROF(GG_CAT_MOVE, 0, 12.0 * B * H * W + 64.0 * B * Ho * Wo, stream);
    constexpr int RB = 4;
    const size_t lds = (size_t)3 * (2 * RB + 1) * W * sizeof(float);
    if (stride == 2 && (W & 3) == 0 && ((uintptr_t)x & 15) == 0 && lds <= 64 * 1024) {
        const int nbands = (int)gg_cdiv(Ho, RB);
        hipLaunchKernelGGL(im2col_nchw3_s2_lds_kernel<RB>, dim3((unsigned)(B * nbands)), dim3(256), lds, (hipStream_t)stream, x, (bf16*)col, B, H, W,
                           Ho, Wo, nbands);
    } else {
        hipLaunchKernelGGL(im2col_nchw3_kernel, dim3(grid_for((int64_t)B * Ho * Wo * 4, 65536)), dim3(256), 0, (hipStream_t)stream, x, (bf16*)col, B,
                           H, W, Ho, Wo, stride);
    }
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_im2col_nhwc_bf16(const void* x, void* col, int B, int H, int W, int C, int stride, void* stream) {
    GG_CHECK(x && col && B > 0 && (C & 7) == 0 && (stride == 1 || stride == 2), "gg_im2col_nhwc_bf16: bad args (C %% 8)");
    GG_CHECK((int64_t)B * H * W * 9 * (C / 8) < ((int64_t)1 << 32), "gg_im2col_nhwc_bf16: tensor too large for 32-bit indexing");
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    GG_PROF(GG_CAT_MOVE, 0, 2.0 * B * H * W * C + 18.0 * B * Ho * Wo * C, stream);
    hipLaunchKernelGGL(im2col_nhwc_kernel, dim3(grid_for((int64_t)B * Ho * Wo * 9 * (C / 8), 65536)), dim3(256), 0,
                       (hipStream_t)stream, (const bf16*)x, (bf16*)col, B, H, W, C, Ho, Wo, stride);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_im2col_nhwc_bn_bf16(const void* y, const float* stat, const float* gamma, const float* beta, int act, void* col, int B,
                                      int H, int W, int C, int stride, void* stream) {
    GG_CHECK(y && stat && gamma && beta && col && B > 0 && (C & 7) == 0 && C <= 512 && (stride == 1 || stride == 2), "gg_im2col_nhwc_bn_bf16: bad args (C %% 8, C <= 512)");
    GG_CHECK(act == GG_ACT_NONE || act == GG_ACT_GELU, "gg_im2col_nhwc_bn_bf16: act must be none or gelu");
    GG_CHECK((int64_t)B * H * W * 9 * (C / 8) < ((int64_t)1 << 32), "gg_im2col_nhwc_bn_bf16: tensor too large for 32-bit indexing");
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    GG_PROF(GG_CAT_MOVE, 0, 2.0 * B * H * W * C + 18.0 * B * Ho * Wo * C, stream);
    hipLaunchKernelGGL(im2col_nhwc_bn_kernel, dim3(grid_for((int64_t)B * Ho * Wo * 9 * (C / 8), 65536)), dim3(256), 0,
                       (hipStream_t)stream, (const bf16*)y, stat, gamma, beta, act, (bf16*)col, B, H, W, C, Ho, Wo, stride);
    GG_LAUNCH_CHECK();
    return 0;
}
// col2im + the BatchNorm-backward reduce of the ConvNorm whose output was gathered: the scattered gradient da is formed per pixel
// (bf16-rounded as the unfused path stores it), multiplied by act'(BN(y)) and written as dz; the per-channel sums (sum dz, sum dz*xhat)
// leave as one partial row per block.  da is never written, the reduce pass never runs.
__global__ __launch_bounds__(256) void col2im_nhwc_bnbwd_kernel(const bf16* __restrict__ dcol, const bf16* __restrict__ y,
                                                                const float* __restrict__ stat, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, int act, bf16* __restrict__ dz,
                                                                float* __restrict__ part, int B, int H, int W, int C, int Ho, int Wo, int CG, int PP) {
    extern __shared__ float c2i_red[];          // [PP][2][C]
    const int g = threadIdx.x % CG, pp = threadIdx.x / CG;
    const int c0 = g * 8;
    float mu[8], rstd[8], ga[8], be[8], s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { mu[j] = stat[c0 + j]; rstd[j] = stat[C + c0 + j]; ga[j] = gamma[c0 + j]; be[j] = beta[c0 + j]; s[j] = q[j] = 0.f; }
    const int64_t npix = (int64_t)B * H * W;
    const unsigned HW = (unsigned)H * (unsigned)W;
    for (int64_t p = (int64_t)blockIdx.x * PP + pp; p < npix; p += (int64_t)gridDim.x * PP) {
        const unsigned pu = (unsigned)p;
        const unsigned b = pu / HW, rem = pu - b * HW;
        const int iy = (int)(rem / (unsigned)W), ix = (int)(rem - (unsigned)iy * (unsigned)W);
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(y + p * C + c0);
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // stride 2: oy = (iy + 1 - ky) / 2 with iy + 1 - ky even -> ky = 1 for even rows, ky in {0, 2} for odd rows; same in x
        const int ky0 = (iy & 1) ? 0 : 1, nky = (iy & 1) ? 2 : 1;
        const int kx0 = (ix & 1) ? 0 : 1, nkx = (ix & 1) ? 2 : 1;
        for (int a = 0; a < nky; ++a) {
            const int ky = ky0 + 2 * a, oy = (iy + 1 - ky) >> 1;
            if (oy < 0 || oy >= Ho) continue;
            for (int c = 0; c < nkx; ++c) {
                const int kx = kx0 + 2 * c, ox = (ix + 1 - kx) >> 1;
                if (ox < 0 || ox >= Wo) continue;
                const bf16x8 d = *reinterpret_cast<const bf16x8*>(dcol + (((int64_t)b * Ho + oy) * Wo + ox) * (9 * C) + (ky * 3 + kx) * C + c0);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)d[j];
            }
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = ((float)v[j] - mu[j]) * rstd[j];
            const float dpre = (float)(bf16)acc[j] * gg_act_grad(ga[j] * xh + be[j], act);
            o[j] = (bf16)dpre;
            s[j] += dpre;
            q[j] += dpre * xh;
        }
        *reinterpret_cast<bf16x8*>(dz + p * C + c0) = o;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { c2i_red[(pp * 2 + 0) * C + c0 + j] = s[j]; c2i_red[(pp * 2 + 1) * C + c0 + j] = q[j]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        float t = 0.f;
        for (int k = 0; k < PP; ++k) t += c2i_red[k * 2 * C + i];
        part[(int64_t)blockIdx.x * 2 * C + i] = t;
    }
}
// stride-2 3x3 only (PatchEmbed).  part: [nparts][2][C] with nparts rows as given (<= 65535); follow with gg_bn_bwd_finalize(part, nparts, ...)
extern "C" int gg_col2im_nhwc_bnbwd_bf16(const void* dcol, const void* y, const float* stat, const float* gamma, const float* beta, int act,
                                         void* dz, float* part, int nparts, int B, int H, int W, int C, void* stream) {
    GG_CHECK(dcol && y && stat && gamma && beta && dz && part && nparts > 0 && nparts <= 65535 && B > 0 && (C & 7) == 0 && C / 8 <= 256,
             "gg_col2im_nhwc_bnbwd_bf16: bad args");
    GG_CHECK((int64_t)B * H * W < ((int64_t)1 << 32), "gg_col2im_nhwc_bnbwd_bf16: tensor too large for 32-bit pixel indexing");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int CG = C / 8, PP = std::max(1, 256 / CG);
    const size_t lds = (size_t)PP * 2 * C * sizeof(float);
    GG_CHECK(lds <= 64 * 1024, "gg_col2im_nhwc_bnbwd_bf16: C too large");
    GG_PROF(GG_CAT_MOVE, 0, 6.0 * B * H * W * C + 18.0 * B * Ho * Wo * C, stream);
    hipLaunchKernelGGL(col2im_nhwc_bnbwd_kernel, dim3((unsigned)nparts), dim3(CG * PP), lds, (hipStream_t)stream, (const bf16*)dcol, (const bf16*)y,
                       stat, gamma, beta, act, (bf16*)dz, part, B, H, W, C, Ho, Wo, CG, PP);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_col2im_nhwc_bf16(const void* dcol, void* dx, int B, int H, int W, int C, int stride, void* stream) {
    GG_CHECK(dcol && dx && B > 0 && (C & 7) == 0 && (stride == 1 || stride == 2), "gg_col2im_nhwc_bf16: bad args");
    GG_CHECK((int64_t)B * H * W * (C / 8) < ((int64_t)1 << 32), "gg_col2im_nhwc_bf16: tensor too large for 32-bit indexing");
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    GG_PROF(GG_CAT_MOVE, 0, 2.0 * B * H * W * C + 18.0 * B * Ho * Wo * C, stream);
    hipLaunchKernelGGL(col2im_nhwc_kernel, dim3(grid_for((int64_t)B * H * W * (C / 8), 65536)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)dcol, (bf16*)dx, B, H, W, C, Ho, Wo, stride);
    GG_LAUNCH_CHECK();
    return 0;
}

struct DwGeom { int CG, PP, threads, pix_per_block, nblocks; };
static DwGeom dw_geom(int64_t npix, int C, int lds_floats_per_pp) {
    DwGeom g;
    g.CG = C / 8;
    g.PP = std::max(1, std::min(256 / g.CG, 15360 / lds_floats_per_pp));   // <= 60 KB of dynamic LDS
    g.threads = g.CG * g.PP;
    int64_t target_blocks = 4096;
    int64_t ppb = std::max<int64_t>(gg_cdiv(npix, target_blocks), (int64_t)g.PP * 4);
    ppb = gg_align(ppb, g.PP);
    g.pix_per_block = (int)ppb;
    g.nblocks = (int)gg_cdiv(npix, ppb);
    return g;
}
// stride-1 convs take the column-walking kernel: PX columns x C/NC channel groups per block (<= 256 threads), whole image height.
// NC = 8 channels per thread, 4 for the doubly fused backward kernel.
static int dw_walk_px(int C, int nc = 8) { return std::max(1, 256 / (C / nc)); }
static bool dw_walk_ok(int C, int stride) { return stride == 1 && (C / 8) <= 256 && gg_dev_env("GG_DW_TILED") == nullptr; }
static bool dw_walk_fused4(int C) { return (C / 4) <= 256; }       // both fusions at once run 4 channels per thread
// fused stride-1 forward (producer BatchNorm + activation on load): 4 output columns x 4 channels per thread
static const int kDwMultiCols = 4;
static bool dw_multi_ok(int C) {
    const int CG = C / 4, PX = std::max(1, 256 / std::max(CG, 1));
    return (C & 3) == 0 && CG <= 256 && (size_t)PX * 2 * C * sizeof(float) <= 64 * 1024 && gg_dev_env("GG_DW_TILED") == nullptr &&
           gg_dev_env("GG_DW_NO_MULTI") == nullptr;
}
static bool dw_multi_plain(int C) { return gg_dev_env("GG_DW_NO_MULTI_PLAIN") == nullptr && dw_multi_ok(C); }      // the plain stride-1 forward too: half the loads per result (-8..15 % on 14x14 / 28x28 maps)
static bool dw_multi_bwd(int C) { return gg_dev_env("GG_DW_NO_MULTI_BWD") == nullptr && dw_multi_ok(C); }          // stride-1 data gradients (plain and fused)
static const int kDwMultiColsEpi = 2;       // variants with the act'(BN) epilogue: 2 columns per thread (4 spill at 256 registers)
static int dw_multi_nbx(int W, int C, int ncol = kDwMultiCols) { return (int)gg_cdiv(W, std::max(1, 256 / (C / 4)) * ncol); }
// stride-2 forward: the walking kernel when its per-thread state fits (8 channels per thread, <= 256 channel groups) and the image fits the
// 30-bit offsets; GG_DW_TILED keeps the LDS-tiled kernel
static bool dw_s2_walk_ok(int C) { return (C / 8) <= 256 && (C & 7) == 0 && gg_dev_env("GG_DW_TILED") == nullptr && gg_dev_env("GG_DW_S2_TILED") == nullptr; }
static int dwconv_s2_walk_launch(const void* x, const float* wt, void* y, int B, int H, int W, int C, float* colstats, void* stream,
                                 const DwWalkFuse* fuse) {
    GG_CHECK((int64_t)H * W * C * 2 < 0x40000000LL, "dwconv: image too large for 30-bit offsets");
    GG_CHECK(((uintptr_t)wt & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "dwconv: operands must be 16-byte aligned");
    DwWalkFuse f;
    memset(&f, 0, sizeof(f));
    if (fuse) f = *fuse;
    const bool in1 = f.in_stat != nullptr;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int CG = C / 8, PX = dw_walk_px(C, 8), nbx = (int)gg_cdiv(Wo, PX);
    GG_PROF(GG_CAT_DWCONV, 18.0 * B * Ho * Wo * C, 2.0 * B * C * ((double)H * W + (double)Ho * Wo), stream);
    const size_t lds = ((size_t)PX * 2 * C + (in1 ? 2 * (size_t)C : 0)) * sizeof(float);
    GG_CHECK(lds <= 64 * 1024, "dwconv: C=%d needs %zu bytes of LDS", C, lds);
    const dim3 grid((unsigned)(B * nbx)), block(CG * PX);
    if (in1) hipLaunchKernelGGL((dwconv3x3_s2_walk_kernel<true>), grid, block, lds, (hipStream_t)stream, (const bf16*)x, wt, (bf16*)y, H, W, C, Ho, Wo,
                                CG, PX, nbx, colstats, f);
    else hipLaunchKernelGGL((dwconv3x3_s2_walk_kernel<false>), grid, block, lds, (hipStream_t)stream, (const bf16*)x, wt, (bf16*)y, H, W, C, Ho, Wo,
                            CG, PX, nbx, colstats, f);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_dwconv_stat_rows(int B, int Ho, int Wo, int C, int stride) {
    if (stride == 1 && dw_multi_plain(C)) return B * dw_multi_nbx(Wo, C);
    if (stride == 2 && dw_s2_walk_ok(C)) return B * (int)gg_cdiv(Wo, dw_walk_px(C, 8));
    if (dw_walk_ok(C, stride)) return B * (int)gg_cdiv(Wo, dw_walk_px(C));
    return B * (int)gg_cdiv(Ho, 8);
}
extern "C" int gg_dwconv_tiled_stat_rows(int B, int Ho) { return B * (int)gg_cdiv(Ho, 8); }
/* rows written by gg_dwconv3x3_bwd_data_fused with an output-side fusion */
extern "C" int gg_dwconv_fused_stat_rows(int B, int H, int W, int C, int with_input_fusion) {
    if (dw_multi_bwd(C)) return B * dw_multi_nbx(W, C, kDwMultiColsEpi);
    if (!dw_walk_ok(C, 1)) return B * (int)gg_cdiv(H, 8);
    (void)with_input_fusion;
    const int nc = dw_walk_fused4(C) ? 4 : 8;
    return B * (int)gg_cdiv(W, dw_walk_px(C, nc));
}
/* partial-statistics rows gg_dwconv3x3_fwd_fused writes */
extern "C" int gg_dwconv_fwd_fused_stat_rows(int B, int H, int W, int C, int stride) {
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    if (stride == 2) return gg_dwconv_stat_rows(B, Ho, Wo, C, 2);
    if (dw_multi_ok(C)) return B * dw_multi_nbx(W, C);
    return gg_dwconv_fused_stat_rows(B, H, W, C, 1);
}
static int dwconv_multi_launch(const void* x, const float* wt, void* y, int B, int H, int W, int C, int flip, float* colstats, void* stream,
                               const DwWalkFuse& f) {
    GG_CHECK((int64_t)H * W * C * 2 < 0x40000000LL, "dwconv: image too large for 30-bit offsets");
    GG_CHECK(((uintptr_t)wt & 7) == 0 && ((uintptr_t)x & 7) == 0 && ((uintptr_t)y & 7) == 0, "dwconv: operands must be 8-byte aligned");
    const bool in1 = f.in_stat != nullptr, in2 = f.in_coef != nullptr, epi = f.ep_y != nullptr;
    GG_CHECK(!(in1 && (in2 || epi)), "dwconv: producer fusion excludes the backward fusions");
    const int CG = C / 4, PX = std::max(1, 256 / CG), nbx = dw_multi_nbx(W, C, epi ? kDwMultiColsEpi : kDwMultiCols);
    GG_PROF(GG_CAT_DWCONV, 18.0 * B * H * W * C, 2.0 * B * C * (double)H * W * (2 + in2 + epi), stream);
    const size_t lds = (size_t)PX * 2 * C * sizeof(float);
    const dim3 grid((unsigned)(B * nbx)), block(CG * PX);
#define GG_DWM(N_, I_, E_) hipLaunchKernelGGL((dwconv3x3_s1_multi_kernel<N_, I_, E_>), grid, block, lds, (hipStream_t)stream, (const bf16*)x, wt, \
                                              (bf16*)y, H, W, C, CG, PX, nbx, flip, colstats, f)
    if (in1) GG_DWM(kDwMultiCols, 1, false);
    else if (in2 && epi) GG_DWM(kDwMultiColsEpi, 2, true);
    else if (in2) GG_DWM(kDwMultiCols, 2, false);
    else if (epi) GG_DWM(kDwMultiColsEpi, 0, true);
    else GG_DWM(kDwMultiCols, 0, false);
#undef GG_DWM
    GG_LAUNCH_CHECK();
    return 0;
}
static int dwconv_walk_launch(const void* x, const float* wt, void* y, int B, int H, int W, int C, int flip, float* colstats, void* stream,
                              const DwWalkFuse* fuse = nullptr) {
    GG_CHECK((int64_t)H * W * C * 2 < 0x40000000LL, "dwconv: image too large for 30-bit offsets");
    GG_CHECK(((uintptr_t)wt & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "dwconv: operands must be 16-byte aligned");
    DwWalkFuse f;
    memset(&f, 0, sizeof(f));
    if (fuse) f = *fuse;
    const bool in2 = f.in_coef != nullptr, epi = f.ep_y != nullptr, in1 = f.in_stat != nullptr;
    const int nc = ((in1 || in2 || epi) && dw_walk_fused4(C)) ? 4 : 8;      // every fused variant: 4 channels per thread
    const int CG = C / nc, PX = dw_walk_px(C, nc), nbx = (int)gg_cdiv(W, PX);
    GG_PROF(GG_CAT_DWCONV, 18.0 * B * H * W * C, 2.0 * B * C * (double)H * W * (2 + in2 + epi), stream);
    const size_t lds = ((size_t)PX * 2 * C + ((in1 || in2 || epi) ? 16 * (size_t)C : 0)) * sizeof(float);
    GG_CHECK(lds <= 64 * 1024, "dwconv: C=%d needs %zu bytes of LDS", C, lds);
    const dim3 grid((unsigned)(B * nbx)), block(CG * PX);
#define GG_DW_WALK(I_, E_, N_) hipLaunchKernelGGL((dwconv3x3_walk_kernel<I_, E_, N_>), grid, block, lds, (hipStream_t)stream, (const bf16*)x, wt, \
                                                  (bf16*)y, H, W, C, CG, PX, nbx, flip, colstats, f)
    if (in1) { GG_CHECK(!in2 && !epi, "dwconv: producer fusion excludes the backward fusions"); if (nc == 4) GG_DW_WALK(1, false, 2); else GG_DW_WALK(1, false, 4); }
    else if (in2 && epi) { if (nc == 4) GG_DW_WALK(2, true, 2); else GG_DW_WALK(2, true, 4); }
    else if (in2) { if (nc == 4) GG_DW_WALK(2, false, 2); else GG_DW_WALK(2, false, 4); }
    else if (epi) { if (nc == 4) GG_DW_WALK(0, true, 2); else GG_DW_WALK(0, true, 4); }
    else GG_DW_WALK(0, false, 4);
#undef GG_DW_WALK
    GG_LAUNCH_CHECK();
    return 0;
}

// stride-2 data gradient geometry: 8-channel groups x pixel lanes, <= 4096 blocks (= partial statistics rows with ep_y)
static bool dw_s2_ok(int C) { return (C / 8) <= 256 && (16 + 2 * std::max(1, 256 / (C / 8))) * (int64_t)C * 4 <= 60 * 1024; }
static int dw_s2_blocks(int B, int H, int W, int C) {          // trips are 2x2 input quads = output-resolution pixels
    return (int)std::min<int64_t>(gg_cdiv((int64_t)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1), std::max(1, 256 / (C / 8))), 4096);
}
extern "C" int gg_dwconv_s2_fused_stat_rows(int B, int H, int W, int C) { return dw_s2_blocks(B, H, W, C); }
static int dwconv_s2_bwd_launch(const void* dy, const float* wt, void* dx, int B, int H, int W, int C, const DwS2Fuse* fuse, void* stream) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int CG = C / 8, PP = std::max(1, 256 / CG);
    GG_CHECK((int64_t)B * H * W * C * 2 < ((int64_t)1 << 31), "dwconv stride-2 data gradient: tensor too large for 32-bit offsets");
    DwS2Fuse f;
    memset(&f, 0, sizeof(f));
    if (fuse) f = *fuse;
    const bool in2 = f.in_coef != nullptr, epi = f.ep_y != nullptr;
    const size_t lds = ((size_t)16 * C + (epi ? (size_t)PP * 2 * C : 0)) * sizeof(float);
    const dim3 grid((unsigned)dw_s2_blocks(B, H, W, C)), block(CG * PP);
#define GG_S2(I_, E_) hipLaunchKernelGGL((dwconv3x3_s2_bwd_data_kernel<I_, E_>), grid, block, lds, (hipStream_t)stream, (const bf16*)dy, wt, \
                                         (bf16*)dx, B, H, W, C, Ho, Wo, CG, PP, f)
    if (in2 && epi) GG_S2(true, true); else if (in2) GG_S2(true, false); else if (epi) GG_S2(false, true); else GG_S2(false, false);
#undef GG_S2
    GG_LAUNCH_CHECK();
    return 0;
}
// stride-2 data gradient with the BatchNorm-backward passes on both sides folded in (PatchMerging.conv2 with frozen taps); same
// contract as gg_dwconv3x3_bwd_data_fused, partial rows = gg_dwconv_s2_fused_stat_rows
extern "C" int gg_dwconv3x3_s2_bwd_data_fused(const void* dz_in, const void* y_in, const float* in_coef, const float* wt, void* out, int B,
                                              int H, int W, int C, const void* ep_y, const float* ep_stat, const float* ep_gamma,
                                              const float* ep_beta, int ep_act, float* ep_part, void* stream) {
    GG_CHECK(dz_in && wt && out && B > 0 && (C & 7) == 0 && dw_s2_ok(C), "gg_dwconv3x3_s2_bwd_data_fused: bad args");
    GG_CHECK(!in_coef || y_in, "gg_dwconv3x3_s2_bwd_data_fused: in_coef needs y_in");
    GG_CHECK(!ep_y || (ep_stat && ep_gamma && ep_beta && ep_part), "gg_dwconv3x3_s2_bwd_data_fused: epilogue needs stat/gamma/beta/partials");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    GG_PROF(GG_CAT_DWCONV, 18.0 * B * Ho * Wo * C, 2.0 * B * C * ((double)H * W * (ep_y ? 2 : 1) + (double)Ho * Wo * (in_coef ? 2 : 1)), stream);
    DwS2Fuse f;
    memset(&f, 0, sizeof(f));
    f.y_in = in_coef ? (const bf16*)y_in : nullptr; f.in_coef = in_coef;
    f.ep_y = (const bf16*)ep_y; f.ep_stat = ep_stat; f.ep_gamma = ep_gamma; f.ep_beta = ep_beta; f.ep_act = ep_act; f.part = ep_part;
    return dwconv_s2_bwd_launch(dz_in, wt, out, B, H, W, C, &f, stream);
}
struct DwFuse {
    const void* in2 = nullptr; const float* in_coef = nullptr;
    const void* ep_y = nullptr; const float* ep_stat = nullptr; const float* ep_gamma = nullptr; const float* ep_beta = nullptr; int ep_act = 0;
};
static int dwconv_tiled_launch(const void* x, const float* wt, void* y, int B, int H, int W, int C, int stride, int flip,
                               const float* in_stat, const float* in_gamma, const float* in_beta, int in_act, float* colstats,
                               void* stream, const DwFuse& f = DwFuse()) {
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const int nbands = (int)gg_cdiv(Ho, 8), nchunks = (int)gg_cdiv(C, 64);
    const int64_t blocks = (int64_t)B * nbands * nchunks;
    GG_CHECK(blocks < ((int64_t)1 << 31), "dwconv: grid too large");
    GG_PROF(GG_CAT_DWCONV, 18.0 * B * Ho * Wo * C, 2.0 * B * C * ((double)H * W + (double)Ho * Wo), stream);
    const int mode = (f.in_coef || f.ep_y) ? 2 : (in_stat ? 1 : 0);
#define GG_DW_LAUNCH(S_, M_)                                                                                                          \
    hipLaunchKernelGGL((dwconv3x3_tiled_kernel<S_, M_>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, wt, \
                       (bf16*)y, B, H, W, C, Ho, Wo, nbands, nchunks, flip, in_stat, in_gamma, in_beta, in_act, colstats,               \
                       (const bf16*)f.in2, f.in_coef, (const bf16*)f.ep_y, f.ep_stat, f.ep_gamma, f.ep_beta, f.ep_act)
    if (stride == 1) {
        if (mode == 0) GG_DW_LAUNCH(1, 0); else if (mode == 1) GG_DW_LAUNCH(1, 1); else GG_DW_LAUNCH(1, 2);
    } else {
        GG_CHECK(mode != 2, "dwconv: backward fusions are stride-1 only");
        if (mode == 0) GG_DW_LAUNCH(2, 0); else GG_DW_LAUNCH(2, 1);
    }
#undef GG_DW_LAUNCH
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_dwconv3x3_fwd(const void* x, const float* wt, void* y, int B, int H, int W, int C, int stride, float* colstats,
                                void* stream) {
    GG_CHECK(x && wt && y && B > 0 && (C & 7) == 0 && (stride == 1 || stride == 2), "gg_dwconv3x3_fwd: bad args");
    if (stride == 1 && dw_multi_plain(C)) { DwWalkFuse wf; memset(&wf, 0, sizeof(wf)); return dwconv_multi_launch(x, wt, y, B, H, W, C, 0, colstats, stream, wf); }
    if (dw_walk_ok(C, stride)) return dwconv_walk_launch(x, wt, y, B, H, W, C, 0, colstats, stream);
    if (stride == 2 && dw_s2_walk_ok(C)) return dwconv_s2_walk_launch(x, wt, y, B, H, W, C, colstats, stream, nullptr);
    return dwconv_tiled_launch(x, wt, y, B, H, W, C, stride, 0, nullptr, nullptr, nullptr, 0, colstats, stream);
}
// fused producer: x is the PRE-BatchNorm output of the previous ConvNorm; act(BN(x)) is formed while staging
extern "C" int gg_dwconv3x3_fwd_fused(const void* x, const float* in_stat, const float* in_gamma, const float* in_beta, int in_act,
                                      const float* wt, void* y, int B, int H, int W, int C, int stride, float* colstats, void* stream) {
    GG_CHECK(x && wt && y && in_stat && in_gamma && in_beta && B > 0 && (C & 7) == 0 && (stride == 1 || stride == 2),
             "gg_dwconv3x3_fwd_fused: bad args");
    if (stride == 1 && dw_multi_ok(C)) {
        DwWalkFuse wf;
        memset(&wf, 0, sizeof(wf));
        wf.in_stat = in_stat; wf.in_gamma = in_gamma; wf.in_beta = in_beta; wf.in_act = in_act;
        return dwconv_multi_launch(x, wt, y, B, H, W, C, 0, colstats, stream, wf);
    }
    if (dw_walk_ok(C, stride)) {
        DwWalkFuse wf;
        memset(&wf, 0, sizeof(wf));
        wf.in_stat = in_stat; wf.in_gamma = in_gamma; wf.in_beta = in_beta; wf.in_act = in_act;
        return dwconv_walk_launch(x, wt, y, B, H, W, C, 0, colstats, stream, &wf);
    }
    if (stride == 2 && dw_s2_walk_ok(C)) {
        DwWalkFuse wf;
        memset(&wf, 0, sizeof(wf));
        wf.in_stat = in_stat; wf.in_gamma = in_gamma; wf.in_beta = in_beta; wf.in_act = in_act;
        return dwconv_s2_walk_launch(x, wt, y, B, H, W, C, colstats, stream, &wf);
    }
    return dwconv_tiled_launch(x, wt, y, B, H, W, C, stride, 0, in_stat, in_gamma, in_beta, in_act, colstats, stream);
}
// stride-1 data gradient with the BatchNorm-backward passes on both sides folded in (see the kernel comment):
//   input  = coef0*dz_in + coef1*y_in + coef2   when in_coef != NULL (else dz_in is used as is)
//   output = conv_T(input) * act'(BN(ep_y))      and stats rows (sum, sum*xhat) when ep_y != NULL (else the plain gradient)
extern "C" int gg_dwconv3x3_bwd_data_fused(const void* dz_in, const void* y_in, const float* in_coef, const float* wt, void* out, int B, int H,
                                           int W, int C, const void* ep_y, const float* ep_stat, const float* ep_gamma,
                                           const float* ep_beta, int ep_act, float* ep_part, void* stream) {
    GG_CHECK(dz_in && wt && out && B > 0 && (C & 7) == 0, "gg_dwconv3x3_bwd_data_fused: bad args");
    GG_CHECK(!in_coef || y_in, "gg_dwconv3x3_bwd_data_fused: in_coef needs y_in");
    GG_CHECK(!ep_y || (ep_stat && ep_gamma && ep_beta && ep_part), "gg_dwconv3x3_bwd_data_fused: epilogue needs stat/gamma/beta/partials");
    if (dw_multi_bwd(C)) {
        DwWalkFuse wf;
        memset(&wf, 0, sizeof(wf));
        wf.in2 = in_coef ? (const bf16*)y_in : nullptr; wf.in_coef = in_coef;
        wf.ep_y = (const bf16*)ep_y; wf.ep_stat = ep_stat; wf.ep_gamma = ep_gamma; wf.ep_beta = ep_beta; wf.ep_act = ep_act;
        return dwconv_multi_launch(dz_in, wt, out, B, H, W, C, 1, ep_y ? ep_part : nullptr, stream, wf);
    }
    if (dw_walk_ok(C, 1)) {
        DwWalkFuse wf;
        memset(&wf, 0, sizeof(wf));
        wf.in2 = in_coef ? (const bf16*)y_in : nullptr; wf.in_coef = in_coef;
        wf.ep_y = (const bf16*)ep_y; wf.ep_stat = ep_stat; wf.ep_gamma = ep_gamma; wf.ep_beta = ep_beta; wf.ep_act = ep_act;
        return dwconv_walk_launch(dz_in, wt, out, B, H, W, C, 1, ep_y ? ep_part : nullptr, stream, &wf);
    }
    DwFuse f;
    f.in2 = in_coef ? y_in : nullptr; f.in_coef = in_coef;
    f.ep_y = ep_y; f.ep_stat = ep_stat; f.ep_gamma = ep_gamma; f.ep_beta = ep_beta; f.ep_act = ep_act;
    return dwconv_tiled_launch(dz_in, wt, out, B, H, W, C, 1, 1, nullptr, nullptr, nullptr, 0, ep_y ? ep_part : nullptr, stream, f);
}
extern "C" int gg_dwconv3x3_bwd_data(const void* dy, const float* wt, void* dx, int B, int H, int W, int C, int stride, void* stream) {
    GG_CHECK(dy && wt && dx && B > 0 && (C & 7) == 0 && (stride == 1 || stride == 2), "gg_dwconv3x3_bwd_data: bad args");
    GG_CHECK((int64_t)B * H * W * (C / 8) < ((int64_t)1 << 32), "gg_dwconv3x3_bwd_data: tensor too large for 32-bit indexing");
    if (stride == 1 && dw_multi_bwd(C)) { DwWalkFuse wf; memset(&wf, 0, sizeof(wf)); return dwconv_multi_launch(dy, wt, dx, B, H, W, C, 1, nullptr, stream, wf); }
    if (dw_walk_ok(C, stride)) return dwconv_walk_launch(dy, wt, dx, B, H, W, C, 1, nullptr, stream);
    if (stride == 1)     // data gradient of a stride-1 depthwise conv == the same conv with flipped taps
        return dwconv_tiled_launch(dy, wt, dx, B, H, W, C, 1, 1, nullptr, nullptr, nullptr, 0, nullptr, stream);
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    GG_PROF(GG_CAT_DWCONV, 18.0 * B * Ho * Wo * C, 2.0 * B * C * ((double)H * W + (double)Ho * Wo), stream);
    if (stride == 2 && dw_s2_ok(C) && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)wt & 15) == 0)
        return dwconv_s2_bwd_launch(dy, wt, dx, B, H, W, C, nullptr, stream);
    hipLaunchKernelGGL(dwconv3x3_bwd_data_kernel, dim3(grid_for((int64_t)B * H * W * (C / 8), 65536)), dim3(256), 0,
                       (hipStream_t)stream, (const bf16*)dy, wt, (bf16*)dx, B, H, W, C, Ho, Wo, stride);
    GG_LAUNCH_CHECK();
    return 0;
}
// walking weight-gradient geometry: channels per thread, columns per block (LDS holds [PX][9][C] floats), image groups
struct WgGeom { int nc, CG, PX, nbx, BG; };
static WgGeom wg_geom(int B, int W, int C) {
    WgGeom g;
    g.nc = (C / 4) <= 256 ? 4 : 8;
    g.CG = C / g.nc;
    g.PX = std::max(1, std::min(256 / g.CG, (int)(15360 / (9 * (int64_t)C))));
    g.nbx = (int)gg_cdiv(W, g.PX);
    g.BG = std::max(1, std::min(B, 2048 / g.nbx));
    return g;
}
extern "C" int64_t gg_dwconv_wgrad_scratch_floats(int B, int H, int W, int C, int stride) {
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const WgGeom w = wg_geom(B, W, C);
    const int64_t nb = std::max<int64_t>(dw_geom((int64_t)B * Ho * Wo, C, 9 * C).nblocks, (int64_t)w.nbx * w.BG);
    return (nb + GG_REDUCE_SLICES) * 9 * C;
}
extern "C" int gg_dwconv3x3_bwd_weight(const void* x, const void* dy, int B, int H, int W, int C, int stride, float* scratch,
                                       float* grad, int accumulate, void* stream) {
    GG_CHECK(x && dy && scratch && grad && B > 0 && (C & 7) == 0, "gg_dwconv3x3_bwd_weight: bad args");
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    GG_PROF(GG_CAT_DWCONV, 18.0 * B * Ho * Wo * C, 2.0 * B * C * ((double)H * W + (double)Ho * Wo), stream);
    if (dw_walk_ok(C, stride) && (int64_t)H * W * C * 2 < 0x40000000LL && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0) {
        const WgGeom w = wg_geom(B, W, C);
        const size_t ldsw = (size_t)w.PX * 9 * C * sizeof(float);
        const dim3 grid(w.nbx, w.BG), block(w.CG * w.PX);
        if (w.nc == 4) hipLaunchKernelGGL((dwconv3x3_wgrad_walk_kernel<2>), grid, block, ldsw, (hipStream_t)stream, (const bf16*)x, (const bf16*)dy, B, H, W, C, w.CG, w.PX, w.nbx, scratch);
        else hipLaunchKernelGGL((dwconv3x3_wgrad_walk_kernel<4>), grid, block, ldsw, (hipStream_t)stream, (const bf16*)x, (const bf16*)dy, B, H, W, C, w.CG, w.PX, w.nbx, scratch);
        const float* rows; int nrows;
        gg_reduce_rows(scratch, w.nbx * w.BG, 9 * C, (hipStream_t)stream, &rows, &nrows);
        hipLaunchKernelGGL(dwconv_wgrad_final_kernel, dim3((unsigned)gg_cdiv(9 * C, 256)), dim3(256), 0, (hipStream_t)stream, rows,
                           nrows, C, grad, accumulate);
        GG_LAUNCH_CHECK();
        return 0;
    }
    DwGeom g = dw_geom((int64_t)B * Ho * Wo, C, 9 * C);
    size_t lds = (size_t)g.PP * 9 * C * sizeof(float);
    GG_CHECK(lds <= 64 * 1024, "gg_dwconv3x3_bwd_weight: LDS budget exceeded for C=%d", C);
    hipLaunchKernelGGL(dwconv3x3_bwd_weight_kernel, dim3(g.nblocks), dim3(g.threads), lds, (hipStream_t)stream, (const bf16*)x,
                       (const bf16*)dy, B, H, W, C, Ho, Wo, stride, g.CG, g.PP, g.pix_per_block, scratch);
    const float* rows; int nrows;
    gg_reduce_rows(scratch, g.nblocks, 9 * C, (hipStream_t)stream, &rows, &nrows);
    hipLaunchKernelGGL(dwconv_wgrad_final_kernel, dim3((unsigned)gg_cdiv(9 * C, 256)), dim3(256), 0, (hipStream_t)stream, rows,
                       nrows, C, grad, accumulate);
    GG_LAUNCH_CHECK();
    return 0;
}
