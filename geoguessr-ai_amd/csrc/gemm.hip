// bf16 MFMA GEMM  C[M,N] = A[M,K] * B[N,K]^T  with fused epilogues (gfx950).
//
// One kernel serves every dense contraction on the hot path: 1x1 convs and im2col'd 3x3 convs
// of PatchEmbed / MBConv / PatchMerging, the qkv / proj / fc1 / fc2 Linears of TinyViT and CLIP,
// the geocell head, and (with pre-transposed operands + split-K) every dgrad / wgrad.
//
// Tile 128(M) x 128(N) x 32(K), 256 threads = 4 waves in 2x2, each wave 64x64 = 4x4 tiles of
// v_mfma_f32_16x16x32_bf16.  The MFMA is issued as D = Wfrag x Xfrag, i.e. D[i=n][j=m], so one
// lane ends up with 4 consecutive n of a single row m -> 8-byte ge4_t / 16-byte f32x4 stores
// along the contiguous dimension of C and vectorised bias / residual / pre-activation access.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include "../../include/gg.h"

#define BK 64

// Element type of this build.  The file is compiled twice: as is (bf16: the bf16 mode's GEMM, gg_gemm_nt) and once more from gemm_f16.hip with
// GG_GEMM_ELEM_F16 inside namespace gg_f16 (fp16: the CLIP tower's fp16 inference mode, gg_gemm_nt_f16).  Tiling, LDS swizzle, buffer-load staging and the
// epilogue classes only see 16-bit elements in 16-byte chunks; what differs is named here: the element / vector types and the MFMA instruction.
#ifdef GG_GEMM_ELEM_F16
typedef f16 ge_t;
typedef f16x8 ge8_t;
typedef f16x4 ge4_t;
typedef f16 ge2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 ge_mfma(ge8_t a, ge8_t b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
#else
typedef bf16 ge_t;
typedef bf16x8 ge8_t;
typedef bf16x4 ge4_t;
typedef bf16x2 ge2_t;
__device__ __forceinline__ f32x4 ge_mfma(ge8_t a, ge8_t b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
#endif

struct GemmParams {
    const ge_t* A; int64_t lda;
    const ge_t* B; int64_t ldb;
    void* C; int64_t ldc;
    int M, N, K;
    const float* bias;
    int act;
    ge_t* preact;                 // optional copy of (acc+bias) before the activation
    const float* rowscale; int rows_per_scale;
    const ge_t* residual; int64_t ldr;
    const ge_t* dact_preact; int dact;   // out = v * act'(dact_preact[m][n]) (backward through fc1's activation)
    float* colstats;              // [tilesM][2][N]: per M-tile column sum / sum of squares of (acc)
    int out_f32;
    int split_k, k_per_split;     // split_k > 1: C is f32 [split][M][N] partials
    int tilesM, tilesN;
    int debug;                    // timing experiments only: 1 = no operand loads, 2 = no result stores
    const ge_t* A2; int k_split;  // optional second A source for contraction columns k >= k_split (same lda)
    const ge_t* bn_y; const float* bn_stat; const float* bn_gamma; const float* bn_beta; int bn_act;   // EPI_BNBWD
    const float* a_stat; const float* a_gamma; const float* a_beta; int a_act;   // PRO: A := act(BN(A)) while staging
    int group_m;                  // LDS-DMA form: tile order walks groups of group_m M-tiles column by column (0: row-major)
};

// LDS image of an R x 64 operand tile: unpadded 128-byte rows, the eight 16-byte chunks of a row XOR-swizzled with
//   T(row) = 2*bit1(row) + 4*bit3(row)
// which makes every 16-lane group of the fragment ds_read_b128 (lanes = 16 rows x 4 k-chunks) hit 16 distinct
// 16-byte bank slots, and keeps the staging ds_write_b128 (8 lanes = one whole row) conflict-free.
__device__ __forceinline__ int lds_chunk_off(int row, int kc) {
    return row * BK + ((kc ^ ((row & 2) | ((row >> 1) & 4))) << 3);
}

// Tile BM x BN x 64, 256 threads = WM x WN waves, each wave (BM/WM) x (BN/WN) as TM x TN tiles of
// v_mfma_f32_16x16x32_bf16.  Two instantiations: 128x128 (2x2 waves, 64 accumulator registers: the MFMA-bound shapes)
// and 128x64 (4x1 waves, 32 accumulator registers -> 5 workgroups per CU: the HBM-bound small-K / small-N shapes, where
// bytes in flight per CU, not MFMA rate, set the speed).
// Epilogue classes are compile-time: with every activation / gradient variant inlined behind runtime switches the
// kernel was 19 000 instructions (150 KB) and ran out of the instruction cache even on the plain path.
enum { EPI_PLAIN = 0,    // raw accumulators (+ optional BatchNorm column statistics): convs, plain dgrads
       EPI_LINEAR = 1,   // + bias?  * rowscale?  + residual?                      : qkv / proj / fc2, their dgrads
       EPI_GELU = 2,     // + bias?, optional pre-activation copy, exact GELU        : fc1
       EPI_QGELU = 3,    // + bias?, quick_gelu                                      : CLIP fc1
       EPI_DGELU = 4,    // * GELU'(saved pre-activation) * rowscale?                : fc2 dgrad
       EPI_F32 = 5,      // f32 output (+ bias?) or split-K partial slabs            : head logits, wgrads
       EPI_BNBWD = 6 };  // * act'(BN(y)) of the ConvNorm the gradient flows into, + its backward column sums: conv dgrads

// The second tensor of the row phase (residual / saved pre-activation / saved BatchNorm input) for this thread's row-phase
// slots: range-checked 16-byte buffer loads, all passes in flight at once; rows beyond M and chunks beyond N read as zeros.
template <int BM, int BN, int EPI>
__device__ __forceinline__ void gemm_ext_load(const GemmParams& p, int m0, int n0, ge8_t (&ex)[BM / (256 / (BN / 8))]) {
    constexpr int CPR = BN / 8, RPP = 256 / CPR, NP = BM / RPP;
    const int chunk = threadIdx.x % CPR, rr = threadIdx.x / CPR;
    const int n = n0 + chunk * 8;
    const ge_t* ext = EPI == EPI_DGELU ? p.dact_preact : (EPI == EPI_LINEAR ? p.residual : (EPI == EPI_BNBWD ? p.bn_y : nullptr));
    const int64_t lde = EPI == EPI_LINEAR ? p.ldr : p.ldc;
    if ((EPI == EPI_DGELU || EPI == EPI_LINEAR || EPI == EPI_BNBWD) && ext) {
        if (((lde & 7) == 0) && ((p.N & 7) == 0) && (((uintptr_t)ext & 15) == 0)) {
            // range-checked 16-byte buffer loads, all passes in flight before the first use; rows beyond M and column
            // chunks beyond N come back as zeros
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const unsigned bytesE = (unsigned)min(p.M - m0, BM) * (unsigned)lde * 2u;
            const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc((void*)(ext + (int64_t)m0 * lde), 0, (int)bytesE, 0x00020000);
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                const unsigned vo = ((unsigned)(pass * RPP + rr) * (unsigned)lde + (unsigned)n) * 2u;
                const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(rsE, (int)(n < p.N ? vo : 0xFFFFFFF0u), 0, 0);
                ex[pass] = __builtin_bit_cast(ge8_t, raw);
            }
        } else {
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                const int m = m0 + pass * RPP + rr;
                ge8_t d = {0, 0, 0, 0, 0, 0, 0, 0};
                if (m < p.M) { for (int j = 0; j < 8; ++j) if (n + j < p.N) d[j] = ext[(int64_t)m * lde + n + j]; }
                ex[pass] = d;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int EPI, bool PRELOADED>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, ge_t* smem, f32x4 (&acc)[BN / WN / 16][BM / WM / 16], int m0, int n0,
                                              int tm, int z, int wm, int wn, int lr, int lg, ge8_t (&ex)[BM / (256 / (BN / 8))], const float* btab) {
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int WROWS = BM / WM, WCOLS = BN / WN;
    // ---------------- lane holds C[m = .. + mt*16 + lr][n = .. + nt*16 + lg*4 + r] ----------------
    // Fragment phase: bias / per-row scale on the fp32 accumulators.  Everything it needs from memory is fetched up
    // front as one batch of independent loads: issued inside the (mt, nt) loop they serialise 16 L2 round trips per tile,
    // which cost more than the whole k-loop of a K = 384 GEMM.
    constexpr bool has_bias = EPI == EPI_LINEAR || EPI == EPI_GELU || EPI == EPI_QGELU || EPI == EPI_F32;
    float bs[TN][4];
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const bool bias_vec = has_bias && p.bias && (p.N & 3) == 0 && ((uintptr_t)p.bias & 15) == 0;
    const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, bias_vec ? p.N * 4 : 0, 0x00020000);
    if (has_bias) {
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
            const int n = n0 + wn * WCOLS + nt * 16 + lg * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) bs[nt][r] = 0.f;
            if (!p.bias) continue;
            if (bias_vec) {          // one range-checked 16-byte load per fragment column group (4 loads, not 16, per tile and thread)
                const f32x4 b4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsBias, n * 4, 0, 0));
#pragma unroll
                for (int r = 0; r < 4; ++r) bs[nt][r] = b4[r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) bs[nt][r] = p.bias[min(n + r, p.N - 1)];
            }
        }
    }
    if (EPI == EPI_F32) {
        // f32 results (head logits, split-K slabs) go out straight from the fragments: 16-byte stores
        const bool vec_ok = (p.ldc & 3) == 0;
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const int m = m0 + wm * WROWS + mt * 16 + lr;
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
                const int n = n0 + wn * WCOLS + nt * 16 + lg * 4;
                f32x4 v = acc[nt][mt];
                if (m >= p.M || n >= p.N) continue;
                float* Cf = reinterpret_cast<float*>(p.C) + ((int64_t)(p.split_k > 1 ? z : 0) * p.M + m) * p.ldc + n;
                if (p.split_k <= 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += bs[nt][r];
                }
                if (vec_ok && n + 3 < p.N) *reinterpret_cast<f32x4*>(Cf) = v;
                else { for (int r = 0; r < 4; ++r) if (n + r < p.N) Cf[r] = v[r]; }
            }
        }
        return;
    }
    // bf16 results are staged through LDS (the k-loop buffers are dead) so that global traffic is 16 bytes per lane
    // along full tile rows instead of 8-byte fragments of 16 different rows.  The element-wise tail that needs a second
    // tensor (GELU + pre-activation copy, GELU' * saved pre-activation, residual add) runs on the staged, bf16-rounded
    // tile in that row layout -- the same rounding points as the reference's bf16 autocast graph, where the Linear's
    // output is a bf16 tensor before the activation / residual op reads it.
    constexpr int CS = BN + 8;                      // staged tile row stride (elements), 16-B aligned
    constexpr int CPR = BN / 8;                     // 16-byte chunks per staged row
    constexpr int RPP = 256 / CPR;                  // rows per cooperative pass
    constexpr int NP = BM / RPP;
    ge_t* Cs = smem;
    float rsv[TM];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
        const int m = min(m0 + wm * WROWS + mt * 16 + lr, p.M - 1);
        rsv[mt] = ((EPI == EPI_LINEAR || EPI == EPI_DGELU) && p.rowscale) ? p.rowscale[m / p.rows_per_scale] : 1.f;
    }
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
            f32x4 v = acc[nt][mt];
            if (has_bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += bs[nt][r];
            }
            if (EPI == EPI_QGELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gg_quick_gelu(v[r]);
            }
            if (EPI == EPI_LINEAR || EPI == EPI_DGELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= rsv[mt];
            }
            ge4_t o = {(ge_t)v[0], (ge_t)v[1], (ge_t)v[2], (ge_t)v[3]};
            *reinterpret_cast<ge4_t*>(Cs + (wm * WROWS + mt * 16 + lr) * CS + wn * WCOLS + nt * 16 + lg * 4) = o;
        }
    // ---------------- row phase: thread = (row rr + pass*RPP, 16-byte column chunk) ----------------
    const int chunk = threadIdx.x % CPR, rr = threadIdx.x / CPR;
    const int n = n0 + chunk * 8;
    const bool wide = ((p.ldc & 7) == 0) && (n + 7 < p.N);
    f32x2 bsc[4], bsh[4], brs[4], bnm[4];      // EPI_BNBWD: z = y*bsc + bsh, xhat = y*brs + bnm for this thread's 8 columns
    if (EPI == EPI_BNBWD) {
        // btab = [bsc | bsh | brs | bnm][BN], one column per thread, written by the kernel before its k-loop (the k-loop's
        // barriers order it): 32 scalar parameter loads per thread and tile cost more than the operand loads of a K = 96 dgrad
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(btab + 0 * BN + chunk * 8 + 4 * q), b = *reinterpret_cast<const f32x4*>(btab + 1 * BN + chunk * 8 + 4 * q);
            const f32x4 c = *reinterpret_cast<const f32x4*>(btab + 2 * BN + chunk * 8 + 4 * q), d = *reinterpret_cast<const f32x4*>(btab + 3 * BN + chunk * 8 + 4 * q);
            bsc[2 * q] = (f32x2){a[0], a[1]}; bsc[2 * q + 1] = (f32x2){a[2], a[3]};
            bsh[2 * q] = (f32x2){b[0], b[1]}; bsh[2 * q + 1] = (f32x2){b[2], b[3]};
            brs[2 * q] = (f32x2){c[0], c[1]}; brs[2 * q + 1] = (f32x2){c[2], c[3]};
            bnm[2 * q] = (f32x2){d[0], d[1]}; bnm[2 * q + 1] = (f32x2){d[2], d[3]};
        }
    }
    if (!PRELOADED) gemm_ext_load<BM, BN, EPI>(p, m0, n0, ex);
    __syncthreads();
    const bool stats = (EPI == EPI_PLAIN || EPI == EPI_BNBWD) && p.colstats != nullptr;
    f32x2 cs2[4], cq2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cs2[j] = cq2[j] = (f32x2)(0.f);
    const bool bn_gelu = p.bn_act == GG_ACT_GELU;
    auto store8 = [&](ge_t* base, int m, const ge8_t& v, bool stream_out = false) {
        ge_t* g = base + (int64_t)m * p.ldc + n;
        if (wide) {
            if (stream_out) __builtin_nontemporal_store(v, reinterpret_cast<ge8_t*>(g));    // read back only in the backward pass
            else *reinterpret_cast<ge8_t*>(g) = v;
        } else { for (int j = 0; j < 8; ++j) if (n + j < p.N) g[j] = v[j]; }
    };
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        const int row = pass * RPP + rr, m = m0 + row;
        if (m >= p.M || n >= p.N) continue;
        ge8_t v = *reinterpret_cast<const ge8_t*>(Cs + row * CS + chunk * 8);
        if (EPI == EPI_BNBWD && !(p.debug & 8)) {
            // dz = da * act'(gamma*xhat + beta); column sums of dz and dz*xhat (of the stored, bf16-rounded dz)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 y = {(float)ex[pass][2 * j], (float)ex[pass][2 * j + 1]};
                const f32x2 d = (f32x2){(float)v[2 * j], (float)v[2 * j + 1]} * gg_act_grad_v2(y * bsc[j] + bsh[j], bn_gelu);
                const ge_t d0 = (ge_t)d.x, d1 = (ge_t)d.y;
                v[2 * j] = d0; v[2 * j + 1] = d1;
                const f32x2 dr = {(float)d0, (float)d1};
                cs2[j] += dr; cq2[j] += dr * (y * brs[j] + bnm[j]);
            }
        } else if (stats) {
            // BatchNorm partial statistics of the stored (bf16-rounded) conv output, taken on the way out
#pragma unroll
            for (int j = 0; j < 4; ++j) { const f32x2 f = {(float)v[2 * j], (float)v[2 * j + 1]}; cs2[j] += f; cq2[j] += f * f; }
        }
        if (p.debug & 2) continue;
        if (EPI == EPI_GELU) {
            if (p.preact) store8(p.preact, m, v, true);
            if (p.act == GG_ACT_QUICK_GELU) {       // CLIP fc1 in training (uniform branch: the pre-activation copy lives in this epilogue class)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (ge_t)gg_quick_gelu((float)v[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const f32x2 r = gg_gelu_v2((f32x2){(float)v[j], (float)v[j + 1]});
                    v[j] = (ge_t)r.x; v[j + 1] = (ge_t)r.y;
                }
            }
        }
        if (EPI == EPI_DGELU) {
            if (p.dact == GG_ACT_QUICK_GELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (ge_t)((float)v[j] * gg_quick_gelu_grad((float)ex[pass][j]));
            } else {
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const f32x2 r = (f32x2){(float)v[j], (float)v[j + 1]} * gg_gelu_grad_v2((f32x2){(float)ex[pass][j], (float)ex[pass][j + 1]});
                    v[j] = (ge_t)r.x; v[j + 1] = (ge_t)r.y;
                }
            }
        }
        if (EPI == EPI_LINEAR && p.residual) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (ge_t)((float)v[j] + (float)ex[pass][j]);
        }
        store8(reinterpret_cast<ge_t*>(p.C), m, v);
    }
    if (stats && !(p.debug & 16)) {
        __syncthreads();
        float cs[8], cq[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { cs[j] = cs2[j >> 1][j & 1]; cq[j] = cq2[j >> 1][j & 1]; }
        // threads sharing a column chunk: lanes l, l+CPR, ... within a wave, then the 4 waves through LDS
        float* red = reinterpret_cast<float*>(smem);         // [4 waves][2][BN]
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int o = CPR; o < 64; o <<= 1) { cs[j] += __shfl_xor(cs[j], o, 64); cq[j] += __shfl_xor(cq[j], o, 64); }
        }
        const int wv = threadIdx.x >> 6;
        if ((threadIdx.x & 63) < CPR) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                red[(wv * 2 + 0) * BN + chunk * 8 + j] = cs[j];
                red[(wv * 2 + 1) * BN + chunk * 8 + j] = cq[j];
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * BN) {
            const int which = threadIdx.x / BN, col = threadIdx.x % BN;
            const float sum = red[(0 * 2 + which) * BN + col] + red[(1 * 2 + which) * BN + col] +
                              red[(2 * 2 + which) * BN + col] + red[(3 * 2 + which) * BN + col];
            if (n0 + col < p.N) p.colstats[(int64_t)tm * 2 * p.N + which * p.N + n0 + col] = sum;
        }
    }
}

// PRO: the A operand is the SAVED PRE-BatchNorm output of the previous ConvNorm and act(gamma*(a-mean)*rstd+beta) is applied
// to each 16-byte chunk between the register prefetch and the LDS store -- the separate BatchNorm-apply pass and the
// activation tensor it would write disappear (frozen ConvNorm chains: nothing downstream needs that tensor).
template <int BM, int BN, int WM, int WN, int MINW, int EPI, bool PRO = false>
__global__ __launch_bounds__(256, MINW) void gemm_nt_kernel(GemmParams p) {
    // one operand stage; the next k-tile travels through registers while this one is consumed.  The epilogue reuses
    // the buffer as a [BM][BN+8] bf16 staging tile.
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int LA = BM * 8 / 256, LB = BN * 8 / 256;     // 16-byte chunks per thread per k-tile
    constexpr int OPER = (BM + BN) * BK, STAGE = BM * (BN + 8);
    __shared__ __attribute__((aligned(16))) ge_t smem[OPER > STAGE ? OPER : STAGE];
    __shared__ __attribute__((aligned(16))) float ptab[PRO ? 2048 : 4];      // PRO: [scale[K] | shift[K]], K <= 1024
    ge_t* As = smem;
    ge_t* Bs = smem + BM * BK;
    if (PRO) {
        for (int k = threadIdx.x; k < p.K; k += 256) {
            const float sc = p.a_gamma[k] * p.a_stat[p.K + k];
            ptab[k] = sc;
            ptab[p.K + k] = p.a_beta[k] - p.a_stat[k] * sc;
        }
        __syncthreads();
    }

    const int tiles = p.tilesM * p.tilesN;
    const int bid = gg_xcd_remap(blockIdx.x, tiles);
    const int tm = bid / p.tilesN, tn = bid % p.tilesN;   // consecutive ids share the A row panel
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.y;
    const int kbeg = z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lg = lane >> 4;

    // ---- operand staging: chunk c = tid + 256*i -> tile row (tid>>3) + 32*i, 16-byte k-chunk tid&7 ----
    // Raw buffer loads: the descriptor starts at this tile's first row and ends after the matrix's last valid row, so
    // rows beyond M / N come back as zeros from the hardware range check (no branches, no clamps); the running k offset
    // is the scalar soffset; a k-chunk beyond K is pushed out of range through its voffset.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int srow = threadIdx.x >> 3, skc = threadIdx.x & 7;
    // descriptor = the valid rows of THIS tile only, so offsets stay 32-bit for any matrix size (host checks 128*ld*2 < 2^32)
    const unsigned bytesA = (unsigned)min(p.M - m0, BM) * (unsigned)p.lda * 2u;
    const unsigned bytesB = (unsigned)min(p.N - n0, BN) * (unsigned)p.ldb * 2u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)m0 * p.lda), 0, (int)bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (int64_t)n0 * p.ldb), 0, (int)bytesB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)((p.A2 ? p.A2 : p.A) + (int64_t)m0 * p.lda), 0, (int)bytesA, 0x00020000);
    unsigned voa[LA], vob[LB];
    int lds_a[LA], lds_b[LB];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        voa[i] = (unsigned)(srow + 32 * i) * (unsigned)p.lda * 2u + skc * 16u;
        lds_a[i] = lds_chunk_off(srow + 32 * i, skc);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        vob[i] = (unsigned)(srow + 32 * i) * (unsigned)p.ldb * 2u + skc * 16u;
        lds_b[i] = lds_chunk_off(srow + 32 * i, skc);
    }
    // fragment read offsets (elements): row = w*rows + i*16 + lr; only bits 1,3 of lr enter the swizzle
    const int sw = (lr & 2) | ((lr >> 1) & 4);
    const int a_base = (wm * (BM / WM) + lr) * BK, b_base = (wn * (BN / WN) + lr) * BK;
    const int kc0 = ((0 + lg) ^ sw) << 3, kc1 = ((4 + lg) ^ sw) << 3;

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr bool EARLY = false;          // fetching the epilogue's second tensor before the k-loop: measured no gain, +24 VGPRs
    ge8_t ex[BM / (256 / (BN / 8))];
    __shared__ __attribute__((aligned(16))) float btab[EPI == EPI_BNBWD ? 4 * BN : 4];
    if (EPI == EPI_BNBWD && threadIdx.x < BN) {
        const int c = min(n0 + (int)threadIdx.x, p.N - 1);
        const float mu = p.bn_stat[c], rstd = p.bn_stat[p.N + c], ga = p.bn_gamma[c], be = p.bn_beta[c];
        btab[0 * BN + threadIdx.x] = ga * rstd; btab[1 * BN + threadIdx.x] = be - mu * ga * rstd;
        btab[2 * BN + threadIdx.x] = rstd; btab[3 * BN + threadIdx.x] = -mu * rstd;
    }
    if (EARLY && !(p.debug & 4)) gemm_ext_load<BM, BN, EPI>(p, m0, n0, ex);
    if (EARLY && (p.debug & 4)) { for (auto& e : ex) e = (ge8_t){1, 1, 1, 1, 1, 1, 1, 1}; }
    u32x4 ra[LA], rb[LB];
    const int nk = (kend - kbeg + BK - 1) / BK;
    auto load_tile = [&](int kt) {
        const int k0 = kbeg + kt * BK;
        const bool kin = (k0 + skc * 8) < kend;                       // K % 8 == 0: a chunk is entirely in or out
        const int so = k0 * 2;
        if (p.A2 && k0 >= p.k_split) {       // uniform: k_split is a multiple of the k-tile
#pragma unroll
            for (int i = 0; i < LA; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA2, (int)(kin ? voa[i] : 0xFFFFFFF0u), so - p.k_split * 2, 0);
        } else {
#pragma unroll
            for (int i = 0; i < LA; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(kin ? voa[i] : 0xFFFFFFF0u), so, 0);
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(kin ? vob[i] : 0xFFFFFFF0u), so, 0);
    };
    if (nk > 0) load_tile(0);
    const bool pro_gelu = p.a_act == GG_ACT_GELU;
    for (int kt = 0; kt < nk; ++kt) {
        if (PRO) {
            // this thread's chunk covers channels k0 + 8*skc .. +7 of every staged row
            const int kc = min(kbeg + kt * BK + skc * 8, p.K - 8);
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(ptab + kc), s1 = *reinterpret_cast<const f32x4*>(ptab + kc + 4);
            const f32x4 h0 = *reinterpret_cast<const f32x4*>(ptab + p.K + kc), h1 = *reinterpret_cast<const f32x4*>(ptab + p.K + kc + 4);
            const f32x2 sc[4] = {{s0[0], s0[1]}, {s0[2], s0[3]}, {s1[0], s1[1]}, {s1[2], s1[3]}};
            const f32x2 sh[4] = {{h0[0], h0[1]}, {h0[2], h0[3]}, {h1[0], h1[1]}, {h1[2], h1[3]}};
#pragma unroll
            for (int i = 0; i < LA; ++i) {
                const bool rok = m0 + srow + 32 * i < p.M;       // rows beyond M must stay zero (they feed the column statistics)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 v = {__uint_as_float(ra[i][q] << 16), __uint_as_float(ra[i][q] & 0xffff0000u)};
                    const f32x2 r = gg_act_v2(v * sc[q] + sh[q], pro_gelu);
                    const ge_t lo = (ge_t)r.x, hi = (ge_t)r.y;
                    const unsigned pk = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
                    ra[i][q] = rok ? pk : 0u;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < LA; ++i) *reinterpret_cast<u32x4*>(As + lds_a[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < LB; ++i) *reinterpret_cast<u32x4*>(Bs + lds_b[i]) = rb[i];
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1);
        const int ksteps = (kend - (kbeg + kt * BK)) > 32 ? 2 : 1;     // skip the all-zero upper half of a K tail
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks < ksteps) {
                const int kc = ks ? kc1 : kc0;
                ge8_t xf[TM], wf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const ge8_t*>(As + a_base + i * 16 * BK + kc);
#pragma unroll
                for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const ge8_t*>(Bs + b_base + i * 16 * BK + kc);
#pragma unroll
                for (int nt = 0; nt < TN; ++nt)
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt)
                        acc[nt][mt] = ge_mfma(wf[nt], xf[mt], acc[nt][mt]);
            }
        }
        __syncthreads();
    }
    gemm_epilogue<BM, BN, WM, WN, EPI, EARLY>(p, smem, acc, m0, n0, tm, z, wm, wn, lr, lg, ex, btab);
}

// ------------------------------------------------------------------------------------------- NT, LDS-DMA form (the MFMA-bound shapes)
// The kernel above is LDS-port bound on the transformer shapes (K >= 192): per 128 x 128 x 64 k-tile it writes 32 KB with ds_write_b128 (~79 B/clk/CU:
// 415 cycles) and reads 64 KB of fragments (256 cycles) under 512 cycles of matrix time.  Operands delivered by LDS-DMA (`buffer_load ... lds`, 16 B per lane:
// no staging registers, no ds_write; rows beyond M / N and chunks beyond K arrive as zeros from the buffer range check) remove the write side, and then the
// operand stream itself is the limit (tools/dma_bw.hip: the same tile walk without any matrix work).  What that stream delivers depends on the PIECE a DMA
// instruction fetches: 16 rows x 64 B (a 32-deep k-stage) tops out at 12-15 TB/s chip-wide whatever the tile and ring, 8 rows x 128 B (whole cache lines, a
// 64-deep stage) reaches 19-27 TB/s given enough bytes in flight -- half-line pieces cost the texture-address path a full line's cycles.  So: 64-deep stages
// with 128-byte LDS rows, and the largest tile whose two-stage ring still lets TWO workgroups share a CU (2 x 80 KB = the whole LDS): 192 x 128, four waves of
// 96 x 64.  Two independent workgroups per CU (rather than one of eight waves on a 256 x 256 tile) because a K = 384 ... 768 tile is short: its prologue
// (first operands: ~1 us) and its epilogue would leave the matrix pipe idle 25-50 % of the time with nobody else on the CU.
// LDS image: rows of 128 bytes, the eight 16-byte chunks of a row XOR-swizzled with T(row) = 2 bit1(row) + 4 bit3(row) on the SOURCE side of the DMA (the
// LDS side of a DMA is lane-linear), which makes every 16-lane group of the fragment ds_read_b128 hit 16 distinct 16-byte bank slots (the register-staged
// kernel's layout).
// Schedule of one stage (two 32-deep k-steps, ONE raw s_barrier, one stage of DMA in flight): B fragments of a k-step (4) and the first two A fragments are in
// registers when the step starts; A fragments stream through a 4-slot register ring two m-tiles ahead of their MFMAs; the second k-step's B fragments are read
// during the first.  After the fourth m-tile of the second k-step every fragment read of the stage has been issued: wait for them and for this wave's DMAs of
// the next stage, barrier, refill this stage's buffer with stage s + 2, read the next stage's first fragments under the last eight MFMAs.
// Epilogue of the LDS-DMA form: the same arithmetic and rounding points as gemm_epilogue (fragment phase: + bias, quick_gelu, * rowscale on the f32 accumulators,
// round to the 16-bit type; row phase on the rounded tile: pre-activation copy, GELU, x GELU'(saved pre-activation), + residual) -- results are bit-identical to
// the register-staged kernel's -- but WAVE-PRIVATE and without general-shape fallbacks.  A wave turns its own (16 TM) x 64 sub-tile from the MFMA layout (a lane: 4
// columns of one row) into rows (8 lanes x 16 B = one 128-byte line of C; an instruction covers 8 rows) through a private LDS scratch, 32 rows at a time: no
// workgroup barrier, and while it runs the workgroup's ring is free to receive the NEXT tile's first stage.  The host sends a launch here only when every row of
// C / the second tensor is 16-byte aligned and N % 8 == 0, so a 16-byte column chunk is entirely inside or outside N; rows beyond M and chunks beyond N are
// dropped by the buffer range check (loads return zeros, stores do nothing): no branches, no scalar tails (the general epilogue compiled to ~200 exec-masked
// branches per tile).  Its memory operands (bias, row scales, second tensor) are fetched by epi_fetch BEFORE the last k-stage's MFMAs.
typedef unsigned int gemm_u32x4_t __attribute__((ext_vector_type(4)));
// (the operand registers are plain local arrays of the kernel, passed by reference: gathered in a struct the compiler kept part of them in scratch memory)
// bias and row scales (MFMA layout): ordinary loads issued before the last k-stage's MFMAs, pinned in registers by gemm_dma_epi_ready before the next tile's DMA
// is issued (a wait the compiler places behind an LDS-DMA covers the DMA too: vmcnt retires in order)
template <int EPI, int TM>
__device__ __forceinline__ void gemm_dma_epi_fetch(const GemmParams& p, f32x4 (&bs)[4], float (&rsv)[TM], int m0, int n0, int wm, int wn, int lane) {
    asm volatile("" : "+v"(lane));        // (opaque: what is derived from it below is recomputed per tile, not hoisted out of the tile loop and kept in registers / spilled)
    const int lr = lane & 15, lg = lane >> 4;
    const int mw = m0 + wm * (TM * 16), nw = n0 + wn * 64;         // the wave's sub-tile
    if (EPI == EPI_LINEAR || EPI == EPI_GELU || EPI == EPI_QGELU) {
        const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.bias ? p.N * 4 : 0, 0x00020000);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bs[nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsBias, (nw + nt * 16 + lg * 4) * 4, 0, 0));     // (no bias: zeros)
    }
    if ((EPI == EPI_LINEAR || EPI == EPI_DGELU) && p.rowscale) {
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) rsv[mt] = p.rowscale[min(mw + mt * 16 + lr, p.M - 1) / p.rows_per_scale];
    }
}
template <int EPI, int TM>
__device__ __forceinline__ void gemm_dma_epi_ready(const GemmParams& p, f32x4 (&bs)[4], float (&rsv)[TM]) {
    if (EPI == EPI_LINEAR || EPI == EPI_GELU || EPI == EPI_QGELU) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) asm volatile("" : "+v"(bs[nt]));
    }
    if ((EPI == EPI_LINEAR || EPI == EPI_DGELU) && p.rowscale) {
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) asm volatile("" : "+v"(rsv[mt]));
    }
}
// The second tensor of the row phase (residual / saved pre-activation), 2 TM x 16 B per lane: 48 - 64 registers that do not fit beside the k-loop's fragments, so it is
// requested after the k-loop, right BEHIND the next tile's first DMA stage: vmcnt retires in order, so the row phase's wait for it also covers that stage -- the
// two round trips overlap, and one of them (not two) is exposed per tile.  (Requested ahead of the DMA and counted by hand in inline asm, the compiler moved the
// destination registers between the load and the wait.)
template <int EPI, bool EXT, int TM>
__device__ __forceinline__ void gemm_dma_ext_fetch(const GemmParams& p, gemm_u32x4_t (&ex)[2 * TM], int m0, int n0, int wm, int wn, int lane) {
    const ge_t* ext = EPI == EPI_DGELU ? p.dact_preact : (EPI == EPI_LINEAR ? p.residual : nullptr);
    if (!EXT) return;
    asm volatile("" : "+v"(lane));
    const int64_t lde = EPI == EPI_LINEAR ? p.ldr : p.ldc;
    const int mw = m0 + wm * (TM * 16), nw = n0 + wn * 64;
    const unsigned rows = (unsigned)__builtin_amdgcn_readfirstlane(max(min(p.M - mw, TM * 16), 0));      // (readfirstlane: the compiler forms the clamp on the vector ALU and would wrap every buffer access using the descriptor in a waterfall loop)
    const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc((void*)(ext + (int64_t)mw * lde), 0, (int)(rows * (unsigned)lde * 2u), 0x00020000);
    const int n = nw + (lane & 7) * 8;
    const unsigned vo0 = ((unsigned)(lane >> 3) * (unsigned)lde + (unsigned)n) * 2u;
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i)
        ex[i] = __builtin_amdgcn_raw_buffer_load_b128(rsE, (int)(n < p.N ? vo0 + (unsigned)i * 8u * (unsigned)lde * 2u : 0xFFFFFFF0u), 0, 0);
}
// scratch: this wave's two 32-row x 144-byte LDS images
template <int EPI, bool EXT, int TM>
__device__ __forceinline__ void gemm_dma_epilogue(const GemmParams& p, ge_t* scratch, f32x4 (&acc)[4][TM], const f32x4 (&bs)[4], const float (&rsv)[TM], const gemm_u32x4_t (&ex)[2 * TM],
                                                  int m0, int n0, int wm, int wn, int lane) {
    constexpr int CS = 72;                                          // scratch row stride (elements): 144 bytes, 16-byte aligned
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    asm volatile("" : "+v"(lane));
    const int lr = lane & 15, lg = lane >> 4;
    const int mw = m0 + wm * (TM * 16), nw = n0 + wn * 64;
    const unsigned rows = (unsigned)__builtin_amdgcn_readfirstlane(max(min(p.M - mw, TM * 16), 0));      // (readfirstlane: the compiler forms the clamp on the vector ALU and would wrap every buffer access using the descriptor in a waterfall loop)
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<ge_t*>(p.C) + (int64_t)mw * p.ldc), 0, (int)(rows * (unsigned)p.ldc * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)((EPI == EPI_GELU && p.preact ? p.preact : reinterpret_cast<ge_t*>(p.C)) + (int64_t)mw * p.ldc), 0,
                                                                         (int)(rows * (unsigned)p.ldc * 2u), 0x00020000);
    const int n = nw + (lane & 7) * 8;
    const bool nin = n < p.N;
    const unsigned vo0 = ((unsigned)(lane >> 3) * (unsigned)p.ldc + (unsigned)n) * 2u;
    const unsigned vstep = 8u * (unsigned)p.ldc * 2u;
    const bool scaled = (EPI == EPI_LINEAR || EPI == EPI_DGELU) && p.rowscale != nullptr;
#pragma unroll
    for (int c = 0; c < TM / 2; ++c) {                              // 32 rows = m-tiles 2 c, 2 c + 1
        ge_t* Cs = scratch + (c & 1) * 32 * CS;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int mt = 2 * c + h;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                f32x4 v = acc[nt][mt];
                if (EPI == EPI_LINEAR || EPI == EPI_GELU || EPI == EPI_QGELU) v += bs[nt];
                if (EPI == EPI_QGELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gg_quick_gelu(v[r]);
                }
                if (scaled) v *= rsv[mt];
                const ge4_t q = {(ge_t)v[0], (ge_t)v[1], (ge_t)v[2], (ge_t)v[3]};
                *reinterpret_cast<ge4_t*>(Cs + (h * 16 + lr) * CS + nt * 16 + lg * 4) = q;
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {                               // 8 rows per instruction
            const int pass = 4 * c + g;
            ge8_t v = *reinterpret_cast<const ge8_t*>(Cs + (g * 8 + (lane >> 3)) * CS + (lane & 7) * 8);
            const unsigned vo = nin ? vo0 + (unsigned)pass * vstep : 0xFFFFFFF0u;
            const ge8_t e = __builtin_bit_cast(ge8_t, ex[EXT ? pass : 0]);
            if (EPI == EPI_GELU) {
                if (p.preact) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsP, (int)vo, 0, 2);      // (nt: read back only in the backward pass)
                if (p.act == GG_ACT_QUICK_GELU) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (ge_t)gg_quick_gelu((float)v[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const f32x2 r = gg_gelu_v2((f32x2){(float)v[j], (float)v[j + 1]});
                        v[j] = (ge_t)r.x; v[j + 1] = (ge_t)r.y;
                    }
                }
            }
            if (EPI == EPI_DGELU) {
                if (p.dact == GG_ACT_QUICK_GELU) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (ge_t)((float)v[j] * gg_quick_gelu_grad((float)e[j]));
                } else {
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const f32x2 r = (f32x2){(float)v[j], (float)v[j + 1]} * gg_gelu_grad_v2((f32x2){(float)e[j], (float)e[j + 1]});
                        v[j] = (ge_t)r.x; v[j + 1] = (ge_t)r.y;
                    }
                }
            }
            if (EPI == EPI_LINEAR && EXT) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (ge_t)((float)v[j] + (float)e[j]);
            }
            // result stores non-temporal: the tile's rows are not read again by this launch, and kept out of the L2's way they do not evict the weight panel and the A rows
            // the concurrent tiles share (c4 fp16 12.1 -> 11.9 ms, bf16 11.8 -> 11.55: same-box A/B, profiles/r06_c4_nt_stores.txt; dev: GG_GEMM_DEBUG=32 restores the default policy)
            if (p.debug & 32) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsC, (int)vo, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsC, (int)vo, 0, 2);
        }
    }
}

template <int N> __device__ __forceinline__ void gemm_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One workgroup per tile.  (A persistent form -- two workgroups per CU walking a per-XCD tile queue, the next tile's first stage issued under the epilogue -- was
// built and measured: the same 12.05 ms for the CLIP tower at batch 1024 as this one.  The operand stream and the chip's clock under the load set the pace,
// not the 1 us between a workgroup's end and its successor's start.)
// Two geometries of the same code (TM m-tiles of 16 rows per wave, NWN waves along N; two wave rows):
//   TM = 6, NWN = 2: 192 x 128 tile, four waves, 2 x 80 KB of LDS = two workgroups per CU (above);
//   TM = 8, NWN = 4: 256 x 256 tile, eight waves of 128 x 64, 128 KB = ONE workgroup per CU: 128 instead of 77 flop per operand byte and 8 instead of 10 DMA
//     pieces per 64 / 48 MFMAs of a wave -- the long-K, wide-N shapes whose k-loop outweighs the exposed prologue / epilogue of a lone workgroup.
template <int EPI, bool EXT = false, int TM = 6, int NWN = 2, int ABL = 0>      // EXT: the row phase reads a second tensor (residual / saved pre-activation); ABL (dev, tools/ablate_gemm16.sh): 1 = no operand DMA, 32 = no MFMAs, 64 = per-tile cycle trace into p.colstats
__global__ __launch_bounds__(128 * NWN, 2) void gemm_nt_dma_kernel(GemmParams p) {
    constexpr int NW = 2 * NWN, NTHR = 64 * NW;
    constexpr int BM = 2 * TM * 16, BN = NWN * 64, SK = 64, NST = 2;
    constexpr int TA = BM * SK, TB = BN * SK, STAGE = TA + TB;            // elements per stage (40 KB / 64 KB)
    static_assert(NST * STAGE * 2 <= 163840 / (NWN == 2 ? 2 : 1) && NW * 2 * 32 * 72 <= NST * STAGE, "LDS: two workgroups of the 192 x 128 form share a CU; the epilogue's wave-private scratch reuses the ring");
    constexpr int TN = 4;
    constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW, DPS = PA + PB;       // DMA pieces (8 rows x 128 B) per wave and stage: 6 A + 4 B / 4 A + 4 B
    static_assert(PA * NW * 8 == BM && PB * NW * 8 == BN, "the waves must divide the pieces of both operand tiles");
    __shared__ __attribute__((aligned(16))) ge_t smem[NST * STAGE];
    const int tiles = p.tilesM * p.tilesN;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int lr = lane & 15, lg = lane >> 4;
    // an XCD runs 64 consecutive ids at a time: row-major order makes them one A panel x 64 B panels (wide N: every B panel is fetched by one workgroup
    // only, the whole weight matrix streams through the L2 once per M-tile row); in groups of group_m M-tiles walked column by column they are
    // group_m A panels x 64 / group_m B panels
    auto tile_mn = [&](int t, int& m0, int& n0) {
        const int bid = gg_xcd_remap(t, tiles);
        int tm, tn;
        if (p.group_m > 1) {
            const int per = p.group_m * p.tilesN, g = bid / per, r = bid - g * per;
            const int first = g * p.group_m, gsz = min(p.tilesM - first, p.group_m);
            tn = r / gsz; tm = first + (r - tn * gsz);
        } else { tm = bid / p.tilesN; tn = bid - tm * p.tilesN; }
        m0 = tm * BM; n0 = tn * BN;
    };
    // DMA geometry: piece pc = wave + NW j covers tile rows 8 pc .. 8 pc + 7; lane -> (row 8 pc + lane / 8, LDS chunk slot lane % 8) and fetches SOURCE chunk
    // slot ^ T(row); bit 1 of the row is bit 4 of the lane, bit 3 of the row is bit 0 of the piece = bit 0 of the wave (NW j is even)
    const int dchunk = (lane & 7) ^ (((lane >> 3) & 2) | ((wave & 1) << 2));
    unsigned voffA[PA], voffB[PB];
#pragma unroll
    for (int j = 0; j < PA; ++j) voffA[j] = (unsigned)((wave + NW * j) * 8 + (lane >> 3)) * (unsigned)p.lda * 2u + dchunk * 16u;
#pragma unroll
    for (int j = 0; j < PB; ++j) voffB[j] = (unsigned)((wave + NW * j) * 8 + (lane >> 3)) * (unsigned)p.ldb * 2u + dchunk * 16u;
    auto issue_stage = [&](const __amdgpu_buffer_rsrc_t& rsA, const __amdgpu_buffer_rsrc_t& rsB, int st, ge_t* base) {
        if (ABL & 1) return;
        const int k0 = st * SK;
        const bool kin = k0 + dchunk * 8 < p.K;                  // K % 8 == 0: a chunk is entirely inside or outside K
#pragma unroll
        for (int j = 0; j < PA; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(base + (wave + NW * j) * 512), 16,
                                                     (int)(kin ? voffA[j] : 0xFFFFFFF0u), k0 * 2, 0, 0);
#pragma unroll
        for (int j = 0; j < PB; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(base + TA + (wave + NW * j) * 512), 16,
                                                     (int)(kin ? voffB[j] : 0xFFFFFFF0u), k0 * 2, 0, 0);
    };
    auto make_rs = [&](int m0, int n0, __amdgpu_buffer_rsrc_t& rsA, __amdgpu_buffer_rsrc_t& rsB) {
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)m0 * p.lda), 0, (int)((unsigned)min(p.M - m0, BM) * (unsigned)p.lda * 2u), 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (int64_t)n0 * p.ldb), 0, (int)((unsigned)min(p.N - n0, BN) * (unsigned)p.ldb * 2u), 0x00020000);
    };
    // fragment addresses (elements): row 16 t + lr of an operand tile, k-step ks, k-chunk lg -> chunk slot (4 ks + lg) ^ T(lr)
    const int sw = (lr & 2) | ((lr >> 1) & 4);
    const int kc0 = ((0 + lg) ^ sw) << 3, kc1 = ((4 + lg) ^ sw) << 3;
    const int a_off = (wm * (TM * 16) + lr) * SK, b_off = TA + (wn * 64 + lr) * SK;
    const int nk = (p.K + SK - 1) / SK;
    const int t = blockIdx.x;
    int m0, n0;
    tile_mn(t, m0, n0);
    __amdgpu_buffer_rsrc_t rsA, rsB;
    make_rs(m0, n0, rsA, rsB);
    unsigned long long tr_t0 = 0, tr_first = 0, tr_wait = 0, tr_issue = 0, tr_loop = 0, tr_c0 = 0, tr_c1 = 0, tr_rt0 = 0;
    if (ABL & 64) { tr_t0 = __builtin_readcyclecounter(); tr_rt0 = __builtin_amdgcn_s_memrealtime(); }
    issue_stage(rsA, rsB, 0, smem);
    if (nk > 1) issue_stage(rsA, rsB, 1, smem + STAGE);
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (nk > 1) gemm_wait_vmcnt<DPS>(); else gemm_wait_vmcnt<0>();     // stage 0 has landed (stage 1 may be in flight)
    __builtin_amdgcn_s_barrier();
    if (ABL & 64) { tr_first = __builtin_readcyclecounter(); tr_wait = 0; tr_issue = 0; }
    ge8_t ar[4], bq[2][TN];
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) bq[0][nt] = *reinterpret_cast<const ge8_t*>(smem + b_off + nt * 16 * SK + kc0);
    ar[0] = *reinterpret_cast<const ge8_t*>(smem + a_off + kc0);
    ar[1] = *reinterpret_cast<const ge8_t*>(smem + a_off + 16 * SK + kc0);
    f32x4 e_bs[4]; float e_rsv[TM]; gemm_u32x4_t e_ex[2 * TM];            // the epilogue's memory operands
    // STEADY: stage s + 2 exists (no conditions inside).  The sched_barriers pin the order [read A two m-tiles ahead; 4 MFMAs of m-tile mt]: left alone
    // the compiler sinks each read to just above its first use and waits for it there.  The next stage's first fragments are read unconditionally (the last
    // stage reads stale bytes it never uses): a branch around them makes the compiler's wait-count merge at the join pessimistic.
    auto stage = [&](auto steady, int s, int slot) {
        constexpr bool STEADY = decltype(steady)::value;
        ge_t* const cur = smem + slot * STAGE;
        ge_t* const nxt = smem + (slot ^ 1) * STAGE;
#pragma unroll
        for (int i = 0; i < 2 * TM; ++i) {                     // i = TM ks + mt
            const int ks = i / TM, mt = i % TM;
            if (i + 2 < 2 * TM) {
                const int i2 = i + 2;
                ar[i2 & 3] = *reinterpret_cast<const ge8_t*>(cur + a_off + (i2 % TM) * 16 * SK + (i2 / TM ? kc1 : kc0));
            }
            if (i == 2) {                                       // the second k-step's B fragments
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) bq[1][nt] = *reinterpret_cast<const ge8_t*>(cur + b_off + nt * 16 * SK + kc1);
            }
            if (i == 2 * TM - 2) {
                // every fragment read of stage s has been issued (the last A fragment one m-tile ago)
                if (ABL & 64) tr_c0 = __builtin_readcyclecounter();
                if (STEADY || s + 1 < nk) gemm_wait_vmcnt<0>(); // this wave's DMAs of stage s + 1 have landed (nothing else is in flight: ring of two)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                   // everybody's have, and everybody has read its fragments of stage s
                if (ABL & 64) { tr_c1 = __builtin_readcyclecounter(); tr_wait += tr_c1 - tr_c0; }
                if (STEADY || s + 2 < nk) issue_stage(rsA, rsB, s + 2, cur);
                if (ABL & 64) tr_issue += __builtin_readcyclecounter() - tr_c1;
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) bq[0][nt] = *reinterpret_cast<const ge8_t*>(nxt + b_off + nt * 16 * SK + kc0);
                ar[0] = *reinterpret_cast<const ge8_t*>(nxt + a_off + kc0);
                ar[1] = *reinterpret_cast<const ge8_t*>(nxt + a_off + 16 * SK + kc0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(ABL & 32))
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) acc[nt][mt] = ge_mfma(bq[ks][nt], ar[i & 3], acc[nt][mt]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int s = 0;
    for (; s + 3 < nk; s += 2) {
        stage(std::true_type{}, s, 0);
        stage(std::true_type{}, s + 1, 1);
    }
    for (; s < nk; ++s) {
        if (s == nk - 1) gemm_dma_epi_fetch<EPI, TM>(p, e_bs, e_rsv, m0, n0, wm, wn, lane);      // nothing else is in flight now: lands under the last stage's MFMAs
        stage(std::false_type{}, s, s & 1);
    }
    __builtin_amdgcn_s_barrier();                               // the ring is idle: no DMA in flight, every fragment read
    if (ABL & 64) tr_loop = __builtin_readcyclecounter();
    gemm_dma_epi_ready<EPI, TM>(p, e_bs, e_rsv);
    gemm_dma_ext_fetch<EPI, EXT, TM>(p, e_ex, m0, n0, wm, wn, lane);
    gemm_dma_epilogue<EPI, EXT, TM>(p, smem + wave * (2 * 32 * 72), acc, e_bs, e_rsv, e_ex, m0, n0, wm, wn, lane);
    if ((ABL & 64) && threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* tt = reinterpret_cast<unsigned long long*>(p.colstats) + (size_t)t * 8;
        const unsigned long long now = __builtin_readcyclecounter();
        tt[0] = tr_first - tr_t0; tt[1] = tr_loop - tr_first; tt[2] = tr_wait; tt[3] = tr_issue; tt[4] = now - tr_loop; tt[5] = tr_rt0;
        tt[6] = hw | ((unsigned long long)(xcc & 0xF) << 32); tt[7] = __builtin_amdgcn_s_memrealtime();
    }
}

// ------------------------------------------------------------------------------------------- TN GEMM (weight gradients)
// dW[N,K] = sum_m dY[m,n] * X[m,k]: both operands are row-major over the REDUCTION index m, so their MFMA fragments
// (8 consecutive m for a fixed n / k) are read from the row-major LDS tiles with the transposing ds_read_b64_tr_b16 --
// no transposed copies of dY and X in HBM.  Output tile 128(n) x 128(k), 2x2 waves; the m range is split over
// blockIdx.y and each split writes an fp32 slab part[split][N][K] (deterministic; reduced by splitk_reduce_kernel).
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ge8_t tn_frag(const ge_t* T, int RS, int col, int lr, int lg) {
    // lane (lr, lg) <- T[4lg + q][col + lr], T[16 + 4lg + q][col + lr]  (q = 0..3); lane 4q+p of a 16-lane group supplies the
    // address of row q, columns 4p..4p+3
    const ge_t* p0 = T + (4 * lg + (lr >> 2)) * RS + col + 4 * (lr & 3);
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 16 * RS));
    union { s16x4 s[2]; ge8_t v; } u;
    u.s[0] = a; u.s[1] = b;
    return u.v;
}
struct TnParams {
    const ge_t* dY; int64_t ldy; const ge_t* X; int64_t ldx;
    int M, N, K;
    const float* rowscale; int rows_per_scale;
    float* part;          // [splits][N][K]
    int tilesN, tilesK, m_per_split;
    const ge_t* Y2; const float* coef;    // optional: the dY operand is  coef0*dY + coef1*Y2 + coef2  per column (BatchNorm backward's
                                          // apply step of a ConvNorm whose dy only feeds this weight gradient), bf16-rounded like the stored dy
};
template <bool BN>        // BN: dY := coef0*dY + coef1*Y2 + coef2 while loading (its 44 extra registers stay out of the plain kernel: 3 vs 2 waves/SIMD)
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TnParams p) {
    // 64-row m-step = two MFMA k-steps per barrier pair.  LDS rows of 288 bytes: the 16 rows x 32 bytes that one transposing
    // read touches then fall on 16 disjoint groups of 8 banks (a 272-byte stride overlaps neighbouring rows: 2-way conflicts)
    constexpr int TB = 128, MS = 64, RS = TB + 16;
    constexpr int LS = MS / 16;                            // 16-byte chunks per thread per operand per step
    __shared__ __attribute__((aligned(16))) ge_t Ys[MS * RS];
    __shared__ __attribute__((aligned(16))) ge_t Xs[MS * RS];
    // flattened (split, tile) order laid out XCD by XCD: the tiles of one split re-read the same row range of dY (tilesK times) and
    // X (tilesN times); dispatched round-robin they would pull it into all eight L2s
    const int ntile = p.tilesN * p.tilesK;
    const int lb = gg_xcd_remap(blockIdx.x, gridDim.x);
    const int split_id = lb / ntile, tile_id = lb - split_id * ntile;
    const int tn = tile_id / p.tilesK, tk = tile_id % p.tilesK;
    const int n0 = tn * TB, k0 = tk * TB;
    const int mbeg = split_id * p.m_per_split, mend = min(p.M, mbeg + p.m_per_split);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int lr = lane & 15, lg = lane >> 4;
    // staging: MS rows x 16 chunks (16 B) per operand -> LS per thread: row = c >> 4 (+16 i), chunk = c & 15.  Range-checked
    // buffer loads over this split's rows: rows beyond mend read as zeros, a column chunk beyond N / K is pushed out of range
    // through its offset -- no branches around the loads (N, K multiples of 8; host checks the split's byte range < 2^32)
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int srow = threadIdx.x >> 4, sch = threadIdx.x & 15;
    const bool yok = (n0 + sch * 8) < p.N, xok = (k0 + sch * 8) < p.K;
    const int nrows = max(mend - mbeg, 0);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dY + (int64_t)mbeg * p.ldy), 0, (int)((unsigned)nrows * (unsigned)p.ldy * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(p.X + (int64_t)mbeg * p.ldx), 0, (int)((unsigned)nrows * (unsigned)p.ldx * 2u), 0x00020000);
    unsigned voy[LS], vox[LS];
#pragma unroll
    for (int i = 0; i < LS; ++i) {
        voy[i] = yok ? ((unsigned)(srow + 16 * i) * (unsigned)p.ldy + (unsigned)(n0 + sch * 8)) * 2u : 0xFFFFFFF0u;
        vox[i] = xok ? ((unsigned)(srow + 16 * i) * (unsigned)p.ldx + (unsigned)(k0 + sch * 8)) * 2u : 0xFFFFFFF0u;
    }
    const unsigned stepY = (unsigned)MS * (unsigned)p.ldy * 2u, stepX = (unsigned)MS * (unsigned)p.ldx * 2u;
    const float inv_rps = p.rowscale ? 1.f / (float)p.rows_per_scale : 0.f;
    const __amdgpu_buffer_rsrc_t rsY2 = __builtin_amdgcn_make_buffer_rsrc((void*)((p.Y2 ? p.Y2 : p.dY) + (int64_t)mbeg * p.ldy), 0, (int)((unsigned)nrows * (unsigned)p.ldy * 2u), 0x00020000);
    float ca[8], cb[8], cc[8];            // this thread's 8 columns (sch is fixed per thread)
    if (BN) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = min(n0 + sch * 8 + j, p.N - 1);
            ca[j] = p.coef[c]; cb[j] = p.coef[p.N + c]; cc[j] = p.coef[2 * p.N + c];
        }
    }
    ge8_t ry[LS], rx[LS], ry2[LS];
    float rsc[LS];
    auto load_step = [&](int m0) {
        const unsigned st = (unsigned)(m0 - mbeg) / MS;
#pragma unroll
        for (int i = 0; i < LS; ++i) {
            // a pushed-out chunk stays out of range: its offset is only advanced when valid
            ry[i] = __builtin_bit_cast(ge8_t, __builtin_amdgcn_raw_buffer_load_b128(rsY, (int)(yok ? voy[i] + st * stepY : 0xFFFFFFF0u), 0, 0));
            rx[i] = __builtin_bit_cast(ge8_t, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(xok ? vox[i] + st * stepX : 0xFFFFFFF0u), 0, 0));
            if (BN) ry2[i] = __builtin_bit_cast(ge8_t, __builtin_amdgcn_raw_buffer_load_b128(rsY2, (int)(yok ? voy[i] + st * stepY : 0xFFFFFFF0u), 0, 0));
        }
        if (p.rowscale) {
#pragma unroll
            for (int i = 0; i < LS; ++i) {
                const int m = min(m0 + srow + 16 * i, p.M - 1);
                int q = (int)((float)m * inv_rps);                       // m / rows_per_scale without the integer division
                q += ((q + 1) * p.rows_per_scale <= m) - (q * p.rows_per_scale > m);
                rsc[i] = p.rowscale[q];
            }
        }
    };
    // element-wise work on the fetched dY rows, run when they are consumed (just before the LDS store), not where they are
    // requested: touching them inside load_step would wait for the loads before the MFMAs they are meant to travel under
    auto finish_step = [&](int m0) {
        if (BN) {          // rows beyond the split / columns beyond N must stay zero: the affine's constant term does not apply there
#pragma unroll
            for (int i = 0; i < LS; ++i) {
                const bool ok = yok && (m0 + srow + 16 * i) < mend;
#pragma unroll
                for (int j = 0; j < 8; ++j) ry[i][j] = ok ? (ge_t)fmaf(ca[j], (float)ry[i][j], fmaf(cb[j], (float)ry2[i][j], cc[j])) : (ge_t)0.f;
            }
        }
        if (p.rowscale) {
#pragma unroll
            for (int i = 0; i < LS; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) ry[i][j] = (ge_t)((float)ry[i][j] * rsc[i]);
            }
        }
    };
    f32x4 acc[4][4];     // [k tile][n tile]: D[i = n][j = k] with A = dY^T fragment, B = X fragment
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (mbeg < mend) load_step(mbeg);
    for (int m0 = mbeg; m0 < mend; m0 += MS) {
        finish_step(m0);
#pragma unroll
        for (int i = 0; i < LS; ++i) {
            *reinterpret_cast<ge8_t*>(Ys + (srow + 16 * i) * RS + sch * 8) = ry[i];
            *reinterpret_cast<ge8_t*>(Xs + (srow + 16 * i) * RS + sch * 8) = rx[i];
        }
        __syncthreads();
        if (m0 + MS < mend) load_step(m0 + MS);
#pragma unroll
        for (int ms = 0; ms < MS / 32; ++ms) {
            if (m0 + ms * 32 >= mend) break;         // uniform: the tail step may hold only one k-step of rows
            ge8_t yf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                yf[i] = tn_frag(Ys + ms * 32 * RS, RS, wn * 64 + i * 16, lr, lg);
                xf[i] = tn_frag(Xs + ms * 32 * RS, RS, wk * 64 + i * 16, lr, lg);
            }
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[kt][nt] = ge_mfma(yf[nt], xf[kt], acc[kt][nt]);   // rows n, cols k
        }
        __syncthreads();
    }
    // lane holds D[n = .. + nt*16 + 4lg + r][k = .. + kt*16 + lr]
    float* out = p.part + (int64_t)split_id * p.N * p.K;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int k = k0 + wk * 64 + kt * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + nt * 16 + lg * 4 + r;
                if (n < p.N && k < p.K) out[(int64_t)n * p.K + k] = acc[kt][nt][r];
            }
        }
}

// sum split-K partials: out[i] = (accumulate ? out[i] : 0) + sum_z part[z][i]
__global__ void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t n, int splits,
                                     int accumulate, float scale) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < splits; ++z) s += part[(int64_t)z * n + i];
        s *= scale;
        out[i] = accumulate ? out[i] + s : s;
    }
}
// 16 bytes per lane, 4 slabs in flight (n % 4 == 0, 16-byte aligned slabs and output)
__global__ __launch_bounds__(256) void splitk_reduce_v4_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t n4,
                                                               int splits, int accumulate, float scale) {
    const f32x4* P = reinterpret_cast<const f32x4*>(part);
    f32x4* O = reinterpret_cast<f32x4*>(out);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int z = 0;
        for (; z + 4 <= splits; z += 4) {
            const f32x4 a = P[(int64_t)z * n4 + i], b = P[(int64_t)(z + 1) * n4 + i], c = P[(int64_t)(z + 2) * n4 + i], d = P[(int64_t)(z + 3) * n4 + i];
            s += (a + b) + (c + d);
        }
        for (; z < splits; ++z) s += P[(int64_t)z * n4 + i];
        s *= scale;
        O[i] = accumulate ? O[i] + s : s;
    }
}

// bf16 [R, C] (row stride ld) -> bf16 [C, R] (row stride ldo), optional per-row scale (drop-path) applied while
// transposing.  Columns/rows beyond the source are not written: the caller zero-pads ldo.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const ge_t* __restrict__ in, int64_t ld, ge_t* __restrict__ out,
                                                             int64_t ldo, int R, int C, const float* rowscale,
                                                             int rows_per_scale) {
    __shared__ ge_t tile[64][64 + 2];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) {
            v = (float)in[(int64_t)r * ld + c];
            if (rowscale) v *= rowscale[r / rows_per_scale];
        }
        tile[i][tx] = (ge_t)v;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) out[(int64_t)c * ldo + r] = tile[tx][i];
    }
}

// f32 [R, C] -> bf16 [R, ldo] (cast) and/or bf16 [C, ldt] (cast + transpose): weight-cache refresh.
__global__ __launch_bounds__(256) void cast_transpose_f32_kernel(const float* __restrict__ in, int R, int C, ge_t* __restrict__ out,
                                                                 int64_t ldo, ge_t* __restrict__ outT, int64_t ldt) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) {
            v = in[(int64_t)r * C + c];
            if (out) out[(int64_t)r * ldo + c] = (ge_t)v;
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    if (outT) {
        for (int i = ty; i < 64; i += 4) {
            const int c = c0 + i, r = r0 + tx;
            if (c < C && r < R) outT[(int64_t)c * ldt + r] = (ge_t)tile[tx][i];
        }
    }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ in, ge_t* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (ge_t)in[i];
}
__global__ void cast_bf16_f32_kernel(const ge_t* __restrict__ in, float* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (float)in[i];
}

// column sums of a bf16 [M, C] matrix (bias gradients): partials [gridDim.y][C] then a finalize pass.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const ge_t* __restrict__ x, int64_t ld, int M, int C,
                                                             const float* rowscale, int rows_per_scale,
                                                             float* __restrict__ part, int rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s = 0.f;
    for (int r = r0; r < r1; ++r) {
        float v = (float)x[(int64_t)r * ld + c];
        if (rowscale) v *= rowscale[r / rows_per_scale];
        s += v;
    }
    part[(int64_t)blockIdx.y * C + c] = s;
}
// 16-byte version: thread = (8-column group of 32, row lane of 8); a block covers 256 columns x rows_per_block rows
__global__ __launch_bounds__(256) void colsum_partial_v8_kernel(const ge_t* __restrict__ x, int64_t ld, int M, int C,
                                                                const float* rowscale, int rows_per_scale,
                                                                float* __restrict__ part, int rows_per_block) {
    __shared__ float red[8][256];
    const int cg = threadIdx.x & 31, pp = threadIdx.x >> 5;
    const int c0 = blockIdx.x * 256 + cg * 8;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.f;
    if (c0 < C) {
#pragma unroll 4
        for (int r = r0 + pp; r < r1; r += 8) {
            const ge8_t v = *reinterpret_cast<const ge8_t*>(x + (int64_t)r * ld + c0);
            const float rs = rowscale ? rowscale[r / rows_per_scale] : 1.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] = fmaf((float)v[j], rs, s[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[pp][cg * 8 + j] = s[j];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
        part[(int64_t)blockIdx.y * C + c] = t;
    }
}
__global__ void colsum_final_kernel(const float* __restrict__ part, int nparts, int C, float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int i = 0; i < nparts; ++i) s += (double)part[(int64_t)i * C + c];
    out[c] = accumulate ? out[c] + (float)s : (float)s;
}

// ------------------------------------------------------------------------------------------- host
// This file is also compiled a second time with the 16-bit operand type swapped to fp16 (gemm_f16.hip: `bf16` / the MFMA builtin redefined,
// everything inside namespace gg_f16): GG_GEMM_NT_NAME is the exported name of that build's NT entry point, and only it is exported there.
#ifndef GG_GEMM_NT_NAME
#define GG_GEMM_NT_NAME gg_gemm_nt
#endif
extern "C" int GG_GEMM_NT_NAME(const GgGemmArgs* a, void* stream) {
    GG_CHECK(a && a->A && a->B && a->C, "gg_gemm_nt: null operand");
#ifdef GG_GEMM_SECOND_TYPE
    GG_CHECK(!a->a_bn_stat && !a->A2 && !a->bn_y && !a->colstats && a->split_k <= 1, "gg_gemm_nt_f16: the BatchNorm-fused / two-source / split-K forms exist in the ge_t build only");
#endif
    GG_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "gg_gemm_nt: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
    GG_CHECK((a->K & 7) == 0 && (a->lda & 7) == 0 && (a->ldb & 7) == 0,
             "gg_gemm_nt: K, lda, ldb must be multiples of 8 (16-byte rows): K=%d lda=%lld ldb=%lld", a->K,
             (long long)a->lda, (long long)a->ldb);
    GG_CHECK(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0, "gg_gemm_nt: A/B must be 16-byte aligned");
    GG_CHECK(a->lda >= (a->A2 ? std::max(a->k_split, a->K - a->k_split) : a->K) && a->ldb >= a->K && a->ldc >= a->N,
             "gg_gemm_nt: leading dimension too small");
    GG_CHECK(a->lda * 256 < 0xFFFFFF00LL && a->ldb * 256 < 0xFFFFFF00LL && (int64_t)a->K * 2 < 0x7FFFFFFFLL,
             "gg_gemm_nt: leading dimension too large for 32-bit tile offsets (ld < 16.7M elements)");
    GG_CHECK((!a->residual || a->ldr * 256 < 0xFFFFFF00LL) && (!a->dact_preact || a->ldc * 256 < 0xFFFFFF00LL),
             "gg_gemm_nt: residual / pre-activation leading dimension too large for 32-bit tile offsets");
    const int split = a->split_k > 1 ? a->split_k : 1;
    if (split > 1)
        GG_CHECK(!a->bias && !a->act && !a->preact && !a->residual && !a->colstats && !a->dact && !a->rowscale,
                 "gg_gemm_nt: split-K writes raw f32 partials, no epilogue allowed");
    if (a->rowscale) GG_CHECK(a->rows_per_scale > 0, "gg_gemm_nt: rows_per_scale must be > 0");
    GG_CHECK(!a->dact_preact || a->dact == GG_ACT_GELU || a->dact == GG_ACT_QUICK_GELU, "gg_gemm_nt: dact must be GELU or QuickGELU");
    GG_CHECK(!(a->dact_preact && (a->bias || a->act || a->residual || a->preact)), "gg_gemm_nt: dact excludes bias/act/residual/preact");
    GG_CHECK(!(a->act && (a->rowscale || a->residual)), "gg_gemm_nt: an activation epilogue excludes rowscale/residual");
    GG_CHECK(!a->preact || a->act == GG_ACT_GELU || a->act == GG_ACT_QUICK_GELU, "gg_gemm_nt: preact is only available with an activation epilogue");
    GG_CHECK(!a->colstats || !(a->bias || a->act || a->rowscale || a->residual || a->dact_preact || a->out_f32),
             "gg_gemm_nt: colstats is only available on the plain and BatchNorm-backward ge_t epilogues");
    GG_CHECK(!a->out_f32 || !(a->act || a->rowscale || a->residual || a->dact_preact || a->preact), "gg_gemm_nt: f32 output supports bias only");
    if (a->A2)
        GG_CHECK(a->k_split > 0 && a->k_split < a->K && (a->k_split % BK) == 0 && split == 1 && ((uintptr_t)a->A2 & 15) == 0,
                 "gg_gemm_nt: A2 needs 0 < k_split < K, k_split %% 64 == 0, no split-K, 16-byte alignment (k_split=%d K=%d)", a->k_split, a->K);
    if (a->bn_y) {
        GG_CHECK(a->bn_stat && a->bn_gamma && a->bn_beta && a->colstats, "gg_gemm_nt: the BatchNorm-backward epilogue needs stat, gamma, beta, colstats");
        GG_CHECK(!(a->bias || a->act || a->rowscale || a->residual || a->dact_preact || a->out_f32 || a->preact) && split == 1,
                 "gg_gemm_nt: the BatchNorm-backward epilogue excludes every other epilogue option");
        GG_CHECK(a->ldc * 256 < 0xFFFFFF00LL, "gg_gemm_nt: ldc too large for 32-bit tile offsets");
    }
    if (a->a_bn_stat) {
        GG_CHECK(a->a_bn_gamma && a->a_bn_beta && a->K <= 1024 && split == 1 && !a->A2,
                 "gg_gemm_nt: the BatchNorm prologue needs gamma/beta, K <= 1024, no split-K, a single A source");
        GG_CHECK(!(a->bias || a->act || a->rowscale || a->residual || a->dact_preact || a->out_f32 || a->preact || a->bn_y),
                 "gg_gemm_nt: the BatchNorm prologue is built for the plain (+ column statistics) epilogue only");
    }
    GemmParams p;
    p.group_m = 0;
    p.a_stat = a->a_bn_stat; p.a_gamma = a->a_bn_gamma; p.a_beta = a->a_bn_beta; p.a_act = a->a_bn_act;
    p.A2 = (const ge_t*)a->A2; p.k_split = a->k_split;
    p.bn_y = (const ge_t*)a->bn_y; p.bn_stat = a->bn_stat; p.bn_gamma = a->bn_gamma; p.bn_beta = a->bn_beta; p.bn_act = a->bn_act;
    p.A = (const ge_t*)a->A; p.lda = a->lda; p.B = (const ge_t*)a->B; p.ldb = a->ldb;
    p.C = a->C; p.ldc = a->ldc; p.M = a->M; p.N = a->N; p.K = a->K;
    p.bias = a->bias; p.act = a->act; p.preact = (ge_t*)a->preact;
    p.rowscale = a->rowscale; p.rows_per_scale = a->rows_per_scale;
    p.residual = (const ge_t*)a->residual; p.ldr = a->ldr;
    p.dact_preact = (const ge_t*)a->dact_preact; p.dact = a->dact_preact ? a->dact : 0;
    p.colstats = a->colstats; p.out_f32 = a->out_f32;
    p.split_k = split;
    int kps = (int)gg_cdiv(a->K, split);
    kps = (int)gg_align(kps, BK);
    p.k_per_split = kps;
    // tile choice: HBM-bound shapes (short K loop or a single narrow N tile) take the light 128x64 tile
    static const char* dbg = gg_dev_env("GG_GEMM_DEBUG");
    p.debug = dbg ? atoi(dbg) : 0;
    static const char* force = gg_dev_env("GG_GEMM_TILE");
    const int rem = a->N % 128;
    bool narrow = a->N <= 64 || (rem != 0 && rem <= 64);     // a 128-wide tile would be at most half full
    if (force) narrow = force[0] == 'n';
    const int bn = narrow ? 64 : 128;
    p.tilesM = (int)gg_cdiv(a->M, 128); p.tilesN = (int)gg_cdiv(a->N, bn);
    // algorithmic bytes: A, B, C once each (bf16; f32 C = 4 bytes) plus every second tensor the epilogue reads or writes
    const double mn = (double)a->M * a->N;
    // algorithmic flops: the two-source dgrad (A2) contracts [dz | y] against the folded [c0 W ; c1 W] weights -- a doubled K that exists only
    // because the BatchNorm-backward apply step was folded into the MFMA; the algorithm's contraction is k_split long (what the fp32 mode declares)
    GG_PROF(GG_CAT_GEMM, 2.0 * a->M * (double)a->N * (a->A2 ? a->k_split : a->K),
            2.0 * ((double)a->M * a->K + (double)a->N * a->K) + ((a->out_f32 || split > 1) ? 4.0 : 2.0) * mn * std::max(1, split) +
                2.0 * mn * ((a->preact != nullptr) + (a->residual != nullptr) + (a->dact_preact != nullptr) + (a->bn_y != nullptr)),
            stream);
    dim3 grid(p.tilesM * p.tilesN, split);
    int epi;
    if (a->out_f32 || split > 1) epi = EPI_F32;
    else if (p.bn_y) epi = EPI_BNBWD;
    else if (p.dact) epi = EPI_DGELU;
    else if (a->act == GG_ACT_GELU || (a->act == GG_ACT_QUICK_GELU && a->preact)) epi = EPI_GELU;      // (row-phase epilogue: pre-activation copy)
    else if (a->act == GG_ACT_QUICK_GELU) epi = EPI_QGELU;
    else if (a->bias || a->rowscale || a->residual) epi = EPI_LINEAR;
    else epi = EPI_PLAIN;
    hipStream_t st = (hipStream_t)stream;
    // the MFMA-bound shapes (the transformer Linears and their data gradients) take the LDS-DMA form: 256 x 128 tiles
    // (its epilogue has no general-shape fallbacks: whole 16-byte column chunks, 16-byte aligned rows of every tensor it touches)
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    const char* dma_sw = gg_dev_env("GG_GEMM_DMA");        // (not cached: the dev tools flip it between launches of one process)
    const int dma_var = dma_sw ? atoi(dma_sw) : 1;
    bool dma = !p.a_stat && !p.A2 && !p.bn_y && (!p.colstats || dma_var == 6) && split == 1 && !a->out_f32 && a->K >= 192 && a->N >= 128 && a->M >= 1024 &&
               (rem == 0 || rem > 64) && a->lda * 512 < 0xFFFFFF00LL && a->ldb * 256 < 0xFFFFFF00LL && a->ldc * 512 < 0xFFFFFF00LL &&
               (!a->residual || (a->ldr * 512 < 0xFFFFFF00LL && (a->ldr & 7) == 0 && al16(a->residual))) && (a->N & 7) == 0 && (a->ldc & 7) == 0 && al16(a->C) &&
               (!a->preact || al16(a->preact)) && (!a->dact_preact || al16(a->dact_preact)) && (!a->bias || al16(a->bias));
    if (dma_sw) dma = dma && dma_var != 0;
    if (const char* only = gg_dev_env("GG_GEMM_DMA_ONLY")) { int on = 0, ok = 0, oe = -1; sscanf(only, "%d,%d,%d", &on, &ok, &oe); dma = dma && a->N == on && a->K == ok && (oe < 0 || oe == epi); }      // (dev: bisecting)
    if (dma) {
        // the 256 x 256 geometry (one workgroup per CU) where its k-loop outweighs a lone workgroup's exposed prologue / epilogue: measured (tools/bench_gemm16.py,
        // profiles/r05_gemm16_forms.txt) 1.37-1.41 PFLOP/s against 1.13-1.15 at K = 4096 / 8192, a tie at K = 3072 (CLIP fc2), 5-12 % slower at K = 384 ... 768
        // -- except under the heaviest epilogue (GELU + pre-activation copy: two output tensors), where its eight waves finish the tile 3-5 % sooner
        const char* tile_sw = gg_dev_env("GG_GEMM_DMA_TILE");
        bool big = (a->K >= 4096 || (epi == EPI_GELU && a->preact && a->N >= 1536)) && a->N % 256 == 0 && (int64_t)gg_cdiv(a->M, 256) * gg_cdiv(a->N, 256) >= 512 &&
                   a->ldb * 512 < 0xFFFFFF00LL;
        if (tile_sw) big = atoi(tile_sw) == 256 && a->ldb * 512 < 0xFFFFFF00LL;
        const int bm = big ? 256 : 192, bn = big ? 256 : 128;
        p.tilesM = (int)gg_cdiv(a->M, bm); p.tilesN = (int)gg_cdiv(a->N, bn);
        const dim3 g2((unsigned)(p.tilesM * p.tilesN)), blk(big ? 512 : 256);
        const char* gm_sw = gg_dev_env("GG_GEMM_GM");
        p.group_m = gm_sw ? atoi(gm_sw) : (p.tilesN > (big ? 4 : 8) ? (big ? 4 : 8) : 0);
        if (dma_var > 1 && epi == EPI_PLAIN && !big) {       // dev: ablations / trace (tools/ablate_gemm16.sh, tools/trace_gemm16.py)
            if (dma_var == 3) hipLaunchKernelGGL((gemm_nt_dma_kernel<EPI_PLAIN, false, 6, 2, 1>), g2, blk, 0, st, p);
            else if (dma_var == 4) hipLaunchKernelGGL((gemm_nt_dma_kernel<EPI_PLAIN, false, 6, 2, 32>), g2, blk, 0, st, p);
            else if (dma_var == 6) hipLaunchKernelGGL((gemm_nt_dma_kernel<EPI_PLAIN, false, 6, 2, 64>), g2, blk, 0, st, p);
            else hipLaunchKernelGGL((gemm_nt_dma_kernel<EPI_PLAIN, false, 6, 2, 33>), g2, blk, 0, st, p);
            GG_LAUNCH_CHECK();
            return 0;
        }
#define GG_DMA_LAUNCH(E, X)                                                                                             \
    do {                                                                                                                \
        if (big) hipLaunchKernelGGL((gemm_nt_dma_kernel<E, X, 8, 4>), g2, blk, 0, st, p);                               \
        else hipLaunchKernelGGL((gemm_nt_dma_kernel<E, X, 6, 2>), g2, blk, 0, st, p);                                   \
    } while (0)
        switch (epi) {
            case EPI_PLAIN: GG_DMA_LAUNCH(EPI_PLAIN, false); break;
            case EPI_LINEAR:
                if (p.residual) GG_DMA_LAUNCH(EPI_LINEAR, true);
                else GG_DMA_LAUNCH(EPI_LINEAR, false);
                break;
            case EPI_GELU: GG_DMA_LAUNCH(EPI_GELU, false); break;
            case EPI_QGELU: GG_DMA_LAUNCH(EPI_QGELU, false); break;
            default: GG_DMA_LAUNCH(EPI_DGELU, true); break;
        }
#undef GG_DMA_LAUNCH
        GG_LAUNCH_CHECK();
        return 0;
    }
#define GG_LAUNCH_EPI(E)                                                                                            \
    do {                                                                                                              \
        if (narrow) hipLaunchKernelGGL((gemm_nt_kernel<128, 64, 4, 1, 5, E>), grid, dim3(256), 0, st, p);             \
        else hipLaunchKernelGGL((gemm_nt_kernel<128, 128, 2, 2, 3, E>), grid, dim3(256), 0, st, p);                   \
    } while (0)
    if (p.a_stat) {       // (implies EPI_PLAIN)
        if (narrow) hipLaunchKernelGGL((gemm_nt_kernel<128, 64, 4, 1, 5, EPI_PLAIN, true>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((gemm_nt_kernel<128, 128, 2, 2, 3, EPI_PLAIN, true>), grid, dim3(256), 0, st, p);
        GG_LAUNCH_CHECK();
        return 0;
    }
    switch (epi) {
        case EPI_PLAIN: GG_LAUNCH_EPI(EPI_PLAIN); break;
        case EPI_LINEAR: GG_LAUNCH_EPI(EPI_LINEAR); break;
        case EPI_GELU: GG_LAUNCH_EPI(EPI_GELU); break;
        case EPI_QGELU: GG_LAUNCH_EPI(EPI_QGELU); break;
        case EPI_DGELU: GG_LAUNCH_EPI(EPI_DGELU); break;
        case EPI_BNBWD: GG_LAUNCH_EPI(EPI_BNBWD); break;
        default: GG_LAUNCH_EPI(EPI_F32); break;
    }
#undef GG_LAUNCH_EPI
    GG_LAUNCH_CHECK();
    return 0;
}
#ifndef GG_GEMM_SECOND_TYPE      // (the helpers below are type-independent or bf16-only: exported by the bf16 build)
extern "C" int gg_gemm_colstats_rows(int M) { return (int)gg_cdiv(M, 128); }
extern "C" int gg_stat_rows_capacity(int rows) { return rows + GG_REDUCE_SLICES; }

// many slabs of a small matrix (patch_embed.conv1: 1536 x 6 KB): 64 float4 columns x 16 slab lanes per block, so a thread walks
// splits / 16 slabs instead of all of them (the flat kernel spent 170 us on 9 MB here)
__global__ __launch_bounds__(1024) void splitk_reduce_tall_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t n4,
                                                                  int splits, int accumulate, float scale) {
    __shared__ f32x4 red[16][64];
    const f32x4* P = reinterpret_cast<const f32x4*>(part);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + tx;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n4)
        for (int z = ty; z < splits; z += 16) s += P[(int64_t)z * n4 + i];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < n4) {
#pragma unroll
        for (int k = 1; k < 16; ++k) s += red[k][tx];
        s *= scale;
        f32x4* O = reinterpret_cast<f32x4*>(out);
        O[i] = accumulate ? O[i] + s : s;
    }
}
extern "C" int gg_splitk_reduce(const float* part, float* out, int64_t n, int splits, int accumulate, float scale, void* stream) {
    GG_CHECK(part && out && n > 0 && splits > 0, "gg_splitk_reduce: bad args");
    GG_PROF(GG_CAT_MOVE, 0, 4.0 * n * (splits + 1), stream);
    int blocks = (int)std::min<int64_t>(gg_cdiv(n, 256), 4096);
    const bool v4 = (n & 3) == 0 && ((uintptr_t)part & 15) == 0 && ((uintptr_t)out & 15) == 0;
    if (v4 && splits >= 64 && n / 4 <= 16384)
        hipLaunchKernelGGL(splitk_reduce_tall_kernel, dim3((unsigned)gg_cdiv(n / 4, 64)), dim3(1024), 0, (hipStream_t)stream, part, out, n / 4, splits, accumulate, scale);
    else if (v4)
        hipLaunchKernelGGL(splitk_reduce_v4_kernel, dim3((unsigned)std::min<int64_t>(gg_cdiv(n / 4, 256), 16384)), dim3(256), 0, (hipStream_t)stream, part, out, n / 4, splits, accumulate, scale);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, part, out, n, splits, accumulate, scale);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_transpose_bf16(const void* in, int64_t ld, void* out, int64_t ldo, int R, int C, const float* rowscale,
                                 int rows_per_scale, void* stream) {
    GG_CHECK(in && out && R > 0 && C > 0 && ld >= C && ldo >= R, "gg_transpose_bf16: bad args");
    GG_PROF(GG_CAT_MOVE, 0, 4.0 * R * C, stream);
    dim3 grid((unsigned)gg_cdiv(C, 64), (unsigned)gg_cdiv(R, 64));
    GG_CHECK(grid.y <= 65535 * 32u, "gg_transpose_bf16: too many rows");
    // gridDim.y limit is 2^31-1 on HIP for y? keep it safe: y <= 65535 requires R <= 4.19M; larger R is chunked
    const int max_rows = 65535 * 64;
    for (int r0 = 0; r0 < R; r0 += max_rows) {
        const int rr = std::min(max_rows, R - r0);
        dim3 g((unsigned)gg_cdiv(C, 64), (unsigned)gg_cdiv(rr, 64));
        hipLaunchKernelGGL(transpose_bf16_kernel, g, dim3(256), 0, (hipStream_t)stream,
                           (const ge_t*)in + (int64_t)r0 * ld, ld, (ge_t*)out + r0, ldo, rr, C,
                           rowscale ? rowscale + 0 : nullptr, rows_per_scale);
        // rowscale index uses the chunk-local row: only valid when r0 is a multiple of rows_per_scale
        if (rowscale) GG_CHECK(r0 == 0, "gg_transpose_bf16: rowscale with > 4.19M rows unsupported");
    }
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_cast_transpose_f32(const float* in, int R, int C, void* out, int64_t ldo, void* outT, int64_t ldt, void* stream) {
    GG_CHECK(in && R > 0 && C > 0 && (out || outT), "gg_cast_transpose_f32: bad args");
    dim3 grid((unsigned)gg_cdiv(C, 64), (unsigned)gg_cdiv(R, 64));
    hipLaunchKernelGGL(cast_transpose_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, R, C, (ge_t*)out, ldo, (ge_t*)outT, ldt);
    GG_LAUNCH_CHECK();
    return 0;
}
// f32 [R, C] -> f32 [C, ldt] (ldt >= R; columns R..ldt-1 of the output are left as they are): the f32 head weight's transposed copy
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, int R, int C, float* __restrict__ outT, int64_t ldt) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? in[(int64_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) outT[(int64_t)c * ldt + r] = tile[tx][i];
    }
}
extern "C" int gg_transpose_f32(const float* in, int R, int C, float* outT, int64_t ldt, void* stream) {
    GG_CHECK(in && outT && R > 0 && C > 0 && ldt >= R, "gg_transpose_f32: bad args");
    GG_PROF(GG_CAT_MOVE, 0, 8.0 * R * C, stream);
    dim3 grid((unsigned)gg_cdiv(C, 64), (unsigned)gg_cdiv(R, 64));
    GG_CHECK(grid.y <= 65535u, "gg_transpose_f32: too many rows");
    hipLaunchKernelGGL(transpose_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, R, C, outT, ldt);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream) {
    GG_CHECK(in && out && n > 0, "gg_cast_f32_to_bf16: bad args");
    int blocks = (int)std::min<int64_t>(gg_cdiv(n, 256), 8192);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, (ge_t*)out, n);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_cast_bf16_to_f32(const void* in, float* out, int64_t n, void* stream) {
    GG_CHECK(in && out && n > 0, "gg_cast_bf16_to_f32: bad args");
    int blocks = (int)std::min<int64_t>(gg_cdiv(n, 256), 8192);
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const ge_t*)in, out, n);
    GG_LAUNCH_CHECK();
    return 0;
}
// out[C] (+)= column sums of x[M,C]; scratch must hold gg_colsum_scratch_floats(M, C) floats
extern "C" int64_t gg_colsum_scratch_floats(int M, int C) { return ((int64_t)gg_cdiv(M, 512) + GG_REDUCE_SLICES) * C; }
extern "C" int gg_colsum_bf16(const void* x, int64_t ld, int M, int C, const float* rowscale, int rows_per_scale,
                              float* scratch, float* out, int accumulate, void* stream) {
    GG_CHECK(x && scratch && out && M > 0 && C > 0, "gg_colsum_bf16: bad args");
    GG_PROF(GG_CAT_NORM, 0, 2.0 * M * C, stream);
    const int rpb = 512;
    const int nparts = (int)gg_cdiv(M, rpb);
    GG_CHECK(nparts <= 65535, "gg_colsum_bf16: M too large");
    if ((C & 7) == 0 && (ld & 7) == 0 && ((uintptr_t)x & 15) == 0)
        hipLaunchKernelGGL(colsum_partial_v8_kernel, dim3((unsigned)gg_cdiv(C, 256), nparts), dim3(256), 0, (hipStream_t)stream,
                           (const ge_t*)x, ld, M, C, rowscale, rows_per_scale, scratch, rpb);
    else
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)gg_cdiv(C, 256), nparts), dim3(256), 0, (hipStream_t)stream,
                           (const ge_t*)x, ld, M, C, rowscale, rows_per_scale, scratch, rpb);
    const float* rows; int nrows;
    gg_reduce_rows(scratch, nparts, C, (hipStream_t)stream, &rows, &nrows);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)gg_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, rows, nrows, C, out, accumulate);
    GG_LAUNCH_CHECK();
    return 0;
}

// dW partial slabs: part[splits][N][K] = per-split  dY[M,N]^T . X[M,K]   (rowscale: optional per-sample scale on dY rows)
extern "C" int gg_gemm_tn_splits(int M, int N, int K) {
    const int64_t tiles = gg_cdiv(N, 128) * gg_cdiv(K, 128);
    int64_t s = std::max<int64_t>(1, std::min<int64_t>(gg_cdiv(1536, tiles), gg_cdiv(M, 1024)));
    const int64_t cap = ((int64_t)64 << 20) / ((int64_t)N * K * 4);        // 64 MiB of slabs at most
    return (int)std::max<int64_t>(1, std::min<int64_t>(s, cap));
}
static int gemm_tn_launch(const void* dY, int64_t ldy, const void* Y2, const float* coef, const void* X, int64_t ldx, int M, int N, int K,
                          const float* rowscale, int rows_per_scale, float* partials, int splits, void* stream);
extern "C" int gg_gemm_tn(const void* dY, int64_t ldy, const void* X, int64_t ldx, int M, int N, int K, const float* rowscale,
                          int rows_per_scale, float* partials, int splits, void* stream) {
    return gemm_tn_launch(dY, ldy, nullptr, nullptr, X, ldx, M, N, K, rowscale, rows_per_scale, partials, splits, stream);
}
// weight gradient of a ConvNorm straight from BatchNorm backward's (dz, y, coef): dW = (coef0*dz + coef1*y + coef2)^T X; the dy
// tensor and the apply pass that would write it do not exist (valid when dy feeds nothing else, e.g. the network's first conv)
extern "C" int gg_gemm_tn_bn(const void* dz, const void* y, int64_t ldy, const float* coef, const void* X, int64_t ldx, int M, int N, int K,
                             float* partials, int splits, void* stream) {
    GG_CHECK(y && coef && ((uintptr_t)y & 15) == 0, "gg_gemm_tn_bn: y / coef missing or misaligned");
    return gemm_tn_launch(dz, ldy, y, coef, X, ldx, M, N, K, nullptr, 0, partials, splits, stream);
}
static int gemm_tn_launch(const void* dY, int64_t ldy, const void* Y2, const float* coef, const void* X, int64_t ldx, int M, int N, int K,
                          const float* rowscale, int rows_per_scale, float* partials, int splits, void* stream) {
    GG_CHECK(dY && X && partials && M > 0 && N > 0 && K > 0 && splits > 0, "gg_gemm_tn: bad args");
    GG_CHECK((N & 7) == 0 && (K & 7) == 0 && (ldy & 7) == 0 && (ldx & 7) == 0, "gg_gemm_tn: N, K, ldy, ldx must be multiples of 8");
    GG_CHECK(((uintptr_t)dY & 15) == 0 && ((uintptr_t)X & 15) == 0, "gg_gemm_tn: operands must be 16-byte aligned");
    GG_CHECK(!rowscale || rows_per_scale > 0, "gg_gemm_tn: rows_per_scale");
    TnParams p;
    p.dY = (const ge_t*)dY; p.ldy = ldy; p.X = (const ge_t*)X; p.ldx = ldx; p.M = M; p.N = N; p.K = K;
    p.rowscale = rowscale; p.rows_per_scale = rows_per_scale; p.part = partials; p.Y2 = (const ge_t*)Y2; p.coef = coef;
    p.tilesN = (int)gg_cdiv(N, 128); p.tilesK = (int)gg_cdiv(K, 128);
    p.m_per_split = (int)gg_align(gg_cdiv(M, splits), 64);
    GG_CHECK((int64_t)p.m_per_split * std::max(ldy, ldx) * 2 < ((int64_t)1 << 32), "gg_gemm_tn: a split's rows must span < 4 GiB per operand (use more splits)");
    GG_CHECK(splits <= 65535, "gg_gemm_tn: too many splits");
    GG_PROF(GG_CAT_GEMM, 2.0 * M * (double)N * K, 2.0 * M * ((double)N * (Y2 ? 2 : 1) + K) + 4.0 * splits * (double)N * K, stream);
    GG_CHECK((int64_t)p.tilesN * p.tilesK * splits < ((int64_t)1 << 31), "gg_gemm_tn: grid too large");
    const dim3 grid((unsigned)(p.tilesN * p.tilesK * splits));
    if (coef) hipLaunchKernelGGL(gemm_tn_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(gemm_tn_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, p);
    GG_LAUNCH_CHECK();
    return 0;
}
#endif  // GG_GEMM_SECOND_TYPE
