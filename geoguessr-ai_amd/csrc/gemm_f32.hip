// fp32 MFMA GEMMs of the reference-precision mode (gfx950):  C[M,N] = epi(A[M,K] . B[N,K]^T)  and the weight-gradient form
// dW[N,K] = dY[M,N]^T . X[M,K], all operands and results f32, arithmetic on v_mfma_f32_16x16x4_f32 (exact f32 products, f32
// accumulation: bit-for-bit a k-ordered fmaf chain -- the precision of the reference's own torch fp32 matmuls).
//
// The reference runs TinyViT / the geocell head in fp32 end to end; this file is the contraction kernel of the mode
// that does the same.  At 64 FLOP/clk/SIMD the f32 MFMA is 1/16 of the bf16 rate, so unlike the bf16 kernel of gemm.hip every shape
// of the model is MFMA-bound here (ridge point 157 TFLOP/s / 8 TB/s = 20 flop/B, the model's leanest GEMM has 2K = 192 flop per
// 12 B): the kernel keeps the matrix pipe fed and otherwise stays simple -- results leave straight from the accumulator fragments.
//
// NT tile 128 x BN x 32 (BN = 128: 2x2 waves, 64x64 per wave; BN = 64: 4x1 waves, 32x64 per wave), 256 threads.  An operand tile is
// R rows x 32 floats = R x 128 bytes, the same byte image as the bf16 kernel's R x 64 tile, so it uses the same 16-byte XOR swizzle
// (conflict-free ds_write_b128 staging and ds_read_b128 fragment reads).  One ds_read_b128 hands a lane 4 consecutive k of its row:
// they feed 4 successive MFMA k-steps (lane (r, g) owns k = 4g + s in step s for BOTH operands, so every k is summed exactly once).
#include <map>
#include <mutex>
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include "../../include/gg.h"

#define FBK 32   // floats per k-tile

namespace {

struct F32GemmParams {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    float* C; int64_t ldc;
    int M, N, K;
    const float* bias;
    float* preact;
    const float* rowscale; int rows_per_scale;
    const float* residual; int64_t ldr;
    const float* dact_preact;
    float* colstats;              // [tilesM][2][N]
    int tilesM, tilesN;
    int debug;                    // timing experiments only (GG_GEMM_F32_DEBUG): 1 no operand loads, 2 no result stores, 4 no LDS staging / barriers
    // BatchNorm-backward epilogue (FE_BNBWD): C = dz = acc * act'(BN(bn_y)), colstats <- column sums of (dz, dz*xhat)
    const float* bn_y; const float* bn_stat; const float* bn_gamma; const float* bn_beta; int bn_act;
    // A-operand prologues of the register-staged kernel: PRO 1: A := a_act(BN(A)) (a_stat = [mean | rstd][K], gamma, beta);
    // PRO 2: A := coef0*A + coef1*A2 + coef2 (a_stat = coef [3][K]: BatchNorm backward's apply step formed while staging)
    const float* A2; const float* a_stat; const float* a_gamma; const float* a_beta; int a_act;
    unsigned long long* trace;      // dev (gg_gemm_f32_set_trace): per-workgroup [hw_id, xcc_id, t_start, t_first_data, t_loop_end, t_epilogue_end, tile, 0] (100 MHz ticks)
    int quick;          // FE_GELU / FE_DGELU: the activation is QuickGELU (CLIP) instead of erf GELU
    int swz_plain = 0;  // dev A/B (GG_GEMM_F32_SWZ=0): the ring kernel's tile rows unpermuted (rounds 2-5: 2-way bank conflicts on the fragment reads)
    int splits = 1, k_per_split = 0;      // small-M form: workgroup (split s, tile t) contracts k in [s k_per_split, (s + 1) k_per_split) into slab s of C ([splits][M][ldc])
};

__device__ __forceinline__ int f32_chunk_off(int row, int kc) {      // float offset of 16-byte chunk kc (0..7) of a tile row
    return row * FBK + ((kc ^ ((row & 2) | ((row >> 1) & 4))) << 2);
}

// fp32 GELU family (common.h gg_phi_f32: fp32-accurate, branch-free)
__device__ __forceinline__ float gelu_exact(float x) { return gg_gelu_f32(x); }
__device__ __forceinline__ float gelu_grad_exact(float x) { return gg_gelu_grad_f32(x); }

enum { FE_PLAIN = 0, FE_LINEAR = 1, FE_GELU = 2, FE_QGELU = 3, FE_DGELU = 4, FE_BNBWD = 5 };
__device__ __forceinline__ float act_grad_exact_f(float x, int act) { return gg_act_grad_f32(x, act); }
__device__ __forceinline__ float act_exact_f(float x, int act) { return gg_act_f32(x, act); }

__device__ __forceinline__ float ror8_f(float v) {          // value of lane (l + 8) % 16 of the same 16-lane DPP row
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {        // sum over the 16 lanes of a DPP row, result in every lane of the row
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, false));   // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, false));   // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, false));   // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xF, 0xF, false));   // row_ror:1
    return v;
}

// Epilogue shared by both NT kernels: lane holds C[m = m0 + wm*WROWS + mt*16 + lr][n = n0 + wn*WCOLS + nt*16 + lg*4 + r]; results leave straight
// from the accumulator fragments (16-byte stores, 64-byte runs per row).  `smem`: at least 2*WM*BN floats, no longer read by anyone.
// MSPLIT = 2: the tile's 16-row blocks are processed in two halves (half the second-tensor prefetch registers per n-tile, which lets the 4-per-CU
// form prefetch two n-tiles at a time and fetch them as whole 128-byte lines)
template <int BM, int BN, int WM, int WN, int EPI, int GNMAX = 2, int MSPLIT = 1>
__device__ __forceinline__ void gemm_f32_epilogue(const F32GemmParams& p, float* smem, f32x4 (&acc)[BN / WN / 16][BM / WM / 16], int m0, int n0,
                                                  int tm, int wm, int wn, int lr, int lg) {
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16, TMH = TM / MSPLIT;
    static_assert(TM % MSPLIT == 0, "MSPLIT must divide the m-tiles of a wave");
    constexpr int WROWS = BM / WM, WCOLS = BN / WN;
    const bool vec_c = (p.ldc & 3) == 0;
    // full-line stores: a lane's 16 bytes of two neighbouring n-tiles are regrouped (one DPP row rotate) so that a store instruction writes
    // 8 rows x 128 B (whole cache lines) instead of 16 rows x 64 B (every line in two halves, 4 stores apart)
    const bool pair_store = TN >= 2 && vec_c && (p.ldc & 31) == 0 && (p.debug & 128) == 0;      // (debug 128: the 64-byte-run stores, for A/B)
        float* red = smem;                                    // [WM][2][BN] column partials (the k-loop's last barrier has passed)
    // the epilogue's second tensor (BatchNorm-backward: saved conv output; GELU': saved pre-activation; linear: residual) for the whole
    // tile, all 16 loads in flight at once -- fetched inside the loop below they were 16 serialised memory round trips per tile
    constexpr bool AUX = EPI == FE_BNBWD || EPI == FE_DGELU || EPI == FE_LINEAR;
    constexpr int GN = (TN * TM > 8) ? GNMAX : TN;           // n-tiles per prefetch group: at most 8 x 16 bytes per lane in flight (32 registers; 16 in the 4-per-CU form)
    const float* aux_src = EPI == FE_BNBWD ? p.bn_y : (EPI == FE_DGELU ? p.dact_preact : (EPI == FE_LINEAR ? p.residual : nullptr));
    const int64_t aux_ld = EPI == FE_LINEAR ? p.ldr : p.ldc;
    const bool aux_vec = (aux_ld & 3) == 0;
    f32x4 aux[AUX ? GN : 1][AUX ? TMH : 1];
    // whole-line fetch of the second tensor: with two n-tiles per group a load instruction reads 8 rows x 128 B (lanes lr < 8: tile e of row lr & 7,
    // lanes lr >= 8: tile o of the same row; second load: rows 8..15) and one DPP row rotate hands every lane its own fragment
    const bool pair_aux = AUX && GN == 2 && aux_src != nullptr && aux_vec && (aux_ld & 31) == 0 && (p.N & 31) == 0 && (p.debug & 128) == 0;
    const int hh = lr >> 3, l7 = lr & 7;
#pragma unroll
    for (int mh = 0; mh < MSPLIT; ++mh) {
#pragma unroll
    for (int g0 = 0; g0 < TN; g0 += GN) {
        const bool paired = pair_aux && g0 + 1 < TN;
        if (AUX) {
            if (paired) {
                const int ne = n0 + wn * WCOLS + g0 * 16 + lg * 4;
#pragma unroll
                for (int mtl = 0; mtl < TMH; ++mtl) {
                    const int mb = m0 + wm * WROWS + (mh * TMH + mtl) * 16;
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int m = mb + 8 * half + l7, n = (hh == half) ? ne : ne + 16;
                        f32x4 h = {0.f, 0.f, 0.f, 0.f};
                        if (m < p.M) h = *reinterpret_cast<const f32x4*>(aux_src + (int64_t)m * aux_ld + n);
                        aux[AUX ? half : 0][AUX ? mtl : 0] = h;
                    }
                }
            } else {
#pragma unroll
                for (int gi = 0; gi < GN; ++gi) {
                    const int n = n0 + wn * WCOLS + (g0 + gi) * 16 + lg * 4;
#pragma unroll
                    for (int mtl = 0; mtl < TMH; ++mtl) {
                        const int m = m0 + wm * WROWS + (mh * TMH + mtl) * 16 + lr;
                        f32x4 h = {0.f, 0.f, 0.f, 0.f};
                        if (g0 + gi < TN && aux_src && m < p.M && n < p.N) {
                            const float* g = aux_src + (int64_t)m * aux_ld + n;
                            if (aux_vec && n + 3 < p.N) h = *reinterpret_cast<const f32x4*>(g);
                            else { for (int r = 0; r < 4; ++r) if (n + r < p.N) h[r] = g[r]; }
                        }
                        aux[AUX ? gi : 0][AUX ? mtl : 0] = h;
                    }
                }
            }
            if (paired) {      // lanes' own fragments: tile e from the load that covered this lane's row half, tile o from the partner lane's
#pragma unroll
                for (int mtl = 0; mtl < TMH; ++mtl) {
                    const f32x4 t0 = aux[0][AUX ? mtl : 0], t1 = aux[AUX && GN > 1 ? 1 : 0][AUX ? mtl : 0];
                    const f32x4 r0 = {ror8_f(t0[0]), ror8_f(t0[1]), ror8_f(t0[2]), ror8_f(t0[3])};
                    const f32x4 r1 = {ror8_f(t1[0]), ror8_f(t1[1]), ror8_f(t1[2]), ror8_f(t1[3])};
                    aux[0][AUX ? mtl : 0] = hh ? t1 : t0;
                    aux[AUX && GN > 1 ? 1 : 0][AUX ? mtl : 0] = hh ? r1 : r0;
                }
            }
        }
#pragma unroll
        for (int gi = 0; gi < GN; ++gi) {
            const int nt = g0 + gi;
            if (nt >= TN) break;
            const int n = n0 + wn * WCOLS + nt * 16 + lg * 4;
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (EPI != FE_PLAIN && EPI != FE_DGELU && p.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) b4[r] = p.bias[min(n + r, p.N - 1)];
            }
            f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cq = {0.f, 0.f, 0.f, 0.f};
            f32x4 bsc = cs, bsh = cs, brs = cs, bnm = cs;
            if (EPI == FE_BNBWD) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = min(n + r, p.N - 1);
                    const float mu = p.bn_stat[c], rstd = p.bn_stat[p.N + c];
                    bsc[r] = p.bn_gamma[c] * rstd; bsh[r] = p.bn_beta[c] - mu * bsc[r]; brs[r] = rstd; bnm[r] = -mu * rstd;
                }
            }
#pragma unroll
            for (int mtl = 0; mtl < TMH; ++mtl) {
                const int mt = mh * TMH + mtl;
                const int m = m0 + wm * WROWS + mt * 16 + lr;
                f32x4 v = acc[nt][mt];
                const bool ok = m < p.M && n < p.N;
                const bool full = ok && vec_c && n + 3 < p.N;
                if (EPI == FE_PLAIN) {
                    if (p.colstats) { cs += v; cq += v * v; }   // rows beyond M / columns beyond N hold exact zeros (range-checked operand loads)
                } else if (EPI == FE_BNBWD) {
                    if (ok) {       // dz = da * act'(gamma*xhat + beta); column sums of dz and dz*xhat
                        const f32x4 yv = aux[AUX ? gi : 0][AUX ? mtl : 0];
                        v *= gg_act_grad_f32_v4(yv * bsc + bsh, p.bn_act);
                        cs += v; cq += v * (yv * brs + bnm);
                    }
                } else if (ok) {
                    const float rs = ((EPI == FE_LINEAR || EPI == FE_DGELU) && p.rowscale) ? p.rowscale[m / p.rows_per_scale] : 1.f;
                    v += b4;
                    if (EPI == FE_GELU && pair_store && nt < (TN & ~1)) { acc[nt][mt] = v; continue; }      // pre-activation copy + activation happen at the paired store
                    if (EPI == FE_GELU) {
                        if (p.preact) {
                            float* g = p.preact + (int64_t)m * p.ldc + n;
                            if (full) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(g));
                            else { for (int r = 0; r < 4; ++r) if (n + r < p.N) g[r] = v[r]; }
                        }
                        if (p.quick) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + expf(-1.702f * v[r]));
                        } else {
                            v = gg_act_f32_v4(v, GG_ACT_GELU);
                        }
                    }
                    if (EPI == FE_QGELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + expf(-1.702f * v[r]));
                    }
                    if (EPI == FE_DGELU) {
                        const f32x4 h = aux[AUX ? gi : 0][AUX ? mtl : 0];
                        if (p.quick) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float sg = 1.0f / (1.0f + expf(-1.702f * h[r]));
                                v[r] *= (sg + 1.702f * h[r] * sg * (1.0f - sg)) * rs;
                            }
                        } else {
                            v *= gg_act_grad_f32_v4(h, GG_ACT_GELU) * rs;
                        }
                    }
                    if (EPI == FE_LINEAR) {
                        v *= rs;
                        v += aux[AUX ? gi : 0][AUX ? mtl : 0];            // zeros without a residual
                    }
                }
                if (pair_store && nt < (TN & ~1)) { acc[nt][mt] = v; continue; }      // stored below, two n-tiles at a time
                if (ok && !((p.debug & 2) && v[0] != 12345.678f)) {
                    float* g = p.C + (int64_t)m * p.ldc + n;
                    if (full) *reinterpret_cast<f32x4*>(g) = v;
                    else { for (int r = 0; r < 4; ++r) if (n + r < p.N) g[r] = v[r]; }
                }
            }
            if ((EPI == FE_PLAIN || EPI == FE_BNBWD) && p.colstats) {
                // column sums over this wave's 64 / 32 rows: the 16 row-lanes first, then the WM waves that share columns through LDS
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // the 16 row-lanes of a column are one DPP row: rotate-and-add inside the VALU (row_ror 8/4/2/1) instead of 8 ds_bpermute round trips
                    const float a = row16_sum(cs[r]), b = row16_sum(cq[r]);
                    if (lr == 0) {
                        const int col = wn * WCOLS + nt * 16 + lg * 4 + r;
                        if (mh == 0) { red[(wm * 2 + 0) * BN + col] = a; red[(wm * 2 + 1) * BN + col] = b; }
                        else { red[(wm * 2 + 0) * BN + col] += a; red[(wm * 2 + 1) * BN + col] += b; }
                    }
                }
            }
        }
    }
    }      // mh
    if (pair_store && !(p.debug & 2)) {
        const int h = hh;
#pragma unroll
        for (int e = 0; e + 1 < TN; e += 2) {
            const int ne = n0 + wn * WCOLS + e * 16 + lg * 4;
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const f32x4 ve = acc[e][mt], vo = acc[e + 1][mt];
                // row_ror:8: the partner row's odd tile
                const float x0 = ror8_f(vo[0]), x1 = ror8_f(vo[1]), x2 = ror8_f(vo[2]), x3 = ror8_f(vo[3]);
                const f32x4 X = {x0, x1, x2, x3};
                const int mb = m0 + wm * WROWS + mt * 16;
#pragma unroll
                for (int half = 0; half < 2; ++half) {          // store 1: rows mb .. mb+7, store 2: rows mb+8 .. mb+15; each row = the 128 B of tiles e | e+1
                    const bool own = (h == half);
                    f32x4 d = own ? ve : X;
                    const int n = own ? ne : ne + 16;
                    const int m = mb + 8 * half + l7;
                    if (m < p.M && n < p.N) {
                        const bool full4 = n + 3 < p.N;
                        if (EPI == FE_GELU) {
                            if (p.preact) {
                                float* gp = p.preact + (int64_t)m * p.ldc + n;
                                if (full4) __builtin_nontemporal_store(d, reinterpret_cast<f32x4*>(gp));
                                else { for (int r = 0; r < 4; ++r) if (n + r < p.N) gp[r] = d[r]; }
                            }
                            if (p.quick) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) d[r] = d[r] / (1.0f + expf(-1.702f * d[r]));
                            } else {
                                d = gg_act_f32_v4(d, GG_ACT_GELU);
                            }
                        }
                        float* g = p.C + (int64_t)m * p.ldc + n;
                        if (full4) *reinterpret_cast<f32x4*>(g) = d;
                        else { for (int r = 0; r < 4; ++r) if (n + r < p.N) g[r] = d[r]; }
                    }
                }
            }
        }
    }
        if ((EPI == FE_PLAIN || EPI == FE_BNBWD) && p.colstats) {
            __syncthreads();
            for (int i = threadIdx.x; i < 2 * BN; i += 256) {
                const int which = i / BN, col = i % BN;
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) s += red[(w * 2 + which) * BN + col];
                if (n0 + col < p.N) p.colstats[(int64_t)tm * 2 * p.N + which * p.N + n0 + col] = s;
            }
            __syncthreads();                                  // the next tile's operand staging reuses this LDS
        }
}

// Row-layout epilogue of the ring kernels (interior tiles of the 64-column wave layouts).  fp32 MFMA and VALU instructions share the SIMD's
// vector issue (tools/mfma_shadow.hip, tools/valu_under_mfma.hip), so every VALU instruction of an epilogue is matrix time taken from the
// co-resident workgroups; what the epilogue above spends on DPP lane exchanges (full-line stores and fetches), per-element bounds checks and
// 64-bit address arithmetic goes away here:
//   * the accumulators cross the wave's private slice of the (now idle) ring through LDS -- ds_write_b128 in the MFMA layout (lane = 16 x row,
//     4 columns), ds_read_b128 as whole rows (16 lanes x 16 B = the wave's 64 columns of one row; a store instruction = 4 rows x 256 B) -- LDS
//     instructions, no VALU; 16-byte slots XOR-swizzled with (row & 7): conflict-free writes, reads at most 2-way on 2 of 8 rows;
//   * all element-wise work happens in the row layout: a lane's 4 columns are the same for every row, so bias / BatchNorm coefficients are loaded
//     once per tile, the second tensor (residual, pre-activation, saved conv output) is fetched row-wise (whole lines) straight into place;
//   * addresses = scalar tile base (SGPR arithmetic) + one per-lane 32-bit offset.
// LDSF = floats of LDS the ring leaves (>= 4 waves x 16 rows x 64).  Results differ from the epilogue above only in the summation order of the
// column statistics.
template <int BM, int WM, int WN, int TM, int TN, int EPI, int LDSF>
__device__ __forceinline__ void gemm_f32_epilogue_rows(const F32GemmParams& p, float* smem, f32x4 (&acc)[TN][TM], int m0, int n0, int tm, int wave, int lane) {
    // TN = 4: the wave's 64 columns = 16 slots of 16 bytes, a row instruction = 4 rows x 16 slots; TN = 6 (the 96-column tile): 24 slots, a row
    // instruction = 2 rows x 24 slots on 48 of the 64 lanes
    static_assert(TN == 4 || TN == 6, "row-layout epilogue: 64- or 96-column wave tiles");
    constexpr int SLOTS = TN * 4, PITCH = SLOTS * 16, RPI = 64 / SLOTS, WCOLS = TN * 16;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    constexpr int WROWS = BM / WM;
    constexpr int CAP = LDSF / 4 / (SLOTS * 64);                               // 16-row blocks the wave's LDS slice holds
    constexpr int MTP = CAP >= TM ? TM : ((CAP >= 2 && TM % 2 == 0) ? 2 : 1);   // ... per pass
    static_assert(MTP >= 1 && TM % MTP == 0, "LDS slice too small");
    constexpr int NPASS = TM / MTP, RP = MTP * 16, NI = RP / RPI;               // rows per pass, row-layout instructions per pass (RPI rows each)
    constexpr bool AUX = EPI == FE_BNBWD || EPI == FE_DGELU || EPI == FE_LINEAR;
    constexpr bool STATS = EPI == FE_PLAIN || EPI == FE_BNBWD;
    constexpr int G = EPI == FE_BNBWD ? 2 : 4;       // row instructions per group (second-tensor fetches in flight; BNBWD keeps 16 coefficient registers)
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lg = lane >> 4;        // MFMA layout: row lr of a 16-row block, columns nt*16 + lg*4 ..
    const int r4 = lane / SLOTS, sl = lane % SLOTS;  // row layout: row r4 of RPI, 16-byte slot sl of the wave's columns
    const bool active = lane < RPI * SLOTS;
    char* const lds = reinterpret_cast<char*>(smem) + wave * (RP * PITCH);
    constexpr int NRB = 8 / RPI;                     // (row & 7) of instruction i, lane row r4: ((i % NRB) * RPI) | r4
    unsigned wb[2], rb[NRB];
#pragma unroll
    for (int b = 0; b < 2; ++b) wb[b] = (unsigned)(lr * PITCH + ((((b << 2) | lg) ^ (lr & 7)) << 4));
#pragma unroll
    for (int b = 0; b < NRB; ++b) rb[b] = (unsigned)(r4 * PITCH + ((sl ^ ((b * RPI) | r4)) << 4));
    // the first pass's accumulators leave for LDS before anything else is loaded: they are dead from here on, which keeps the per-tile constants
    // below (up to 24 registers in the BatchNorm-backward form) inside the 128 registers of the 4-per-CU kernels
    auto to_lds = [&](int ps) {
#pragma unroll
        for (int mtl = 0; mtl < MTP; ++mtl)
#pragma unroll
            for (int nt = 0; nt < TN; ++nt)
                *reinterpret_cast<f32x4*>(lds + wb[nt & 1] + mtl * (16 * PITCH) + (nt >> 1) * 128) = acc[nt][ps * MTP + mtl];
    };
    to_lds(0);
    __builtin_amdgcn_sched_barrier(0);
    const int mw0 = m0 + wm * WROWS, nw0 = n0 + wn * WCOLS;
    const float* aux_src = EPI == FE_BNBWD ? p.bn_y : (EPI == FE_DGELU ? p.dact_preact : (EPI == FE_LINEAR ? p.residual : nullptr));
    const int64_t aux_ld = EPI == FE_LINEAR ? p.ldr : p.ldc;
    // addressing: one buffer descriptor per tensor on the wave's first element, scalar row offset (soffset), per-lane offset of (row r4, slot sl)
    const unsigned strideC = (unsigned)p.ldc * 4u, strideX = (unsigned)aux_ld * 4u;
    const unsigned offC = (unsigned)r4 * strideC + (unsigned)sl * 16u, offX = (unsigned)r4 * strideX + (unsigned)sl * 16u;
    // STORES take a descriptor rebased per row group (scalar adds) and NO register soffset: with a register soffset the compiler assumes the
    // hardware has read a 16-byte store's data by the next instruction (LLVM's VMEM-store hazard rule exempts that form) and reused the data
    // registers for an LDS address in the following instruction -- on gfx950 one workgroup in ~2000 then stored the address
    char* const cptr = reinterpret_cast<char*>(p.C + (int64_t)mw0 * p.ldc + nw0);
    char* const pptr = reinterpret_cast<char*>((EPI == FE_GELU && p.preact ? p.preact : p.C) + (int64_t)mw0 * p.ldc + nw0);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)((AUX && aux_src ? aux_src : p.C) + (int64_t)mw0 * aux_ld + nw0), 0, (int)(WROWS * strideX), 0x00020000);
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if ((EPI == FE_LINEAR || EPI == FE_GELU || EPI == FE_QGELU) && p.bias) b4 = *reinterpret_cast<const f32x4*>(p.bias + nw0 + min(sl, SLOTS - 1) * 4);
    f32x4 bsc = b4, bsh = b4, brs = b4, bnm = b4, cs = {0.f, 0.f, 0.f, 0.f}, cq = cs;
    if (EPI == FE_BNBWD) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(p.bn_stat + nw0 + sl * 4), rstd = *reinterpret_cast<const f32x4*>(p.bn_stat + p.N + nw0 + sl * 4);      // (sl < SLOTS always: lane % SLOTS)
        bsc = *reinterpret_cast<const f32x4*>(p.bn_gamma + nw0 + sl * 4) * rstd;
        bsh = *reinterpret_cast<const f32x4*>(p.bn_beta + nw0 + sl * 4) - mu * bsc;
        brs = rstd; bnm = -mu * rstd;
    }
    // per-row scale (DropPath): lane l holds the scale of the wave's row l; a row instruction fetches its rows' values with one ds_bpermute
    const bool scaled = (EPI == FE_LINEAR || EPI == FE_DGELU) && p.rowscale != nullptr;
    int rs_lane = 0x3F800000;
    if (scaled) rs_lane = __builtin_bit_cast(int, p.rowscale[min(mw0 + lane, p.M - 1) / p.rows_per_scale]);
    const int rs_addr = r4 * 4;
    const bool quick = p.quick != 0;
    // FLAG: the one per-launch option that would otherwise become per-element selects -- PLAIN: column statistics wanted; LINEAR: a residual is added
    auto run = [&](auto flag) {
        constexpr bool FLAG = decltype(flag)::value;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            if (ps > 0) to_lds(ps);
#pragma unroll
            for (int i0 = 0; i0 < NI; i0 += G) {
                f32x4 aux[AUX ? G : 1];
                if (AUX && (EPI != FE_LINEAR || FLAG)) {
#pragma unroll
                    for (int j = 0; j < G; ++j)
                        aux[AUX ? j : 0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(active ? offX : 0xFFFFFFF0u), (int)((unsigned)(ps * RP + (i0 + j) * RPI) * strideX), 0));
                }
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int i = i0 + j, rel = ps * RP + i * RPI;        // the RPI rows mw0 + rel .., this lane: + r4
                    f32x4 v = *reinterpret_cast<const f32x4*>(lds + rb[i % NRB] + i * (RPI * PITCH));
                    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)(cptr + (size_t)rel * strideC), 0, (int)(RPI * strideC), 0x00020000);
                    float rs = 1.f;
                    if (EPI == FE_LINEAR || EPI == FE_DGELU) {
                        if (scaled) rs = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(rs_addr + rel * 4, rs_lane));
                    }
                    if (EPI == FE_PLAIN) {
                        if (FLAG) { cs += v; cq += v * v; }
                    } else if (EPI == FE_LINEAR) {
                        v = (v + b4) * rs;
                        if (FLAG) v += aux[AUX ? j : 0];
                    } else if (EPI == FE_GELU || EPI == FE_QGELU) {
                        v += b4;
                        if (EPI == FE_GELU && p.preact) {
                            const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)(pptr + (size_t)rel * strideC), 0, (int)(RPI * strideC), 0x00020000);
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsP, (int)(active ? offC : 0xFFFFFFF0u), 0, 2 /* nt */);
                        }
                        if (EPI == FE_QGELU || quick) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + expf(-1.702f * v[r]));
                        } else {
                            v = gg_act_f32_v4(v, GG_ACT_GELU);
                        }
                    } else if (EPI == FE_DGELU) {
                        const f32x4 h = aux[AUX ? j : 0];
                        if (quick) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float sg = 1.0f / (1.0f + expf(-1.702f * h[r]));
                                v[r] *= (sg + 1.702f * h[r] * sg * (1.0f - sg)) * rs;
                            }
                        } else {
                            v *= gg_act_grad_f32_v4(h, GG_ACT_GELU) * rs;
                        }
                    } else {      // FE_BNBWD: dz = da * act'(gamma*xhat + beta); column sums of dz and dz*xhat
                        const f32x4 yv = aux[AUX ? j : 0];
                        v *= gg_act_grad_f32_v4(yv * bsc + bsh, p.bn_act);
                        cs += v; cq += v * (yv * brs + bnm);
                    }
                    if (STATS) asm volatile("" : "+v"(cs), "+v"(cq));      // pin the running sums here (left alone, the compiler sinks all 16 row terms to the reduction below and spills them)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsC, (int)(active ? offC : 0xFFFFFFF0u), 0, 2 /* nt */);      // (idle lanes of the 24-slot form: out of range, dropped); nt: result tiles stream through L2, the weight panel stays (fc1 / fc2 of stage 2: -2.7 %)
                }
            }
        }
    };
    const bool flag = EPI == FE_PLAIN ? p.colstats != nullptr : (EPI == FE_LINEAR ? p.residual != nullptr : true);
    if (EPI != FE_PLAIN && EPI != FE_LINEAR) run(std::true_type{});
    else if (flag) run(std::true_type{});
    else run(std::false_type{});
    if (STATS && (EPI == FE_BNBWD || p.colstats)) {      // (the BatchNorm-backward form always has them)
        // per-lane sums (4 columns, this lane's rows) -> LDS [wave][r4][2][64] -> thread (which, column): over the WM waves and the 4 row lanes
        __syncthreads();                                      // every wave is done with its transposition slice
        float* red = smem;
        if (active) {
            *reinterpret_cast<f32x4*>(red + ((wave * RPI + r4) * 2 + 0) * WCOLS + sl * 4) = cs;
            *reinterpret_cast<f32x4*>(red + ((wave * RPI + r4) * 2 + 1) * WCOLS + sl * 4) = cq;
        }
        __syncthreads();
        constexpr int BNC = WN * WCOLS;
        for (int t = threadIdx.x; t < 2 * BNC; t += 256) {
            const int which = t / BNC, col = t % BNC, wn_ = col / WCOLS, c = col % WCOLS;
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w)
#pragma unroll
                for (int q = 0; q < RPI; ++q) sum += red[(((w * WN + wn_) * RPI + q) * 2 + which) * WCOLS + c];
            p.colstats[(int64_t)tm * 2 * p.N + which * p.N + n0 + col] = sum;
        }
    }
}

// PERSISTENT workgroups: the grid is min(tiles, resident workgroups) and a workgroup walks tiles t, t + grid, ...  The operand
// registers that prefetch the next k-tile are idle during a tile's last k-iteration, so they fetch the NEXT tile's first k-tile
// there: a tile's prologue (first-load latency, ~10 % of a K = 384 tile, ~25 % of a K = 96 tile when measured by ablation) and its
// epilogue stores overlap with matrix work instead of adding to it.  Result-independent of the grid size.
template <int BN, int WM, int WN, int EPI, int PRO = 0, int BNC = BN>      // BNC < BN: compute BNC columns of a BN-row B panel (see the ring kernel)
__global__ __launch_bounds__(256, PRO != 0 ? 2 : (BN == 128 ? 3 : 4)) void gemm_nt_f32_kernel(F32GemmParams p) {
    constexpr int BM = 128;
    constexpr int TM = BM / WM / 16, TN = BNC / WN / 16;
    constexpr int LA = BM * 8 / 256, LB = BN * 8 / 256;           // 16-byte chunks per thread per k-tile
    constexpr int WROWS = BM / WM, WCOLS = BNC / WN;
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * FBK];
    __shared__ __attribute__((aligned(16))) float ptab[PRO ? 3 * 1024 : 4];      // PRO: per-k coefficients [3][K], K <= 1024
    float* As = smem;
    float* Bs = smem + BM * FBK;
    const int tiles = p.tilesM * p.tilesN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lg = lane >> 4;
    if (PRO == 1) {
        for (int k = threadIdx.x; k < p.K; k += 256) {
            const float sc = p.a_gamma[k] * p.a_stat[p.K + k];
            ptab[k] = sc; ptab[1024 + k] = p.a_beta[k] - p.a_stat[k] * sc;
        }
        __syncthreads();
    }
    if (PRO == 2) {
        for (int k = threadIdx.x; k < p.K; k += 256) { ptab[k] = p.a_stat[k]; ptab[1024 + k] = p.a_stat[p.K + k]; ptab[2048 + k] = p.a_stat[2 * p.K + k]; }
        __syncthreads();
    }

    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int srow = threadIdx.x >> 3, skc = threadIdx.x & 7;
    unsigned voa[LA], vob[LB];
    int lds_a[LA], lds_b[LB];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        voa[i] = (unsigned)(srow + 32 * i) * (unsigned)p.lda * 4u + skc * 16u;
        lds_a[i] = f32_chunk_off(srow + 32 * i, skc);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        vob[i] = (unsigned)(srow + 32 * i) * (unsigned)p.ldb * 4u + skc * 16u;
        lds_b[i] = f32_chunk_off(srow + 32 * i, skc);
    }
    const int sw = (lr & 2) | ((lr >> 1) & 4);
    const int a_base = (wm * (BM / WM) + lr) * FBK, b_base = (wn * WCOLS + lr) * FBK;
    const int kc0 = ((0 + lg) ^ sw) << 2, kc1 = ((4 + lg) ^ sw) << 2;
    const int nk = (p.K + FBK - 1) / FBK;
    const bool vec_c = (p.ldc & 3) == 0;

    u32x4 ra[LA], rb[LB], ra2[PRO == 2 ? LA : 1];
    // operand loads of k-tile kt of the tile at (tm, tn): raw buffer loads, descriptor = the valid rows of that tile (rows beyond M / N
    // read as zeros through the hardware range check), k offset in the scalar soffset, chunks beyond K pushed out of range
    auto load_tile = [&](int tm_, int tn_, int kt) {
        if (p.debug & 1) return;
        const int m0_ = tm_ * BM, n0_ = tn_ * BNC;
        const unsigned bytesA = (unsigned)min(p.M - m0_, BM) * (unsigned)p.lda * 4u;
        const unsigned bytesB = (unsigned)min(p.N - n0_, BNC) * (unsigned)p.ldb * 4u;
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)m0_ * p.lda), 0, (int)bytesA, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (int64_t)n0_ * p.ldb), 0, (int)bytesB, 0x00020000);
        const int k0 = kt * FBK;
        const bool kin = (k0 + skc * 4) < p.K;                    // K % 4 == 0: a chunk is entirely in or out
        const int so = k0 * 4;
#pragma unroll
        for (int i = 0; i < LA; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(kin ? voa[i] : 0xFFFFFFF0u), so, 0);
        if (PRO == 2) {
            const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 + (int64_t)m0_ * p.lda), 0, (int)bytesA, 0x00020000);
#pragma unroll
            for (int i = 0; i < LA; ++i) ra2[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA2, (int)(kin ? voa[i] : 0xFFFFFFF0u), so, 0);
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(kin ? vob[i] : 0xFFFFFFF0u), so, 0);
    };
    if (p.debug & 1) {
#pragma unroll
        for (int i = 0; i < LA; ++i) ra[i] = (u32x4){1, 2, 3, 4};
#pragma unroll
        for (int i = 0; i < LB; ++i) rb[i] = (u32x4){1, 2, 3, 4};
    }
    int t = blockIdx.x;
    int bid = gg_xcd_remap(t, tiles);
    int tm = bid / p.tilesN, tn = bid % p.tilesN;
    load_tile(tm, tn, 0);
    while (true) {
        const int t_next = t + (int)gridDim.x;
        const bool has_next = t_next < tiles;
        const int bid_n = gg_xcd_remap(has_next ? t_next : t, tiles);
        const int tm_n = bid_n / p.tilesN, tn_n = bid_n % p.tilesN;
        const int m0 = tm * BM, n0 = tn * BNC;
        f32x4 acc[TN][TM];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt) {
            if (PRO) {
                // this thread's chunk covers contraction columns k0 + 4 skc .. + 3 of every staged A row; rows beyond M and chunks beyond K stay
                // zero (they feed the column statistics / must not add the affine's constant term)
                const int kq = kt * FBK + skc * 4;
                const bool kin = kq < p.K;
                const int kc = min(kq, p.K - 4);
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(ptab + kc), t1 = *reinterpret_cast<const f32x4*>(ptab + 1024 + kc);
                const f32x4 t2 = PRO == 2 ? *reinterpret_cast<const f32x4*>(ptab + 2048 + kc) : t0;
#pragma unroll
                for (int i = 0; i < LA; ++i) {
                    const bool rok = kin && (m0 + srow + 32 * i < p.M);
                    f32x4 v = __builtin_bit_cast(f32x4, ra[i]);
                    if (PRO == 1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = act_exact_f(fmaf(v[j], t0[j], t1[j]), p.a_act);
                    } else {
                        v = t0 * v + (t1 * __builtin_bit_cast(f32x4, ra2[i]) + t2);
                    }
                    if (!rok) v = (f32x4){0.f, 0.f, 0.f, 0.f};
                    ra[i] = __builtin_bit_cast(u32x4, v);
                }
            }
            if (!(p.debug & 4)) {
#pragma unroll
                for (int i = 0; i < LA; ++i) *reinterpret_cast<u32x4*>(As + lds_a[i]) = ra[i];
#pragma unroll
                for (int i = 0; i < LB; ++i) *reinterpret_cast<u32x4*>(Bs + lds_b[i]) = rb[i];
                __syncthreads();
            }
            if (kt + 1 < nk) load_tile(tm, tn, kt + 1);
            else if (has_next) load_tile(tm_n, tn_n, 0);              // the next tile's first operands travel under this tile's last MFMAs + epilogue
            const int rounds = (p.K - kt * FBK) > 16 ? 2 : 1;            // skip the all-zero upper half of a K tail
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks < rounds) {
                    const int kc = ks ? kc1 : kc0;
                    f32x4 xf[TM], wf[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const f32x4*>(As + a_base + i * 16 * FBK + kc);
#pragma unroll
                    for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const f32x4*>(Bs + b_base + i * 16 * FBK + kc);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
                            for (int mt = 0; mt < TM; ++mt)
                                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][s], xf[mt][s], acc[nt][mt], 0, 0, 0);
                }
            }
            if (!(p.debug & 4)) __syncthreads();
        }

        gemm_f32_epilogue<BM, BNC, WM, WN, EPI>(p, smem, acc, m0, n0, tm, wm, wn, lr, lg);
        if (!has_next) break;
        t = t_next; tm = tm_n; tn = tn_n;
    }
}

// ---------------------------------------------------------------------------------------------- NT, LDS-DMA ring (default)
// Measured on the register-staged kernel above (GG_GEMM_F32_DEBUG ablations, M = 200 704, N = 1152, K = 384): 119 TFLOP/s as is, 131
// without operand loads, 132 without stores, 136 without either, 144 without LDS staging / barriers, against 154 sustained by the
// bare MFMA loop -- its one-k-tile prefetch distance (1.7 us) is shorter than a loaded HBM miss, and every k-tile pays two barriers
// plus an exposed ds_read.  This kernel removes both: operands travel global -> LDS by LDS-DMA (`buffer_load ... lds`: no
// staging registers, hardware range check = zero fill) into a ring of 4 stages of 16 k (16 KB each), 3 stages ahead of the matrix
// work (counted vmcnt, one raw s_barrier per stage); the fragments of stage s+1 are read while the 64 MFMAs of stage s issue
// (two fragment register sets), so a wave's MFMA stream has no memory wait inside the k-loop.  64 KB LDS -> 2 workgroups per CU.
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int IPW> __device__ __forceinline__ void wait_stages_outstanding(int n) {      // at most n later stages (IPW DMAs each) in flight
    if (n >= 3) wait_vmcnt<3 * IPW>();
    else if (n == 2) wait_vmcnt<2 * IPW>();
    else if (n == 1) wait_vmcnt<IPW>();
    else wait_vmcnt<0>();
}
// BNC < BN: the tile computes BNC columns (N = 96 / 576 ...: no MFMA work on padding) while the LDS image and the DMA pattern stay those of
// the BN-row B panel -- rows >= BNC are outside the buffer resource's range and arrive as zeros without touching memory.
// SB (single fragment buffer, NST = 2, OCC = 4): the small-K form.  With K <= 192 a tile is 6 - 12 stages long and a workgroup spends as long
// waiting for its first operands and draining its epilogue as in the k-loop; what hides that is MORE resident workgroups, not a deeper
// pipeline inside one.  One fragment register set (32 instead of 64 registers: 4 waves per SIMD) and a 2-stage ring (32 KB: 4 workgroups
// per CU); the fragment reads of a stage are exposed to this wave, the three other waves of the SIMD fill the matrix pipe meanwhile.
// PRO (SB form only): the A operand is transformed on its way from LDS to the matrix pipe -- PRO 1: A := a_act(A * sc[k] + sh[k]) (BatchNorm +
// activation of the previous ConvNorm: MBConv conv3 reads conv2's saved pre-BatchNorm output), PRO 2: A := c0[k] A + c1[k] A2 + c2[k] (BatchNorm
// backward's apply step; A2 rides the ring as a third operand panel).  The per-k coefficients live in LDS (K <= PTK = 384: every MBConv /
// PatchMerging shape of the 5M / 11M / 21M models); rows beyond M and chunks beyond K are zeroed AFTER the transform (they feed the column
// statistics / must not add the affine's constant term).  With WN == 1 (the 4 x 1 wave layout used for N <= 96) every A element is
// transformed by exactly one wave.
// REPI (SB form, 64-column wave layouts): every tile of the launch is interior and aligned (checked by the host) and leaves through the row-layout
// epilogue; the general epilogue is not compiled in (the two together exceed the 128 registers of the 4-per-CU form)
// BM_ = 64 (with BN = 64, 2 x 2 waves of 32 x 32): the small-M form -- a launch with fewer 128-row tiles than the chip has CUs (one
// serving panorama: M = 784 / 196 rows in stages 2 / 3) is cut into four times as many workgroups; a tile is MFMA-time-bound on its own CU (128 x 128 x 1536
// = 82 us at one CU's fp32 matrix rate), so what shortens the launch is more CUs, not a better pipeline
template <int BN, int WM, int WN, int EPI, int NST = 4, int OCC = 2, int BNC = BN, bool SB = false, int PRO = 0, bool REPI = false, int BM_ = 128>
__global__ __launch_bounds__(256, OCC) void gemm_nt_f32_ring_kernel(F32GemmParams p) {
    constexpr int BM = BM_, SK = 16, PTK = 384;
    constexpr int TM = BM / WM / 16, TN = BNC / WN / 16;
    constexpr int WCOLS = BNC / WN;
    constexpr int ROWS = BM + BN + (PRO == 2 ? BM : 0), STAGE = ROWS * SK;      // floats per stage
    constexpr int IPW = ROWS / 16 / 4;                            // 1-KiB DMA instructions per wave per stage (16 rows x 64 B each)
    constexpr int JA = BM / 64, JB = JA + BN / 64;                // of a wave's blocks the first JA are A rows, then B rows up to JB, then A2 rows
    static_assert(PRO == 0 || SB, "the A prologues are built on the single-buffer form");
    __shared__ __attribute__((aligned(16))) float smem[NST * STAGE];
    __shared__ __attribute__((aligned(16))) float ptab[PRO ? 3 * PTK : 4];
    const int tiles = p.tilesM * p.tilesN;
    unsigned long long tr0 = 0, tr1 = 0, tr2 = 0;
    unsigned long long mt0 = 0;
    unsigned long long tw_dma = 0, tw_bar = 0, tw_mfma = 0;       // dev trace: cycles wave 0 spent waiting for its DMAs / at the stage barrier (SB: + fragment reads, MFMA issue)
    if (p.trace) { tr0 = wall_clock64(); mt0 = __builtin_readcyclecounter(); }
    int bx = blockIdx.x;
    if constexpr (BM_ == 64) {
        if (p.splits > 1) {          // split-K: this workgroup's share of the contraction, its own result slab
            const int sp = bx / tiles, ks = sp * p.k_per_split;
            bx -= sp * tiles;
            p.A += ks; p.B += ks; p.K = min(p.k_per_split, p.K - ks);
            p.C += (int64_t)sp * p.M * p.ldc;
        }
    }
    const int bid = gg_xcd_remap(bx, tiles);
    const int tm = bid / p.tilesN, tn = bid % p.tilesN;
    const int m0 = tm * BM, n0 = tn * BNC;
    // the wave index as a scalar: everything derived from it (LDS destinations of the DMAs, the wave's rows and columns) stays in SGPRs.  fp32 MFMA
    // and VALU instructions share the SIMD's vector issue (tools/mfma_shadow.hip: every VALU instruction next to an MFMA adds its own 5 - 8
    // cycles, tools/valu_under_mfma.hip: another wave's VALU does not issue at all while MFMAs are pending), so a VALU instruction in the k-loop
    // or the epilogue is matrix time lost, a SALU / LDS / memory instruction is not
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lg = lane >> 4;
    const unsigned bytesA = (unsigned)min(p.M - m0, BM) * (unsigned)p.lda * 4u;
    const unsigned bytesB = (unsigned)min(p.N - n0, BNC) * (unsigned)p.ldb * 4u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)m0 * p.lda), 0, (int)bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (int64_t)n0 * p.ldb), 0, (int)bytesB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA2 = PRO == 2 ? __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 + (int64_t)m0 * p.lda), 0, (int)bytesA, 0x00020000) : rsA;
    if (PRO == 1) {
        for (int k = threadIdx.x; k < p.K; k += 256) {
            const float sc = p.a_gamma[k] * p.a_stat[p.K + k];
            ptab[k] = sc; ptab[PTK + k] = p.a_beta[k] - p.a_stat[k] * sc;
        }
    }
    if (PRO == 2) {
        for (int k = threadIdx.x; k < p.K; k += 256) { ptab[k] = p.a_stat[k]; ptab[PTK + k] = p.a_stat[p.K + k]; ptab[2 * PTK + k] = p.a_stat[2 * p.K + k]; }
    }
    if (PRO) __syncthreads();
    // DMA geometry: block blk = wave + 4 j covers tile rows 16 blk .. 16 blk + 15 (A rows first, then B rows, then A2 rows) x 16 k: lane -> (row lane/4, 16-byte chunk lane%4)
    // 64-byte tile rows (16 floats): the LDS side of a DMA is lane-linear, so the bank permutation is applied on the SOURCE side -- LDS slot (lane & 3) of row lane / 4
    // receives global chunk slot ^ perm(row quad) -- and undone in the fragment reads below.  perm = {0, 2, 3, 1} is what ds_read_b128's 16-lane service groups
    // ({0-3, 12-15, 20-27}, ...) need for 16 rows x one k-chunk; read straight (as rounds 2-5 did) every fragment read was a 2-way bank conflict
    const int drow = lane >> 2, dch = (lane & 3) ^ (p.swz_plain ? 0 : ((0x78 >> (2 * (lane >> 4))) & 3));
    unsigned voff[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
        const int blk = wave + 4 * j;
        const int row = (j < JA ? blk : (j < JB ? blk - BM / 16 : blk - (BM + BN) / 16)) * 16 + drow;
        voff[j] = (unsigned)row * (unsigned)((j < JA || j >= JB) ? p.lda : p.ldb) * 4u + dch * 16u;
    }
    auto issue_stage_at = [&](int st, float* base) {
        const int k0 = st * SK;
        if (k0 + SK <= p.K) {                                     // (uniform) every chunk of the stage is inside K: no per-lane masking
#pragma unroll
            for (int j = 0; j < IPW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(j < JA ? rsA : (j < JB ? rsB : rsA2), (__attribute__((address_space(3))) void*)(base + (wave + 4 * j) * 256), 16,
                                                         (int)voff[j], k0 * 4, 0, 0);
        } else {
            const bool kin = (k0 + dch * 4) < p.K;                // K % 4 == 0: a chunk is entirely in or out
#pragma unroll
            for (int j = 0; j < IPW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(j < JA ? rsA : (j < JB ? rsB : rsA2), (__attribute__((address_space(3))) void*)(base + (wave + 4 * j) * 256), 16,
                                                         (int)(kin ? voff[j] : 0xFFFFFFF0u), k0 * 4, 0, 0);
        }
    };
    auto issue_stage = [&](int st) { issue_stage_at(st, smem + (st % NST) * STAGE); };
    const int fch = lg ^ (p.swz_plain ? 0 : ((0x78 >> (2 * ((lr >> 2) & 3))) & 3));      // position of this lane's k-chunk lg in its row
    const int a_off = (wm * (BM / WM) + lr) * SK + fch * 4, b_off = BM * SK + (wn * WCOLS + lr) * SK + fch * 4;
    auto frag_read_at = [&](const float* base, f32x4 (&xf)[TM], f32x4 (&wf)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) xf[i] = *reinterpret_cast<const f32x4*>(base + a_off + i * 16 * SK);
#pragma unroll
        for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const f32x4*>(base + b_off + i * 16 * SK);
    };
    auto frag_read = [&](int st, f32x4 (&xf)[TM], f32x4 (&wf)[TN]) { frag_read_at(smem + (st % NST) * STAGE, xf, wf); };
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = (p.K + SK - 1) / SK;
    if (SB) {
        issue_stage(0);
        wait_vmcnt<0>();
    } else {
#pragma unroll
        for (int st = 0; st < NST; ++st)
            if (st < nk) issue_stage(st);
        wait_stages_outstanding<IPW>(min(nk, NST) - 1);         // stage 0 has landed (this wave's part) ...
    }
    __builtin_amdgcn_s_barrier();                                 // ... and everybody else's
    if (SB) {
        f32x4 xs[TM], ws[TN];
        if (p.trace) tr1 = wall_clock64();
        unsigned long long c0 = 0, c1 = 0, c2 = 0;
        // one k-stage; PAR = the ring buffer it lives in (NST == 2), a compile-time constant so that the LDS addresses are immediates
        auto stage = [&](int s, auto par) {
            constexpr int PAR = decltype(par)::value;
            float* const cur = smem + PAR * STAGE;
            float* const nxt = smem + (1 - PAR) * STAGE;
            if (p.trace) c0 = __builtin_readcyclecounter();
            if (s > 0) {
                wait_vmcnt<0>();                                  // this wave's DMAs of stage s have landed ...
                __builtin_amdgcn_s_barrier();                     // ... everybody's have, and every wave has consumed stage s - 1 (its MFMAs needed the reads)
            }
            if (p.trace) c1 = __builtin_readcyclecounter();
            if (s + 1 < nk) issue_stage_at(s + 1, nxt);           // into the buffer stage s - 1 occupied
            frag_read_at(cur, xs, ws);
            if (p.trace) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); c2 = __builtin_readcyclecounter(); tw_dma += c1 - c0; tw_bar += c2 - c1; }
            if (PRO) {
                const int kq = s * SK + lg * 4;                   // this lane's fragments hold contraction columns kq .. kq + 3 of 16 rows each
                const bool kin = kq < p.K;
                const int kc = min(kq, p.K - 4);
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(ptab + kc), c1 = *reinterpret_cast<const f32x4*>(ptab + PTK + kc);
                const f32x4 c2 = PRO == 2 ? *reinterpret_cast<const f32x4*>(ptab + 2 * PTK + kc) : c0;
                const float* base2 = cur + (BM + BN) * SK;
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
                    const bool rok = kin && (m0 + wm * (BM / WM) + mt * 16 + lr < p.M);
                    f32x4 v = xs[mt];
                    if (PRO == 1) {
                        v = gg_act_f32_v4(v * c0 + c1, p.a_act);
                    } else {
                        const f32x4 x2 = *reinterpret_cast<const f32x4*>(base2 + a_off + mt * 16 * SK);
                        v = c0 * v + (c1 * x2 + c2);
                    }
                    xs[mt] = rok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int nt = 0; nt < TN; ++nt)
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[nt][q], xs[mt][q], acc[nt][mt], 0, 0, 0);
            if (p.trace) tw_mfma += __builtin_readcyclecounter() - c2;
        };
        static_assert(!SB || NST == 2, "the single-buffer form walks a 2-stage ring");
        for (int s = 0; s < nk; s += 2) {
            stage(s, std::integral_constant<int, 0>{});
            if (s + 1 < nk) stage(s + 1, std::integral_constant<int, 1>{});
        }
        __builtin_amdgcn_s_barrier();
        if (p.trace) tr2 = wall_clock64();
        // second-tensor fetch as whole lines (two n-tiles per group, M in two halves) where the registers allow it: the GELU' and residual epilogues
        // of the 128 x 128 tile; the BatchNorm-backward epilogue keeps one n-tile per group (its per-column coefficients need the registers)
        constexpr bool PAIR_AUX = (EPI == FE_DGELU || EPI == FE_LINEAR) && TM == 4 && PRO == 0;
        if constexpr (REPI) {
            gemm_f32_epilogue_rows<BM, WM, WN, TM, TN, EPI, NST * STAGE>(p, smem, acc, m0, n0, tm, wave, lane);
        } else
        gemm_f32_epilogue<BM, BNC, WM, WN, EPI, (PAIR_AUX ? 2 : 1), (PAIR_AUX ? 2 : 1)>(p, smem, acc, m0, n0, tm, wm, wn, lr, lg);
        if (p.trace && threadIdx.x == 0) {        // dev trace (gg_gemm_f32_set_trace), same record as the double-buffered form; no per-stage wait split
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            const unsigned long long issued = wall_clock64();     // the epilogue's instructions are issued; what follows is the store drain
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned long long* t = p.trace + (size_t)blockIdx.x * 8;
            t[0] = hw | ((unsigned long long)(xcc & 0xF) << 32); t[1] = __builtin_readcyclecounter() - mt0; t[2] = tr0; t[3] = tr1; t[4] = tr2; t[5] = wall_clock64();
            // wave 0 inside the k-loop (shader-clock cycles): [6] = waiting for DMA + barrier | (LDS fragment reads << 32), [7] = MFMA issue | ((issued - tr2) << 32, 10 ns ticks)
            t[6] = (tw_dma & 0xFFFFFFFFull) | (tw_bar << 32); t[7] = (tw_mfma & 0xFFFFFFFFull) | ((issued - tr2) << 32);
        }
        return;
    }
    f32x4 xa[TM], wa[TN], xb[TM], wb[TN];
    if (p.trace) tr1 = wall_clock64();
    frag_read(0, xa, wa);
    auto step = [&](int s, f32x4 (&xc)[TM], f32x4 (&wc)[TN], f32x4 (&xn)[TM], f32x4 (&wn_)[TN]) {
        const bool more = s + 1 < nk;
        unsigned long long c0 = 0, c1 = 0;
        if (p.trace) c0 = __builtin_readcyclecounter();
        if (more) wait_stages_outstanding<IPW>(min(nk - 1, s + NST - 1) - (s + 1));     // this wave's DMAs of stage s+1 have landed
        if (p.trace) c1 = __builtin_readcyclecounter();
        __builtin_amdgcn_s_waitcnt(0xC07F);                       // lgkmcnt(0): ... and its fragment reads of stage s (whose buffer is refilled below)
        __builtin_amdgcn_s_barrier();
        if (p.trace) { tw_dma += c1 - c0; tw_bar += __builtin_readcyclecounter() - c1; }
        if (s + NST < nk) issue_stage(s + NST);                   // into the buffer stage s occupied
        if (more) frag_read(s + 1, xn, wn_);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int nt = 0; nt < TN; ++nt)
#pragma unroll
                for (int mt = 0; mt < TM; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[nt][q], xc[mt][q], acc[nt][mt], 0, 0, 0);
    };
    for (int s = 0; s < nk; s += 2) {
        step(s, xa, wa, xb, wb);
        if (s + 1 < nk) step(s + 1, xb, wb, xa, wa);
    }
    __builtin_amdgcn_s_barrier();                                 // every wave is done with the ring: the epilogue may reuse it
    if (p.trace) tr2 = wall_clock64();
    gemm_f32_epilogue<BM, BNC, WM, WN, EPI>(p, smem, acc, m0, n0, tm, wm, wn, lr, lg);
    if (p.trace && threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the epilogue's stores have been acknowledged
        unsigned long long* t = p.trace + (size_t)blockIdx.x * 8;
        t[0] = hw | ((unsigned long long)(xcc & 0xF) << 32); t[1] = __builtin_readcyclecounter() - mt0; t[2] = tr0; t[3] = tr1; t[4] = tr2; t[5] = wall_clock64(); t[6] = tw_dma; t[7] = tw_bar;
    }
}

// ------------------------------------------------------------------------------------------- TN (weight gradients)
// part[split][N][K] = sum over the split's rows m of dY[m][n] * X[m][k].  Both operands are row-major over the reduction index, and the
// f32 MFMA takes ONE float per lane per operand (A[i = n][k = m], B[k = m][j = k]), so fragments are plain ds_read_b32 of the
// row-major LDS tiles: lane (lr, lg) reads T[4*step + lg][col + lr] -- no transposed copies, no transposing reads.  Row stride
// 128 + 16 floats: the two m-rows a 32-lane half touches land on disjoint bank halves.
struct F32TnParams {
    const float* dY; int64_t ldy; const float* X; int64_t ldx;
    int M, N, K;
    const float* rowscale; int rows_per_scale;
    float* part;
    int tilesN, tilesK, m_per_split;
    const float* Y2; const float* coef;      // BN form: dY := coef0*dY + coef1*Y2 + coef2 per column while staging (Y2 laid out like dY)
};
// SMALL (N <= 64 and K <= 64, e.g. patch_embed.conv1: 48 x 32 over 12.8 M rows): one 16x16-fragment block covers the whole result, so the four
// waves split the ROWS of every 32-row step instead of the (n, k) plane and each writes its own slab (slab = 4 * block + wave; the
// caller's slab reduce sums them like any other split).  Fragments entirely beyond N / K are skipped in both forms.
// BN: the left operand is BatchNorm backward's apply step dy = c0*dz + c1*y + c2 (gg_bn_bwd_finalize's coef) of the two tensors (dz, y), formed
// per 16-byte chunk between the global load and the LDS store -- the weight gradient of a ConvNorm straight from (dz, y), no dy tensor.  Rows
// beyond the split read as zero in X, so the c2 they would contribute to dy meets a zero and drops out.
template <bool SMALL, bool BN = false, bool FE = false>
__global__ __launch_bounds__(256, 3) void gemm_tn_f32_kernel(F32TnParams p) {
    constexpr int TB = 128, MS = 32, RS = TB + 16;
    constexpr int LS = MS * (TB / 4) / 256;                 // 16-byte chunks per thread per operand per step (= 4)
    __shared__ __attribute__((aligned(16))) float Ys[MS * RS];
    __shared__ __attribute__((aligned(16))) float Xs[MS * RS];
    const int ntile = p.tilesN * p.tilesK;
    const int lb = gg_xcd_remap(blockIdx.x, gridDim.x);
    const int split_id = lb / ntile, tile_id = lb - split_id * ntile;
    const int tn = tile_id / p.tilesK, tk = tile_id % p.tilesK;
    const int n0 = tn * TB, k0 = tk * TB;
    const int mbeg = split_id * p.m_per_split, mend = min(p.M, mbeg + p.m_per_split);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // scalar: the fragment limits below stay in SGPRs
    const int wn = SMALL ? 0 : wave >> 1, wk = SMALL ? 0 : wave & 1;
    const int lr = lane & 15, lg = lane >> 4;
    const int nt_lim = min(4, max(0, (p.N - n0 - wn * 64 + 15) >> 4)), kt_lim = min(4, max(0, (p.K - k0 - wk * 64 + 15) >> 4));   // wave-uniform
    const int srow = threadIdx.x >> 5, sch = threadIdx.x & 31;          // 8 rows x 32 chunks per pass, LS passes
    const bool yok = (n0 + sch * 4) < p.N, xok = (k0 + sch * 4) < p.K;
    const int nrows = max(mend - mbeg, 0);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dY + (int64_t)mbeg * p.ldy), 0, (int)((unsigned)nrows * (unsigned)p.ldy * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(p.X + (int64_t)mbeg * p.ldx), 0, (int)((unsigned)nrows * (unsigned)p.ldx * 4u), 0x00020000);
    unsigned voy[LS], vox[LS];
#pragma unroll
    for (int i = 0; i < LS; ++i) {
        voy[i] = ((unsigned)(srow + 8 * i) * (unsigned)p.ldy + (unsigned)(n0 + sch * 4)) * 4u;
        vox[i] = ((unsigned)(srow + 8 * i) * (unsigned)p.ldx + (unsigned)(k0 + sch * 4)) * 4u;
    }
    const unsigned stepY = (unsigned)MS * (unsigned)p.ldy * 4u, stepX = (unsigned)MS * (unsigned)p.ldx * 4u;
    f32x4 ry[LS], rx[LS], ry2[BN ? LS : 1];
    float rsc[LS];
    f32x4 cf0 = {0.f, 0.f, 0.f, 0.f}, cf1 = cf0, cf2 = cf0;
    __amdgpu_buffer_rsrc_t rsY2 = rsY;
    if (BN) {
        rsY2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Y2 + (int64_t)mbeg * p.ldy), 0, (int)((unsigned)nrows * (unsigned)p.ldy * 4u), 0x00020000);
        if (yok) {
            cf0 = *reinterpret_cast<const f32x4*>(p.coef + n0 + sch * 4);
            cf1 = *reinterpret_cast<const f32x4*>(p.coef + p.N + n0 + sch * 4);
            cf2 = *reinterpret_cast<const f32x4*>(p.coef + 2 * p.N + n0 + sch * 4);
        }
    }
    auto load_step = [&](int m0) {
        const unsigned st = (unsigned)(m0 - mbeg) / MS;
#pragma unroll
        for (int i = 0; i < LS; ++i) {
            ry[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsY, (int)(yok ? voy[i] + st * stepY : 0xFFFFFFF0u), 0, 0));
            if (BN) ry2[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsY2, (int)(yok ? voy[i] + st * stepY : 0xFFFFFFF0u), 0, 0));
            rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(xok ? vox[i] + st * stepX : 0xFFFFFFF0u), 0, 0));
        }
        if (p.rowscale) {
#pragma unroll
            for (int i = 0; i < LS; ++i) rsc[i] = p.rowscale[min(m0 + srow + 8 * i, p.M - 1) / p.rows_per_scale];
        }
    };
    f32x4 acc[4][4];     // [k tile][n tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (mbeg < mend) load_step(mbeg);
    // FE (host: N and K multiples of 64): every wave's 64 x 64 tile is either entirely live or entirely outside the matrix, so the MFMAs of a live wave
    // run back to back; the general form wraps each of the 16 MFMAs of a k-step in its own exec-mask branch (16 basic blocks per step)
    const bool live = nt_lim > 0 && kt_lim > 0;
    for (int m0 = mbeg; m0 < mend; m0 += MS) {
        if (BN) {
#pragma unroll
            for (int i = 0; i < LS; ++i) ry[i] = cf0 * ry[i] + (cf1 * ry2[i] + cf2);
        }
        if (p.rowscale) {
#pragma unroll
            for (int i = 0; i < LS; ++i) ry[i] *= rsc[i];
        }
#pragma unroll
        for (int i = 0; i < LS; ++i) {
            *reinterpret_cast<f32x4*>(Ys + (srow + 8 * i) * RS + sch * 4) = ry[i];
            *reinterpret_cast<f32x4*>(Xs + (srow + 8 * i) * RS + sch * 4) = rx[i];
        }
        __syncthreads();
        if (m0 + MS < mend) load_step(m0 + MS);
        const int steps = min(MS, mend - m0 + 3) / 4;              // rows beyond mend are zero anyway; skip whole empty steps
        const int ss0 = SMALL ? 2 * wave : 0, ss1 = (FE && !live) ? 0 : (SMALL ? min(steps, 2 * wave + 2) : steps);
#pragma unroll 2
        for (int ss = ss0; ss < ss1; ++ss) {
            float yf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                yf[i] = Ys[(4 * ss + lg) * RS + wn * 64 + i * 16 + lr];
                xf[i] = Xs[(4 * ss + lg) * RS + wk * 64 + i * 16 + lr];
            }
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    if (FE || (kt < kt_lim && nt < nt_lim))
                        acc[kt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[kt], yf[nt], acc[kt][nt], 0, 0, 0);   // rows k, cols n
        }
        __syncthreads();
    }
    // lane holds D[k = .. + kt*16 + 4lg + r][n = .. + nt*16 + lr]: its four values are consecutive along a row of part[n][k] -- one 16-byte store per
    // fragment (K % 4 == 0: a chunk is entirely inside or outside).  With the operands the other way round a lane held four ROWS of one column and the
    // slab left in 4-byte stores: the head's weight gradient (N = 12647, M = 256 rows to reduce) spent 0.45 of its 0.5 ms storing
    float* out = p.part + (int64_t)(SMALL ? split_id * 4 + wave : split_id) * p.N * p.K;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int k = k0 + wk * 64 + kt * 16 + lg * 4;
            const int n = n0 + wn * 64 + nt * 16 + lr;
            if (n < p.N && k < p.K) *reinterpret_cast<f32x4*>(out + (int64_t)n * p.K + k) = acc[kt][nt];
        }
}

// column sums of an f32 [M, C] matrix (bias gradients): thread = (4-column group of 64, row lane of 4)
__global__ __launch_bounds__(256) void colsum_partial_f32_kernel(const float* __restrict__ x, int64_t ld, int M, int C,
                                                                 const float* rowscale, int rows_per_scale, float* __restrict__ part,
                                                                 int rows_per_block) {
    __shared__ float red[4][256];
    const int cg = threadIdx.x & 63, pp = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 256 + cg * 4;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (c0 < C) {
#pragma unroll 4
        for (int r = r0 + pp; r < r1; r += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (int64_t)r * ld + c0);
            const float rs = rowscale ? rowscale[r / rows_per_scale] : 1.f;
            s += v * rs;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[pp][cg * 4 + j] = s[j];
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) part[(int64_t)blockIdx.y * C + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ void colsum_final_f32_kernel(const float* __restrict__ part, int nparts, int C, float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int i = 0; i < nparts; ++i) s += (double)part[(int64_t)i * C + c];
    out[c] = accumulate ? out[c] + (float)s : (float)s;
}

}  // namespace

// ------------------------------------------------------------------------------------------- host
static unsigned long long* g_f32_trace = nullptr;
// dev: per-workgroup timeline of the ring kernel (tools/trace_gemm_f32.py); buf = 8 x uint64 per tile or NULL to switch it off
extern "C" int gg_gemm_f32_set_trace(void* buf) { g_f32_trace = (unsigned long long*)buf; return 0; }
// split-K of the small-M form: C[m][n] = (sum over the slabs in slab order + bias) * rowscale + residual -- the linear epilogue's arithmetic after a
// deterministic reduction (no atomics: a step stays repeatable bit for bit)
__global__ __launch_bounds__(256) void splitk_nt_reduce_f32_kernel(const float* __restrict__ part, int splits, int M, int N, float* __restrict__ C, int64_t ldc,
                                                                   const float* __restrict__ bias, const float* __restrict__ rowscale, int rows_per_scale,
                                                                   const float* __restrict__ residual, int64_t ldr) {
    const int n4 = N >> 2;
    const int64_t total = (int64_t)M * n4, slab = (int64_t)M * N;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4), n = (int)(i - (int64_t)m * n4) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(part + (int64_t)m * N + n);
        for (int sidx = 1; sidx < splits; ++sidx) v += *reinterpret_cast<const f32x4*>(part + sidx * slab + (int64_t)m * N + n);
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
        if (rowscale) v *= rowscale[m / rows_per_scale];
        if (residual) v += *reinterpret_cast<const f32x4*>(residual + (int64_t)m * ldr + n);
        *reinterpret_cast<f32x4*>(C + (int64_t)m * ldc + n) = v;
    }
}
// slabs of the split-K form: one lazily allocated 16 MiB buffer per (device, stream), at most 8 of them (least recently used released), all released by gg_graph_clear.  Launches of one stream run in order, so they can share their slabs; two
// streams (or threads) issuing qualifying GEMMs concurrently get different buffers.  Under stream capture nothing can be allocated and the launch may be replayed
// on any stream next to anything: the graph cache (graph.cpp) hands every captured graph slabs of its OWN (gg_gemm_f32_capture_scratch, allocated before the
// capture starts when the key's eager run used the split form, freed with the graph), so a replay computes exactly what the eager call did; a capture that
// was given none runs unsplit.
constexpr int64_t kSplitScratchFloats = (int64_t)4 << 20;
static std::mutex g_split_mu;
static std::map<hipStream_t, float*> g_capture_slabs;
struct StreamSlab { float* buf; uint64_t tick; };
static std::map<std::pair<int, hipStream_t>, StreamSlab> g_stream_slabs;
static uint64_t g_slab_tick = 0;
constexpr size_t kMaxStreamSlabs = 8;
static thread_local long tl_splitk_uses = 0;
long gg_gemm_f32_splitk_uses() { return tl_splitk_uses; }                       // (graph.h) split-form launches issued by this thread so far
size_t gg_gemm_f32_splitk_bytes() { return (size_t)kSplitScratchFloats * sizeof(float); }
void gg_gemm_f32_capture_scratch(hipStream_t cap, float* buf) {                 // buf == nullptr: forget the stream
    std::lock_guard<std::mutex> lk(g_split_mu);
    if (buf) g_capture_slabs[cap] = buf; else g_capture_slabs.erase(cap);
}
static float* splitk_scratch(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lk(g_split_mu);
    if (cs != hipStreamCaptureStatusNone) {
        auto it = g_capture_slabs.find(st);
        return it == g_capture_slabs.end() ? nullptr : it->second;
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    auto it = g_stream_slabs.find({dev, st});
    if (it != g_stream_slabs.end()) { it->second.tick = ++g_slab_tick; return it->second.buf; }
    // bounded: at most kMaxStreamSlabs (device, stream) entries; the least recently used one is released (hipFree waits for the device, so work that still reads the
    // slab has finished) -- a process that creates a stream per request does not grow by 16 MiB per stream handle
    if (g_stream_slabs.size() >= kMaxStreamSlabs) {
        auto lru = g_stream_slabs.begin();
        for (auto j = g_stream_slabs.begin(); j != g_stream_slabs.end(); ++j) if (j->second.tick < lru->second.tick) lru = j;
        int cur = dev;
        if (lru->first.first != cur) (void)hipSetDevice(lru->first.first);
        if (lru->second.buf) (void)hipFree(lru->second.buf);
        if (lru->first.first != cur) (void)hipSetDevice(cur);
        g_stream_slabs.erase(lru);
    }
    float* b = nullptr;
    if (hipMalloc((void**)&b, kSplitScratchFloats * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); b = nullptr; }
    g_stream_slabs[{dev, st}] = StreamSlab{b, ++g_slab_tick};
    return b;
}
// teardown (gg_graph_clear): every per-stream slab is released; the next qualifying GEMM allocates again
void gg_gemm_f32_release_scratch() {
    std::lock_guard<std::mutex> lk(g_split_mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto& kv : g_stream_slabs) {
        if (!kv.second.buf) continue;
        (void)hipSetDevice(kv.first.first);
        (void)hipFree(kv.second.buf);
    }
    (void)hipSetDevice(cur);
    g_stream_slabs.clear();
}

extern "C" int gg_gemm_nt_f32(const GgGemmArgs* a, void* stream) {
    GG_CHECK(a && a->A && a->B && a->C, "gg_gemm_nt_f32: null operand");
    GG_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "gg_gemm_nt_f32: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
    GG_CHECK((a->K & 3) == 0 && (a->lda & 3) == 0 && (a->ldb & 3) == 0,
             "gg_gemm_nt_f32: K, lda, ldb must be multiples of 4 (16-byte rows): K=%d lda=%lld ldb=%lld", a->K, (long long)a->lda, (long long)a->ldb);
    GG_CHECK(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0 && ((uintptr_t)a->C & 15) == 0, "gg_gemm_nt_f32: A/B/C must be 16-byte aligned");
    GG_CHECK(a->lda >= a->K && a->ldb >= a->K && a->ldc >= a->N, "gg_gemm_nt_f32: leading dimension too small");
    GG_CHECK(a->lda * 512 < 0xFFFFFF00LL && a->ldb * 512 < 0xFFFFFF00LL && (int64_t)a->K * 4 < 0x7FFFFFFFLL,
             "gg_gemm_nt_f32: leading dimension too large for 32-bit tile offsets (ld < 8.3M elements)");
    GG_CHECK(a->split_k <= 1, "gg_gemm_nt_f32: no split-K form");
    if (a->bn_y) {
        GG_CHECK(a->bn_stat && a->bn_gamma && a->bn_beta && a->colstats && ((uintptr_t)a->bn_y & 15) == 0,
                 "gg_gemm_nt_f32: the BatchNorm-backward epilogue needs stat, gamma, beta, colstats");
        GG_CHECK(!(a->bias || a->act || a->rowscale || a->residual || a->dact_preact || a->preact), "gg_gemm_nt_f32: the BatchNorm-backward epilogue excludes every other epilogue option");
    }
    if (a->a_bn_stat) {        // A prologues (register-staged kernel): BN+act of one source, or the affine of two sources (A2 != NULL)
        GG_CHECK(a->K <= 1024 && (a->A2 || (a->a_bn_gamma && a->a_bn_beta)), "gg_gemm_nt_f32: the A prologue needs K <= 1024 and gamma/beta (or A2 + coef)");
        GG_CHECK(!a->A2 || ((uintptr_t)a->A2 & 15) == 0, "gg_gemm_nt_f32: A2 must be 16-byte aligned");
        if (a->A2) GG_CHECK(!(a->act || a->dact_preact || a->preact || a->colstats || a->bn_y), "gg_gemm_nt_f32: the two-source prologue is built with the linear epilogue");
        else GG_CHECK(!(a->bias || a->act || a->rowscale || a->residual || a->dact_preact || a->preact || a->bn_y), "gg_gemm_nt_f32: the BatchNorm prologue is built with the plain (+ column statistics) epilogue");
    } else GG_CHECK(!a->A2, "gg_gemm_nt_f32: A2 without coefficients");
    if (a->rowscale) GG_CHECK(a->rows_per_scale > 0, "gg_gemm_nt_f32: rows_per_scale must be > 0");
    GG_CHECK(!a->dact_preact || a->dact == GG_ACT_GELU || a->dact == GG_ACT_QUICK_GELU, "gg_gemm_nt_f32: dact must be GELU or QuickGELU");
    GG_CHECK(!(a->dact_preact && (a->bias || a->act || a->residual || a->preact)), "gg_gemm_nt_f32: dact excludes bias/act/residual/preact");
    GG_CHECK(!(a->act && (a->rowscale || a->residual)), "gg_gemm_nt_f32: an activation epilogue excludes rowscale/residual");
    GG_CHECK(!a->preact || a->act == GG_ACT_GELU || a->act == GG_ACT_QUICK_GELU, "gg_gemm_nt_f32: preact is only available with an activation epilogue");
    GG_CHECK(!a->colstats || !(a->bias || a->act || a->rowscale || a->residual || a->dact_preact), "gg_gemm_nt_f32: colstats needs the plain or BatchNorm-backward epilogue");
    GG_CHECK(!a->preact || ((uintptr_t)a->preact & 15) == 0, "gg_gemm_nt_f32: preact must be 16-byte aligned");
    GG_CHECK(!a->residual || ((uintptr_t)a->residual & 15) == 0, "gg_gemm_nt_f32: residual must be 16-byte aligned");
    GG_CHECK(!a->dact_preact || ((uintptr_t)a->dact_preact & 15) == 0, "gg_gemm_nt_f32: dact_preact must be 16-byte aligned");
    F32GemmParams p;
    p.A = (const float*)a->A; p.lda = a->lda; p.B = (const float*)a->B; p.ldb = a->ldb; p.C = (float*)a->C; p.ldc = a->ldc;
    p.M = a->M; p.N = a->N; p.K = a->K; p.bias = a->bias; p.preact = (float*)a->preact;
    p.rowscale = a->rowscale; p.rows_per_scale = a->rows_per_scale; p.residual = (const float*)a->residual; p.ldr = a->ldr;
    p.dact_preact = (const float*)a->dact_preact; p.colstats = a->colstats;
    p.quick = (a->dact_preact ? a->dact : a->act) == GG_ACT_QUICK_GELU;
    static const char* swz_env = gg_dev_env("GG_GEMM_F32_SWZ");
    p.swz_plain = swz_env && atoi(swz_env) == 0;
    p.bn_y = (const float*)a->bn_y; p.bn_stat = a->bn_stat; p.bn_gamma = a->bn_gamma; p.bn_beta = a->bn_beta; p.bn_act = a->bn_act;
    p.A2 = (const float*)a->A2; p.a_stat = a->a_bn_stat; p.a_gamma = a->a_bn_gamma; p.a_beta = a->a_bn_beta; p.a_act = a->a_bn_act;
    const int rem = a->N % 128;
    bool narrow = a->N <= 64 || (rem != 0 && rem <= 64);
    // 96-column tiles (ring kernel only) when they cover N with less padding than both 128 and 64 would (N = 96, 288, ...: +14 % at N = 96;
    // at equal padding the 64-wide tile's 3 workgroups per CU win, N = 576: 126 vs 118 TFLOP/s)
    static const char* w96_env = gg_dev_env("GG_GEMM_F32_NO_W96");
    const int64_t pad96 = gg_cdiv(a->N, 96) * 96, pad128 = gg_cdiv(a->N, 128) * 128, pad64 = gg_cdiv(a->N, 64) * 64;
    const bool wide96 = !w96_env && a->N > 64 && pad96 < pad128 && pad96 < pad64;
    if (wide96) narrow = false;
    const int bn = wide96 ? 96 : (narrow ? 64 : 128);
    p.tilesM = (int)gg_cdiv(a->M, 128); p.tilesN = (int)gg_cdiv(a->N, bn);
    // small-M form (64 x 64 tiles): launches that would leave CUs idle; not for the forms that write per-128-row column statistics
    static const char* small_env = gg_dev_env("GG_GEMM_F32_NO_SMALL");
    static const char* small_max_env = gg_dev_env("GG_GEMM_F32_SMALL_MAX");
    static const int small_max = small_max_env ? atoi(small_max_env) : 256;      // fewer 128-row tiles than CUs
    const bool small = !small_env && !a->a_bn_stat && !a->colstats && !a->bn_y && (int64_t)p.tilesM * p.tilesN <= small_max;
    if (small) { p.tilesM = (int)gg_cdiv(a->M, 64); p.tilesN = (int)gg_cdiv(a->N, 64); }
    // split-K on top: few tiles and a long contraction (the head's data gradient: M = batch, N = 576, K = 12648 -- 36 tiles walking 790 k-stages each;
    // fc2 of stage 3 at one panorama) -- plain / linear epilogues only, applied by the reduction
    static const char* nosplit_env = getenv("GG_GEMM_SPLITK") && atoi(getenv("GG_GEMM_SPLITK")) == 0 ? "1" : gg_dev_env("GG_GEMM_F32_NO_SPLITK");
    float* slabs = nullptr;
    int nsplit = 1;
    if (small && !nosplit_env && a->K >= 1024 && (int64_t)p.tilesM * p.tilesN <= 64 && !(a->act || a->preact || a->dact_preact) && (a->N & 3) == 0 && (a->ldc & 3) == 0 &&
        (!a->residual || (a->ldr & 3) == 0) && (!a->bias || ((uintptr_t)a->bias & 15) == 0)) {
        const int tiles64 = p.tilesM * p.tilesN;
        int want = std::min<int64_t>(std::max(1, 256 / tiles64), a->K / 256);
        want = (int)std::min<int64_t>(want, kSplitScratchFloats / ((int64_t)a->M * a->N));
        if (want > 1) {
            const int kps = (int)gg_align(gg_cdiv(a->K, want), 16);
            nsplit = (int)gg_cdiv(a->K, kps);
            if (nsplit > 1 && (slabs = splitk_scratch((hipStream_t)stream)) != nullptr) { p.splits = nsplit; p.k_per_split = kps; ++tl_splitk_uses; }
            else nsplit = 1;
        }
    }
    static const char* dbg = gg_dev_env("GG_GEMM_F32_DEBUG");
    p.debug = dbg ? atoi(dbg) : 0;
    p.trace = g_f32_trace;
    const double mn = (double)a->M * a->N;
    GG_PROF(GG_CAT_GEMM, 2.0 * a->M * (double)a->N * a->K,
            4.0 * ((double)a->M * a->K * (a->A2 ? 2 : 1) + (double)a->N * a->K + mn) +
                4.0 * mn * ((a->preact != nullptr) + (a->residual != nullptr) + (a->dact_preact != nullptr) + (a->bn_y != nullptr)),
            stream);
    int epi;
    if (a->bn_y) epi = FE_BNBWD;
    else if (a->dact_preact) epi = FE_DGELU;
    else if (a->act == GG_ACT_GELU || (a->act == GG_ACT_QUICK_GELU && a->preact)) epi = FE_GELU;
    else if (a->act == GG_ACT_QUICK_GELU) epi = FE_QGELU;
    else if (a->bias || a->rowscale || a->residual) epi = FE_LINEAR;
    else epi = FE_PLAIN;
    // default: the LDS-DMA ring kernel, one workgroup per tile.  GG_GEMM_F32_RING=0: the register-staged kernel with persistent
    // workgroups (at most the resident count, each walks tiles t, t + grid, ...) -- kept for A/B timing and the DEBUG ablations
    static const char* ring_env = gg_dev_env("GG_GEMM_F32_RING");
    static const char* np_env = gg_dev_env("GG_GEMM_F32_NO_PERSIST");
    const bool ring = ((!(ring_env && ring_env[0] == '0') && (p.debug & 5) == 0) || wide96) && !a->a_bn_stat;
    const int resident = 256 * (a->a_bn_stat ? 2 : (narrow ? 4 : 3));      // workgroups the persistent variants keep resident (launch bounds)
    dim3 grid((ring || np_env || p.tilesM * p.tilesN <= resident) ? p.tilesM * p.tilesN : resident);
    hipStream_t st = (hipStream_t)stream;
    // the single-fragment-buffer form (4 workgroups per CU; 6 for the 128 x 64 tile) is the default for every K: +2...+7 % on the model's shapes
    // against the double-buffered 3-per-CU form (tools/ab_gemm_sb.sh).  GG_GEMM_F32_SB=<K threshold> (0: off) for A/B runs
    static const char* sb_env = gg_dev_env("GG_GEMM_F32_SB");
    const int sb_k = sb_env ? atoi(sb_env) : (1 << 30);
    const bool sb = ring && !wide96 && a->K <= sb_k;
    // row-layout epilogue (gemm_f32_epilogue_rows): every tile interior, 16-byte rows everywhere, at most three DropPath scales per wave.
    // GG_GEMM_F32_ROWS_EPI=0: the general epilogue (A/B, parity of the two forms)
    static const char* rows_env = gg_dev_env("GG_GEMM_F32_ROWS_EPI");
    const float* xsrc = a->bn_y ? (const float*)a->bn_y : (a->dact_preact ? (const float*)a->dact_preact : (const float*)a->residual);
    const int64_t xld = (a->bn_y || a->dact_preact) ? a->ldc : a->ldr;
    const bool rows_ok = !(rows_env && rows_env[0] == '0') && p.debug == 0 && !p.trace && a->M % 128 == 0 && a->N % bn == 0 && (a->ldc & 3) == 0 && a->ldc < (1 << 23) &&
                          (!xsrc || ((xld & 3) == 0 && xld < (1 << 23))) && (!a->rowscale || a->rows_per_scale >= 32) && (!a->bias || ((uintptr_t)a->bias & 15) == 0) &&
                          (!a->bn_y || ((((uintptr_t)a->bn_stat | (uintptr_t)a->bn_gamma | (uintptr_t)a->bn_beta) & 15) == 0)) && (!a->colstats || ((uintptr_t)a->colstats & 3) == 0);
    const bool rows_epi = sb && rows_ok;
    const bool sb96 = ring && wide96 && rows_ok && a->K <= sb_k;      // N = 96, 288: the 4 x 1 wave layout of the single-buffer form with the row-layout epilogue
#define GG_LAUNCH_F32(E)                                                                                          \
    do {                                                                                                          \
        if (small) hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<64, 2, 2, E, 4, 4, 64, false, 0, false, 64>), dim3(p.tilesM * p.tilesN), dim3(256), 0, st, p); \
        else if (sb && narrow && rows_epi) hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<64, 4, 1, E, 2, (E == FE_BNBWD ? 4 : ((E == FE_DGELU || E == FE_LINEAR) ? 5 : 6)), 64, true, 0, true>), grid, dim3(256), 0, st, p); \
        else if (sb && rows_epi) hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<128, 2, 2, E, 2, 4, 128, true, 0, true>), grid, dim3(256), 0, st, p); \
        else if (sb && narrow) hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<64, 4, 1, E, 2, (E == FE_BNBWD ? 4 : ((E == FE_DGELU || E == FE_LINEAR) ? 5 : 6)), 64, true>), grid, dim3(256), 0, st, p); \
        else if (sb) hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<128, 2, 2, E, 2, 4, 128, true>), grid, dim3(256), 0, st, p); \
        else if (sb96) hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<128, 4, 1, E, 2, 4, 96, true, 0, true>), grid, dim3(256), 0, st, p); \
        else if (wide96) hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<128, 2, 2, E, 3, 3, 96>), grid, dim3(256), 0, st, p); \
        else if (ring && narrow) hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<64, 4, 1, E, 4, 2>), grid, dim3(256), 0, st, p);    \
        else if (ring) hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<128, 2, 2, E, 3, 3>), grid, dim3(256), 0, st, p);       \
        else if (narrow) hipLaunchKernelGGL((gemm_nt_f32_kernel<64, 4, 1, E>), grid, dim3(256), 0, st, p);           \
        else hipLaunchKernelGGL((gemm_nt_f32_kernel<128, 2, 2, E>), grid, dim3(256), 0, st, p);                      \
    } while (0)
    static const char* pro_ring_env = gg_dev_env("GG_GEMM_F32_PRO_RING");      // A/B: "0" = the register-staged prologue kernels
    if (a->a_bn_stat && a->K <= 384 && !(pro_ring_env && pro_ring_env[0] == '0') && (p.debug & 5) == 0) {
        // prologue GEMMs on the LDS-DMA ring (single-buffer form): the transform is applied to the A fragments after the LDS read
        dim3 rgrid(p.tilesM * p.tilesN);
        if (a->A2 && rows_ok) {
            if (wide96) hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<128, 4, 1, FE_LINEAR, 2, 3, 96, true, 2, true>), rgrid, dim3(256), 0, st, p);
            else if (narrow) hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<64, 4, 1, FE_LINEAR, 2, 4, 64, true, 2, true>), rgrid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<128, 2, 2, FE_LINEAR, 2, 3, 128, true, 2, true>), rgrid, dim3(256), 0, st, p);
        } else if (rows_ok) {
            if (wide96) hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<128, 4, 1, FE_PLAIN, 2, 4, 96, true, 1, true>), rgrid, dim3(256), 0, st, p);
            else if (narrow) hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<64, 4, 1, FE_PLAIN, 2, 4, 64, true, 1, true>), rgrid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<128, 2, 2, FE_PLAIN, 2, 4, 128, true, 1, true>), rgrid, dim3(256), 0, st, p);
        } else if (a->A2) {        // PRO 2, linear epilogue; the third operand panel makes a stage 24 KB: 3 workgroups per CU
            if (wide96) hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<128, 4, 1, FE_LINEAR, 2, 3, 96, true, 2>), rgrid, dim3(256), 0, st, p);
            else if (narrow) hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<64, 4, 1, FE_LINEAR, 2, 4, 64, true, 2>), rgrid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<128, 2, 2, FE_LINEAR, 2, 3, 128, true, 2>), rgrid, dim3(256), 0, st, p);
        } else {            // PRO 1, plain (+ column statistics) epilogue
            if (wide96) hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<128, 4, 1, FE_PLAIN, 2, 4, 96, true, 1>), rgrid, dim3(256), 0, st, p);
            else if (narrow) hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<64, 4, 1, FE_PLAIN, 2, 4, 64, true, 1>), rgrid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<128, 2, 2, FE_PLAIN, 2, 4, 128, true, 1>), rgrid, dim3(256), 0, st, p);
        }
        GG_LAUNCH_CHECK();
        return 0;
    }
    if (a->a_bn_stat) {       // prologue kernels: (PRO 1, plain epilogue) / (PRO 2, linear epilogue)
        if (a->A2) {
            if (wide96) hipLaunchKernelGGL((gemm_nt_f32_kernel<128, 2, 2, FE_LINEAR, 2, 96>), grid, dim3(256), 0, st, p);
            else if (narrow) hipLaunchKernelGGL((gemm_nt_f32_kernel<64, 4, 1, FE_LINEAR, 2>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((gemm_nt_f32_kernel<128, 2, 2, FE_LINEAR, 2>), grid, dim3(256), 0, st, p);
        } else {
            if (wide96) hipLaunchKernelGGL((gemm_nt_f32_kernel<128, 2, 2, FE_PLAIN, 1, 96>), grid, dim3(256), 0, st, p);
            else if (narrow) hipLaunchKernelGGL((gemm_nt_f32_kernel<64, 4, 1, FE_PLAIN, 1>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((gemm_nt_f32_kernel<128, 2, 2, FE_PLAIN, 1>), grid, dim3(256), 0, st, p);
        }
        GG_LAUNCH_CHECK();
        return 0;
    }
    if (nsplit > 1) {        // slabs through the plain epilogue, then the reduction applies bias / rowscale / residual
        F32GemmParams q = p;
        q.C = slabs; q.ldc = a->N; q.bias = nullptr; q.rowscale = nullptr; q.residual = nullptr;
        hipLaunchKernelGGL((gemm_nt_f32_ring_kernel<64, 2, 2, FE_PLAIN, 4, 4, 64, false, 0, false, 64>), dim3(p.tilesM * p.tilesN * nsplit), dim3(256), 0, st, q);
        const int64_t n4 = (int64_t)a->M * (a->N >> 2);
        hipLaunchKernelGGL(splitk_nt_reduce_f32_kernel, dim3((unsigned)std::min<int64_t>(gg_cdiv(n4, 256), 2048)), dim3(256), 0, st, slabs, nsplit, a->M, a->N, (float*)a->C, a->ldc,
                           a->bias, a->rowscale, a->rows_per_scale, (const float*)a->residual, a->ldr);
        GG_LAUNCH_CHECK();
        return 0;
    }
    switch (epi) {
        case FE_BNBWD: GG_LAUNCH_F32(FE_BNBWD); break;
        case FE_PLAIN: GG_LAUNCH_F32(FE_PLAIN); break;
        case FE_LINEAR: GG_LAUNCH_F32(FE_LINEAR); break;
        case FE_GELU: GG_LAUNCH_F32(FE_GELU); break;
        case FE_QGELU: GG_LAUNCH_F32(FE_QGELU); break;
        default: GG_LAUNCH_F32(FE_DGELU); break;
    }
#undef GG_LAUNCH_F32
    GG_LAUNCH_CHECK();
    return 0;
}

static int gemm_tn_f32_launch(const void* dY, const void* Y2, const float* coef, int64_t ldy, const void* X, int64_t ldx, int M, int N, int K, const float* rowscale,
                              int rows_per_scale, float* partials, int splits, void* stream);
extern "C" int gg_gemm_tn_f32(const void* dY, int64_t ldy, const void* X, int64_t ldx, int M, int N, int K, const float* rowscale,
                              int rows_per_scale, float* partials, int splits, void* stream) {
    return gemm_tn_f32_launch(dY, nullptr, nullptr, ldy, X, ldx, M, N, K, rowscale, rows_per_scale, partials, splits, stream);
}
// weight gradient of a ConvNorm from (dz, y, coef): the left operand is coef0*dz + coef1*y + coef2 per column (y laid out like dz, ld = ldy)
extern "C" int gg_gemm_tn_bn_f32(const void* dz, const void* y, int64_t ldy, const float* coef, const void* X, int64_t ldx, int M, int N, int K,
                                 float* partials, int splits, void* stream) {
    GG_CHECK(y && coef && ((uintptr_t)y & 15) == 0 && ((uintptr_t)coef & 15) == 0, "gg_gemm_tn_bn_f32: y / coef missing or misaligned");
    return gemm_tn_f32_launch(dz, y, coef, ldy, X, ldx, M, N, K, nullptr, 0, partials, splits, stream);
}
static int gemm_tn_f32_launch(const void* dY, const void* Y2, const float* coef, int64_t ldy, const void* X, int64_t ldx, int M, int N, int K, const float* rowscale,
                              int rows_per_scale, float* partials, int splits, void* stream) {
    GG_CHECK(dY && X && partials && M > 0 && N > 0 && K > 0 && splits > 0, "gg_gemm_tn_f32: bad args");
    GG_CHECK((N & 3) == 0 && (K & 3) == 0 && (ldy & 3) == 0 && (ldx & 3) == 0, "gg_gemm_tn_f32: N, K, ldy, ldx must be multiples of 4");
    GG_CHECK(((uintptr_t)dY & 15) == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)partials & 15) == 0, "gg_gemm_tn_f32: operands and partials must be 16-byte aligned");
    GG_CHECK(!rowscale || rows_per_scale > 0, "gg_gemm_tn_f32: rows_per_scale");
    F32TnParams p;
    p.dY = (const float*)dY; p.ldy = ldy; p.X = (const float*)X; p.ldx = ldx; p.M = M; p.N = N; p.K = K;
    p.rowscale = rowscale; p.rows_per_scale = rows_per_scale; p.part = partials;
    p.Y2 = (const float*)Y2; p.coef = coef;
    p.tilesN = (int)gg_cdiv(N, 128); p.tilesK = (int)gg_cdiv(K, 128);
    const bool small = N <= 64 && K <= 64 && (splits & 3) == 0;      // gg_gemm_tn_f32_splits returns a multiple of 4 for these shapes
    const int blocks = small ? splits / 4 : splits;
    p.m_per_split = (int)gg_align(gg_cdiv(M, blocks), 32);
    GG_CHECK((int64_t)p.m_per_split * std::max(ldy, ldx) * 4 < ((int64_t)1 << 32), "gg_gemm_tn_f32: a split's rows must span < 4 GiB per operand (use more splits)");
    GG_CHECK((int64_t)p.tilesN * p.tilesK * blocks < ((int64_t)1 << 31), "gg_gemm_tn_f32: grid too large");
    GG_PROF(GG_CAT_GEMM, 2.0 * M * (double)N * K, 4.0 * M * ((double)N * (Y2 ? 2 : 1) + K) + 4.0 * splits * (double)N * K, stream);
    const bool fe = !small && (N & 63) == 0 && (K & 63) == 0;       // wave tiles all-or-nothing: the branch-free MFMA block
    const dim3 tgrid((unsigned)(p.tilesN * p.tilesK * blocks));
    if (Y2) {
        if (small) hipLaunchKernelGGL((gemm_tn_f32_kernel<true, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
        else if (fe) hipLaunchKernelGGL((gemm_tn_f32_kernel<false, true, true>), tgrid, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((gemm_tn_f32_kernel<false, true>), tgrid, dim3(256), 0, (hipStream_t)stream, p);
    } else if (small) hipLaunchKernelGGL((gemm_tn_f32_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    else if (fe) hipLaunchKernelGGL((gemm_tn_f32_kernel<false, false, true>), tgrid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((gemm_tn_f32_kernel<false>), tgrid, dim3(256), 0, (hipStream_t)stream, p);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_gemm_tn_f32_splits(int M, int N, int K) {
    const int64_t tiles = gg_cdiv(N, 128) * gg_cdiv(K, 128);
    const int64_t cap = std::max<int64_t>(1, ((int64_t)128 << 20) / ((int64_t)N * K * 4));        // 128 MiB of slabs at most (scratch.splitk)
    const int64_t smax = std::max<int64_t>(1, std::min<int64_t>(cap, gg_cdiv(M, 512)));              // at least 512 rows per split
    // fill whole rounds of the 768 resident workgroups (3 per CU): tiles * splits just below a multiple of 768 -- 90 tiles x 12 splits ran
    // 1080 workgroups = 1.4 rounds (70 % of the second round idle); x 17 = 1530 = 1.99 rounds
    int64_t s = 1;
    double best = 0.0;
    for (int64_t c = 1; c <= smax && c * tiles <= 4 * 768; ++c) {
        const double eff = (double)(c * tiles) / (768.0 * (double)gg_cdiv(c * tiles, 768));
        if (eff > best + 0.02) { best = eff; s = c; }
    }
    if (N <= 64 && K <= 64 && M >= 4096) s = 4 * std::max<int64_t>(1, std::min<int64_t>(s, cap / 4));   // row-split form: 4 slabs (one per wave) per block
    return (int)s;
}

extern "C" int gg_colsum_f32(const float* x, int64_t ld, int M, int C, const float* rowscale, int rows_per_scale, float* scratch,
                             float* out, int accumulate, void* stream) {
    GG_CHECK(x && scratch && out && M > 0 && C > 0 && (C & 3) == 0 && (ld & 3) == 0 && ((uintptr_t)x & 15) == 0, "gg_colsum_f32: bad args (C, ld %% 4)");
    GG_PROF(GG_CAT_NORM, 0, 4.0 * M * C, stream);
    const int rpb = 512;
    const int nparts = (int)gg_cdiv(M, rpb);
    GG_CHECK(nparts <= 65535, "gg_colsum_f32: M too large");
    hipLaunchKernelGGL(colsum_partial_f32_kernel, dim3((unsigned)gg_cdiv(C, 256), nparts), dim3(256), 0, (hipStream_t)stream, x, ld, M, C,
                       rowscale, rows_per_scale, scratch, rpb);
    const float* rows; int nrows;
    gg_reduce_rows(scratch, nparts, C, (hipStream_t)stream, &rows, &nrows);
    hipLaunchKernelGGL(colsum_final_f32_kernel, dim3((unsigned)gg_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, rows, nrows, C, out, accumulate);
    GG_LAUNCH_CHECK();
    return 0;
}
