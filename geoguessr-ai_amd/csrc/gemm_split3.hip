// The fp32_split mode's GEMM ("the bf16 x 3 split"): an fp32-accurate GEMM on the bf16 matrix pipe.
//
//   C[M,N] (f32) = A[M,K] . B[N,K]^T   with   A = a1 + a2 + a3,  B = b1 + b2 + b3   (three bf16 terms each: 24 significand bits)
//                = (a1 b3 + a2 b2 + a3 b1) + (a1 b2 + a2 b1) + a1 b1                  (six bf16 products, f32 accumulation; the dropped terms are < 2^-24)
//
// The fp32 MFMA of gfx950 runs at the vector ALU's rate (157 TFLOP/s; the model's large GEMMs already sit at the clock-limited 125-138); the bf16 MFMA
// runs at 2.5 PFLOP/s, so six products are worth up to 417 TFLOP/s of fp32-equivalent work -- if a kernel keeps that pipe fed.  Run as ONE bf16 GEMM
// over a 6 K contraction the library's 128^2 two-barrier kernel gives 133-141 (it re-reads every fragment from LDS once per product).  This kernel
// stages the three planes of both operands together, so a fragment read from LDS feeds three (A) or up to three (B) products:
//   * operands: pre-split bf16 planes in HBM, [3][rows][ld] (gg_split3_bf16 makes them; a production form would have the PRODUCER's epilogue write them);
//   * tile 256 x 128, k-stage 32 (one v_mfma_f32_16x16x32_bf16 step), 512 threads = 4 x 2 waves (64 x 64 per wave: 4 x 4 MFMA tiles x 6 products = 96 MFMAs per
//     24 fragment reads and stage; the 128 x 128 / 64 x 32-per-wave form moves 1.5 x the LDS bytes per MFMA and sits at the LDS-port / MFMA balance point);
//   * LDS-DMA (`buffer_load ... lds`, 16 B per lane, no VGPR staging) into a 2-stage ring of 6 plane tiles (72 KB per stage: 144 KB, one workgroup per CU), rows of 64 bytes with the
//     16-byte chunk index XOR-ed with s3_swz((row >> 2) & 3, p.swz_plain) on the SOURCE side of the DMA, so the fragment reads (ds_read_b128: 16 rows x one chunk) are
//     conflict-free without padding (s3_swz: the permutation the instruction's 16-lane service groups need);
//   * one raw s_barrier per stage, the next stage's DMA in flight under the current stage's 96 MFMAs; two waves per SIMD.
// Epilogues: the Linear family of the model (bias, GELU with a saved pre-activation, rowscale + residual, x GELU'), results as f32 and / or as planes for the
// next split GEMM.  The fp32_split mode of csrc/tinyvit.hip runs on the forms further down (gemm_nt_split3a / b_kernel: A as f32, split in the loader); this
// plane-fed form is what they are checked against bit for bit (tests/test_gpu_kernels.py) and what tools/bench_split3.py times next to the f32-MFMA GEMM.
// What the A-as-f32 forms add (round 6): compile-time epilogue classes (split3_epilogue_rows_ec), non-temporal result stores, an optional act(BatchNorm(A)) prologue in
// the loader (gg_gemm_nt_split3_af32_pro), the conflict-free chunk permutation s3_swz, a 32 x 32 x 16 MFMA variant kept as a measured alternative (gemm_nt_split3w_kernel);
// what bounds them -- the chip's power limit on random data, not a pipe -- is in DESIGN.md 5.
#include "common.h"
#include <type_traits>
#include <algorithm>
#include <mutex>
#include <vector>
#include <stdlib.h>
#include <string.h>
#include "../../include/gg.h"

namespace {

struct Split3Params {
    const bf16* A; int64_t lda, plane_a;      // planes a1, a2, a3 at A + i * plane_a (elements)
    const float* Af; int64_t ldaf;            // ... or the f32 operand itself (gemm_nt_split3a_kernel splits it while it is staged)
    const bf16* B; int64_t ldb, plane_b;
    float* C; int64_t ldc;             // f32 result (may be null when only planes are wanted)
    const float* bias;
    int M, N, K, tilesM, tilesN;
    // epilogue family of the model's Linears (the f32 GEMM's classes): v = acc + bias; preact copy; act; * rowscale[m / rows_per_scale]; + residual;
    // or v = (acc) * act'(dact_preact) * rowscale (the dgrad through an activation); result as f32 and / or as three bf16 planes [3][M][ldp]
    int act; float* preact;
    const float* rowscale; int rows_per_scale;
    const float* residual; int64_t ldr;
    const float* dact_preact; int dact;
    bf16* c_planes; int64_t ldp;
    const float* a_stat; const float* a_gamma; const float* a_beta; int a_act;      // A prologue (gemm_nt_split3a_kernel<..., PRO>): A := a_act(BatchNorm(A)), a_stat = [mean | rstd][K]
    int swz_plain;                     // dev A/B: 1 = the round-5 chunk swizzle (plain XOR with the row quad: 2-way bank conflicts on every fragment read)
    float* colstats;                   // BatchNorm partials [ceil(M / 128)][2][N] (column sums of the result and of its square per 128-row block; plain epilogue only) or null
};

// one 4-column group of one row in the MFMA result layout (lane: row m, columns n .. n + 3)
__device__ __forceinline__ void split3_epilogue4(const Split3Params& p, f32x4 v, int m, int n) {
    if (m >= p.M || n >= p.N) return;
    const bool full = n + 3 < p.N;
    if (p.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += p.bias[n + r];
    }
    auto ld4 = [&](const float* base, int64_t ld) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        const float* q = base + (int64_t)m * ld + n;
        if (full && (ld & 3) == 0) t = *reinterpret_cast<const f32x4*>(q);
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < p.N) t[r] = q[r];
        }
        return t;
    };
    auto st4 = [&](float* base, int64_t ld, f32x4 t) {
        float* q = base + (int64_t)m * ld + n;
        if (full && (ld & 3) == 0) *reinterpret_cast<f32x4*>(q) = t;
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < p.N) q[r] = t[r];
        }
    };
    if (p.preact) st4(p.preact, p.ldc, v);
    if (p.dact_preact) v = v * gg_act_grad_f32_v4(ld4(p.dact_preact, p.ldc), p.dact);
    else if (p.act) v = gg_act_f32_v4(v, p.act);
    if (p.rowscale) v = v * p.rowscale[m / p.rows_per_scale];
    if (p.residual) v = v + ld4(p.residual, p.ldr);
    if (p.C) st4(p.C, p.ldc, v);
    if (p.c_planes) {
        bf16x4 p1, p2, p3;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bf16 s1_, s2_, s3_;
            gg_split3_rne(v[r], s1_, s2_, s3_);
            p1[r] = s1_; p2[r] = s2_; p3[r] = s3_;
        }
        bf16* q = p.c_planes + (int64_t)m * p.ldp + n;
        const int64_t plane = (int64_t)p.M * p.ldp;
        if (full && (p.ldp & 3) == 0) {
            *reinterpret_cast<bf16x4*>(q) = p1; *reinterpret_cast<bf16x4*>(q + plane) = p2; *reinterpret_cast<bf16x4*>(q + 2 * plane) = p3;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < p.N) { q[r] = p1[r]; q[plane + r] = p2[r]; q[2 * plane + r] = p3[r]; }
        }
    }
}

// Epilogue classes known at compile time (EC; the host picks one when the shape allows the vector path: N % 8 == 0, row pitches % 4 == 0): the generic row epilogue
// below decides bias / activation / row scale / residual / planes / statistics per row behind ~40 runtime branches, and the compiler cannot move a row's global loads across
// them.  Here the aux rows a thread needs (the saved pre-activation of the GELU' data gradient, the residual) are ALL requested before the first tile row is touched (8 rows
// x 32 bytes per thread in flight), and the loop body is branch-free.  Arithmetic, operation order and rounding are the generic path's: results are bit-identical.
//   1: C = acc + bias                                    (qkv, plain data gradients)
//   2: pre = acc + bias;  C = GELU(pre)                  (fc1 forward; pre-activation saved)
//   3: C = acc * GELU'(saved pre-activation) [* scale]   (fc2 data gradient)
//   4: C = (acc + bias) [* scale] + residual             (proj / fc2 forward)
template <int BM, int BN, int NTHR, int HINT, int EC>
__device__ __forceinline__ void split3_epilogue_rows_ec(const Split3Params& p, const float* Ct, int m0, int n0) {
    constexpr int LDT = BN + 4, CPR = BN / 8, RPP = NTHR / CPR, NR = (BM + RPP - 1) / RPP;
    if ((int)threadIdx.x >= RPP * CPR) return;
    const int chunk = threadIdx.x % CPR, r0 = threadIdx.x / CPR;
    const int n = n0 + chunk * 8;
    if (n >= p.N) return;                                           // (N % 8 == 0: a chunk is entirely inside or outside)
    auto ldg = [&](const float* q) -> f32x4 {
        if constexpr (HINT & 4) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q));
        else return *reinterpret_cast<const f32x4*>(q);
    };
    auto stg = [&](float* q, f32x4 t) {
        if constexpr (HINT & 1) __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(q));
        else *reinterpret_cast<f32x4*>(q) = t;
    };
    f32x4 aux[(EC == 3 || EC == 4) ? NR : 1][2];
    if constexpr (EC == 3 || EC == 4) {
        const float* base = EC == 3 ? p.dact_preact : p.residual;
        const int64_t ld = EC == 3 ? p.ldc : p.ldr;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int rr = r0 + i * RPP, m = m0 + rr;
            if (rr < BM && m < p.M) { aux[i][0] = ldg(base + (int64_t)m * ld + n); aux[i][1] = ldg(base + (int64_t)m * ld + n + 4); }
            else aux[i][0] = aux[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (EC != 3 && p.bias) { b0 = *reinterpret_cast<const f32x4*>(p.bias + n); b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4); }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int rr = r0 + i * RPP, m = m0 + rr;
        if (rr >= BM || m >= p.M) break;
        f32x4 v0 = *reinterpret_cast<const f32x4*>(Ct + rr * LDT + chunk * 8) + b0, v1 = *reinterpret_cast<const f32x4*>(Ct + rr * LDT + chunk * 8 + 4) + b1;
        float* const c = p.C + (int64_t)m * p.ldc + n;
        if constexpr (EC == 2) {
            float* const pre = p.preact + (int64_t)m * p.ldc + n;
            stg(pre, v0); stg(pre + 4, v1);
            v0 = gg_act_f32_v4(v0, 1); v1 = gg_act_f32_v4(v1, 1);
        }
        if constexpr (EC == 3) { v0 = v0 * gg_act_grad_f32_v4(aux[i][0], 1); v1 = v1 * gg_act_grad_f32_v4(aux[i][1], 1); }
        if constexpr (EC == 3 || EC == 4) {
            if (p.rowscale) { const float sc = p.rowscale[m / p.rows_per_scale]; v0 = v0 * sc; v1 = v1 * sc; }
        }
        if constexpr (EC == 4) { v0 = v0 + aux[i][0]; v1 = v1 + aux[i][1]; }
        stg(c, v0); stg(c + 4, v1);
    }
}

// Row-layout epilogue of a BM x BN tile: the accumulators cross LDS (the idle ring) so that every global access of the epilogue is a run of whole rows --
// a lane owns 8 consecutive columns of a row, 16 lanes one 128-column row: 512-byte f32 runs and 256-byte plane runs instead of the MFMA layout's
// 64- / 32-byte pieces of 16 different rows per instruction (the plane-writing epilogues of fc1 / the fc2 dgrad were slower than the f32 GEMM's with those)
// HINT bits (cache policy experiments): 1 = result / pre-activation stores non-temporal, 4 = residual / saved-pre-activation loads non-temporal
template <int BM, int BN, int NTHR, int HINT = 0>
__device__ __forceinline__ void split3_epilogue_rows(const Split3Params& p, float* Ct, int m0, int n0) {
    constexpr int LDT = BN + 4;                                   // padded row: conflict-free 16-byte column writes from the MFMA layout
    constexpr int CPR = BN / 8;                                   // 8-column chunks per row
    constexpr int RPP = NTHR / CPR;                               // rows per pass
    const bool live = (int)threadIdx.x < RPP * CPR;                // (BN = 96: 12 chunks per row do not divide the block)
    if (!live && !p.colstats) return;                              // (no barrier follows without column statistics)
    const int chunk = threadIdx.x % CPR, r0 = threadIdx.x / CPR;
    const int n = n0 + chunk * 8;
    const bool nfull = n + 7 < p.N;
    float bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = (p.bias && n + j < p.N) ? p.bias[n + j] : 0.f;
    const int64_t plane = (int64_t)p.M * p.ldp;
    float cs[2][8], cq[2][8];                                        // column sums of this thread's rows, per 128-row block (p.colstats)
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[0][j] = cs[1][j] = cq[0][j] = cq[1][j] = 0.f;
    for (int rr = r0; rr < BM && live; rr += RPP) {
        const int m = m0 + rr;
        if (m >= p.M || n >= p.N) continue;
        float v[8];
        {
            const f32x4 a = *reinterpret_cast<const f32x4*>(Ct + rr * LDT + chunk * 8), b = *reinterpret_cast<const f32x4*>(Ct + rr * LDT + chunk * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = a[j] + bv[j]; v[4 + j] = b[j] + bv[4 + j]; }
        }
        auto ld8 = [&](const float* base, int64_t ld, float (&t)[8]) {
            const float* q = base + (int64_t)m * ld + n;
            if (nfull && (ld & 3) == 0) {
                f32x4 a, b;
                if constexpr (HINT & 4) { a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q)); b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q + 4)); }
                else { a = *reinterpret_cast<const f32x4*>(q); b = *reinterpret_cast<const f32x4*>(q + 4); }
#pragma unroll
                for (int j = 0; j < 4; ++j) { t[j] = a[j]; t[4 + j] = b[j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = n + j < p.N ? q[j] : 0.f;
            }
        };
        auto st8 = [&](float* base, int64_t ld, const float (&t)[8]) {
            float* q = base + (int64_t)m * ld + n;
            if (nfull && (ld & 3) == 0) {
                if constexpr (HINT & 1) {
                    __builtin_nontemporal_store((f32x4){t[0], t[1], t[2], t[3]}, reinterpret_cast<f32x4*>(q));
                    __builtin_nontemporal_store((f32x4){t[4], t[5], t[6], t[7]}, reinterpret_cast<f32x4*>(q + 4));
                } else {
                    *reinterpret_cast<f32x4*>(q) = (f32x4){t[0], t[1], t[2], t[3]};
                    *reinterpret_cast<f32x4*>(q + 4) = (f32x4){t[4], t[5], t[6], t[7]};
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (n + j < p.N) q[j] = t[j];
            }
        };
        if (p.preact) st8(p.preact, p.ldc, v);
        // (the packed forms: GELU / GELU' on two lanes of a v_pk_* instruction -- the scalar forms were a third of these launches' time at K = 192)
        if (p.dact_preact) {
            float t[8];
            ld8(p.dact_preact, p.ldc, t);
            const f32x4 g0 = gg_act_grad_f32_v4((f32x4){t[0], t[1], t[2], t[3]}, p.dact), g1 = gg_act_grad_f32_v4((f32x4){t[4], t[5], t[6], t[7]}, p.dact);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] *= g0[j]; v[4 + j] *= g1[j]; }
        } else if (p.act) {
            const f32x4 a0 = gg_act_f32_v4((f32x4){v[0], v[1], v[2], v[3]}, p.act), a1 = gg_act_f32_v4((f32x4){v[4], v[5], v[6], v[7]}, p.act);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = a0[j]; v[4 + j] = a1[j]; }
        }
        if (p.rowscale) {
            const float sc = p.rowscale[m / p.rows_per_scale];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] *= sc;
        }
        if (p.residual) {
            float t[8];
            ld8(p.residual, p.ldr, t);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += t[j];
        }
        if (p.C) st8(p.C, p.ldc, v);
        if (p.colstats) {
            const float hi = rr >= 128 ? 1.f : 0.f, lo = 1.f - hi;
#pragma unroll
            for (int j = 0; j < 8; ++j) { cs[0][j] += lo * v[j]; cq[0][j] += lo * v[j] * v[j]; cs[1][j] += hi * v[j]; cq[1][j] += hi * v[j] * v[j]; }
        }
        if (p.c_planes) {
            bf16x8 p1, p2, p3;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bf16 s1_, s2_, s3_;
                gg_split3_rne(v[j], s1_, s2_, s3_);
                p1[j] = s1_; p2[j] = s2_; p3[j] = s3_;
            }
            bf16* q = p.c_planes + (int64_t)m * p.ldp + n;
            if (nfull && (p.ldp & 7) == 0) {
                *reinterpret_cast<bf16x8*>(q) = p1; *reinterpret_cast<bf16x8*>(q + plane) = p2; *reinterpret_cast<bf16x8*>(q + 2 * plane) = p3;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (n + j < p.N) { q[j] = p1[j]; q[plane + j] = p2[j]; q[2 * plane + j] = p3[j]; }
            }
        }
    }
    if (p.colstats) {
        // thread partials -> LDS [block][which][r0][BN] (the tile image is dead after the barrier) -> one thread per (block, which, column) adds the RPP rows in order
        constexpr int PARTS = BM / 128;
        __syncthreads();
        if (live) {
#pragma unroll
            for (int part = 0; part < PARTS; ++part)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    Ct[((part * 2 + 0) * RPP + r0) * BN + chunk * 8 + j] = cs[part][j];
                    Ct[((part * 2 + 1) * RPP + r0) * BN + chunk * 8 + j] = cq[part][j];
                }
        }
        __syncthreads();
        const int nparts = (p.M + 127) / 128;
        for (int t = threadIdx.x; t < PARTS * 2 * BN; t += NTHR) {
            const int part = t / (2 * BN), which = (t / BN) & 1, col = t % BN;
            float sum = 0.f;
            for (int q = 0; q < RPP; ++q) sum += Ct[((part * 2 + which) * RPP + q) * BN + col];
            const int prow = m0 / 128 + part;
            if (prow < nparts && n0 + col < p.N) p.colstats[((int64_t)prow * 2 + which) * p.N + n0 + col] = sum;
        }
    }
}

constexpr int S3_BM = 128, S3_BN = 128, S3_SK = 32;          // tile, k-stage (bf16 elements): a row of a plane tile is 64 bytes
constexpr int S3_TILE = S3_BM * S3_SK;                         // bf16 elements of one plane tile (8 KB)
constexpr int S3_STAGE = 6 * S3_TILE;                          // a1 a2 a3 b1 b2 b3
// Chunk swizzle of a plane tile (rows of 64 bytes = four 16-byte chunks): the chunk that holds k-chunk c of row r sits at position c ^ s3_swz((r >> 2) & 3).
// ds_read_b128 is serviced in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, {32-35, 44-47, 52-59}, {36-43, 48-51, 60-63} -- over 64 banks of
// 4 bytes (MI355X_MICROARCH.md, LDS table), so the 16 lanes of a group (all 16 rows lr of a fragment: rows 0-3 and 12-15 with one k-chunk lg, rows 4-11 with
// lg ^ 1) must fall on 16 different 16-byte slots of a 256-byte window: slot = 4 (r & 3) + position, i.e. the four row quads q = r >> 2 of a group need four
// different positions.  The plain XOR (position = c ^ q) maps (q 0, lg 0) and (q 1, lg 1) to the same position: a 2-way conflict in every group, every
// fragment read took 8 instead of 4 LDS cycles (SQ_LDS_BANK_CONFLICT = 4 cycles per ds_read_b128, 39 % of the LDS-active cycles:
// profiles/r06_split_sq_counters.txt).  The permutation q -> {0, 2, 3, 1} makes every group a permutation of the 16 slots.
__device__ __forceinline__ int s3_swz(int q, int plain = 0) { return plain ? q : (0x78 >> (2 * q)) & 3; }
template <int N> __device__ __forceinline__ void wait_outstanding() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// BM x BN tile, WM x WN waves (each (BM / WM) x (BN / WN)), NST ring stages; PIPE: fragments of stage s + 1 are read into a second register set under the MFMAs of stage s
template <int BM, int BN, int WM, int WN, bool PIPE, int NST>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_split3_kernel(Split3Params p) {
    constexpr int NW = WM * WN, TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int SLA = BM / 16, SLB = BN / 16;                  // 16-row wave-slices (1 KB) of an A / B plane tile
    constexpr int IA = SLA / NW, IB = SLB / NW;                  // DMA instructions per A / B plane tile and wave
    constexpr int DPS = 3 * (IA + IB);                           // ... per stage and wave
    constexpr int TA = BM * S3_SK, TB = BN * S3_SK, STAGE = 3 * (TA + TB);
    static_assert(IA * NW == SLA && IB * NW == SLB && IA >= 1 && IB >= 1, "waves must divide the slices of both plane tiles");
    static_assert(!PIPE || NST == 3, "the pipelined form walks a 3-stage ring");
    extern __shared__ __attribute__((aligned(16))) bf16 s3mem[];
    const int tiles = p.tilesM * p.tilesN;
    const int bid = gg_xcd_remap(blockIdx.x, tiles);
    const int tm = bid / p.tilesN, tn = bid % p.tilesN;
    const int m0 = tm * BM, n0 = tn * BN;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lg = lane >> 4;
    // DMA geometry: slice sl = wave + NW j covers tile rows 16 sl .. 16 sl + 15: lane -> (row 16 sl + lane / 4, LDS chunk slot lane % 4); the slot holds SOURCE chunk
    // slot ^ ((row >> 2) & 3) = slot ^ (lane >> 4) (16 sl does not touch bits 2-3)
    const int dchunk = (lane & 3) ^ s3_swz(lane >> 4, p.swz_plain);
    const unsigned rowsA = (unsigned)min(p.M - m0, BM), rowsB = (unsigned)min(p.N - n0, BN);
    __amdgpu_buffer_rsrc_t rs[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        rs[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + i * p.plane_a + (int64_t)m0 * p.lda), 0, (int)(rowsA * (unsigned)p.lda * 2u), 0x00020000);
        rs[3 + i] = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + i * p.plane_b + (int64_t)n0 * p.ldb), 0, (int)(rowsB * (unsigned)p.ldb * 2u), 0x00020000);
    }
    unsigned voffA[IA], voffB[IB];
#pragma unroll
    for (int j = 0; j < IA; ++j) voffA[j] = (unsigned)((wave + NW * j) * 16 + (lane >> 2)) * (unsigned)p.lda * 2u + dchunk * 16u;
#pragma unroll
    for (int j = 0; j < IB; ++j) voffB[j] = (unsigned)((wave + NW * j) * 16 + (lane >> 2)) * (unsigned)p.ldb * 2u + dchunk * 16u;
    auto issue_stage = [&](int st, bf16* base) {
        const int k0 = st * S3_SK;
        const bool kin = k0 + dchunk * 8 < p.K;                 // K % 8 == 0: a chunk is entirely inside or outside (outside: range check -> zeros)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < IA; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[i], (__attribute__((address_space(3))) void*)(base + i * TA + (wave + NW * j) * 512), 16,
                                                         (int)(kin ? voffA[j] : 0xFFFFFFF0u), k0 * 2, 0, 0);
#pragma unroll
            for (int j = 0; j < IB; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[3 + i], (__attribute__((address_space(3))) void*)(base + 3 * TA + i * TB + (wave + NW * j) * 512), 16,
                                                         (int)(kin ? voffB[j] : 0xFFFFFFF0u), k0 * 2, 0, 0);
        }
    };
    // fragment addresses: row (16 t + lr) of a plane tile, k-chunk lg -> slot lg ^ ((lr >> 2) & 3) (the tile index t does not touch bits 2-3 of the row)
    const int fslot = (lg ^ s3_swz((lr >> 2) & 3, p.swz_plain)) * 8;
    const int a_off = (wm * (BM / WM) + lr) * S3_SK + fslot, b_off = 3 * TA + (wn * (BN / WN) + lr) * S3_SK + fslot;
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = (p.K + S3_SK - 1) / S3_SK;
    auto frag_read = [&](const bf16* cur, bf16x8 (&xf)[3][TM], bf16x8 (&wf)[3][TN]) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) xf[pl][mt] = *reinterpret_cast<const bf16x8*>(cur + pl * TA + a_off + mt * 16 * S3_SK);
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) wf[pl][nt] = *reinterpret_cast<const bf16x8*>(cur + pl * TB + b_off + nt * 16 * S3_SK);
        }
    };
    // small terms first: (a1 b3 + a2 b2 + a3 b1), (a1 b2 + a2 b1), a1 b1.  D = Wfrag x Xfrag: lane owns 4 consecutive n of row m = lr
    auto mfma_stage = [&](const bf16x8 (&xf)[3][TM], const bf16x8 (&wf)[3][TN]) {
#define S3_MFMA(PA, PB)                                                                                              \
    _Pragma("unroll") for (int nt = 0; nt < TN; ++nt) _Pragma("unroll") for (int mt = 0; mt < TM; ++mt)             \
        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[PB][nt], xf[PA][mt], acc[nt][mt], 0, 0, 0)
        S3_MFMA(0, 2); S3_MFMA(1, 1); S3_MFMA(2, 0);
        S3_MFMA(0, 1); S3_MFMA(1, 0);
        S3_MFMA(0, 0);
#undef S3_MFMA
    };
    issue_stage(0, s3mem);
    if constexpr (PIPE) {
        // 3-stage ring, software-pipelined: iteration s multiplies the fragments of stage s (already in registers) while the fragments of stage s + 1 are read
        // from LDS into the other register set and the DMAs of stages s + 2 / s + 3 are in flight
        bf16x8 xa[3][TM], wa[3][TN], xb[3][TM], wb[3][TN];
        if (nk > 1) issue_stage(1, s3mem + STAGE);
        if (nk > 2) issue_stage(2, s3mem + 2 * STAGE);
        if (nk > 2) wait_outstanding<2 * DPS>(); else if (nk > 1) wait_outstanding<DPS>(); else wait_outstanding<0>();
        __builtin_amdgcn_s_barrier();
        frag_read(s3mem, xa, wa);
        auto iter = [&](int s, int slot, bf16x8 (&xc)[3][TM], bf16x8 (&wc)[3][TN], bf16x8 (&xn)[3][TM], bf16x8 (&wn_)[3][TN]) {
            const bool more = s + 1 < nk;
            if (more) { if (s + 2 < nk) wait_outstanding<DPS>(); else wait_outstanding<0>(); }      // stage s + 1 landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // own fragment reads of stage s are in registers ...
            __builtin_amdgcn_s_barrier();                           // ... everybody's are: slot(s) may be refilled; everybody's DMAs of stage s + 1 have landed
            if (s + 3 < nk) issue_stage(s + 3, s3mem + slot * STAGE);
            if (more) frag_read(s3mem + (slot == 2 ? 0 : slot + 1) * STAGE, xn, wn_);
            mfma_stage(xc, wc);
        };
        int slot = 0;
        for (int s = 0; s < nk; s += 2) {
            iter(s, slot, xa, wa, xb, wb);
            slot = slot == 2 ? 0 : slot + 1;
            if (s + 1 < nk) { iter(s + 1, slot, xb, wb, xa, wa); slot = slot == 2 ? 0 : slot + 1; }
        }
    } else if constexpr (NST == 3) {
        // 3-stage ring, two stages in flight: the DMA of stage s + 2 is issued when stage s starts
        if (nk > 1) issue_stage(1, s3mem + STAGE);
        int cb = 0;
        for (int s = 0; s < nk; ++s) {
            bf16* const cur = s3mem + cb * STAGE;
            if (s + 1 < nk) wait_outstanding<DPS>(); else wait_outstanding<0>();      // this wave's DMAs of stage s have landed (those of stage s + 1 may be in flight) ...
            __builtin_amdgcn_s_barrier();                           // ... everybody's have, and every wave has read its fragments of stage s - 1
            if (s + 2 < nk) issue_stage(s + 2, s3mem + (cb == 0 ? 2 : cb - 1) * STAGE);      // into the slot stage s - 1 occupied
            cb = cb == 2 ? 0 : cb + 1;
            bf16x8 xf[3][TM], wf[3][TN];
            frag_read(cur, xf, wf);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mfma_stage(xf, wf);
        }
    } else {
        // 2-stage ring: the DMA of stage s + 1 is issued when stage s starts (into the slot stage s - 1 occupied) and has the stage's MFMAs to land.
        // Within a stage the fragment reads are ordered by first use (a1, b3 | a2, b2 | a3, b1) so the LDS port works under the MFMAs of the previous product
        // group, and the last product (a1 b1) of stage s is held back behind the barrier of stage s + 1: it runs while that stage's first fragments (a1, b3)
        // and a2, b2 are read, so the matrix pipe does not idle across the barrier
        bf16x8 xf[3][TM], wf[3][TN], xn[TM];
        auto rd_a = [&](const bf16* cur, int pl, bf16x8 (&d)[TM]) {
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) d[mt] = *reinterpret_cast<const bf16x8*>(cur + pl * TA + a_off + mt * 16 * S3_SK);
        };
        auto rd_b = [&](const bf16* cur, int pl, bf16x8 (&d)[TN]) {
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) d[nt] = *reinterpret_cast<const bf16x8*>(cur + pl * TB + b_off + nt * 16 * S3_SK);
        };
#define S3_MFMA(PA, PB)                                                                                              \
    _Pragma("unroll") for (int nt = 0; nt < TN; ++nt) _Pragma("unroll") for (int mt = 0; mt < TM; ++mt)             \
        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[PB][nt], xf[PA][mt], acc[nt][mt], 0, 0, 0)
        wait_outstanding<0>();
        __builtin_amdgcn_s_barrier();
        if (nk > 1) issue_stage(1, s3mem + STAGE);
        rd_a(s3mem, 0, xf[0]); rd_b(s3mem, 2, wf[2]);
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(xf[0][i]));      // the compiler waits for these reads here: nothing is pending at the loop head
#pragma unroll
        for (int i = 0; i < TN; ++i) asm volatile("" : "+v"(wf[2][i]));
        for (int s = 0; s < nk; ++s) {
            const bf16* const cur = s3mem + (s & 1) * STAGE;
            rd_a(cur, 1, xf[1]); rd_b(cur, 1, wf[1]); rd_a(cur, 2, xf[2]); rd_b(cur, 0, wf[0]);
            S3_MFMA(0, 2); S3_MFMA(1, 1); S3_MFMA(2, 0);
            S3_MFMA(0, 1); S3_MFMA(1, 0);
            if (s + 1 < nk) {
                wait_outstanding<0>();                              // this wave's DMAs of stage s + 1 have landed ...
                __builtin_amdgcn_s_barrier();                       // ... everybody's have, and every wave holds all its fragments of stage s: its slot may be refilled
                if (s + 2 < nk) issue_stage(s + 2, s3mem + (s & 1) * STAGE);
                const bf16* const nxt = s3mem + ((s + 1) & 1) * STAGE;
                rd_a(nxt, 0, xn); rd_b(nxt, 2, wf[2]);              // b3 of stage s is dead; a1 is still needed by the held-back product
                S3_MFMA(0, 0);
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) xf[0][mt] = xn[mt];
            } else {
                S3_MFMA(0, 0);
            }
        }
#undef S3_MFMA
    }
    // epilogue: lane holds C[m = m0 + (BM / WM) wm + 16 mt + lr][n = n0 + (BN / WN) wn + 16 nt + 4 lg + r]
    constexpr bool ROWS = (size_t)BM * (BN + 4) * 4 <= (size_t)NST * STAGE * 2;      // the idle ring holds the f32 tile: row-layout epilogue
    if constexpr (ROWS) {
        float* Ct = reinterpret_cast<float*>(s3mem);
        __builtin_amdgcn_s_barrier();                             // every wave has read its last fragments: the ring is free
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
                *reinterpret_cast<f32x4*>(Ct + (wm * (BM / WM) + mt * 16 + lr) * (BN + 4) + wn * (BN / WN) + nt * 16 + lg * 4) = acc[nt][mt];
        __syncthreads();
        split3_epilogue_rows<BM, BN, 64 * NW>(p, Ct, m0, n0);
    } else {
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
                split3_epilogue4(p, acc[nt][mt], m0 + wm * (BM / WM) + mt * 16 + lr, n0 + wn * (BN / WN) + nt * 16 + lg * 4);
    }
}

// Persistent form of the 2 x 4-wave kernel: one workgroup per CU walks tiles t, t + grid, ...; the first two stages of the NEXT tile are issued before the
// current tile's epilogue, so the result stores and the next operands' latency overlap (with one 144 KB workgroup per CU nothing else would hide them:
// at K = 384 a tile is 12 stages = 9 us of MFMAs next to ~1.5 us of first-operand latency and ~1 us of stores)
__global__ __launch_bounds__(512) void gemm_nt_split3_persistent_kernel(Split3Params p) {
    constexpr int WM = 2, WN = 4, NW = 8, TM = 4, TN = 2, DPS = 6;
    extern __shared__ __attribute__((aligned(16))) bf16 s3mem[];
    const int tiles = p.tilesM * p.tilesN;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lg = lane >> 4;
    const int dchunk = (lane & 3) ^ s3_swz(lane >> 4, p.swz_plain);
    const unsigned row = (unsigned)(wave * 16 + (lane >> 2));
    const unsigned voffA = row * (unsigned)p.lda * 2u + dchunk * 16u, voffB = row * (unsigned)p.ldb * 2u + dchunk * 16u;
    const int fslot = (lg ^ s3_swz((lr >> 2) & 3, p.swz_plain)) * 8;
    const int a_off = (wm * 64 + lr) * S3_SK + fslot, b_off = (wn * 32 + lr) * S3_SK + fslot;
    const int nk = (p.K + S3_SK - 1) / S3_SK;
    __amdgpu_buffer_rsrc_t rs[6];
    auto set_tile = [&](int t, int& m0, int& n0) {
        const int bid = gg_xcd_remap(t, tiles);
        m0 = (bid / p.tilesN) * S3_BM; n0 = (bid % p.tilesN) * S3_BN;
        const unsigned rowsA = (unsigned)min(p.M - m0, S3_BM), rowsB = (unsigned)min(p.N - n0, S3_BN);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            rs[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + i * p.plane_a + (int64_t)m0 * p.lda), 0, (int)(rowsA * (unsigned)p.lda * 2u), 0x00020000);
            rs[3 + i] = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + i * p.plane_b + (int64_t)n0 * p.ldb), 0, (int)(rowsB * (unsigned)p.ldb * 2u), 0x00020000);
        }
    };
    auto issue_stage = [&](int st, bf16* base) {
        const int k0 = st * S3_SK;
        const bool kin = k0 + dchunk * 8 < p.K;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[i], (__attribute__((address_space(3))) void*)(base + i * S3_TILE + wave * 512), 16,
                                                     (int)(kin ? (i < 3 ? voffA : voffB) : 0xFFFFFFF0u), k0 * 2, 0, 0);
    };
    int m0, n0;
    int t = blockIdx.x;
    if (t >= tiles) return;
    set_tile(t, m0, n0);
    issue_stage(0, s3mem);
    if (nk > 1) issue_stage(1, s3mem + S3_STAGE);
    for (; t < tiles; t += gridDim.x) {
        f32x4 acc[TN][TM];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int cb = 0;
        for (int s = 0; s < nk; ++s) {
            bf16* const cur = s3mem + cb * S3_STAGE;
            // (stage 0 of a tile: the previous tile's result stores were issued after this tile's first DMAs -- in-order return, so wait for everything)
            if (s + 1 < nk && s > 0) wait_outstanding<DPS>(); else wait_outstanding<0>();
            __builtin_amdgcn_s_barrier();
            if (s + 2 < nk) issue_stage(s + 2, s3mem + (cb == 0 ? 2 : cb - 1) * S3_STAGE);
            cb = cb == 2 ? 0 : cb + 1;
            bf16x8 xf[3][TM], wf[3][TN];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) xf[pl][mt] = *reinterpret_cast<const bf16x8*>(cur + pl * S3_TILE + a_off + mt * 16 * S3_SK);
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) wf[pl][nt] = *reinterpret_cast<const bf16x8*>(cur + (3 + pl) * S3_TILE + b_off + nt * 16 * S3_SK);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define S3_MFMA(PA, PB)                                                                                              \
    _Pragma("unroll") for (int nt = 0; nt < TN; ++nt) _Pragma("unroll") for (int mt = 0; mt < TM; ++mt)             \
        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[PB][nt], xf[PA][mt], acc[nt][mt], 0, 0, 0)
            S3_MFMA(0, 2); S3_MFMA(1, 1); S3_MFMA(2, 0);
            S3_MFMA(0, 1); S3_MFMA(1, 0);
            S3_MFMA(0, 0);
#undef S3_MFMA
        }
        const int em0 = m0, en0 = n0;
        // next tile's first operands: every wave has read the last stage's fragments once it passes this barrier, so slots 0 / 1 are free
        // (the next tile starts its ring at slot 0 again)
        __builtin_amdgcn_s_barrier();
        if (t + (int)gridDim.x < tiles) {
            set_tile(t + gridDim.x, m0, n0);
            issue_stage(0, s3mem);
            if (nk > 1) issue_stage(1, s3mem + S3_STAGE);
        }
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) split3_epilogue4(p, acc[nt][mt], em0 + wm * 64 + mt * 16 + lr, en0 + wn * 32 + nt * 16 + lg * 4);
    }
}

// ------------------------------------------------------------------------------------------- A as f32, split while it is staged
// The form every Linear of the model can take: the activation operand stays the f32 tensor its producer wrote (no plane copy in HBM: 6 instead of 4 bytes per
// element, and a producer epilogue that has to write it), only the WEIGHT comes as cached planes.  Tile 256 x 128, 4 x 2 waves of 64 x 64, k-stage 32, 2-stage
// LDS ring of 72 KB as above.  B planes by LDS-DMA.  A: a thread loads one f32x4 of four rows per stage (8 lanes cover the 128 bytes = one cache line of a
// row), TWO stages ahead of its use, and -- one stage ahead -- splits it into three bf16x4 (20 vector instructions, which ride under the 96 MFMAs of the
// running stage: the bf16 MFMA does not share its issue with the vector ALU the way the f32 MFMA does) and writes them into the ring with the fragment
// reads' chunk swizzle.  In-order VMEM completion makes one counted wait per stage enough: at the top of stage s the queue holds A(s+1), B(s), A(s+2);
// vmcnt(4) leaves only A(s+2) in flight.
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// TN_: 16-column MFMA tiles per wave (4: 128-column tile; 3: 96 -- for N = 192, which 128-wide tiles cover with a quarter of the work wasted)
// PRO: the A operand is a saved pre-BatchNorm convolution output and the product wants act(BatchNorm(A)) (MBConv conv3 reading BN2 + GELU of the depthwise conv): the
// transform rides on the loader, in front of the split -- per element one FMA with the column's (scale, shift) from an LDS table + the f32-accurate GELU; with one N-tile
// (N = 96) every element is transformed exactly once and the activation tensor is never written (the f32-MFMA form of this fusion applied it to the fragments of every wave)
template <int ABL, int TN_ = 4, int EC = 0, int PRO = 0>      // PRO: 0 off; 1 + activation code of the prologue (1 BatchNorm only, 2 + GELU, 3 + QuickGELU): compile-time, so a stage carries one activation's code
__global__ __launch_bounds__(512) void gemm_nt_split3a_kernel(Split3Params p) {
    constexpr int TN = TN_, BM = 256, BN = 32 * TN, WN = 2, NW = 8, TM = 4;
    constexpr int TA = BM * S3_SK, TB = BN * S3_SK, STAGE = 3 * (TA + TB);
    constexpr int A_AUX = (ABL & 32) ? 2 : 0;                      // (cache-policy experiment: the f32 operand's loads non-temporal)
    extern __shared__ __attribute__((aligned(16))) bf16 s3mem[];
    const int tiles = p.tilesM * p.tilesN;
    const int bid = gg_xcd_remap(blockIdx.x, tiles);
    const int tm = bid / p.tilesN, tn = bid % p.tilesN;
    const int m0 = tm * BM, n0 = tn * BN;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lg = lane >> 4;
    const unsigned rowsA = (unsigned)min(p.M - m0, BM), rowsB = (unsigned)min(p.N - n0, BN);
    // B planes: 16-row slices by LDS-DMA (one per plane and wave: 8 slices = 128 rows), source chunk = slot ^ ((row >> 2) & 3)
    const int dchunk = (lane & 3) ^ s3_swz(lane >> 4, p.swz_plain);
    __amdgpu_buffer_rsrc_t rsB[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        rsB[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + i * p.plane_b + (int64_t)n0 * p.ldb), 0, (int)(rowsB * (unsigned)p.ldb * 2u), 0x00020000);
    const unsigned voffB = (unsigned)(wave * 16 + (lane >> 2)) * (unsigned)p.ldb * 2u + dchunk * 16u;
    // A: thread -> f32x4 number kq of rows (tid >> 3) + 64 j; its bf16x4 goes to chunk (kq >> 1) ^ ((row >> 2) & 3), half kq & 1 of the plane row
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Af + (int64_t)m0 * p.ldaf), 0, (int)(rowsA * (unsigned)p.ldaf * 4u), 0x00020000);
    const int kq = threadIdx.x & 7, arow = threadIdx.x >> 3;
    unsigned voffA[4];
    int ldsA[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = arow + 64 * j;
        voffA[j] = (unsigned)row * (unsigned)p.ldaf * 4u + kq * 16u;
        ldsA[j] = row * S3_SK + (((kq >> 1) ^ s3_swz((row >> 2) & 3, p.swz_plain)) << 3) + ((kq & 1) << 2);
    }
    // PRO: (scale, shift) per contraction column behind the ring: [2][KT] floats, zero beyond K (the masked stages then transform zeros into act(0) = 0)
    float* const ptab = reinterpret_cast<float*>(s3mem + 2 * STAGE);
    const int KT = PRO ? ((p.K + 31) & ~31) + 3 * S3_SK : 0;
    if constexpr (PRO) {
        for (int k = threadIdx.x; k < KT; k += 512) {
            float sc = 0.f, sh = 0.f;
            if (k < p.K) { sc = p.a_gamma[k] * p.a_stat[p.K + k]; sh = p.a_beta[k] - p.a_stat[k] * sc; }
            ptab[k] = sc; ptab[KT + k] = sh;
        }
        __syncthreads();
    }
    auto pro4 = [&](f32x4 v, int st) -> f32x4 {                     // act(v * scale + shift) for this thread's four columns of stage st
        if constexpr (PRO) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(ptab + st * S3_SK + kq * 4), sh = *reinterpret_cast<const f32x4*>(ptab + KT + st * S3_SK + kq * 4);
            // (scalar FMAs on purpose: beside MFMAs a v_pk_fma_f32 costs far more than the two v_fma_f32 it replaces -- MI355X_MICROARCH.md, cycle constants)
            if constexpr (PRO >= 5) return gg_act_f32_v4(v * sc + sh, PRO - 5);      // (dev A/B: the packed form)
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = gg_act_f32(fmaf(v[e], sc[e], sh[e]), PRO - 1);
            return o;
        } else return v;
    };
    auto issue_b = [&](int st, bf16* base) {
        const int k0 = st * S3_SK;
        const bool kin = k0 + dchunk * 8 < p.K;
        if (wave < BN / 16) {                                   // (wave-uniform: BN / 16 slices of 16 rows per plane)
#pragma unroll
            for (int i = 0; i < 3; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB[i], (__attribute__((address_space(3))) void*)(base + 3 * TA + i * TB + wave * 512), 16,
                                                         (int)(kin ? voffB : 0xFFFFFFF0u), k0 * 2, 0, 0);
        }
    };
    auto load_a = [&](int st, f32x4 (&r)[4]) {
        const int k0 = st * S3_SK;
        const bool kin = k0 + kq * 4 < p.K;                     // K % 4 == 0: an f32x4 is entirely inside or outside
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(kin ? voffA[j] : 0xFFFFFFF0u), k0 * 4, A_AUX));
    };
    auto split_store = [&](const f32x4 (&r)[4], bf16* base) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x4 p1, p2, p3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bf16 s1_, s2_, s3_;
                gg_split3_rne(r[j][e], s1_, s2_, s3_);
                p1[e] = s1_; p2[e] = s2_; p3[e] = s3_;
            }
            *reinterpret_cast<bf16x4*>(base + ldsA[j]) = p1;
            *reinterpret_cast<bf16x4*>(base + TA + ldsA[j]) = p2;
            *reinterpret_cast<bf16x4*>(base + 2 * TA + ldsA[j]) = p3;
        }
    };
    const int fslot = (lg ^ s3_swz((lr >> 2) & 3, p.swz_plain)) * 8;
    const int a_off = (wm * 64 + lr) * S3_SK + fslot, b_off = 3 * TA + (wn * 16 * TN + lr) * S3_SK + fslot;
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = (p.K + S3_SK - 1) / S3_SK;
    f32x4 ra[2][4];
    issue_b(0, s3mem);
    load_a(0, ra[0]);
    load_a(1, ra[1]);
    wait_vm<4>();                                                   // B(0) and A(0)
    if constexpr (PRO) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[0][j] = pro4(ra[0][j], 0);
    }
    split_store(ra[0], s3mem);
    load_a(2, ra[0]);
    // Every stage runs the same branch-free code, written in the order it should issue: 24 quads of MFMAs (one n-tile x four m-tiles of one plane product), and
    // between them, in 12 slices, the split of A(s + 1) (two elements per slice), its plane writes and the reload of the freed registers with A(s + 3); the
    // fragment reads of the second and third product group ride under the first.  A scheduling barrier after every quad keeps that order (left to itself the
    // compiler puts all vector work in front of one solid run of 96 MFMAs: ~700 cycles per stage with the matrix pipe idle; sched_group_barrier patterns of
    // this size were honoured for one build and dropped by the next).  Stages beyond K are harmless: their loads / DMAs fail the k-mask (no memory access,
    // zeros), their plane writes go to a slot nobody reads; so the per-stage VMEM count is constant and one counted wait serves all.
    auto stage = [&](int s, f32x4 (&rnext)[4]) {
        // rnext holds A(s + 1) (loaded two stages ago); after its split it is reloaded with A(s + 3)
        bf16* const cur = s3mem + (s & 1) * STAGE;
        bf16* const nxt = s3mem + ((s + 1) & 1) * STAGE;
        wait_vm<4>();                                               // B(s) and A(s + 1) have landed; A(s + 2) may be in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // own plane writes of A(s) are done ...
        __builtin_amdgcn_s_barrier();                               // ... everybody's are, and every wave has read its fragments of stage s - 1: that slot is free
        if (!(ABL & 8)) issue_b(s + 1, nxt);
        bf16x8 xf[3][TM], wf[3][TN];
        if (ABL & 2) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int t = 0; t < 4; ++t) { xf[pl][t] = __builtin_bit_cast(bf16x8, acc[0][t]); if (t < TN) wf[pl][t] = __builtin_bit_cast(bf16x8, acc[1][t]); }
        }
        auto rd_a1 = [&](int pl, int mt) { if (!(ABL & 2)) xf[pl][mt] = *reinterpret_cast<const bf16x8*>(cur + pl * TA + a_off + mt * 16 * S3_SK); };
        auto rd_b1 = [&](int pl, int nt) { if (!(ABL & 2)) wf[pl][nt] = *reinterpret_cast<const bf16x8*>(cur + pl * TB + b_off + nt * 16 * S3_SK); };
#pragma unroll
        for (int t = 0; t < 4; ++t) { rd_a1(0, t); if (t < TN) rd_b1(2, t); }
        const int k3 = (s + 3) * S3_SK;
        const bool kin3 = k3 + kq * 4 < p.K;
        bf16x4 p1, p2, p3;
        constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};      // small terms first: (a1 b3 + a2 b2 + a3 b1), (a1 b2 + a2 b1), a1 b1
#pragma unroll
        for (int g = 0; g < 6; ++g) {
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
                    if (!(ABL & 1)) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[PB[g]][nt], xf[PA[g]][mt], acc[nt][mt], 0, 0, 0);
                    else { acc[nt][mt][0] += (float)wf[PB[g]][nt][0]; acc[nt][mt][1] += (float)xf[PA[g]][mt][0]; }
                }
                constexpr int Q = 6 * TN;                           // quads per stage; the split's 12 slices go to the quads with (12 qi) mod Q < 12
                const int qi = g * TN + nt;
                if (g < 2) {                                        // a2, b2 under the first product group; a3, b1 under the second: three reads per quad
#pragma unroll
                    for (int idx = 3 * nt; idx < 3 * nt + 3; ++idx) {
                        if (idx < 4) rd_a1(g + 1, idx);
                        else if (idx - 4 < TN) rd_b1(1 - g, idx - 4);
                    }
                }
                if ((qi * 12) % Q < 12 && !(ABL & 4)) {
                    const int ms = (qi * 12) / Q, jj = ms / 3, part = ms % 3;
                    if (part < 2) {
                        if (PRO && part == 0) rnext[jj] = pro4(rnext[jj], s + 1);      // (the row's four columns at once: two packed GELU evaluations)
#pragma unroll
                        for (int e = 2 * part; e < 2 * part + 2; ++e) {
                            bf16 s1_, s2_, s3_;
                            gg_split3_rne(rnext[jj][e], s1_, s2_, s3_);
                            p1[e] = s1_; p2[e] = s2_; p3[e] = s3_;
                        }
                    } else {
                        *reinterpret_cast<bf16x4*>(nxt + ldsA[jj]) = p1;
                        *reinterpret_cast<bf16x4*>(nxt + TA + ldsA[jj]) = p2;
                        *reinterpret_cast<bf16x4*>(nxt + 2 * TA + ldsA[jj]) = p3;
                        rnext[jj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(kin3 ? voffA[jj] : 0xFFFFFFF0u), k3 * 4, A_AUX));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int s = 0; s < nk; s += 2) {
        stage(s, ra[1]);
        if (s + 1 < nk) stage(s + 1, ra[0]);
    }
    wait_vm<0>();                                                   // (the masked loads / DMAs of the stages beyond K: nothing may land in the ring after this)
    // epilogue: the accumulators cross the idle ring to whole rows
    float* Ct = reinterpret_cast<float*>(s3mem);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
            *reinterpret_cast<f32x4*>(Ct + (wm * 64 + mt * 16 + lr) * (BN + 4) + wn * 16 * TN + nt * 16 + lg * 4) = acc[nt][mt];
    __syncthreads();
    if constexpr (EC == 0) split3_epilogue_rows<BM, BN, 512, ((ABL & 16) ? 1 : 0) | ((ABL & 64) ? 4 : 0)>(p, Ct, m0, n0);
    else split3_epilogue_rows_ec<BM, BN, 512, ((ABL & 16) ? 1 : 0) | ((ABL & 64) ? 4 : 0), EC>(p, Ct, m0, n0);
}

// The 256 x 128 form on v_mfma_f32_32x32x16_bf16 (the default for 128-column tiles).  Same tile, ring, loader, split and epilogue as gemm_nt_split3a_kernel; a wave's
// 64 x 64 is 2 x 2 tiles of 32 x 32 and a 32-deep stage is two 16-deep MFMA steps: 48 instructions of 32 cycles instead of 96 of 16.  Why: an MFMA holds the SIMD's
// vector issue for 8 cycles whatever its shape (MI355X_MICROARCH.md, cycle constants), and this kernel carries ~125 vector instructions per wave and stage next to its
// MFMAs (the split of A: cvt / shift / subtract, 11 per pair of elements, plus the first-term clamp).  With 16-cycle MFMAs the two waves of a SIMD ask for
// 2 x (96 x 8 + ~125 x 4.5) = 2 650 of the stage's 3 072 issue cycles -- any slip idles the matrix pipe, and it ran at 61-67 % (SQ counters:
// profiles/r06_split_sq_counters.txt: wait_inst 0.46, no LDS conflicts left); with 32-cycle MFMAs the same work asks for 1 900.
// Fragments: lane l holds row (l & 31), k = 8 (l >> 5) + 0..7 of a 32 x 16 operand block; the two 16-byte chunks of a k-step sit at positions (2 ks + (l >> 5)) ^ ((row >> 2) & 3)
// of the row's 64 bytes -- for 32-row fragments the plain XOR is what ds_read_b128's 16-lane service groups need (rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of one chunk:
// row quads {0, 3, 5, 6} / {1, 2, 4, 7} take four different positions).  Result tile: lane holds column m = l & 31, rows n = (r & 3) + 8 (r >> 2) + 4 (l >> 5).
// The k-order of the accumulation differs from the 16 x 16 x 32 kernels' (two 16-deep steps per product instead of one 32-deep): results agree with theirs to rounding, not bit for bit.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int HINT>
__global__ __launch_bounds__(512) void gemm_nt_split3w_kernel(Split3Params p) {
    constexpr int BM = 256, BN = 128, WN = 2, TM = 2, TN = 2;
    constexpr int TA = BM * S3_SK, TB = BN * S3_SK, STAGE = 3 * (TA + TB);
    constexpr int A_AUX = (HINT & 32) ? 2 : 0;
    extern __shared__ __attribute__((aligned(16))) bf16 s3mem[];
    const int tiles = p.tilesM * p.tilesN;
    const int bid = gg_xcd_remap(blockIdx.x, tiles);
    const int tm = bid / p.tilesN, tn = bid % p.tilesN;
    const int m0 = tm * BM, n0 = tn * BN;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l32 = lane & 31, lh = lane >> 5;
    const unsigned rowsA = (unsigned)min(p.M - m0, BM), rowsB = (unsigned)min(p.N - n0, BN);
    // B planes: 16-row slices by LDS-DMA (one per plane and wave), lane -> (row 16 sl + lane / 4, slot lane % 4); the slot holds SOURCE chunk slot ^ ((row >> 2) & 3) = slot ^ (lane >> 4)
    const int dchunk = (lane & 3) ^ (lane >> 4);
    __amdgpu_buffer_rsrc_t rsB[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        rsB[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + i * p.plane_b + (int64_t)n0 * p.ldb), 0, (int)(rowsB * (unsigned)p.ldb * 2u), 0x00020000);
    const unsigned voffB = (unsigned)(wave * 16 + (lane >> 2)) * (unsigned)p.ldb * 2u + dchunk * 16u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Af + (int64_t)m0 * p.ldaf), 0, (int)(rowsA * (unsigned)p.ldaf * 4u), 0x00020000);
    const int kq = threadIdx.x & 7, arow = threadIdx.x >> 3;
    unsigned voffA[4];
    int ldsA[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = arow + 64 * j;
        voffA[j] = (unsigned)row * (unsigned)p.ldaf * 4u + kq * 16u;
        ldsA[j] = row * S3_SK + (((kq >> 1) ^ ((row >> 2) & 3)) << 3) + ((kq & 1) << 2);
    }
    auto issue_b = [&](int st, bf16* base) {
        const int k0 = st * S3_SK;
        const bool kin = k0 + dchunk * 8 < p.K;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB[i], (__attribute__((address_space(3))) void*)(base + 3 * TA + i * TB + wave * 512), 16,
                                                     (int)(kin ? voffB : 0xFFFFFFF0u), k0 * 2, 0, 0);
    };
    auto load_a = [&](int st, f32x4 (&r)[4]) {
        const int k0 = st * S3_SK;
        const bool kin = k0 + kq * 4 < p.K;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(kin ? voffA[j] : 0xFFFFFFF0u), k0 * 4, A_AUX));
    };
    auto split_store = [&](const f32x4 (&r)[4], bf16* base) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x4 p1, p2, p3;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bf16 s1_, s2_, s3_;
                gg_split3_rne(r[j][e], s1_, s2_, s3_);
                p1[e] = s1_; p2[e] = s2_; p3[e] = s3_;
            }
            *reinterpret_cast<bf16x4*>(base + ldsA[j]) = p1;
            *reinterpret_cast<bf16x4*>(base + TA + ldsA[j]) = p2;
            *reinterpret_cast<bf16x4*>(base + 2 * TA + ldsA[j]) = p3;
        }
    };
    // fragment t = 2 tile + ks of a plane: row 32 tile + l32 of the wave's 64, chunk position (2 ks + lh) ^ ((l32 >> 2) & 3)
    const int fsw = (l32 >> 2) & 3;
    const int fpos0 = ((0 + lh) ^ fsw) * 8, fpos1 = ((2 + lh) ^ fsw) * 8;
    const int a_off = (wm * 64 + l32) * S3_SK, b_off = 3 * TA + (wn * 64 + l32) * S3_SK;
    f32x16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int nk = (p.K + S3_SK - 1) / S3_SK;
    f32x4 ra[2][4];
    issue_b(0, s3mem);
    load_a(0, ra[0]);
    load_a(1, ra[1]);
    wait_vm<4>();                                                   // B(0) and A(0)
    split_store(ra[0], s3mem);
    load_a(2, ra[0]);
    // One stage = 24 units of two MFMAs (one 32 x 32 tile of one plane product, both k-steps), in issue order; between them, on every second unit, a slice of the split
    // of A(s + 1) (two elements / the plane writes + the reload with A(s + 3)); the fragment reads of the second and third product group ride under the first and second.
    auto stage = [&](int s, f32x4 (&rnext)[4]) {
        bf16* const cur = s3mem + (s & 1) * STAGE;
        bf16* const nxt = s3mem + ((s + 1) & 1) * STAGE;
        wait_vm<4>();                                               // B(s) and A(s + 1) have landed; A(s + 2) may be in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // own plane writes of A(s) are done ...
        __builtin_amdgcn_s_barrier();                               // ... everybody's are, and every wave has read its fragments of stage s - 1: that slot is free
        issue_b(s + 1, nxt);
        bf16x8 xf[3][4], wf[3][4];
        auto rd_a1 = [&](int pl, int t) { xf[pl][t] = *reinterpret_cast<const bf16x8*>(cur + pl * TA + a_off + (t >> 1) * 32 * S3_SK + ((t & 1) ? fpos1 : fpos0)); };
        auto rd_b1 = [&](int pl, int t) { wf[pl][t] = *reinterpret_cast<const bf16x8*>(cur + pl * TB + b_off + (t >> 1) * 32 * S3_SK + ((t & 1) ? fpos1 : fpos0)); };
#pragma unroll
        for (int t = 0; t < 4; ++t) { rd_a1(0, t); rd_b1(2, t); }
        const int k3 = (s + 3) * S3_SK;
        const bool kin3 = k3 + kq * 4 < p.K;
        bf16x4 p1, p2, p3;
        constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};      // small terms first: (a1 b3 + a2 b2 + a3 b1), (a1 b2 + a2 b1), a1 b1
#pragma unroll
        for (int g = 0; g < 6; ++g) {
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[PB[g]][2 * nt], xf[PA[g]][2 * mt], acc[nt][mt], 0, 0, 0);
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[PB[g]][2 * nt + 1], xf[PA[g]][2 * mt + 1], acc[nt][mt], 0, 0, 0);
                    const int u = nt * TM + mt, qi = g * 4 + u;     // unit 0 .. 23
                    if (g < 2) {                                    // a2, b2 under the first product group; a3, b1 under the second: two reads per unit
#pragma unroll
                        for (int idx = 2 * u; idx < 2 * u + 2; ++idx) {
                            if (idx < 4) rd_a1(g + 1, idx);
                            else rd_b1(1 - g, idx - 4);
                        }
                    }
                    if ((qi & 1) == 0) {                            // the split's 12 slices on the even units
                        const int ms = qi >> 1, jj = ms / 3, part = ms % 3;
                        if (part < 2) {
#pragma unroll
                            for (int e = 2 * part; e < 2 * part + 2; ++e) {
                                bf16 s1_, s2_, s3_;
                                gg_split3_rne(rnext[jj][e], s1_, s2_, s3_);
                                p1[e] = s1_; p2[e] = s2_; p3[e] = s3_;
                            }
                        } else {
                            *reinterpret_cast<bf16x4*>(nxt + ldsA[jj]) = p1;
                            *reinterpret_cast<bf16x4*>(nxt + TA + ldsA[jj]) = p2;
                            *reinterpret_cast<bf16x4*>(nxt + 2 * TA + ldsA[jj]) = p3;
                            rnext[jj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(kin3 ? voffA[jj] : 0xFFFFFFF0u), k3 * 4, A_AUX));
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    for (int s = 0; s < nk; s += 2) {
        stage(s, ra[1]);
        if (s + 1 < nk) stage(s + 1, ra[0]);
    }
    wait_vm<0>();                                                   // (the masked loads / DMAs of the stages beyond K: nothing may land in the ring after this)
    // epilogue: the accumulators cross the idle ring to whole rows (lane: column m = l32 of the tile, rows n = 8 q + 4 lh + 0..3)
    float* Ct = reinterpret_cast<float*>(s3mem);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4*>(Ct + (wm * 64 + mt * 32 + l32) * (BN + 4) + wn * 64 + nt * 32 + 8 * q + 4 * lh) =
                    (f32x4){acc[nt][mt][4 * q], acc[nt][mt][4 * q + 1], acc[nt][mt][4 * q + 2], acc[nt][mt][4 * q + 3]};
    __syncthreads();
    split3_epilogue_rows<BM, BN, 512, ((HINT & 16) ? 1 : 0) | ((HINT & 64) ? 4 : 0)>(p, Ct, m0, n0);
}

// The same product on a 128 x 128 tile with FOUR waves (2 x 2 of 64 x 64) and 72 KB of LDS, so that TWO workgroups share a CU: with one 144 KB workgroup per CU
// nothing hides a tile's prologue and -- worse -- its epilogue (the GELU / GELU' epilogues of fc1 / the fc2 data gradient are 8 us of vector work per 256 x 128
// tile next to 11 us of MFMAs at K = 384: in the model those launches ran 20-45 % slower than the bias-only shape).  LDS: the A planes single-buffered (24 KB:
// a wave reads ALL its A fragments of a stage, a second barrier frees the buffer, and the split of A(s + 1) is written into it under the stage's MFMAs), the B
// planes double-buffered by LDS-DMA (2 x 24 KB).  Everything else as above.
template <int TN_ = 4, int EC = 0>
__global__ __launch_bounds__(256) void gemm_nt_split3b_kernel(Split3Params p) {
    constexpr int TN = TN_, BM = 128, BN = 32 * TN, WN = 2, TM = 4;
    constexpr int TA = BM * S3_SK, TB = BN * S3_SK;
    extern __shared__ __attribute__((aligned(16))) bf16 s3mem[];
    bf16* const Abuf = s3mem;                                      // [3][TA]
    bf16* const Bbuf = s3mem + 3 * TA;                             // [2][3][TB]
    const int tiles = p.tilesM * p.tilesN;
    const int bid = gg_xcd_remap(blockIdx.x, tiles);
    const int tm = bid / p.tilesN, tn = bid % p.tilesN;
    const int m0 = tm * BM, n0 = tn * BN;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 15, lg = lane >> 4;
    const unsigned rowsA = (unsigned)min(p.M - m0, BM), rowsB = (unsigned)min(p.N - n0, BN);
    const int dchunk = (lane & 3) ^ s3_swz(lane >> 4, p.swz_plain);
    __amdgpu_buffer_rsrc_t rsB[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        rsB[i] = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + i * p.plane_b + (int64_t)n0 * p.ldb), 0, (int)(rowsB * (unsigned)p.ldb * 2u), 0x00020000);
    unsigned voffB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) voffB[j] = (unsigned)((wave + 4 * j) * 16 + (lane >> 2)) * (unsigned)p.ldb * 2u + dchunk * 16u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Af + (int64_t)m0 * p.ldaf), 0, (int)(rowsA * (unsigned)p.ldaf * 4u), 0x00020000);
    const int kq = threadIdx.x & 7, arow = threadIdx.x >> 3;
    unsigned voffA[4];
    int ldsA[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = arow + 32 * j;
        voffA[j] = (unsigned)row * (unsigned)p.ldaf * 4u + kq * 16u;
        ldsA[j] = row * S3_SK + (((kq >> 1) ^ s3_swz((row >> 2) & 3, p.swz_plain)) << 3) + ((kq & 1) << 2);
    }
    auto issue_b = [&](int st) {
        bf16* const base = Bbuf + (st & 1) * 3 * TB;
        const int k0 = st * S3_SK;
        const bool kin = k0 + dchunk * 8 < p.K;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (wave + 4 * j < BN / 16)                       // (wave-uniform)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB[i], (__attribute__((address_space(3))) void*)(base + i * TB + (wave + 4 * j) * 512), 16,
                                                             (int)(kin ? voffB[j] : 0xFFFFFFF0u), k0 * 2, 0, 0);
    };
    auto load_a = [&](int st, f32x4 (&r)[4]) {
        const int k0 = st * S3_SK;
        const bool kin = k0 + kq * 4 < p.K;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(kin ? voffA[j] : 0xFFFFFFF0u), k0 * 4, 0));
    };
    auto split2 = [&](const f32x4& v, int e0, bf16x4& p1, bf16x4& p2, bf16x4& p3) {
#pragma unroll
        for (int e = e0; e < e0 + 2; ++e) {
            bf16 s1_, s2_, s3_;
            gg_split3_rne(v[e], s1_, s2_, s3_);
            p1[e] = s1_; p2[e] = s2_; p3[e] = s3_;
        }
    };
    const int fslot = (lg ^ s3_swz((lr >> 2) & 3, p.swz_plain)) * 8;
    const int a_off = (wm * 64 + lr) * S3_SK + fslot, b_off = (wn * 16 * TN + lr) * S3_SK + fslot;
    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = (p.K + S3_SK - 1) / S3_SK;
    f32x4 ra[2][4];
    issue_b(0);
    load_a(0, ra[0]);
    load_a(1, ra[1]);
    wait_vm<4>();                                                   // B(0) and A(0)
    {
        bf16x4 p1, p2, p3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            split2(ra[0][j], 0, p1, p2, p3); split2(ra[0][j], 2, p1, p2, p3);
            *reinterpret_cast<bf16x4*>(Abuf + ldsA[j]) = p1; *reinterpret_cast<bf16x4*>(Abuf + TA + ldsA[j]) = p2; *reinterpret_cast<bf16x4*>(Abuf + 2 * TA + ldsA[j]) = p3;
        }
    }
    load_a(2, ra[0]);
    auto stage = [&](int s, f32x4 (&rnext)[4]) {
        const bf16* const curB = Bbuf + (s & 1) * 3 * TB;
        wait_vm<4>();                                               // B(s) and A(s + 1) have landed; A(s + 2) may be in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // own plane writes of A(s) are done ...
        __builtin_amdgcn_s_barrier();                               // ... everybody's are; everybody's B(s) is there; every wave is past its B reads of stage s - 1
        issue_b(s + 1);
        bf16x8 xf[3][TM], wf[3][TN];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) xf[pl][mt] = *reinterpret_cast<const bf16x8*>(Abuf + pl * TA + a_off + mt * 16 * S3_SK);
        auto rd_b1 = [&](int pl, int nt) { wf[pl][nt] = *reinterpret_cast<const bf16x8*>(curB + pl * TB + b_off + nt * 16 * S3_SK); };
#pragma unroll
        for (int t = 0; t < TN; ++t) rd_b1(2, t);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the twelve A fragments (and b3) are in registers ...
        __builtin_amdgcn_s_barrier();                               // ... everybody's are: the A buffer is free for the planes of A(s + 1)
        const int k3 = (s + 3) * S3_SK;
        const bool kin3 = k3 + kq * 4 < p.K;
        bf16x4 p1, p2, p3;
        constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
        for (int g = 0; g < 6; ++g) {
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[PB[g]][nt], xf[PA[g]][mt], acc[nt][mt], 0, 0, 0);
                constexpr int Q = 6 * TN;
                const int qi = g * TN + nt;
                if (g < 2 && 2 * nt < TN) { rd_b1(1 - g, 2 * nt); if (2 * nt + 1 < TN) rd_b1(1 - g, 2 * nt + 1); }      // b2 under the first product group, b1 under the second
                if ((qi * 12) % Q < 12) {
                    const int ms = (qi * 12) / Q, jj = ms / 3, part = ms % 3;
                    if (part < 2) split2(rnext[jj], 2 * part, p1, p2, p3);
                    else {
                        *reinterpret_cast<bf16x4*>(Abuf + ldsA[jj]) = p1;
                        *reinterpret_cast<bf16x4*>(Abuf + TA + ldsA[jj]) = p2;
                        *reinterpret_cast<bf16x4*>(Abuf + 2 * TA + ldsA[jj]) = p3;
                        rnext[jj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(kin3 ? voffA[jj] : 0xFFFFFFF0u), k3 * 4, 0));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int s = 0; s < nk; s += 2) {
        stage(s, ra[1]);
        if (s + 1 < nk) stage(s + 1, ra[0]);
    }
    wait_vm<0>();
    float* Ct = reinterpret_cast<float*>(s3mem);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
            *reinterpret_cast<f32x4*>(Ct + (wm * 64 + mt * 16 + lr) * (BN + 4) + wn * 16 * TN + nt * 16 + lg * 4) = acc[nt][mt];
    __syncthreads();
    if constexpr (EC == 0) split3_epilogue_rows<BM, BN, 256, 1>(p, Ct, m0, n0);      // (non-temporal result stores, as the 256 x 128 form)
    else split3_epilogue_rows_ec<BM, BN, 256, 1, EC>(p, Ct, m0, n0);
}

// ------------------------------------------------------------------------------------------- weight gradient (TN), both operands f32
// dW[N][K] = sum_m s_m dY[m][n] X[m][k]  (s = the DropPath row scale or 1): the contraction runs over the ROWS of both operands, so both are split in the loader
// and staged as plane images [32 m][columns] (sub-images of 32 columns: 64-byte rows, chunk swizzle (-(row >> 2)) & 3 as in attention_split.h), and the
// fragments -- 8 consecutive m of one column per lane -- come out of LDS through the transposing read ds_read_tr16_b64 (two per fragment).  Tile 256 (n) x 128
// (k), 4 x 2 waves of 64 x 64, 32 rows per stage, 2-stage ring of 72 KB; D = X^T-fragment x dY^T-fragment, so a lane owns 4 consecutive k of one n (16-byte
// stores).  The rows are cut into `splits` slabs (one workgroup per tile and slab), partial tiles go to the slab scratch and gg_splitk_reduce adds them in slab
// order -- the protocol of gg_gemm_tn_f32, whose place this kernel takes in the fp32_split mode.
typedef short s3_s4 __attribute__((ext_vector_type(4)));
struct Split3TnParams {
    const float* dY; int64_t ldy; const float* X; int64_t ldx;
    int M, N, K; const float* rowscale; int rps;
    float* part; int rows_per_split, tilesN, tilesK;
    int last_scale;                     // (M - 1) / rps: the row-scale index is clamped to it (rows beyond M -- zeros -- in a slab's last stages must not read past the array)
};
constexpr int TN3_SUB = 1024 + 32;                                 // elements of one 32-column sub-image of the TN kernel's plane images (skewed: see the kernel)
__device__ __forceinline__ int s3_img_off(int row, int ch) { return row * 32 + ((ch ^ ((-(row >> 2)) & 3)) << 3); }
template <int SUB_, bool REMAP>
__global__ __launch_bounds__(512) void gemm_tn_split3_kernel(Split3TnParams p) {
    constexpr int BN = 256, BK = 128, WK = 2;
    // plane images of one stage (bf16 elements): dY then X, as sub-images of 32 columns x 32 rows (SUB elements each: 1024, or 1024 + 32 -- the 64-byte skew that puts the
    // two sub-images a 16-lane group of a plane write touches on different halves of the 128-byte bank window; at a pitch of exactly 2 KB every plane write is a 2-way
    // conflict, 576 LDS cycles per stage, and yet the unskewed image is the faster one: see gg_gemm_tn_split3)
    constexpr int SUB = SUB_;
    constexpr int PY = (BN / 32) * SUB, PX = (BK / 32) * SUB, STAGE = 3 * (PY + PX);
    extern __shared__ __attribute__((aligned(16))) bf16 s3mem[];
    // XCD-contiguous walk: the tiles of one row slab are consecutive logical ids, so they run on ONE XCD at about the same time and its L2 serves the slab's dY / X rows to
    // all of them (in dispatch order the K-tiles of a slab sat on different XCDs: dY was fetched once per K-tile and X once per N-tile from HBM, 3-4 x the algorithmic bytes)
    const int tiles = p.tilesN * p.tilesK;
    const int lid = REMAP ? gg_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int t = lid % tiles, slab = lid / tiles;
    const int tn = t / p.tilesK, tk = t % p.tilesK;
    const int n0 = tn * BN, k0 = tk * BK;
    const int m_begin = slab * p.rows_per_split, rows = min(p.M - m_begin, p.rows_per_split);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WK, wk = wave % WK;                       // wave: n range 64 wm, k range 64 wk
    const int lr = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dY + (int64_t)m_begin * p.ldy), 0, (int)((unsigned)rows * (unsigned)p.ldy * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(p.X + (int64_t)m_begin * p.ldx), 0, (int)((unsigned)rows * (unsigned)p.ldx * 4u), 0x00020000);
    // loader: dY f32x4 number c4 of rows (tid >> 6) + 8 j (j < 4); X f32x4 number c4 of rows (tid >> 5) + 16 j (j < 2); columns beyond N / K are masked
    unsigned voff[6];
    int ldso[6];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (threadIdx.x >> 6) + 8 * j, col = (threadIdx.x & 63) * 4;
        voff[j] = (n0 + col < p.N) ? (unsigned)row * (unsigned)p.ldy * 4u + (unsigned)(n0 + col) * 4u : 0xFFFFFFF0u;
        ldso[j] = (col >> 5) * SUB + s3_img_off(row, (col & 31) >> 3) + (col & 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (threadIdx.x >> 5) + 16 * j, col = (threadIdx.x & 31) * 4;
        voff[4 + j] = (k0 + col < p.K) ? (unsigned)row * (unsigned)p.ldx * 4u + (unsigned)(k0 + col) * 4u : 0xFFFFFFF0u;
        ldso[4 + j] = 3 * PY + (col >> 5) * SUB + s3_img_off(row, (col & 31) >> 3) + (col & 4);
    }
    // DropPath row scale of the four dY rows of this thread: index (m / rps) kept as a quotient / remainder pair that advances by 32 rows per stage
    int sq[4], sr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m_begin + (int)(threadIdx.x >> 6) + 8 * j;
        sq[j] = p.rowscale ? m / p.rps : 0; sr[j] = p.rowscale ? m % p.rps : 0;
    }
    const int nk = (rows + 31) >> 5;
    auto load = [&](int st, f32x4 (&r)[6]) {
        // (rows beyond the slab fail the descriptor's range check: zeros)
        const unsigned sy = (unsigned)st * 32u * (unsigned)p.ldy * 4u, sx = (unsigned)st * 32u * (unsigned)p.ldx * 4u;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsY, (int)(st < nk ? voff[j] : 0xFFFFFFF0u), (int)sy, 0));
#pragma unroll
        for (int j = 0; j < 2; ++j) r[4 + j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(st < nk ? voff[4 + j] : 0xFFFFFFF0u), (int)sx, 0));
    };
    auto split2 = [&](const f32x4& v, int e0, bf16x4& p1, bf16x4& p2, bf16x4& p3) {
#pragma unroll
        for (int e = e0; e < e0 + 2; ++e) {
            bf16 s1_, s2_, s3_;
            gg_split3_rne(v[e], s1_, s2_, s3_);
            p1[e] = s1_; p2[e] = s2_; p3[e] = s3_;
        }
    };
    auto scale_rows = [&](f32x4 (&r)[6]) {                          // the scales of the rows just loaded; then the counters move on to the rows two stages later
        if (!p.rowscale) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] *= p.rowscale[min(sq[j], p.last_scale)];
    };
    auto advance_rows = [&]() {
        if (!p.rowscale) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) { sr[j] += 32; while (sr[j] >= p.rps) { sr[j] -= p.rps; ++sq[j]; } }
    };
    auto store_planes = [&](const f32x4 (&r)[6], bf16* base) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            bf16x4 p1, p2, p3;
            split2(r[j], 0, p1, p2, p3); split2(r[j], 2, p1, p2, p3);
            const int pl = j < 4 ? PY : PX;
            *reinterpret_cast<bf16x4*>(base + ldso[j]) = p1; *reinterpret_cast<bf16x4*>(base + pl + ldso[j]) = p2; *reinterpret_cast<bf16x4*>(base + 2 * pl + ldso[j]) = p3;
        }
    };
    // fragment addresses (element offsets within a plane; + 16 rows = + 512): lane (lr, lg) supplies row 4 lg + (lr >> 2), columns 4 (lr & 3) .. of the 16-column block
    int yoff[4], xoff[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        const int row = 4 * lg + (lr >> 2);
        const int cy = 64 * wm + 16 * tt + 4 * (lr & 3), cx = 64 * wk + 16 * tt + 4 * (lr & 3);
        yoff[tt] = (cy >> 5) * SUB + s3_img_off(row, (cy & 31) >> 3) + (cy & 7);
        xoff[tt] = 3 * PY + (cx >> 5) * SUB + s3_img_off(row, (cx & 31) >> 3) + (cx & 7);
    }
    f32x4 acc[4][4];                                                // [n-tile][k-tile]; lane: n = lr, k = 4 lg + r
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ra[2][6];
    load(0, ra[0]);
    load(1, ra[1]);
    wait_vm<6>();
    scale_rows(ra[0]); advance_rows();
    store_planes(ra[0], s3mem);
    load(2, ra[0]);
    auto stage = [&](int s, f32x4 (&rnext)[6]) {
        // rnext holds the rows of stage s + 1 (loaded two stages ago); after their split it is reloaded with stage s + 3
        const bf16* const cur = s3mem + (s & 1) * STAGE;
        bf16* const nxt = s3mem + ((s + 1) & 1) * STAGE;
        wait_vm<6>();                                               // the rows of stage s + 1 have landed; those of stage s + 2 may be in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                               // everybody's planes of stage s are written, everybody's fragments of stage s - 1 are read
        union Fr { s3_s4 h[2]; bf16x8 v; };
        Fr xf[3][4], yf[3][4];                                      // X^T fragments per k-tile (A operand: i = k), dY^T fragments per n-tile (B operand: j = n)
        auto rd = [&](bool isx, int pl, int tt, int half) {
            const bf16* a = cur + (isx ? pl * PX + xoff[tt] : pl * PY + yoff[tt]) + half * 512;
            (isx ? xf[pl][tt] : yf[pl][tt]).h[half] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s3_s4*)a);
        };
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) { rd(true, 0, tt, 0); rd(true, 0, tt, 1); rd(false, 2, tt, 0); rd(false, 2, tt, 1); }
        scale_rows(rnext); advance_rows();
        bf16x4 p1, p2, p3;
        constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};      // small terms first
#pragma unroll
        for (int g = 0; g < 6; ++g) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[PA[g]][kt].v, yf[PB[g]][nt].v, acc[nt][kt], 0, 0, 0);
                const int qi = g * 4 + nt;
                if (g < 2) {                                        // x2, dy2 under the first product group; x3, dy1 under the second: four reads per quad
#pragma unroll
                    for (int idx = 4 * nt; idx < 4 * nt + 4; ++idx) {
                        if (idx < 8) rd(true, g + 1, idx >> 1, idx & 1);
                        else rd(false, 1 - g, (idx - 8) >> 1, idx & 1);
                    }
                }
                if ((qi * 18) % 24 < 18) {                          // the split's 18 slices on 18 of the 24 quads
                    const int ms = (qi * 18) / 24, jj = ms / 3, part = ms % 3;
                    if (part < 2) split2(rnext[jj], 2 * part, p1, p2, p3);
                    else {
                        const int pl = jj < 4 ? PY : PX;
                        *reinterpret_cast<bf16x4*>(nxt + ldso[jj]) = p1; *reinterpret_cast<bf16x4*>(nxt + pl + ldso[jj]) = p2; *reinterpret_cast<bf16x4*>(nxt + 2 * pl + ldso[jj]) = p3;
                        const int st3 = s + 3;
                        if (jj < 4) rnext[jj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsY, (int)(st3 < nk ? voff[jj] : 0xFFFFFFF0u), (int)((unsigned)st3 * 32u * (unsigned)p.ldy * 4u), 0));
                        else rnext[jj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(st3 < nk ? voff[jj] : 0xFFFFFFF0u), (int)((unsigned)st3 * 32u * (unsigned)p.ldx * 4u), 0));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int s = 0; s < nk; s += 2) {
        stage(s, ra[1]);
        if (s + 1 < nk) stage(s + 1, ra[0]);
    }
    wait_vm<0>();
    float* out = p.part + (int64_t)slab * p.N * p.K;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + 64 * wm + 16 * nt + lr;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int k = k0 + 64 * wk + 16 * kt + 4 * lg;
            if (n < p.N && k < p.K) *reinterpret_cast<f32x4*>(out + (int64_t)n * p.K + k) = acc[nt][kt];
        }
    }
}

// x (f32, [rows][ldx]) -> planes [3][rows][cols] bf16: x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, int64_t rows, int cols, int64_t ldx, bf16* __restrict__ out) {
    const int64_t n4 = rows * (cols / 4);
    const int64_t plane = rows * cols;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / (cols / 4);
        const int c = (int)(i - r * (cols / 4)) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
        bf16x4 p1, p2, p3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16 s1_, s2_, s3_;
            gg_split3_rne(v[j], s1_, s2_, s3_);
            p1[j] = s1_; p2[j] = s2_; p3[j] = s3_;
        }
        *reinterpret_cast<bf16x4*>(out + r * cols + c) = p1;
        *reinterpret_cast<bf16x4*>(out + plane + r * cols + c) = p2;
        *reinterpret_cast<bf16x4*>(out + 2 * plane + r * cols + c) = p3;
    }
}

}  // namespace

extern "C" int gg_split3_bf16(const float* x, int64_t rows, int cols, int64_t ldx, void* planes, void* stream) {
    GG_CHECK(x && planes && rows > 0 && cols > 0 && (cols & 3) == 0 && (ldx & 3) == 0 && ldx >= cols && ((uintptr_t)x & 15) == 0 && ((uintptr_t)planes & 7) == 0,
             "gg_split3_bf16: bad args (cols %% 4, ldx %% 4, 16-byte aligned x)");
    GG_PROF(GG_CAT_MOVE, 0, 10.0 * rows * cols, stream);
    const int64_t n4 = rows * (cols / 4);
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)std::min<int64_t>(gg_cdiv(n4, 256), 16384)), dim3(256), 0, (hipStream_t)stream, x, rows, cols, ldx, (bf16*)planes);
    GG_LAUNCH_CHECK();
    return 0;
}

static int split3_launch(Split3Params& p, void* stream) {
    const int M = p.M, N = p.N, K = p.K;
    p.tilesM = (int)gg_cdiv(M, S3_BM); p.tilesN = (int)gg_cdiv(N, S3_BN);
    // form: 0 = 128 x 128 tile, 2 x 4 waves, fragments read before the MFMAs, 3-stage ring; 1 = the same, software-pipelined; 2 = 2 x 2 waves (64 x 64 per
    // wave), pipelined; 3 = form 0 as persistent workgroups with the next tile's first stages issued before the epilogue; 4 = 256 x 128 tile, 4 x 2 waves of
    // 64 x 64 (96 MFMAs per 24 fragment reads, two waves per SIMD), 2-stage ring (default)
    static const char* fenv = gg_dev_env("GG_SPLIT3_FORM");
    const int form = fenv ? atoi(fenv) : (M > 128 ? 4 : 0);      // default: the 256 x 128 tile (64 x 64 per wave), the 128 x 128 one for a single row of tiles
    const int fi = form >= 0 && form <= 4 ? form : 0;
    void (*kern)(Split3Params) = fi == 4 ? gemm_nt_split3_kernel<256, 128, 4, 2, false, 2> : fi == 3 ? gemm_nt_split3_persistent_kernel :
                                 fi == 2 ? gemm_nt_split3_kernel<128, 128, 2, 2, true, 3> : fi == 1 ? gemm_nt_split3_kernel<128, 128, 2, 4, true, 3> :
                                           gemm_nt_split3_kernel<128, 128, 2, 4, false, 3>;
    const int bm = fi == 4 ? 256 : 128;
    p.tilesM = (int)gg_cdiv(M, bm);
    const size_t lds = fi == 4 ? (size_t)2 * 3 * (256 + 128) * S3_SK * sizeof(bf16) : (size_t)3 * S3_STAGE * sizeof(bf16);
    static bool raised[5] = {false, false, false, false, false};
    if (!raised[fi]) {
        GG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess,
                 "gg_gemm_nt_split3: cannot raise the dynamic LDS limit");
        raised[fi] = true;
    }
    // algorithmic work = the fp32 product it replaces: 2 M N K flop; bytes: three bf16 planes per operand + the result (+ the epilogue's tensors)
    const double mn = (double)M * N;
    GG_PROF(GG_CAT_GEMM | GG_CAT_SPLIT_FLAG, 2.0 * M * (double)N * K,
            6.0 * ((double)M * K + (double)N * K) + 4.0 * mn * ((p.C != nullptr) + (p.preact != nullptr) + (p.residual != nullptr) + (p.dact_preact != nullptr)) +
                (p.c_planes ? 6.0 * mn : 0.0), stream);
    hipLaunchKernelGGL(kern, dim3((unsigned)(fi == 3 ? std::min(p.tilesM * p.tilesN, 256) : p.tilesM * p.tilesN)), dim3(fi == 2 ? 256 : 512), lds, (hipStream_t)stream, p);
    GG_LAUNCH_CHECK();
    return 0;
}

extern "C" int gg_gemm_nt_split3_ex(const GgSplit3Args* a, void* stream) {
    GG_CHECK(a && a->a_planes && a->b_planes && (a->C || a->c_planes) && a->M > 0 && a->N > 0 && a->K > 0, "gg_gemm_nt_split3: null pointer / bad shape");
    GG_CHECK((a->K & 7) == 0 && (a->lda & 7) == 0 && (a->ldb & 7) == 0 && a->lda >= a->K && a->ldb >= a->K, "gg_gemm_nt_split3: K, lda, ldb must be multiples of 8, ld >= K");
    GG_CHECK(((uintptr_t)a->a_planes & 15) == 0 && ((uintptr_t)a->b_planes & 15) == 0 && ((uintptr_t)a->C & 15) == 0 && ((uintptr_t)a->c_planes & 7) == 0,
             "gg_gemm_nt_split3: alignment");
    GG_CHECK((int64_t)256 * a->lda * 2 < ((int64_t)1 << 31) && (int64_t)256 * a->ldb * 2 < ((int64_t)1 << 31), "gg_gemm_nt_split3: row pitch too large");
    GG_CHECK((!a->C || a->ldc >= a->N) && (!a->c_planes || a->ldp >= a->N) && (!a->residual || a->ldr >= a->N), "gg_gemm_nt_split3: leading dimension too small");
    GG_CHECK((!a->preact && !a->dact_preact) || a->ldc >= a->N, "gg_gemm_nt_split3: preact / dact_preact use ldc");
    GG_CHECK(!a->rowscale || a->rows_per_scale > 0, "gg_gemm_nt_split3: rowscale needs rows_per_scale");
    GG_CHECK(!(a->dact_preact && a->act), "gg_gemm_nt_split3: act and dact_preact are exclusive");
    Split3Params p;
    p.swz_plain = 0;
    p.a_stat = p.a_gamma = p.a_beta = nullptr; p.a_act = 0;
    p.A = (const bf16*)a->a_planes; p.lda = a->lda; p.plane_a = (int64_t)a->M * a->lda; p.Af = nullptr; p.ldaf = 0;
    p.B = (const bf16*)a->b_planes; p.ldb = a->ldb; p.plane_b = (int64_t)a->N * a->ldb;
    p.C = a->C; p.ldc = a->ldc ? a->ldc : a->N; p.bias = a->bias; p.M = a->M; p.N = a->N; p.K = a->K;
    p.act = a->act; p.preact = a->preact; p.rowscale = a->rowscale; p.rows_per_scale = a->rows_per_scale; p.residual = a->residual; p.ldr = a->ldr;
    p.dact_preact = a->dact_preact; p.dact = a->dact; p.c_planes = (bf16*)a->c_planes; p.ldp = a->ldp; p.colstats = nullptr;
    return split3_launch(p, stream);
}

extern "C" int gg_gemm_nt_split3(const void* a_planes, int64_t lda, const void* b_planes, int64_t ldb, float* C, int64_t ldc, int M, int N, int K,
                                 const float* bias, void* stream) {
    GgSplit3Args a;
    memset(&a, 0, sizeof(a));
    a.a_planes = a_planes; a.lda = lda; a.b_planes = b_planes; a.ldb = ldb; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.bias = bias;
    return gg_gemm_nt_split3_ex(&a, stream);
}

// A as f32 [M][lda] (split in the kernel's loader), B as planes b_plane_stride elements apart (0: N * ldb): args->a_planes / lda are ignored
// colstats (optional; plain epilogue only): BatchNorm partials [ceil(M / 128)][2][N] as gg_gemm_nt_f32 writes them (GgGemmArgs.colstats)
extern "C" int gg_gemm_nt_split3_af32_stats(const GgSplit3Args* a, const float* A, int64_t lda, int64_t b_plane_stride, float* colstats, void* stream);
extern "C" int gg_gemm_nt_split3_af32(const GgSplit3Args* a, const float* A, int64_t lda, int64_t b_plane_stride, void* stream) {
    return gg_gemm_nt_split3_af32_stats(a, A, lda, b_plane_stride, nullptr, stream);
}
static int split3_af32_launch(const GgSplit3Args* a, const float* A, int64_t lda, int64_t b_plane_stride, float* colstats, const float* bn_stat, const float* bn_gamma,
                              const float* bn_beta, int bn_act, void* stream);
extern "C" int gg_gemm_nt_split3_af32_stats(const GgSplit3Args* a, const float* A, int64_t lda, int64_t b_plane_stride, float* colstats, void* stream) {
    return split3_af32_launch(a, A, lda, b_plane_stride, colstats, nullptr, nullptr, nullptr, 0, stream);
}
// A := act(BatchNorm(A)) formed in the loader (bn_stat = [mean | rstd][K] as gg_bn_finalize leaves it; act GG_ACT_NONE / GELU / QUICK_GELU): the split twin of
// GgGemmArgs.a_bn_* (gg_gemm_nt_f32's BatchNorm prologue).  K >= 384 (the 256 x 128 form), K <= 1024, plain epilogue (optionally with colstats)
extern "C" int gg_gemm_nt_split3_af32_pro(const GgSplit3Args* a, const float* A, int64_t lda, int64_t b_plane_stride, const float* bn_stat, const float* bn_gamma,
                                          const float* bn_beta, int bn_act, float* colstats, void* stream) {
    GG_CHECK(bn_stat && bn_gamma && bn_beta, "gg_gemm_nt_split3_af32_pro: the BatchNorm prologue needs stat, gamma, beta");
    GG_CHECK(a && a->K >= 384 && a->K <= 1024, "gg_gemm_nt_split3_af32_pro: K must be 384 ... 1024");
    GG_CHECK(bn_act >= 0 && bn_act <= 2, "gg_gemm_nt_split3_af32_pro: act must be GG_ACT_NONE, GG_ACT_GELU or GG_ACT_QUICK_GELU");
    GG_CHECK(!(a->bias || a->act || a->rowscale || a->residual || a->dact_preact || a->preact || a->c_planes), "gg_gemm_nt_split3_af32_pro: plain epilogue only");
    return split3_af32_launch(a, A, lda, b_plane_stride, colstats, bn_stat, bn_gamma, bn_beta, bn_act, stream);
}
static int split3_af32_launch(const GgSplit3Args* a, const float* A, int64_t lda, int64_t b_plane_stride, float* colstats, const float* bn_stat, const float* bn_gamma,
                              const float* bn_beta, int bn_act, void* stream) {
    GG_CHECK(a && A && a->b_planes && (a->C || a->c_planes) && a->M > 0 && a->N > 0 && a->K > 0, "gg_gemm_nt_split3_af32: null pointer / bad shape");
    GG_CHECK((a->K & 7) == 0 && (lda & 3) == 0 && (a->ldb & 7) == 0 && lda >= a->K && a->ldb >= a->K, "gg_gemm_nt_split3_af32: K %% 8, lda %% 4, ldb %% 8, ld >= K");
    GG_CHECK(((uintptr_t)A & 15) == 0 && ((uintptr_t)a->b_planes & 15) == 0 && ((uintptr_t)a->C & 15) == 0 && ((uintptr_t)a->c_planes & 7) == 0, "gg_gemm_nt_split3_af32: alignment");
    GG_CHECK((int64_t)256 * lda * 4 < ((int64_t)1 << 31) && (int64_t)256 * a->ldb * 2 < ((int64_t)1 << 31), "gg_gemm_nt_split3_af32: row pitch too large");
    GG_CHECK((!a->C || a->ldc >= a->N) && (!a->c_planes || a->ldp >= a->N) && (!a->residual || a->ldr >= a->N), "gg_gemm_nt_split3_af32: leading dimension too small");
    GG_CHECK((!a->preact && !a->dact_preact) || a->ldc >= a->N, "gg_gemm_nt_split3_af32: preact / dact_preact use ldc");
    GG_CHECK(!a->rowscale || a->rows_per_scale > 0, "gg_gemm_nt_split3_af32: rowscale needs rows_per_scale");
    GG_CHECK(!(a->dact_preact && a->act), "gg_gemm_nt_split3_af32: act and dact_preact are exclusive");
    GG_CHECK(!colstats || !(a->bias || a->act || a->rowscale || a->residual || a->dact_preact || a->preact || a->c_planes), "gg_gemm_nt_split3_af32: colstats needs the plain epilogue");
    GG_CHECK(((uintptr_t)a->preact & 15) == 0 && ((uintptr_t)a->residual & 15) == 0 && ((uintptr_t)a->dact_preact & 15) == 0,
             "gg_gemm_nt_split3_af32: preact, residual and dact_preact must be 16-byte aligned (the row epilogue moves them 16 bytes at a time)");
    Split3Params p;
    static const char* senv = gg_dev_env("GG_SPLIT3_SWZ");      // dev A/B: 0 = the plain-XOR chunk swizzle of round 5
    p.swz_plain = senv && atoi(senv) == 0;
    p.colstats = colstats;
    p.a_stat = bn_stat; p.a_gamma = bn_gamma; p.a_beta = bn_beta; p.a_act = bn_act;
    const bool pro = bn_stat != nullptr;
    p.A = nullptr; p.lda = 0; p.plane_a = 0; p.Af = A; p.ldaf = lda;
    p.B = (const bf16*)a->b_planes; p.ldb = a->ldb; p.plane_b = b_plane_stride > 0 ? b_plane_stride : (int64_t)a->N * a->ldb;
    p.C = a->C; p.ldc = a->ldc ? a->ldc : a->N; p.bias = a->bias; p.M = a->M; p.N = a->N; p.K = a->K;
    p.act = a->act; p.preact = a->preact; p.rowscale = a->rowscale; p.rows_per_scale = a->rows_per_scale; p.residual = a->residual; p.ldr = a->ldr;
    p.dact_preact = a->dact_preact; p.dact = a->dact; p.c_planes = (bf16*)a->c_planes; p.ldp = a->ldp;
    // 256 x 128 (one 144 KB workgroup per CU) from K = 384 on; 128 x 128 (two per CU) for the short contractions of stage 1, where a tile lives only 6 stages and
    // the second workgroup hides its prologue / epilogue (K = 192: 1.30 x against 1.22 x the f32-MFMA GEMM; at K >= 384 the big tile wins by 1-2 %)
    static const char* tenv = gg_dev_env("GG_SPLIT3A_TILE");      // dev: 256 / 128 forces one form
    const bool big = pro || (tenv ? atoi(tenv) == 256 : p.K >= 384);
    // 96-column tiles where 128-column ones would waste a fifth or more of their work on columns beyond N that 96-column ones do not (N = 192: 25 % -> -8...-11 %
    // in time; N = 576, 10 %: the narrower tile's higher LDS traffic per MFMA costs more than the waste)
    static const char* nenv = gg_dev_env("GG_SPLIT3A_BN");          // dev: 96 / 128 forces one width
    const double w128 = 1.0 - (double)p.N / ((double)gg_cdiv(p.N, 128) * 128), w96 = 1.0 - (double)p.N / ((double)gg_cdiv(p.N, 96) * 96);
    const bool n96 = nenv ? atoi(nenv) == 96 : (w128 - w96 >= 0.2);
    const int bn = n96 ? 96 : 128;
    p.tilesM = (int)gg_cdiv(p.M, big ? 256 : 128); p.tilesN = (int)gg_cdiv(p.N, bn);
    // dev: ablations of the 256 x 128 form (1-8: results are garbage): 1 no MFMA, 2 no fragment reads, 4 no A path, 8 no B DMA; cache-policy variants (results unchanged):
    // 16 non-temporal result stores (the default), 32 + non-temporal A loads, 64 + non-temporal epilogue loads, 256 = default-policy stores
    static const char* aenv = gg_dev_env("GG_SPLIT3A_ABL");
    const int abl = aenv ? atoi(aenv) : 16;
    // dev: 32 = the 256 x 128 form on v_mfma_f32_32x32x16_bf16 (gemm_nt_split3w_kernel: 10 % fewer wave cycles, the same wall time at the clock the chip then holds)
    static const char* menv = gg_dev_env("GG_SPLIT3A_MFMA");
    const bool wide = menv && atoi(menv) == 32;
    // epilogue class (split3_epilogue_rows_ec) when the shape takes the vector path and the options are one of the model's four combinations; 0 = the generic epilogue
    static const char* eenv = gg_dev_env("GG_SPLIT3_NO_EC");       // dev: the generic epilogue everywhere
    int ec = 0;
    if (!eenv && p.C && !p.c_planes && !p.colstats && (p.N & 7) == 0 && (p.ldc & 3) == 0 && (!p.residual || (p.ldr & 3) == 0) && (!p.bias || ((uintptr_t)p.bias & 15) == 0)) {
        if (p.act == 1 && p.preact && !p.rowscale && !p.residual && !p.dact_preact) ec = 2;
        else if (p.dact_preact && p.dact == 1 && !p.bias && !p.act && !p.preact && !p.residual) ec = 3;
        else if (p.residual && !p.act && !p.preact && !p.dact_preact) ec = 4;
        else if (!p.act && !p.preact && !p.rowscale && !p.residual && !p.dact_preact) ec = 1;
    }
#define S3_EC(K, ...) (ec == 1 ? K<__VA_ARGS__, 1> : ec == 2 ? K<__VA_ARGS__, 2> : ec == 3 ? K<__VA_ARGS__, 3> : ec == 4 ? K<__VA_ARGS__, 4> : K<__VA_ARGS__, 0>)
    static const char* pkenv = gg_dev_env("GG_SPLIT3_PRO_PK");      // dev A/B: the prologue's GELU on packed FMAs
    void (*kern)(Split3Params) = (pro && pkenv && bn_act == 1) ? (n96 ? gemm_nt_split3a_kernel<16, 3, 0, 6> : gemm_nt_split3a_kernel<16, 4, 0, 6>) : pro ? (n96 ? (bn_act == 1 ? gemm_nt_split3a_kernel<16, 3, 0, 2> : bn_act == 2 ? gemm_nt_split3a_kernel<16, 3, 0, 3> : gemm_nt_split3a_kernel<16, 3, 0, 1>)
                                            : (bn_act == 1 ? gemm_nt_split3a_kernel<16, 4, 0, 2> : bn_act == 2 ? gemm_nt_split3a_kernel<16, 4, 0, 3> : gemm_nt_split3a_kernel<16, 4, 0, 1>)) :
                                 !big ? (n96 ? S3_EC(gemm_nt_split3b_kernel, 3) : S3_EC(gemm_nt_split3b_kernel, 4)) : n96 ? S3_EC(gemm_nt_split3a_kernel, 16, 3) :
                                 (!wide && abl == 16) ? S3_EC(gemm_nt_split3a_kernel, 16, 4) :
                                 wide ? (abl == 256 ? gemm_nt_split3w_kernel<0> : abl == 48 ? gemm_nt_split3w_kernel<48> : gemm_nt_split3w_kernel<16>) :
                                 abl == 1 ? gemm_nt_split3a_kernel<1> : abl == 2 ? gemm_nt_split3a_kernel<2> : abl == 4 ? gemm_nt_split3a_kernel<4> :
                                 abl == 8 ? gemm_nt_split3a_kernel<8> : abl == 6 ? gemm_nt_split3a_kernel<6> : abl == 256 ? gemm_nt_split3a_kernel<0> :
                                 abl == 32 ? gemm_nt_split3a_kernel<32> : abl == 48 ? gemm_nt_split3a_kernel<48> : abl == 80 ? gemm_nt_split3a_kernel<80> :
                                 abl == 112 ? gemm_nt_split3a_kernel<112> : gemm_nt_split3a_kernel<16>;
#undef S3_EC
    const size_t lds = (big ? (size_t)2 * 3 * (256 + bn) * S3_SK * sizeof(bf16) : (size_t)3 * (128 + 2 * bn) * S3_SK * sizeof(bf16)) +
                       (pro ? (size_t)2 * (((p.K + 31) & ~31) + 3 * S3_SK) * sizeof(float) : 0);
    GG_CHECK(lds <= 160 * 1024, "gg_gemm_nt_split3_af32: the ring plus the prologue table exceed the LDS");
    {
        static std::mutex raised_mu;
        static std::vector<const void*> raised;                    // kernels whose dynamic LDS limit has been raised (per kernel function, once)
        std::lock_guard<std::mutex> lk(raised_mu);
        const void* kp = reinterpret_cast<const void*>(kern);
        if (std::find(raised.begin(), raised.end(), kp) == raised.end()) {
            GG_CHECK(hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess, "gg_gemm_nt_split3_af32: cannot raise the dynamic LDS limit");
            raised.push_back(kp);
        }
    }
    const double mn = (double)p.M * p.N;
    GG_PROF(GG_CAT_GEMM | GG_CAT_SPLIT_FLAG, 2.0 * p.M * (double)p.N * p.K,
            4.0 * (double)p.M * p.K + 6.0 * (double)p.N * p.K + 4.0 * mn * ((p.C != nullptr) + (p.preact != nullptr) + (p.residual != nullptr) + (p.dact_preact != nullptr)) +
                (p.c_planes ? 6.0 * mn : 0.0), stream);
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.tilesM * p.tilesN)), dim3(big ? 512 : 256), lds, (hipStream_t)stream, p);
    GG_LAUNCH_CHECK();
    return 0;
}

extern "C" int gg_gemm_tn_f32_splits(int M, int N, int K);
// weight gradient of a Linear in the fp32_split mode: partial dW[N][K] slabs like gg_gemm_tn_f32 (same arguments; reduce with gg_splitk_reduce)
extern "C" int gg_gemm_tn_split3_splits(int M, int N, int K) {
    const int64_t tiles = gg_cdiv(N, 256) * gg_cdiv(K, 128);
    const int64_t cap = std::max<int64_t>(1, ((int64_t)128 << 20) / ((int64_t)N * K * 4));        // 128 MiB of slabs at most (scratch.splitk)
    const int64_t smax = std::max<int64_t>(1, std::min<int64_t>(cap, gg_cdiv(M, 1024)));             // at least 32 stages per slab
    int64_t s = 1;
    double best = 0.0;
    for (int64_t c = 1; c <= smax && c * tiles <= 4 * 256; ++c) {                                     // one 144 KB workgroup per CU: whole rounds of 256
        const double eff = (double)(c * tiles) / (256.0 * (double)gg_cdiv(c * tiles, 256));
        if (eff > best + 0.02) { best = eff; s = c; }
    }
    // a quarter more slabs than gg_gemm_tn_f32 would take: at EQUAL slab counts the two kernels' errors against fp64 are within 6 % of each other (the error of
    // either is that of a slab's f32 accumulation chain and falls with the square root of the slab count: tools/tn_split_error_vs_slabs.py); with the shorter
    // chains the split kernel's is below the f32 kernel's on every shape
    s = std::min<int64_t>(cap, std::max<int64_t>(s, (5 * (int64_t)gg_gemm_tn_f32_splits(M, N, K) + 3) / 4));
    const int64_t rows = gg_cdiv(gg_cdiv(M, s), 32) * 32;          // slabs are whole 32-row stages
    return (int)gg_cdiv(M, rows);
}
extern "C" int gg_gemm_tn_split3(const float* dY, int64_t ldy, const float* X, int64_t ldx, int M, int N, int K, const float* rowscale, int rows_per_scale, float* partials,
                                 int splits, void* stream) {
    GG_CHECK(dY && X && partials && M > 0 && N > 0 && K > 0 && splits > 0, "gg_gemm_tn_split3: bad args");
    GG_CHECK((N & 3) == 0 && (K & 3) == 0 && (ldy & 3) == 0 && (ldx & 3) == 0 && ldy >= N && ldx >= K, "gg_gemm_tn_split3: N, K, ldy, ldx must be multiples of 4, ld >= columns");
    GG_CHECK(((uintptr_t)dY & 15) == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)partials & 15) == 0, "gg_gemm_tn_split3: operands and partials must be 16-byte aligned");
    GG_CHECK(!rowscale || rows_per_scale > 0, "gg_gemm_tn_split3: rows_per_scale");
    Split3TnParams p;
    p.dY = dY; p.ldy = ldy; p.X = X; p.ldx = ldx; p.M = M; p.N = N; p.K = K; p.rowscale = rowscale; p.rps = rows_per_scale > 0 ? rows_per_scale : 1; p.part = partials;
    p.last_scale = (M - 1) / p.rps;
    p.rows_per_split = (int)(gg_cdiv(gg_cdiv(M, splits), 32) * 32);
    GG_CHECK((int64_t)p.rows_per_split * (splits - 1) < M, "gg_gemm_tn_split3: more splits than 32-row stages");
    GG_CHECK((int64_t)p.rows_per_split * std::max(ldy, ldx) * 4 < ((int64_t)1 << 31), "gg_gemm_tn_split3: a slab exceeds the 2 GiB descriptor range");
    p.tilesN = (int)gg_cdiv(N, 256); p.tilesK = (int)gg_cdiv(K, 128);
    // dev A/B (same box, tools/bench_split3_tn.py, profiles/r06_tn_split_ab.txt): bit 0 = dispatch-order walk (no XCD-contiguous remap), bit 1 = sub-images skewed by 64 bytes.
    // Default: remap, no skew -- the remap is worth 3-5 % (and cuts the HBM fetch 4 x); the skew removes every LDS bank conflict of the plane writes (SQ_LDS_BANK_CONFLICT
    // 41 M -> 0 per launch) and still LOSES 5-9 %: those conflict cycles were hidden under the MFMAs, the 152 KB ring's larger offsets are not free
    static const char* tenv = gg_dev_env("GG_SPLIT3_TN");
    const int tv = tenv ? atoi(tenv) : 0;
    void (*kern)(Split3TnParams) = tv == 1 ? gemm_tn_split3_kernel<1024, false> : tv == 2 ? gemm_tn_split3_kernel<TN3_SUB, true> : tv == 3 ? gemm_tn_split3_kernel<TN3_SUB, false> :
                                   gemm_tn_split3_kernel<1024, true>;
    const int sub = (tv & 2) ? TN3_SUB : 1024;
    static bool raised[4] = {false, false, false, false};
    if (!raised[tv & 3]) {
        GG_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess,
                 "gg_gemm_tn_split3: cannot raise the dynamic LDS limit");
        raised[tv & 3] = true;
    }
    GG_PROF(GG_CAT_GEMM | GG_CAT_SPLIT_FLAG, 2.0 * M * (double)N * K, 4.0 * ((double)M * N + (double)M * K) + 8.0 * splits * (double)N * K, stream);
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.tilesN * p.tilesK * splits)), dim3(512), (size_t)2 * 3 * (256 / 32 + 128 / 32) * sub * sizeof(bf16), (hipStream_t)stream, p);
    GG_LAUNCH_CHECK();
    return 0;
}
