// bf16 MFMA GEMM  C[M,N] = A[M,K] * B[N,K]^T  with fused epilogues (gfx950).
//
// One kernel serves every dense contraction on the hot path: 1x1 convs and im2col'd 3x3 convs
// of PatchEmbed / MBConv / PatchMerging, the qkv / proj / fc1 / fc2 Linears of TinyViT and CLIP,
// the geocell head, and (with pre-transposed operands + split-K) every dgrad / wgrad.
//
// Tile 128(M) x 128(N) x 32(K), 256 threads = 4 waves in 2x2, each wave 64x64 = 4x4 tiles of
// v_mfma_f32_16x16x32_bf16.  The MFMA is issued as D = Wfrag x Xfrag, i.e. D[i=n][j=m], so one
// lane ends up with 4 consecutive n of a single row m -> 8-byte bf16x4 / 16-byte f32x4 stores
// along the contiguous dimension of C and vectorised bias / residual / pre-activation access.
#include "common.h"
#include "../../include/gg.h"

#define BM 128
#define BN 128
#define BK 64

struct GemmParams {
    const bf16* A; int64_t lda;
    const bf16* B; int64_t ldb;
    void* C; int64_t ldc;
    int M, N, K;
    const float* bias;
    int act;
    bf16* preact;                 // optional copy of (acc+bias) before the activation
    const float* rowscale; int rows_per_scale;
    const bf16* residual; int64_t ldr;
    const bf16* dact_preact; int dact;   // out = v * act'(dact_preact[m][n]) (backward through fc1's activation)
    float* colstats;              // [tilesM][2][N]: per M-tile column sum / sum of squares of (acc)
    int out_f32;
    int split_k, k_per_split;     // split_k > 1: C is f32 [split][M][N] partials
    int tilesM, tilesN;
};

// LDS image of a 128 x 64 operand tile: unpadded 128-byte rows, the eight 16-byte chunks of a row XOR-swizzled with
//   T(row) = 2*bit1(row) + 4*bit3(row)
// which makes every 16-lane group of the fragment ds_read_b128 (lanes = 16 rows x 4 k-chunks) hit 16 distinct
// 16-byte bank slots, and keeps the staging ds_write_b128 (8 lanes = one whole row) conflict-free.
__device__ __forceinline__ int lds_chunk_off(int row, int kc) {
    return row * BK + ((kc ^ ((row & 2) | ((row >> 1) & 4))) << 3);
}

__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, bf16* smem, f32x4 (&acc)[4][4], int m0, int n0, int tm, int z,
                                              int wm, int wn, int lr, int lg) {
    // ---------------- epilogue: lane holds C[m = .. + mt*16 + lr][n = .. + nt*16 + lg*4 + r] ----------------
    // bf16 results are staged through LDS (the k-loop buffers are dead) so that global stores are 16 bytes per lane
    // along full 256-byte tile rows instead of 8-byte fragments of 16 different rows.
    constexpr int CS = BN + 8;                      // staged tile row stride (elements): 272 B, 16-B aligned
    bf16* Cs = smem;                                // [BM][CS] = 34 816 B
    const bool staged = !p.out_f32 && p.split_k <= 1;
    const bool vec_ok = ((p.ldc & 3) == 0) && (p.residual == nullptr || (p.ldr & 3) == 0);
    float csum[4][4], csq[4][4];
    if (p.colstats) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) csum[i][r] = csq[i][r] = 0.f;
    }
    // cooperative wide store of the staged tile: 16 lanes x 16 B per row, 16 rows per pass
    auto flush_tile = [&](bf16* dst) {
        __syncthreads();
        const int chunk = threadIdx.x & 15, rr = threadIdx.x >> 4;
        const int n = n0 + chunk * 8;
        const bool wide = ((p.ldc & 7) == 0) && (n + 7 < p.N);
#pragma unroll
        for (int pass = 0; pass < BM / 16; ++pass) {
            const int row = pass * 16 + rr, m = m0 + row;
            if (m >= p.M || n >= p.N) continue;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(Cs + row * CS + chunk * 8);
            bf16* g = dst + (int64_t)m * p.ldc + n;
            if (wide) *reinterpret_cast<bf16x8*>(g) = v;
            else { for (int j = 0; j < 8; ++j) if (n + j < p.N) g[j] = v[j]; }
        }
        __syncthreads();
    };
    if (staged && p.preact) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n = n0 + wn * 64 + nt * 16 + lg * 4;
                f32x4 v = acc[nt][mt];
                if (p.bias) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += p.bias[n + r];
                }
                bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                *reinterpret_cast<bf16x4*>(Cs + (wm * 64 + mt * 16 + lr) * CS + wn * 64 + nt * 16 + lg * 4) = o;
            }
        flush_tile(p.preact);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm * 64 + mt * 16 + lr;
        const bool mok = m < p.M;
        float rs = 1.f;
        if (p.rowscale && mok) rs = p.rowscale[m / p.rows_per_scale];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wn * 64 + nt * 16 + lg * 4;
            f32x4 v = acc[nt][mt];
            if (p.colstats) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { csum[nt][r] += v[r]; csq[nt][r] += v[r] * v[r]; }
            }
            const bool inb = mok && n < p.N;
            if (p.split_k > 1) {
                if (!inb) continue;
                float* Cz = reinterpret_cast<float*>(p.C) + ((int64_t)z * p.M + m) * p.ldc + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < p.N) Cz[r] = v[r];
                continue;
            }
            const bool full = vec_ok && (n + 3 < p.N);
            if (inb) {
                if (p.bias) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += p.bias[n + r];
                }
                if (p.preact && !staged) {
                    bf16* P = p.preact + (int64_t)m * p.ldc + n;
                    for (int r = 0; r < 4; ++r) if (n + r < p.N) P[r] = (bf16)v[r];
                }
                if (p.act) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gg_act(v[r], p.act);
                }
                if (p.dact) {
                    const bf16* D = p.dact_preact + (int64_t)m * p.ldc + n;
                    if (full) {
                        bf16x4 d = *reinterpret_cast<const bf16x4*>(D);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] *= gg_act_grad((float)d[r], p.dact);
                    } else {
                        for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] *= gg_act_grad((float)D[r], p.dact);
                    }
                }
                if (p.rowscale) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] *= rs;
                }
                if (p.residual) {
                    const bf16* R = p.residual + (int64_t)m * p.ldr + n;
                    if (full) {
                        bf16x4 d = *reinterpret_cast<const bf16x4*>(R);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += (float)d[r];
                    } else {
                        for (int r = 0; r < 4; ++r) if (n + r < p.N) v[r] += (float)R[r];
                    }
                }
            }
            if (staged) {
                bf16x4 o = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                *reinterpret_cast<bf16x4*>(Cs + (wm * 64 + mt * 16 + lr) * CS + wn * 64 + nt * 16 + lg * 4) = o;
            } else if (inb) {
                float* Cf = reinterpret_cast<float*>(p.C) + (int64_t)m * p.ldc + n;
                if (full) *reinterpret_cast<f32x4*>(Cf) = v;
                else { for (int r = 0; r < 4; ++r) if (n + r < p.N) Cf[r] = v[r]; }
            }
        }
    }
    if (staged) flush_tile(reinterpret_cast<bf16*>(p.C));
    if (p.colstats) {
        // rows beyond M and k beyond K contributed exact zeros.  Reduce over the 16 lanes sharing lg,
        // then over the two wm waves through LDS.
        if (!staged) __syncthreads();
        float* red = reinterpret_cast<float*>(smem);   // [2 wm][2 {sum,sq}][128 n]
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = csum[nt][r], q = csq[nt][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
                if (lr == 0) {
                    const int nl = wn * 64 + nt * 16 + lg * 4 + r;
                    red[(wm * 2 + 0) * 128 + nl] = s;
                    red[(wm * 2 + 1) * 128 + nl] = q;
                }
            }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int n = n0 + threadIdx.x;
            if (n < p.N) {
                float* out = p.colstats + (int64_t)tm * 2 * p.N;
                out[n] = red[0 * 128 + threadIdx.x] + red[2 * 128 + threadIdx.x];
                out[p.N + n] = red[1 * 128 + threadIdx.x] + red[3 * 128 + threadIdx.x];
            }
        }
    }
    if (p.colstats) __syncthreads();     // the next tile's operand stores reuse this buffer
}

__global__ __launch_bounds__(256, 3) void gemm_nt_kernel(GemmParams p) {
    // one operand stage (A 16 KiB + B 16 KiB); the next k-tile travels through registers while this one is consumed.
    // The epilogue reuses the buffer as a [128][136] bf16 staging tile (34 816 B).  (A persistent variant with
    // cross-tile prefetch was measured slower at 2 and 3 workgroups per CU: register pressure / spills.)
    __shared__ __attribute__((aligned(16))) bf16 smem[BM * (BN + 8)];
    bf16* As = smem;
    bf16* Bs = smem + BM * BK;

    const int tiles = p.tilesM * p.tilesN;
    const int bid = gg_xcd_remap(blockIdx.x, tiles);
    const int tm = bid / p.tilesN, tn = bid % p.tilesN;   // consecutive ids share the A row panel
    const int m0 = tm * BM, n0 = tn * BN;
    const int z = blockIdx.y;
    const int kbeg = z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lg = lane >> 4;

    // ---- per-thread staging geometry, fixed for the whole k loop: chunk c = tid + 256*i -> row c>>3, k-chunk c&7 ----
    const int srow = threadIdx.x >> 3, skc = threadIdx.x & 7;          // rows srow + 32*i
    const bf16* ga[4]; const bf16* gb[4];
    bool va[4], vb[4];
    int lds_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = srow + 32 * i;
        va[i] = (m0 + row) < p.M;
        vb[i] = (n0 + row) < p.N;
        ga[i] = p.A + (int64_t)min(m0 + row, p.M - 1) * p.lda + kbeg + skc * 8;
        gb[i] = p.B + (int64_t)min(n0 + row, p.N - 1) * p.ldb + kbeg + skc * 8;
        lds_off[i] = lds_chunk_off(row, skc);
    }
    // fragment read offsets (elements): row = w*64 + i*16 + lr; only bits 1,3 of lr enter the swizzle
    const int sw = (lr & 2) | ((lr >> 1) & 4);
    const int a_base = (wm * 64 + lr) * BK, b_base = (wn * 64 + lr) * BK;
    const int kc0 = ((0 + lg) ^ sw) << 3, kc1 = ((4 + lg) ^ sw) << 3;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    bf16x8 ra[4], rb[4];
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const int nk = (kend - kbeg + BK - 1) / BK;
    auto load_tile = [&](int kt) {
        const int koff = kt * BK;
        const bool kok = (kbeg + koff + skc * 8) < kend;     // K % 8 == 0: a chunk is entirely in or out
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = (va[i] && kok) ? *reinterpret_cast<const bf16x8*>(ga[i] + koff) : zero8;
            rb[i] = (vb[i] && kok) ? *reinterpret_cast<const bf16x8*>(gb[i] + koff) : zero8;
        }
    };
    if (nk > 0) load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<bf16x8*>(As + lds_off[i]) = ra[i];
            *reinterpret_cast<bf16x8*>(Bs + lds_off[i]) = rb[i];
        }
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int kc = ks ? kc1 : kc0;
            bf16x8 xf[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xf[i] = *reinterpret_cast<const bf16x8*>(As + a_base + i * 16 * BK + kc);
                wf[i] = *reinterpret_cast<const bf16x8*>(Bs + b_base + i * 16 * BK + kc);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
        }
        __syncthreads();
    }
    gemm_epilogue(p, smem, acc, m0, n0, tm, z, wm, wn, lr, lg);
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

// bf16 [R, C] (row stride ld) -> bf16 [C, R] (row stride ldo), optional per-row scale (drop-path) applied while
// transposing.  Columns/rows beyond the source are not written: the caller zero-pads ldo.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16* __restrict__ in, int64_t ld, bf16* __restrict__ out,
                                                             int64_t ldo, int R, int C, const float* rowscale,
                                                             int rows_per_scale) {
    __shared__ bf16 tile[64][64 + 2];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) {
            v = (float)in[(int64_t)r * ld + c];
            if (rowscale) v *= rowscale[r / rows_per_scale];
        }
        tile[i][tx] = (bf16)v;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) out[(int64_t)c * ldo + r] = tile[tx][i];
    }
}

// f32 [R, C] -> bf16 [R, ldo] (cast) and/or bf16 [C, ldt] (cast + transpose): weight-cache refresh.
__global__ __launch_bounds__(256) void cast_transpose_f32_kernel(const float* __restrict__ in, int R, int C, bf16* __restrict__ out,
                                                                 int64_t ldo, bf16* __restrict__ outT, int64_t ldt) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) {
            v = in[(int64_t)r * C + c];
            if (out) out[(int64_t)r * ldo + c] = (bf16)v;
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    if (outT) {
        for (int i = ty; i < 64; i += 4) {
            const int c = c0 + i, r = r0 + tx;
            if (c < C && r < R) outT[(int64_t)c * ldt + r] = (bf16)tile[tx][i];
        }
    }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ in, bf16* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (bf16)in[i];
}
__global__ void cast_bf16_f32_kernel(const bf16* __restrict__ in, float* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (float)in[i];
}

// column sums of a bf16 [M, C] matrix (bias gradients): partials [gridDim.y][C] then a finalize pass.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16* __restrict__ x, int64_t ld, int M, int C,
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
__global__ void colsum_final_kernel(const float* __restrict__ part, int nparts, int C, float* __restrict__ out, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int i = 0; i < nparts; ++i) s += (double)part[(int64_t)i * C + c];
    out[c] = accumulate ? out[c] + (float)s : (float)s;
}

// ------------------------------------------------------------------------------------------- host
extern "C" int gg_gemm_nt(const GgGemmArgs* a, void* stream) {
    GG_CHECK(a && a->A && a->B && a->C, "gg_gemm_nt: null operand");
    GG_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "gg_gemm_nt: bad shape M=%d N=%d K=%d", a->M, a->N, a->K);
    GG_CHECK((a->K & 7) == 0 && (a->lda & 7) == 0 && (a->ldb & 7) == 0,
             "gg_gemm_nt: K, lda, ldb must be multiples of 8 (16-byte rows): K=%d lda=%lld ldb=%lld", a->K,
             (long long)a->lda, (long long)a->ldb);
    GG_CHECK(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->B & 15) == 0, "gg_gemm_nt: A/B must be 16-byte aligned");
    GG_CHECK(a->lda >= a->K && a->ldb >= a->K && a->ldc >= a->N, "gg_gemm_nt: leading dimension too small");
    const int split = a->split_k > 1 ? a->split_k : 1;
    if (split > 1)
        GG_CHECK(!a->bias && !a->act && !a->preact && !a->residual && !a->colstats && !a->dact && !a->rowscale,
                 "gg_gemm_nt: split-K writes raw f32 partials, no epilogue allowed");
    if (a->rowscale) GG_CHECK(a->rows_per_scale > 0, "gg_gemm_nt: rows_per_scale must be > 0");
    GemmParams p;
    p.A = (const bf16*)a->A; p.lda = a->lda; p.B = (const bf16*)a->B; p.ldb = a->ldb;
    p.C = a->C; p.ldc = a->ldc; p.M = a->M; p.N = a->N; p.K = a->K;
    p.bias = a->bias; p.act = a->act; p.preact = (bf16*)a->preact;
    p.rowscale = a->rowscale; p.rows_per_scale = a->rows_per_scale;
    p.residual = (const bf16*)a->residual; p.ldr = a->ldr;
    p.dact_preact = (const bf16*)a->dact_preact; p.dact = a->dact_preact ? a->dact : 0;
    p.colstats = a->colstats; p.out_f32 = a->out_f32;
    p.split_k = split;
    int kps = (int)gg_cdiv(a->K, split);
    kps = (int)gg_align(kps, BK);
    p.k_per_split = kps;
    p.tilesM = (int)gg_cdiv(a->M, BM); p.tilesN = (int)gg_cdiv(a->N, BN);
    GG_PROF(GG_CAT_GEMM, 2.0 * a->M * (double)a->N * a->K, 2.0 * ((double)a->M * a->K + (double)a->N * a->K + (double)a->M * a->N), stream);
    dim3 grid(p.tilesM * p.tilesN, split);
    hipLaunchKernelGGL(gemm_nt_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_gemm_colstats_rows(int M) { return (int)gg_cdiv(M, BM); }
extern "C" int gg_stat_rows_capacity(int rows) { return rows + GG_REDUCE_SLICES; }

extern "C" int gg_splitk_reduce(const float* part, float* out, int64_t n, int splits, int accumulate, float scale, void* stream) {
    GG_CHECK(part && out && n > 0 && splits > 0, "gg_splitk_reduce: bad args");
    GG_PROF(GG_CAT_MOVE, 0, 4.0 * n * (splits + 1), stream);
    int blocks = (int)std::min<int64_t>(gg_cdiv(n, 256), 4096);
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
                           (const bf16*)in + (int64_t)r0 * ld, ld, (bf16*)out + r0, ldo, rr, C,
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
    hipLaunchKernelGGL(cast_transpose_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, R, C, (bf16*)out, ldo, (bf16*)outT, ldt);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream) {
    GG_CHECK(in && out && n > 0, "gg_cast_f32_to_bf16: bad args");
    int blocks = (int)std::min<int64_t>(gg_cdiv(n, 256), 8192);
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, (bf16*)out, n);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_cast_bf16_to_f32(const void* in, float* out, int64_t n, void* stream) {
    GG_CHECK(in && out && n > 0, "gg_cast_bf16_to_f32: bad args");
    int blocks = (int)std::min<int64_t>(gg_cdiv(n, 256), 8192);
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)in, out, n);
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
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)gg_cdiv(C, 256), nparts), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)x, ld, M, C, rowscale, rows_per_scale, scratch, rpb);
    const float* rows; int nrows;
    gg_reduce_rows(scratch, nparts, C, (hipStream_t)stream, &rows, &nrows);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)gg_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, rows, nrows, C, out, accumulate);
    GG_LAUNCH_CHECK();
    return 0;
}
