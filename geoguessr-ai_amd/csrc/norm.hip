// BatchNorm (train-mode batch statistics), LayerNorm, pooling.  HBM-bound wavefront-reduction kernels.
//
// BatchNorm2d in train mode (timm ConvNorm; reference keeps every BN in train mode):
//   producer kernel (GEMM / dwconv) emits per-block column partials of sum and sum-of-squares of the
//   fp32 pre-BN result  ->  bn_finalize (fp64 reduction, running-stat update)  ->  bn_apply.
// Backward:  bn_bwd_reduce (dz, per-channel sum g, sum g*xhat)  ->  bn_bwd_finalize  ->  bn_bwd_apply.
#include "common.h"
#include "../../include/gg.h"

// storage-type helpers: 8 consecutive elements per lane (16 bytes of bf16, 32 bytes of f32)
template <typename T> struct Vec8;
template <> struct Vec8<bf16> {
    static __device__ __forceinline__ void load(const bf16* p, float (&f)[8]) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (float)v[j];
    }
    static __device__ __forceinline__ void store(bf16* p, const float (&f)[8]) {
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (bf16)f[j];
        *reinterpret_cast<bf16x8*>(p) = v;
    }
};
template <> struct Vec8<f16> {
    static __device__ __forceinline__ void load(const f16* p, float (&f)[8]) {
        const f16x8 v = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (float)v[j];
    }
    static __device__ __forceinline__ void store(f16* p, const float (&f)[8]) {
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (f16)f[j];
        *reinterpret_cast<f16x8*>(p) = v;
    }
};
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&f)[8]) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
        f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
    }
    static __device__ __forceinline__ void store(float* p, const float (&f)[8]) {
        *reinterpret_cast<f32x4*>(p) = (f32x4){f[0], f[1], f[2], f[3]};
        *reinterpret_cast<f32x4*>(p + 4) = (f32x4){f[4], f[5], f[6], f[7]};
    }
};

// activation at the precision of the storage type: the bf16 path's polynomial erf (|err| 2e-5) is far below bf16 resolution; the f32
// (reference-precision) path uses the fp32-accurate forms of common.h (gg_phi_f32: 1.2 ulp of 1)
template <typename T> __device__ __forceinline__ float act_t(float x, int act) { return sizeof(T) == 2 ? gg_act(x, act) : gg_act_f32(x, act); }
template <typename T> __device__ __forceinline__ float act_grad_t(float x, int act) { return sizeof(T) == 2 ? gg_act_grad(x, act) : gg_act_grad_f32(x, act); }

// ------------------------------------------------------------------------------ BN statistics
// part [nparts][2][C] -> stat [2][C] = (mean, rstd); running stats updated with unbiased variance.
__global__ __launch_bounds__(64 * GG_FOLD_TY) void bn_finalize_kernel(const float* __restrict__ part, int nparts, int C, double count, float eps, float momentum,
                                   float* __restrict__ stat, float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    double s, q;
    gg_fold_cols2(part, nparts, 2 * (int64_t)C, c, C + c, c < C, s, q);
    if (c >= C || (threadIdx.x >> 6) != 0) return;
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    stat[c] = (float)mean;
    stat[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
}
__global__ void bn_eval_stat_kernel(const float* __restrict__ rm, const float* __restrict__ rv, int C, float eps, float* __restrict__ stat) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    stat[c] = rm[c];
    stat[C + c] = rsqrtf(rv[c] + eps);
}

// out = act( [residual + rs *] (gamma * (y - mean) * rstd + beta) )
// Thread = (row-lane pp, channel group g): g is fixed per thread, so the folded per-channel scale / shift live in
// registers and the row loop issues only the 16-byte streaming accesses.
template <typename T>
__global__ void bn_apply_kernel(const T* __restrict__ y, const float* __restrict__ stat, const float* __restrict__ gamma,
                                const float* __restrict__ beta, int64_t M, int C, int act, const T* __restrict__ residual,
                                const float* __restrict__ rowscale, int rows_per_scale, T* __restrict__ out, int CG, int PP,
                                int rows_per_block) {
    const int g = threadIdx.x % CG, pp = threadIdx.x / CG;
    const int c0 = g * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = stat[C + c0 + j] * gamma[c0 + j];
        sh[j] = beta[c0 + j] - stat[c0 + j] * sc[j];
    }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(M, r0 + rows_per_block);
    for (int64_t m = r0 + pp; m < r1; m += PP) {
        float v[8], o[8];
        Vec8<T>::load(y + m * C + c0, v);
        if (residual) {
            float r[8];
            Vec8<T>::load(residual + m * C + c0, r);
            const float rs = rowscale ? rowscale[m / rows_per_scale] : 1.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = act_t<T>(r[j] + rs * (v[j] * sc[j] + sh[j]), act);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = act_t<T>(v[j] * sc[j] + sh[j], act);
        }
        Vec8<T>::store(out + m * C + c0, o);
    }
}

// Thread = (row-lane pp, channel group g), g fixed per thread (blockDim.x = CG*PP).
//   z    = gamma*xhat + beta ; pre = residual ? residual + rs*z : z
//   dpre = dout * act'(pre)            -> written to dz (it is also the skip-path gradient of an MBConv)
//   g    = residual ? rs*dpre : dpre   -> partial sums  sum g, sum g*xhat   [gridDim.x][2][C]
template <typename T>
__global__ void bn_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ y, const float* __restrict__ stat,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, int64_t M, int C, int act,
                                     const T* __restrict__ residual, const float* __restrict__ rowscale, int rows_per_scale,
                                     T* __restrict__ dz, float* __restrict__ part, int CG, int PP, int rows_per_block) {
    extern __shared__ float sred[];   // [PP][2][C]
    const int g = threadIdx.x % CG, pp = threadIdx.x / CG;
    const int c0 = g * 8;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(M, r0 + rows_per_block);
    float mu[8], rstd[8], ga[8], be[8], s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mu[j] = stat[c0 + j]; rstd[j] = stat[C + c0 + j]; ga[j] = gamma[c0 + j]; be[j] = beta[c0 + j];
        s[j] = q[j] = 0.f;
    }
    for (int64_t m = r0 + pp; m < r1; m += PP) {
        float d[8], v[8], r[8], o[8];
        Vec8<T>::load(dout + m * C + c0, d);
        Vec8<T>::load(y + m * C + c0, v);
        float rs = 1.f;
        if (residual) {
            Vec8<T>::load(residual + m * C + c0, r);
            if (rowscale) rs = rowscale[m / rows_per_scale];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (v[j] - mu[j]) * rstd[j];
            float pre = ga[j] * xh + be[j];
            if (residual) pre = r[j] + rs * pre;
            const float dpre = d[j] * act_grad_t<T>(pre, act);
            o[j] = dpre;
            const float gg = residual ? rs * dpre : dpre;
            s[j] += gg;
            q[j] += gg * xh;
        }
        if (dz) Vec8<T>::store(dz + m * C + c0, o);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sred[(pp * 2 + 0) * C + c0 + j] = s[j];
        sred[(pp * 2 + 1) * C + c0 + j] = q[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        float t = 0.f;
        for (int k = 0; k < PP; ++k) t += sred[k * 2 * C + i];
        part[(int64_t)blockIdx.x * 2 * C + i] = t;
    }
}
// part [nparts][2][C] (sum g, sum g*xhat) -> coef [3][C] with  dy = coef0*g + coef1*y + coef2  ==
// gamma*rstd*(g - mean(g) - xhat*mean(g*xhat));  optional dgamma (+)= sum g*xhat, dbeta (+)= sum g
__global__ __launch_bounds__(64 * GG_FOLD_TY) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nparts, int C, double count, const float* __restrict__ stat,
                                       const float* __restrict__ gamma, float* __restrict__ coef, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, int accumulate) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    double s, q;
    gg_fold_cols2(part, nparts, 2 * (int64_t)C, c, C + c, c < C, s, q);
    if (c >= C || (threadIdx.x >> 6) != 0) return;
    const double mu = stat[c], rstd = stat[C + c];
    const double a = (double)gamma[c] * rstd, k2 = s / count, k3 = q / count;
    coef[c] = (float)a;
    coef[C + c] = (float)(-a * k3 * rstd);
    coef[2 * C + c] = (float)(-a * k2 + a * k3 * rstd * mu);
    if (dgamma) {
        dgamma[c] = accumulate ? dgamma[c] + (float)q : (float)q;
        dbeta[c] = accumulate ? dbeta[c] + (float)s : (float)s;
    }
}
// the same coefficients from (sum g*x, sum g) with x = gamma*xhat + beta the BatchNorm's OUTPUT (rows left by gg_layernorm_bwd_colsum); the
// BatchNorm's own parameter gradients would need a division by gamma and are not formed: frozen BatchNorm parameters only
__global__ __launch_bounds__(64 * GG_FOLD_TY) void bn_bwd_coef_from_x_kernel(const float* __restrict__ part, int nparts, int C, double count,
                                                                           const float* __restrict__ stat, const float* __restrict__ gamma,
                                                                           const float* __restrict__ beta, float* __restrict__ coef) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    double t, s1;
    gg_fold_cols2(part, nparts, 2 * (int64_t)C, c, C + c, c < C, t, s1);
    if (c >= C || (threadIdx.x >> 6) != 0) return;
    const double mu = stat[c], r = stat[C + c], g = gamma[c];
    const double u = (t - (double)beta[c] * s1) / count;          // gamma * mean(g * xhat)
    coef[c] = (float)(g * r);
    coef[C + c] = (float)(-r * r * u);
    coef[2 * C + c] = (float)(-g * r * s1 / count + r * r * mu * u);
}
// dy = coef0*(rs*dz) + coef1*y + coef2     (same thread geometry as bn_apply)
template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dz, const T* __restrict__ y, const float* __restrict__ coef, int64_t M,
                                    int C, const float* __restrict__ rowscale, int rows_per_scale, T* __restrict__ dy, int CG, int PP,
                                    int rows_per_block) {
    const int g = threadIdx.x % CG, pp = threadIdx.x / CG;
    const int c0 = g * 8;
    float ca[8], cb[8], cc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ca[j] = coef[c0 + j]; cb[j] = coef[C + c0 + j]; cc[j] = coef[2 * C + c0 + j]; }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(M, r0 + rows_per_block);
    for (int64_t m = r0 + pp; m < r1; m += PP) {
        float d[8], v[8], o[8];
        Vec8<T>::load(dz + m * C + c0, d);
        Vec8<T>::load(y + m * C + c0, v);
        const float rs = rowscale ? rowscale[m / rows_per_scale] : 1.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaf(ca[j] * rs, d[j], fmaf(cb[j], v[j], cc[j]));
        Vec8<T>::store(dy + m * C + c0, o);
    }
}

// Folds BatchNorm backward's apply step (dy = c0*dz + c1*y + c2) into the weights of the 1x1-conv dgrad that consumes dy:
//   dx[m][ci] = sum_co dy[m][co] W[co][ci] = [dz | y][m][:] . Bf[ci][:] + bias[ci]
//   Bf[ci][co] = bf16(c0[co] W[co][ci]),  Bf[ci][Cout + co] = bf16(c1[co] W[co][ci]),
//   bias[ci]   = sum_co c2[co] W[co][ci] + (c1[co] W[co][ci] - Bf[ci][Cout + co]) * mean[co]
// The second bias term moves the bf16 rounding error of the y-weights off the batch mean of y (it would otherwise scale
// with |mean|/std of the channel) onto the centred activations.  One block per input channel.
__global__ __launch_bounds__(256) void bn_bwd_fold_kernel(const float* __restrict__ W, const float* __restrict__ coef,
                                                          const float* __restrict__ stat, int Cout, int Cin, bf16* __restrict__ Bf,
                                                          float* __restrict__ bias) {
    __shared__ float red[32];
    const int ci = blockIdx.x;
    float acc = 0.f;
    for (int co = threadIdx.x; co < Cout; co += blockDim.x) {
        const float w = W[(int64_t)co * Cin + ci];
        const float w0 = coef[co] * w, w1 = coef[Cout + co] * w;
        const bf16 b0 = (bf16)w0, b1 = (bf16)w1;
        Bf[(int64_t)ci * 2 * Cout + co] = b0;
        Bf[(int64_t)ci * 2 * Cout + Cout + co] = b1;
        acc += coef[2 * Cout + co] * w + (w1 - (float)b1) * stat[co];
    }
    acc = gg_block_sum<256>(acc, red);
    if (threadIdx.x == 0) bias[ci] = acc;
}

// ------------------------------------------------------------------------------ LayerNorm
// One wave per row, 8-element chunks: lane takes chunks lane, lane+64 (C <= 1024).  gamma / beta live in registers for the wave's whole row walk, and two
// rows are in flight per wave (both rows' loads are issued before the first reduction: at one 1.5 KB row per wave the CLIP tower's LayerNorms ran at 4.1 TB/s).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const TI* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int64_t M, int C, float eps,
                                                            TO* __restrict__ out, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int nch = C >> 3;
    float ga[2][8], be[2][8];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int ch = lane + 64 * k;
#pragma unroll
        for (int j = 0; j < 8; ++j) { ga[k][j] = ch < nch ? gamma[ch * 8 + j] : 0.f; be[k][j] = ch < nch ? beta[ch * 8 + j] : 0.f; }
    }
    for (int64_t m0 = 2 * wave; m0 < M; m0 += 2 * nwaves) {
        float v[2][2][8];                                            // [row][chunk][element]
        float s[2] = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int64_t m = m0 + r;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int ch = lane + 64 * k;
                if (ch < nch && m < M) Vec8<TI>::load(x + m * C + ch * 8, v[r][k]);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[r][k][j] = 0.f;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) s[r] += v[r][k][j];
        float mean[2], rstd[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) mean[r] = gg_wave_sum(s[r]) / (float)C;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int ch = lane + 64 * k;
                if (ch < nch) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float d = v[r][k][j] - mean[r]; q += d * d; }
                }
            }
            rstd[r] = rsqrtf(gg_wave_sum(q) / (float)C + eps);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int64_t m = m0 + r;
            if (m >= M) continue;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int ch = lane + 64 * k;
                if (ch < nch) {
                    float o[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (v[r][k][j] - mean[r]) * rstd[r] * ga[k][j] + be[k][j];
                    Vec8<TO>::store(out + m * C + ch * 8, o);
                }
            }
            if (mean_out && lane == 0) { mean_out[m] = mean[r]; rstd_out[m] = rstd[r]; }
        }
    }
}

// ---- bf16 LayerNorm, 16 lanes per row ----------------------------------------------------------------------------
// A 64-lane wave per row leaves 25-60 % of the lanes idle at C = 192 / 384 (24 / 48 16-byte chunks).  Here a row belongs
// to a 16-lane group (4 rows per wave, 16 per block); lane l of the group owns chunks l, l+16, ... (NCH of them), so a
// group reads 256 contiguous bytes per step, reductions are 4 shuffles, and gamma / beta stay in registers.
__device__ __forceinline__ float gg_group16_sum(float v) {
    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
    return v;
}
// BNIN: x is the saved pre-BatchNorm output of the ConvNorm in front (TinyViT local_conv); the BatchNorm apply runs on the way in
// (bf16-rounded exactly as the separate apply pass stores it) and the applied tensor -- the residual stream -- is written to xout, so
// the apply pass and one read of the stream disappear.
template <typename T> __device__ __forceinline__ float ln_round(float v) { return sizeof(T) == 2 ? (float)(bf16)v : v; }   // storage rounding of T
// PL3 (experiment, DESIGN.md 5 "the bf16 x 3 split"): the result leaves as three bf16 planes [3][M][C] (o = p1 + p2 + p3 to 24 bits) for gg_gemm_nt_split3
// instead of one f32 tensor: 6 instead of 4 bytes per element written by a kernel that reads 4
template <typename T, int NCH, bool BNIN = false, bool PL3 = false>
__global__ __launch_bounds__(256) void layernorm_fwd_g16_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, int64_t M, int C, float eps,
                                                                T* __restrict__ out, float* __restrict__ mean_out,
                                                                float* __restrict__ rstd_out, const float* __restrict__ bn_stat = nullptr,
                                                                const float* __restrict__ bn_gamma = nullptr,
                                                                const float* __restrict__ bn_beta = nullptr, T* __restrict__ xout = nullptr) {
    const int l16 = threadIdx.x & 15;
    const int nch = C >> 3;
    float g[NCH][8], b[NCH][8];
    // BatchNorm scale / shift live in LDS (another 48 registers per thread at C = 384 would cost two waves per SIMD of a streaming kernel)
    __shared__ __attribute__((aligned(16))) float bn_tab[BNIN ? 2 * 640 : 4];
    if (BNIN) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            const float a = bn_stat[C + c] * bn_gamma[c];
            bn_tab[c] = a; bn_tab[640 + c] = bn_beta[c] - bn_stat[c] * a;
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = min(l16 + 16 * k, nch - 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) { g[k][j] = gamma[ch * 8 + j]; b[k][j] = beta[ch * 8 + j]; }
    }
    const float invC = 1.f / (float)C;
    for (int64_t m = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); m < M; m += (int64_t)gridDim.x * 16) {
        float v[NCH][8];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = l16 + 16 * k;
            if (ch < nch) Vec8<T>::load(x + m * C + ch * 8, v[k]);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[k][j] = 0.f;
            }
            if (BNIN && ch < nch) {
                const f32x4 s0 = *reinterpret_cast<const f32x4*>(bn_tab + ch * 8), s1 = *reinterpret_cast<const f32x4*>(bn_tab + ch * 8 + 4);
                const f32x4 h0 = *reinterpret_cast<const f32x4*>(bn_tab + 640 + ch * 8), h1 = *reinterpret_cast<const f32x4*>(bn_tab + 640 + ch * 8 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[k][j] = ln_round<T>(v[k][j] * s0[j] + h0[j]);
                    v[k][j + 4] = ln_round<T>(v[k][j + 4] * s1[j] + h1[j]);
                }
                Vec8<T>::store(xout + m * C + ch * 8, v[k]);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[k][j];
        const float mean = gg_group16_sum(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const bool ok = l16 + 16 * k < nch;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = ok ? v[k][j] - mean : 0.f; v[k][j] = d; q += d * d; }
        }
        const float rstd = rsqrtf(gg_group16_sum(q) * invC + eps);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = l16 + 16 * k;
            if (ch < nch) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = fmaf(v[k][j] * rstd, g[k][j], b[k][j]);
                if constexpr (PL3) {
                    bf16* pl = reinterpret_cast<bf16*>(out);
                    bf16x8 p1, p2, p3;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        bf16 s1_, s2_, s3_;
                        gg_split3_rne(o[j], s1_, s2_, s3_);
                        p1[j] = s1_; p2[j] = s2_; p3[j] = s3_;
                    }
                    *reinterpret_cast<bf16x8*>(pl + m * C + ch * 8) = p1;
                    *reinterpret_cast<bf16x8*>(pl + M * C + m * C + ch * 8) = p2;
                    *reinterpret_cast<bf16x8*>(pl + 2 * M * C + m * C + ch * 8) = p3;
                } else
                Vec8<T>::store(out + m * C + ch * 8, o);
            }
        }
        if (mean_out && l16 == 0) { mean_out[m] = mean; rstd_out[m] = rstd; }
    }
}
// backward, same geometry.  PARAMS 1: accumulate (sum dout*xhat, sum dout) per channel -> part [gridDim.x][2][C] (the LayerNorm's own
// parameter gradients).  PARAMS 2: accumulate (sum dx*x, sum dx) of the RESULT dx against the raw input x instead -- the column sums the
// BatchNorm in front of this LayerNorm needs for its backward (gg_bn_bwd_coef_from_x), so that no separate reduce pass reads dx again
template <typename T, int NCH, int PARAMS>
__global__ __launch_bounds__(256) void layernorm_bwd_g16_kernel(const T* __restrict__ dout, const T* __restrict__ x,
                                                                const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                                const float* __restrict__ gamma, int64_t M, int C,
                                                                const T* __restrict__ dres, T* __restrict__ dx,
                                                                float* __restrict__ part) {
    extern __shared__ float sred[];   // [4 waves][2][C] when PARAMS
    const int l16 = threadIdx.x & 15;
    const int nch = C >> 3;
    float g[NCH][8];
    float dg[PARAMS ? NCH : 1][8], db[PARAMS ? NCH : 1][8];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = min(l16 + 16 * k, nch - 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[k][j] = gamma[ch * 8 + j];
    }
    if (PARAMS) {
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) dg[k][j] = db[k][j] = 0.f;
    }
    const float invC = 1.f / (float)C;
    for (int64_t m = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); m < M; m += (int64_t)gridDim.x * 16) {
        float xh[NCH][8], dxh[NCH][8], rr[NCH][8];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = l16 + 16 * k;
            const bool ok = ch < nch;
            if (ok) {
                Vec8<T>::load(x + m * C + ch * 8, xh[k]);
                Vec8<T>::load(dout + m * C + ch * 8, dxh[k]);
                if (dres) Vec8<T>::load(dres + m * C + ch * 8, rr[k]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) xh[k][j] = dxh[k][j] = rr[k][j] = 0.f;
            }
        }
        const float mean = mean_in[m], rstd = rstd_in[m];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const bool ok = l16 + 16 * k < nch;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dv = dxh[k][j];
                xh[k][j] = ok ? (xh[k][j] - mean) * rstd : 0.f;
                dxh[k][j] = dv * g[k][j];
                s1 += dxh[k][j];
                s2 = fmaf(dxh[k][j], xh[k][j], s2);
                if (PARAMS == 1) { dg[k][j] = fmaf(dv, xh[k][j], dg[k][j]); db[k][j] += dv; }
            }
        }
        s1 = gg_group16_sum(s1) * invC;
        s2 = gg_group16_sum(s2) * invC;
        const float inv_rstd = PARAMS == 2 ? 1.f / rstd : 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = l16 + 16 * k;
            if (ch < nch) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float r = rstd * (dxh[k][j] - s1 - xh[k][j] * s2);
                    if (dres) r += rr[k][j];
                    o[j] = r;
                    if (PARAMS == 2) { dg[k][j] = fmaf(r, fmaf(xh[k][j], inv_rstd, mean), dg[k][j]); db[k][j] += r; }
                }
                Vec8<T>::store(dx + m * C + ch * 8, o);
            }
        }
    }
    if (PARAMS) {
        const int w = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = l16 + 16 * k;
#pragma unroll
            for (int j = 0; j < 8; ++j) {      // the wave's 4 row groups first
                float a = dg[k][j], c = db[k][j];
                a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
                c += __shfl_xor(c, 16, 64); c += __shfl_xor(c, 32, 64);
                if ((threadIdx.x & 63) < 16 && ch < nch) {
                    sred[(w * 2 + 0) * C + ch * 8 + j] = a;
                    sred[(w * 2 + 1) * C + ch * 8 + j] = c;
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
            float t = 0.f;
            for (int k = 0; k < 4; ++k) t += sred[k * 2 * C + i];
            part[(int64_t)blockIdx.x * 2 * C + i] = t;
        }
    }
}

// dx = rstd*(dxh - mean(dxh) - xhat*mean(dxh*xhat)) [+ dres],  dxh = dout*gamma
// param partials: part [gridDim.x][2][C] = (sum dout*xhat, sum dout) over this block's rows (if part != null)
template <typename TI, typename TG>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TG* __restrict__ dout, const TI* __restrict__ x,
                                                            const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                            const float* __restrict__ gamma, int64_t M, int C,
                                                            const TG* __restrict__ dres, TG* __restrict__ dx, float* __restrict__ part) {
    extern __shared__ float sred[];   // [4 waves][2][C] when part != null
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t wave = blockIdx.x * (int64_t)(blockDim.x >> 6) + w;
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int nch = C >> 3;
    float dg[2][8], db[2][8];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) dg[k][j] = db[k][j] = 0.f;
    for (int64_t m = wave; m < M; m += nwaves) {
        const float mean = mean_in[m], rstd = rstd_in[m];
        float xh[2][8], dxh[2][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int ch = lane + 64 * k;
            if (ch < nch) {
                float xv[8], dv[8];
                Vec8<TI>::load(x + m * C + ch * 8, xv);
                Vec8<TG>::load(dout + m * C + ch * 8, dv);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xh[k][j] = (xv[j] - mean) * rstd;
                    dxh[k][j] = dv[j] * gamma[ch * 8 + j];
                    s1 += dxh[k][j];
                    s2 += dxh[k][j] * xh[k][j];
                    dg[k][j] += dv[j] * xh[k][j];
                    db[k][j] += dv[j];
                }
            }
        }
        s1 = gg_wave_sum(s1) / (float)C;
        s2 = gg_wave_sum(s2) / (float)C;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int ch = lane + 64 * k;
            if (ch < nch) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = rstd * (dxh[k][j] - s1 - xh[k][j] * s2);
                if (dres) {
                    float r[8];
                    Vec8<TG>::load(dres + m * C + ch * 8, r);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] += r[j];
                }
                Vec8<TG>::store(dx + m * C + ch * 8, o);
            }
        }
    }
    if (part) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int ch = lane + 64 * k;
            if (ch < nch) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sred[(w * 2 + 0) * C + ch * 8 + j] = dg[k][j];
                    sred[(w * 2 + 1) * C + ch * 8 + j] = db[k][j];
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
            float t = 0.f;
            for (int k = 0; k < 4; ++k) t += sred[k * 2 * C + i];
            part[(int64_t)blockIdx.x * 2 * C + i] = t;
        }
    }
}
// part [nparts][2][C] -> dgamma (+)= part[.][0], dbeta (+)= part[.][1]
__global__ __launch_bounds__(64 * GG_FOLD_TY) void ln_param_final_kernel(const float* __restrict__ part, int nparts, int C, float* __restrict__ dgamma,
                                      float* __restrict__ dbeta, int accumulate) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    double s, q;
    gg_fold_cols2(part, nparts, 2 * (int64_t)C, c, C + c, c < C, s, q);
    if (c >= C || (threadIdx.x >> 6) != 0) return;
    dgamma[c] = accumulate ? dgamma[c] + (float)s : (float)s;
    dbeta[c] = accumulate ? dbeta[c] + (float)q : (float)q;
}

// ------------------------------------------------------------------------------ pooling / means
// x bf16 [B*T, C] -> out f32 [B, C] = mean over T tokens
template <typename TT>
__global__ __launch_bounds__(256) void token_mean_fwd_kernel(const TT* __restrict__ x, float* __restrict__ out, int B, int T, int C) {
    const int cg = C >> 3;
    const int64_t total = (int64_t)B * cg;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const int64_t b = i / cg;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = 0; t < T; ++t) {
            float v[8];
            Vec8<TT>::load(x + (b * T + t) * C + g * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
        const float inv = 1.f / (float)T;
#pragma unroll
        for (int j = 0; j < 8; ++j) out[b * C + g * 8 + j] = acc[j] * inv;
    }
}
// dout f32 [B, C] -> dx bf16 [B*T, C] = dout / T
template <typename TT>
__global__ __launch_bounds__(256) void token_mean_bwd_kernel(const float* __restrict__ dout, TT* __restrict__ dx, int B, int T, int C) {
    const int cg = C >> 3;
    const int64_t total = (int64_t)B * T * cg;
    const float inv = 1.f / (float)T;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const int64_t bt = i / cg;
        const int64_t b = bt / T;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = dout[b * C + g * 8 + j] * inv;
        Vec8<TT>::store(dx + bt * C + g * 8, o);
    }
}
// emb f32 [N, V, C] -> out bf16 [N, ldo] (mean over V views; SuperGuessr panorama combine)
template <typename TT>
__global__ void view_mean_fwd_kernel(const float* __restrict__ emb, TT* __restrict__ out, int64_t ldo, int N, int V, int C) {
    const int64_t total = (int64_t)N * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t n = i / C;
        float s = 0.f;
        for (int v = 0; v < V; ++v) s += emb[(n * V + v) * C + c];
        out[n * ldo + c] = (TT)(s / (float)V);
    }
}
// dmean bf16 [N, ld] -> demb f32 [N, V, C] = dmean / V
template <typename TT>
__global__ void view_mean_bwd_kernel(const TT* __restrict__ dmean, int64_t ld, float* __restrict__ demb, int N, int V, int C) {
    const int64_t total = (int64_t)N * V * C;
    const float inv = 1.f / (float)V;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t n = i / ((int64_t)V * C);
        demb[i] = (float)dmean[n * ld + c] * inv;
    }
}

// ------------------------------------------------------------------------------------------- host
static int grid_for(int64_t n, int cap = 16384) { return (int)std::max<int64_t>(1, std::min<int64_t>(gg_cdiv(n, 256), cap)); }

struct RowGeom { int CG, PP, threads, rows_per_block, nblocks; };
static RowGeom row_geom(int64_t M, int C) {
    RowGeom g;
    g.CG = C / 8;
    g.PP = std::max(1, std::min(256 / g.CG, 15360 / (2 * C)));
    g.threads = g.CG * g.PP;
    int64_t rpb = std::max<int64_t>(gg_cdiv(M, 8192), (int64_t)g.PP * 4);
    rpb = gg_align(rpb, g.PP);
    g.rows_per_block = (int)rpb;
    g.nblocks = (int)gg_cdiv(M, rpb);
    return g;
}
extern "C" int gg_bn_finalize(float* part, int nparts, int C, int64_t count, float eps, float momentum, float* stat,
                              float* running_mean, float* running_var, void* stream) {
    GG_CHECK(part && stat && nparts > 0 && C > 0 && count > 0, "gg_bn_finalize: bad args");
    const float* rows; int nrows;
    gg_reduce_rows(part, nparts, 2 * C, (hipStream_t)stream, &rows, &nrows, GG_REDUCE_DIRECT_MAX);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)gg_cdiv(C, 64)), dim3(64 * GG_FOLD_TY), 0, (hipStream_t)stream, rows, nrows, C,
                       (double)count, eps, momentum, stat, running_mean, running_var);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_bn_eval_stat(const float* running_mean, const float* running_var, int C, float eps, float* stat, void* stream) {
    GG_CHECK(running_mean && running_var && stat && C > 0, "gg_bn_eval_stat: bad args");
    hipLaunchKernelGGL(bn_eval_stat_kernel, dim3((unsigned)gg_cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, running_mean,
                       running_var, C, eps, stat);
    GG_LAUNCH_CHECK();
    return 0;
}
template <typename T>
static int bn_apply_t(const void* y, const float* stat, const float* gamma, const float* beta, int64_t M, int C, int act,
                      const void* residual, const float* rowscale, int rows_per_scale, void* out, void* stream) {
    GG_CHECK(y && stat && gamma && beta && out && M > 0 && (C & 7) == 0, "gg_bn_apply: bad args");
    GG_CHECK(!rowscale || rows_per_scale > 0, "gg_bn_apply: rows_per_scale");
    GG_PROF(GG_CAT_NORM, 0, (residual ? 3.0 : 2.0) * sizeof(T) * M * C, stream);
    GG_CHECK(C <= 2048, "gg_bn_apply: C too large");
    RowGeom g = row_geom(M, C);
    hipLaunchKernelGGL(bn_apply_kernel<T>, dim3(g.nblocks), dim3(g.threads), 0, (hipStream_t)stream, (const T*)y, stat, gamma, beta, M, C,
                       act, (const T*)residual, rowscale, rows_per_scale, (T*)out, g.CG, g.PP, g.rows_per_block);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_bn_apply(const void* y, const float* stat, const float* gamma, const float* beta, int64_t M, int C, int act,
                           const void* residual, const float* rowscale, int rows_per_scale, void* out, void* stream) {
    return bn_apply_t<bf16>(y, stat, gamma, beta, M, C, act, residual, rowscale, rows_per_scale, out, stream);
}
extern "C" int gg_bn_apply_f32(const float* y, const float* stat, const float* gamma, const float* beta, int64_t M, int C, int act,
                               const float* residual, const float* rowscale, int rows_per_scale, float* out, void* stream) {
    return bn_apply_t<float>(y, stat, gamma, beta, M, C, act, residual, rowscale, rows_per_scale, out, stream);
}
extern "C" int64_t gg_bn_bwd_scratch_floats(int64_t M, int C) { return ((int64_t)row_geom(M, C).nblocks + GG_REDUCE_SLICES) * 2 * C + 3 * C; }
extern "C" int gg_bn_bwd_rows(int64_t M, int C) { return row_geom(M, C).nblocks; }
// part: [gg_stat_rows_capacity(gg_bn_bwd_rows(M,C))][2][C]; dz may be NULL (no activation / no residual: dz == dout)
template <typename T>
static int bn_bwd_reduce_t(const void* dout, const void* y, const float* stat, const float* gamma, const float* beta, int64_t M, int C,
                           int act, const void* residual, const float* rowscale, int rows_per_scale, void* dz, float* part, void* stream) {
    GG_CHECK(dout && y && stat && gamma && beta && part && M > 0 && (C & 7) == 0 && C <= 2048, "gg_bn_bwd_reduce: bad args");
    RowGeom g = row_geom(M, C);
    GG_CHECK(g.threads <= 1024, "gg_bn_bwd_reduce: C too large");
    GG_PROF(GG_CAT_NORM, 0, (residual ? 4.0 : (dz ? 3.0 : 2.0)) * sizeof(T) * M * C, stream);
    size_t lds = (size_t)g.PP * 2 * C * sizeof(float);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<T>, dim3(g.nblocks), dim3(g.threads), lds, (hipStream_t)stream, (const T*)dout,
                       (const T*)y, stat, gamma, beta, M, C, act, (const T*)residual, rowscale, rows_per_scale, (T*)dz, part,
                       g.CG, g.PP, g.rows_per_block);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_bn_bwd_reduce(const void* dout, const void* y, const float* stat, const float* gamma, const float* beta, int64_t M, int C,
                                int act, const void* residual, const float* rowscale, int rows_per_scale, void* dz, float* part, void* stream) {
    return bn_bwd_reduce_t<bf16>(dout, y, stat, gamma, beta, M, C, act, residual, rowscale, rows_per_scale, dz, part, stream);
}
// part rows -> coef [3][C] (dy = coef0*g + coef1*y + coef2) and the parameter gradients
extern "C" int gg_bn_bwd_finalize(float* part, int nparts, int C, int64_t count, const float* stat, const float* gamma, float* coef,
                                  float* dgamma, float* dbeta, int accumulate, void* stream) {
    GG_CHECK(part && stat && gamma && coef && nparts > 0 && C > 0 && count > 0, "gg_bn_bwd_finalize: bad args");
    const float* rows; int nrows;
    gg_reduce_rows(part, nparts, 2 * C, (hipStream_t)stream, &rows, &nrows, GG_REDUCE_DIRECT_MAX);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)gg_cdiv(C, 64)), dim3(64 * GG_FOLD_TY), 0, (hipStream_t)stream, rows, nrows, C,
                       (double)count, stat, gamma, coef, dgamma, dbeta, accumulate);
    GG_LAUNCH_CHECK();
    return 0;
}
template <typename T>
static int bn_bwd_apply_t(const void* dz, const void* y, const float* coef, int64_t M, int C, const float* rowscale,
                          int rows_per_scale, void* dy, void* stream) {
    GG_CHECK(dz && y && coef && dy && M > 0 && (C & 7) == 0 && C <= 2048, "gg_bn_bwd_apply: bad args");
    RowGeom g = row_geom(M, C);
    GG_PROF(GG_CAT_NORM, 0, 3.0 * sizeof(T) * M * C, stream);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(g.nblocks), dim3(g.threads), 0, (hipStream_t)stream, (const T*)dz, (const T*)y, coef,
                       M, C, rowscale, rows_per_scale, (T*)dy, g.CG, g.PP, g.rows_per_block);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_bn_bwd_apply(const void* dz, const void* y, const float* coef, int64_t M, int C, const float* rowscale,
                               int rows_per_scale, void* dy, void* stream) {
    return bn_bwd_apply_t<bf16>(dz, y, coef, M, C, rowscale, rows_per_scale, dy, stream);
}
extern "C" int gg_bn_bwd_fold_weights(const float* W, const float* coef, const float* stat, int Cout, int Cin, void* Bf, float* bias,
                                      void* stream) {
    GG_CHECK(W && coef && stat && Bf && bias && Cout > 0 && Cin > 0, "gg_bn_bwd_fold_weights: bad args");
    hipLaunchKernelGGL(bn_bwd_fold_kernel, dim3(Cin), dim3(256), 0, (hipStream_t)stream, W, coef, stat, Cout, Cin, (bf16*)Bf, bias);
    GG_LAUNCH_CHECK();
    return 0;
}
// scratch: [nblocks + slices][2][C] partial rows followed by coef [3][C]
template <typename T>
static int bn_bwd_t(const void* dout, const void* y, const float* stat, const float* gamma, const float* beta, int64_t M, int C,
                    int act, const void* residual, const float* rowscale, int rows_per_scale, void* dz, void* dy, float* scratch,
                    float* dgamma, float* dbeta, int accumulate, void* stream) {
    GG_CHECK(dz && dy && scratch, "gg_bn_bwd: bad args");
    const int nb = row_geom(M, C).nblocks;
    float* part = scratch;
    float* coef = scratch + ((int64_t)nb + GG_REDUCE_SLICES) * 2 * C;
    GG_TRY(bn_bwd_reduce_t<T>(dout, y, stat, gamma, beta, M, C, act, residual, rowscale, rows_per_scale, dz, part, stream));
    GG_TRY(gg_bn_bwd_finalize(part, nb, C, M, stat, gamma, coef, dgamma, dbeta, accumulate, stream));
    return bn_bwd_apply_t<T>(dz, y, coef, M, C, residual ? rowscale : nullptr, rows_per_scale, dy, stream);
}
extern "C" int gg_bn_bwd(const void* dout, const void* y, const float* stat, const float* gamma, const float* beta, int64_t M, int C,
                         int act, const void* residual, const float* rowscale, int rows_per_scale, void* dz, void* dy, float* scratch,
                         float* dgamma, float* dbeta, int accumulate, void* stream) {
    return bn_bwd_t<bf16>(dout, y, stat, gamma, beta, M, C, act, residual, rowscale, rows_per_scale, dz, dy, scratch, dgamma, dbeta, accumulate, stream);
}
/* f32 storage (reference-precision mode): same three passes, exact erf in the activation derivative */
extern "C" int gg_bn_bwd_reduce_f32(const float* dout, const float* y, const float* stat, const float* gamma, const float* beta, int64_t M, int C,
                                    int act, const float* residual, const float* rowscale, int rows_per_scale, float* dz, float* part, void* stream) {
    return bn_bwd_reduce_t<float>(dout, y, stat, gamma, beta, M, C, act, residual, rowscale, rows_per_scale, dz, part, stream);
}
extern "C" int gg_bn_bwd_apply_f32(const float* dz, const float* y, const float* coef, int64_t M, int C, const float* rowscale, int rows_per_scale,
                                   float* dy, void* stream) {
    return bn_bwd_apply_t<float>(dz, y, coef, M, C, rowscale, rows_per_scale, dy, stream);
}
extern "C" int gg_bn_bwd_f32(const float* dout, const float* y, const float* stat, const float* gamma, const float* beta, int64_t M, int C,
                             int act, const float* residual, const float* rowscale, int rows_per_scale, float* dz, float* dy, float* scratch,
                             float* dgamma, float* dbeta, int accumulate, void* stream) {
    return bn_bwd_t<float>(dout, y, stat, gamma, beta, M, C, act, residual, rowscale, rows_per_scale, dz, dy, scratch, dgamma, dbeta, accumulate, stream);
}

static int ln_blocks(int64_t M) { return (int)std::max<int64_t>(1, std::min<int64_t>(gg_cdiv(M, 4), 2048)); }
extern "C" int gg_layernorm_fwd(const void* x, int x_f32, const float* gamma, const float* beta, int64_t M, int C, float eps, void* out,
                                int out_f32, float* mean, float* rstd, void* stream) {
    GG_CHECK(x && gamma && beta && out && M > 0 && (C & 7) == 0 && C <= 1024, "gg_layernorm_fwd: bad args (C %% 8, C <= 1024)");
    dim3 grid(ln_blocks(M)), block(256);
    GG_PROF(GG_CAT_NORM, 0, ((x_f32 ? 4.0 : 2.0) + (out_f32 ? 4.0 : 2.0)) * M * C, stream);
    hipStream_t s = (hipStream_t)stream;
    const int nchl = (C / 8 + 15) / 16;          // 8-element chunks per lane in the 16-lanes-per-row kernels
    if (x_f32 == out_f32 && nchl <= 5) {
        const dim3 g16((unsigned)std::min<int64_t>(gg_cdiv(M, 16), 8192));
#define GG_LN_FWD(N_)                                                                                                                          \
    do {                                                                                                                                       \
        if (x_f32) hipLaunchKernelGGL((layernorm_fwd_g16_kernel<float, N_>), g16, block, 0, s, (const float*)x, gamma, beta, M, C, eps, (float*)out, mean, rstd); \
        else hipLaunchKernelGGL((layernorm_fwd_g16_kernel<bf16, N_>), g16, block, 0, s, (const bf16*)x, gamma, beta, M, C, eps, (bf16*)out, mean, rstd);         \
    } while (0)
        switch (nchl) { case 1: GG_LN_FWD(1); break; case 2: GG_LN_FWD(2); break; case 3: GG_LN_FWD(3); break; case 4: GG_LN_FWD(4); break; default: GG_LN_FWD(5); }
#undef GG_LN_FWD
    } else if (!x_f32 && !out_f32)
        hipLaunchKernelGGL((layernorm_fwd_kernel<bf16, bf16>), grid, block, 0, s, (const bf16*)x, gamma, beta, M, C, eps, (bf16*)out, mean, rstd);
    else if (x_f32 && out_f32)
        hipLaunchKernelGGL((layernorm_fwd_kernel<float, float>), grid, block, 0, s, (const float*)x, gamma, beta, M, C, eps, (float*)out, mean, rstd);
    else if (x_f32 && !out_f32)
        hipLaunchKernelGGL((layernorm_fwd_kernel<float, bf16>), grid, block, 0, s, (const float*)x, gamma, beta, M, C, eps, (bf16*)out, mean, rstd);
    else
        hipLaunchKernelGGL((layernorm_fwd_kernel<bf16, float>), grid, block, 0, s, (const bf16*)x, gamma, beta, M, C, eps, (float*)out, mean, rstd);
    GG_LAUNCH_CHECK();
    return 0;
}
// experiment: f32 LayerNorm whose result leaves as three bf16 planes [3][M][C] (the A operand of gg_gemm_nt_split3)
extern "C" int gg_layernorm_fwd_split3(const float* x, const float* gamma, const float* beta, int64_t M, int C, float eps, void* planes, float* mean, float* rstd,
                                       void* stream) {
    GG_CHECK(x && gamma && beta && planes && M > 0 && (C & 7) == 0 && C <= 640, "gg_layernorm_fwd_split3: bad args (C %% 8, C <= 640)");
    const int nchl = (C / 8 + 15) / 16;
    GG_PROF(GG_CAT_NORM, 0, 10.0 * M * C, stream);
    const dim3 g16((unsigned)std::min<int64_t>(gg_cdiv(M, 16), 8192)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define GG_LN_FWD(N_) hipLaunchKernelGGL((layernorm_fwd_g16_kernel<float, N_, false, true>), g16, block, 0, s, x, gamma, beta, M, C, eps, (float*)planes, mean, rstd)
    switch (nchl) { case 1: GG_LN_FWD(1); break; case 2: GG_LN_FWD(2); break; case 3: GG_LN_FWD(3); break; case 4: GG_LN_FWD(4); break; default: GG_LN_FWD(5); }
#undef GG_LN_FWD
    GG_LAUNCH_CHECK();
    return 0;
}
// ... and the form whose input is BatchNorm(y) of a saved pre-BatchNorm conv output (xout = BN(y) f32: the residual stream; planes = LN(xout))
extern "C" int gg_layernorm_fwd_bn_split3(const float* y, const float* bn_stat, const float* bn_gamma, const float* bn_beta, float* xout, const float* gamma,
                                          const float* beta, int64_t M, int C, float eps, void* planes, float* mean, float* rstd, void* stream) {
    GG_CHECK(y && bn_stat && bn_gamma && bn_beta && xout && gamma && beta && planes && M > 0 && (C & 7) == 0 && C <= 640, "gg_layernorm_fwd_bn_split3: bad args");
    const int nchl = (C / 8 + 15) / 16;
    GG_PROF(GG_CAT_NORM, 0, 14.0 * M * C, stream);
    const dim3 g16((unsigned)std::min<int64_t>(gg_cdiv(M, 16), 8192)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define GG_LN_FWD(N_) hipLaunchKernelGGL((layernorm_fwd_g16_kernel<float, N_, true, true>), g16, block, 0, s, y, gamma, beta, M, C, eps, (float*)planes, mean, rstd, bn_stat, bn_gamma, bn_beta, xout)
    switch (nchl) { case 1: GG_LN_FWD(1); break; case 2: GG_LN_FWD(2); break; case 3: GG_LN_FWD(3); break; case 4: GG_LN_FWD(4); break; default: GG_LN_FWD(5); }
#undef GG_LN_FWD
    GG_LAUNCH_CHECK();
    return 0;
}
// LayerNorm of BatchNorm(y) for a saved pre-BatchNorm conv output y: xout = BN(y) in the storage type (the residual stream), out = LN(xout)
template <typename T>
static int layernorm_fwd_bn_t(const void* y, const float* bn_stat, const float* bn_gamma, const float* bn_beta, void* xout,
                              const float* gamma, const float* beta, int64_t M, int C, float eps, void* out, float* mean, float* rstd,
                              void* stream) {
    GG_CHECK(y && bn_stat && bn_gamma && bn_beta && xout && gamma && beta && out && M > 0 && (C & 7) == 0, "gg_layernorm_fwd_bn: bad args");
    const int nchl = (C / 8 + 15) / 16;
    GG_CHECK(nchl <= 5, "gg_layernorm_fwd_bn: C <= 640");
    GG_PROF(GG_CAT_NORM, 0, 3.0 * sizeof(T) * M * C, stream);
    const dim3 g16((unsigned)std::min<int64_t>(gg_cdiv(M, 16), 8192)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define GG_LN_FWD(N_) hipLaunchKernelGGL((layernorm_fwd_g16_kernel<T, N_, true>), g16, block, 0, s, (const T*)y, gamma, beta, M, C, eps, (T*)out, mean, rstd, bn_stat, bn_gamma, bn_beta, (T*)xout)
    switch (nchl) { case 1: GG_LN_FWD(1); break; case 2: GG_LN_FWD(2); break; case 3: GG_LN_FWD(3); break; case 4: GG_LN_FWD(4); break; default: GG_LN_FWD(5); }
#undef GG_LN_FWD
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_layernorm_fwd_bn(const void* y, const float* bn_stat, const float* bn_gamma, const float* bn_beta, void* xout,
                                   const float* gamma, const float* beta, int64_t M, int C, float eps, void* out, float* mean, float* rstd,
                                   void* stream) {
    return layernorm_fwd_bn_t<bf16>(y, bn_stat, bn_gamma, bn_beta, xout, gamma, beta, M, C, eps, out, mean, rstd, stream);
}
extern "C" int gg_layernorm_fwd_bn_f32(const float* y, const float* bn_stat, const float* bn_gamma, const float* bn_beta, float* xout,
                                       const float* gamma, const float* beta, int64_t M, int C, float eps, float* out, float* mean, float* rstd,
                                       void* stream) {
    return layernorm_fwd_bn_t<float>(y, bn_stat, bn_gamma, bn_beta, xout, gamma, beta, M, C, eps, out, mean, rstd, stream);
}
extern "C" int64_t gg_layernorm_bwd_scratch_floats(int64_t M, int C) { return ((int64_t)ln_blocks(M) + GG_REDUCE_SLICES) * 2 * C; }
// f32 != 0: x, dout, dres, dx are all f32 (head norm); else all bf16
extern "C" int gg_layernorm_bwd(const void* dout, const void* x, int f32, const float* mean, const float* rstd, const float* gamma,
                                int64_t M, int C, const void* dres, void* dx, float* scratch, float* dgamma, float* dbeta,
                                int accumulate, void* stream) {
    GG_CHECK(dout && x && mean && rstd && gamma && dx && M > 0 && (C & 7) == 0 && C <= 1024, "gg_layernorm_bwd: bad args");
    GG_CHECK(!dgamma || (scratch && dbeta), "gg_layernorm_bwd: parameter grads need scratch + dbeta");
    // with parameter gradients every block ends in a 2*C-column reduction + partial row: fewer, longer-running blocks amortise it
    const int nb = dgamma ? std::min(ln_blocks(M), 512) : ln_blocks(M);
    GG_PROF(GG_CAT_NORM, 0, (dres ? 4.0 : 3.0) * (f32 ? 4.0 : 2.0) * M * C, stream);
    float* part = dgamma ? scratch : nullptr;
    size_t lds = dgamma ? (size_t)4 * 2 * C * sizeof(float) : 0;
    hipStream_t s = (hipStream_t)stream;
    const int nchl = (C / 8 + 15) / 16;
    if (nchl <= 5) {
#define GG_LN_BWD2(T_, N_)                                                                                                             \
    do {                                                                                                                               \
        if (part) hipLaunchKernelGGL((layernorm_bwd_g16_kernel<T_, N_, 1>), dim3(nb), dim3(256), lds, s, (const T_*)dout, (const T_*)x, \
                                     mean, rstd, gamma, M, C, (const T_*)dres, (T_*)dx, part);                                          \
        else hipLaunchKernelGGL((layernorm_bwd_g16_kernel<T_, N_, 0>), dim3(nb), dim3(256), 0, s, (const T_*)dout, (const T_*)x,   \
                                mean, rstd, gamma, M, C, (const T_*)dres, (T_*)dx, part);                                               \
    } while (0)
#define GG_LN_BWD(N_) do { if (f32) GG_LN_BWD2(float, N_); else GG_LN_BWD2(bf16, N_); } while (0)
        switch (nchl) { case 1: GG_LN_BWD(1); break; case 2: GG_LN_BWD(2); break; case 3: GG_LN_BWD(3); break; case 4: GG_LN_BWD(4); break; default: GG_LN_BWD(5); }
#undef GG_LN_BWD
#undef GG_LN_BWD2
    } else if (f32)
        hipLaunchKernelGGL((layernorm_bwd_kernel<float, float>), dim3(nb), dim3(256), lds, s, (const float*)dout, (const float*)x, mean, rstd,
                           gamma, M, C, (const float*)dres, (float*)dx, part);
    else
        hipLaunchKernelGGL((layernorm_bwd_kernel<bf16, bf16>), dim3(nb), dim3(256), lds, s, (const bf16*)dout, (const bf16*)x, mean, rstd,
                           gamma, M, C, (const bf16*)dres, (bf16*)dx, part);
    if (dgamma) {
        const float* rows; int nrows;
        gg_reduce_rows(part, nb, 2 * C, s, &rows, &nrows, GG_REDUCE_DIRECT_MAX);
        hipLaunchKernelGGL(ln_param_final_kernel, dim3((unsigned)gg_cdiv(C, 64)), dim3(64 * GG_FOLD_TY), 0, s, rows, nrows, C, dgamma, dbeta, accumulate);
    }
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_layernorm_bwd_colsum_rows(int64_t M) { return std::min(ln_blocks(M), 512); }
extern "C" int gg_layernorm_bwd_colsum(const void* dout, const void* x, int f32, const float* mean, const float* rstd, const float* gamma,
                                       int64_t M, int C, const void* dres, void* dx, float* part, void* stream) {
    GG_CHECK(dout && x && mean && rstd && gamma && dx && part && M > 0 && (C & 7) == 0 && C <= 640, "gg_layernorm_bwd_colsum: bad args (C %% 8, C <= 640)");
    const int nb = gg_layernorm_bwd_colsum_rows(M);
    GG_PROF(GG_CAT_NORM, 0, (dres ? 4.0 : 3.0) * (f32 ? 4.0 : 2.0) * M * C, stream);
    const size_t lds = (size_t)4 * 2 * C * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define GG_LN_BWD2(T_, N_) hipLaunchKernelGGL((layernorm_bwd_g16_kernel<T_, N_, 2>), dim3(nb), dim3(256), lds, s, (const T_*)dout, (const T_*)x, \
                                              mean, rstd, gamma, M, C, (const T_*)dres, (T_*)dx, part)
#define GG_LN_BWD(N_) do { if (f32) GG_LN_BWD2(float, N_); else GG_LN_BWD2(bf16, N_); } while (0)
    switch ((C / 8 + 15) / 16) { case 1: GG_LN_BWD(1); break; case 2: GG_LN_BWD(2); break; case 3: GG_LN_BWD(3); break; case 4: GG_LN_BWD(4); break; default: GG_LN_BWD(5); }
#undef GG_LN_BWD
#undef GG_LN_BWD2
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_bn_bwd_coef_from_x(float* part, int nparts, int C, int64_t count, const float* stat, const float* gamma, const float* beta,
                                     float* coef, void* stream) {
    GG_CHECK(part && stat && gamma && beta && coef && nparts > 0 && C > 0 && count > 0, "gg_bn_bwd_coef_from_x: bad args");
    const float* rows; int nrows;
    gg_reduce_rows(part, nparts, 2 * C, (hipStream_t)stream, &rows, &nrows, GG_REDUCE_DIRECT_MAX);
    hipLaunchKernelGGL(bn_bwd_coef_from_x_kernel, dim3((unsigned)gg_cdiv(C, 64)), dim3(64 * GG_FOLD_TY), 0, (hipStream_t)stream, rows, nrows, C,
                       (double)count, stat, gamma, beta, coef);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_token_mean_fwd(const void* x, float* out, int B, int T, int C, void* stream) {
    GG_CHECK(x && out && B > 0 && T > 0 && (C & 7) == 0, "gg_token_mean_fwd: bad args");
    hipLaunchKernelGGL(token_mean_fwd_kernel<bf16>, dim3(grid_for((int64_t)B * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, out, B, T, C);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_token_mean_bwd(const float* dout, void* dx, int B, int T, int C, void* stream) {
    GG_CHECK(dout && dx && B > 0 && T > 0 && (C & 7) == 0, "gg_token_mean_bwd: bad args");
    hipLaunchKernelGGL(token_mean_bwd_kernel<bf16>, dim3(grid_for((int64_t)B * T * (C / 8))), dim3(256), 0, (hipStream_t)stream, dout, (bf16*)dx, B, T, C);
    GG_LAUNCH_CHECK();
    return 0;
}
// fp16 storage (CLIP tower, act_dtype 2): LayerNorm forward and the token mean
extern "C" int gg_layernorm_fwd_f16(const void* x, const float* gamma, const float* beta, int64_t M, int C, float eps, void* out, void* stream) {
    GG_CHECK(x && gamma && beta && out && M > 0 && (C & 7) == 0 && C <= 1024, "gg_layernorm_fwd_f16: bad args (C %% 8, C <= 1024)");
    GG_PROF(GG_CAT_NORM, 0, 4.0 * M * C, stream);
    hipLaunchKernelGGL((layernorm_fwd_kernel<f16, f16>), dim3(ln_blocks(M)), dim3(256), 0, (hipStream_t)stream, (const f16*)x, gamma, beta, M, C, eps, (f16*)out,
                       (float*)nullptr, (float*)nullptr);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_token_mean_fwd_f16(const void* x, float* out, int B, int T, int C, void* stream) {
    GG_CHECK(x && out && B > 0 && T > 0 && (C & 7) == 0, "gg_token_mean_fwd_f16: bad args");
    hipLaunchKernelGGL(token_mean_fwd_kernel<f16>, dim3(grid_for((int64_t)B * (C / 8))), dim3(256), 0, (hipStream_t)stream, (const f16*)x, out, B, T, C);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_token_mean_fwd_f32(const float* x, float* out, int B, int T, int C, void* stream) {
    GG_CHECK(x && out && B > 0 && T > 0 && (C & 7) == 0, "gg_token_mean_fwd_f32: bad args");
    hipLaunchKernelGGL(token_mean_fwd_kernel<float>, dim3(grid_for((int64_t)B * (C / 8))), dim3(256), 0, (hipStream_t)stream, x, out, B, T, C);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_token_mean_bwd_f32(const float* dout, float* dx, int B, int T, int C, void* stream) {
    GG_CHECK(dout && dx && B > 0 && T > 0 && (C & 7) == 0, "gg_token_mean_bwd_f32: bad args");
    hipLaunchKernelGGL(token_mean_bwd_kernel<float>, dim3(grid_for((int64_t)B * T * (C / 8))), dim3(256), 0, (hipStream_t)stream, dout, dx, B, T, C);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_view_mean_fwd(const float* emb, void* out, int64_t ldo, int N, int V, int C, void* stream) {
    GG_CHECK(emb && out && N > 0 && V > 0 && C > 0 && ldo >= C, "gg_view_mean_fwd: bad args");
    hipLaunchKernelGGL(view_mean_fwd_kernel<bf16>, dim3(grid_for((int64_t)N * C)), dim3(256), 0, (hipStream_t)stream, emb, (bf16*)out, ldo, N, V, C);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_view_mean_bwd(const void* dmean, int64_t ld, float* demb, int N, int V, int C, void* stream) {
    GG_CHECK(dmean && demb && N > 0 && V > 0 && C > 0 && ld >= C, "gg_view_mean_bwd: bad args");
    hipLaunchKernelGGL(view_mean_bwd_kernel<bf16>, dim3(grid_for((int64_t)N * V * C)), dim3(256), 0, (hipStream_t)stream, (const bf16*)dmean, ld, demb, N, V, C);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_view_mean_fwd_f32(const float* emb, float* out, int64_t ldo, int N, int V, int C, void* stream) {
    GG_CHECK(emb && out && N > 0 && V > 0 && C > 0 && ldo >= C, "gg_view_mean_fwd_f32: bad args");
    hipLaunchKernelGGL(view_mean_fwd_kernel<float>, dim3(grid_for((int64_t)N * C)), dim3(256), 0, (hipStream_t)stream, emb, out, ldo, N, V, C);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_view_mean_bwd_f32(const float* dmean, int64_t ld, float* demb, int N, int V, int C, void* stream) {
    GG_CHECK(dmean && demb && N > 0 && V > 0 && C > 0 && ld >= C, "gg_view_mean_bwd_f32: bad args");
    hipLaunchKernelGGL(view_mean_bwd_kernel<float>, dim3(grid_for((int64_t)N * V * C)), dim3(256), 0, (hipStream_t)stream, dmean, ld, demb, N, V, C);
    GG_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------- DropPath keep/scale rows (timm DropPath, scale_by_keep)
// out[s][b] = Bernoulli(1 - rate[s]) / (1 - rate[s]) per (slot, sample): what timm's drop_path draws with x.new_empty(...).bernoulli_(keep)
// / keep (models/tinyvit.py:135 is the call site).  Counter-based generator (two rounds of the splitmix64 finaliser over
// (seed, call counter, element index)): any launch shape gives the same stream, nothing is kept on the device between calls.
__device__ __forceinline__ uint64_t gg_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void drop_path_scales_kernel(const float* __restrict__ rates, int slots, int batch, uint64_t seed, uint64_t counter,
                                        float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slots * batch) return;
    const float keep = 1.0f - rates[i / batch];
    const uint64_t r = gg_mix64(gg_mix64(seed + 0x9E3779B97F4A7C15ull * (counter + 1)) ^ (uint64_t)i * 0xD1342543DE82EF95ull);
    const float u = (float)(r >> 40) * (1.0f / 16777216.0f);          // 24 uniform bits in [0, 1)
    out[i] = (keep >= 1.0f) ? 1.0f : (u < keep ? 1.0f / keep : 0.0f);
}
extern "C" int gg_drop_path_scales(const float* rates, int slots, int batch, uint64_t seed, uint64_t counter, float* out, void* stream) {
    GG_CHECK(rates && out && slots > 0 && batch > 0, "gg_drop_path_scales: bad args");
    hipLaunchKernelGGL(drop_path_scales_kernel, dim3((unsigned)gg_cdiv((int64_t)slots * batch, 256)), dim3(256), 0, (hipStream_t)stream, rates,
                       slots, batch, seed, counter, out);
    GG_LAUNCH_CHECK();
    return 0;
}
