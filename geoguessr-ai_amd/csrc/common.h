// Shared device/host helpers for libgg (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "prof.h"

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16;                                  // fp16 storage of the CLIP tower's fp16 mode (fp32 accumulation everywhere)
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define GG_WAVE 64

// ---- error plumbing (api.cpp) -------------------------------------------------
extern "C" const char* gg_last_error(void);
#define GG_CHECK(cond, ...)                         \
    do {                                            \
        if (!(cond)) {                              \
            gg_set_error(__VA_ARGS__);              \
            return -1;                              \
        }                                           \
    } while (0)
#define GG_HIP(call)                                                                  \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                       \
            gg_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return -2;                                                                \
        }                                                                             \
    } while (0)
#define GG_LAUNCH_CHECK()                                                             \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            gg_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
            return -3;                                                                \
        }                                                                             \
    } while (0)
#define GG_TRY(call)            \
    do {                        \
        int r_ = (call);        \
        if (r_ != 0) return r_; \
    } while (0)

// Development switches (A/B timing of alternative kernels / schedules): read ONLY when GG_DEV_SWITCHES is set, so that a stray variable in a
// production environment cannot move the library off its tested default path.  Every alternative a switch selects is also a shape fallback of
// the default path; tests/test_gpu_switches.py runs the parity step under each group of switches.
#include <stdlib.h>
static inline const char* gg_dev_env(const char* name) {
    static const bool on = getenv("GG_DEV_SWITCHES") != nullptr;
    return on ? getenv(name) : nullptr;
}
// x = a + b + c with three bf16 terms (RNE; exact for every finite f32 whose low terms stay normal: 3 x 8 significand bits).  The first term is formed from x clamped
// to the largest finite bf16 (3.39e38): bf16(x) alone rounds the top half-ulp of the f32 range (|x| >= 3.3961e38) to inf, which would turn a finite operand into
// (inf, -inf, NaN).  +-inf / NaN operands give non-finite planes (NaN in the product where an f32 GEMM may keep +-inf: inf - inf in the residual).
#define GG_BF16_MAX_F 3.3895313892515355e38f
__device__ __forceinline__ void gg_split3_rne(float x, __bf16& a, __bf16& b, __bf16& c) {
    a = (__bf16)__builtin_amdgcn_fmed3f(x, -GG_BF16_MAX_F, GG_BF16_MAX_F);
    const float r1 = x - (float)a;
    b = (__bf16)r1;
    c = (__bf16)(r1 - (float)b);
}
static inline int64_t gg_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t gg_align(int64_t a, int64_t b) { return gg_cdiv(a, b) * b; }

// ---- device math ---------------------------------------------------------------
// erf(x/sqrt2) as an odd polynomial u*P(u^2), u = |x|/sqrt2, fitted on [0, 3] and clamped to 1 on the way out (least-squares Chebyshev fit, degree 17): no
// transcendental, and the FMAs pair up into v_pk_fma_f32.  |erf error| <= 2.1e-5 (1 - erf(3) = 2.2e-5 is the clamp),
// i.e. |GELU error| <= 4.4e-5 absolute, 100x below the bf16 resolution every GELU result is stored with.  libm erff
// and the Abramowitz-Stegun form (rcp + exp) cost 5x / 2.5x more VALU and made the GELU epilogues, not HBM or MFMA,
// the limiter of the fc1 GEMMs and the BatchNorm+GELU kernels (19 G evaluations per 1024-image step).
__device__ __forceinline__ float gg_erf_sqrt2(float x) {
    const float u = fabsf(x) * 0.70710678118654752f;      // no clamp on the way in: beyond the fit range the polynomial rises monotonically
    const float t = u * u;                                // (>= 0.99999 for every u >= 3, +inf on overflow) and the clamp on the way out saturates it
    float p = 3.912539981e-08f;
    p = fmaf(p, t, -1.883036475e-06f);
    p = fmaf(p, t, 4.008835822e-05f);
    p = fmaf(p, t, -5.029218737e-04f);
    p = fmaf(p, t, 4.196857568e-03f);
    p = fmaf(p, t, -2.499890141e-02f);
    p = fmaf(p, t, 1.109308004e-01f);
    p = fmaf(p, t, -3.752196431e-01f);
    p = fmaf(p, t, 1.128250599e+00f);
    return copysignf(fminf(p * u, 1.0f), x);
}
// ---- fp32 (reference-precision) activations ---------------------------------------------------------------------------------
// GELU needs Phi(x) = 0.5 erfc(-x / sqrt2), not erf: h(t) = 0.5 erfc(t) = exp2(P(t)) for t = |x| / sqrt2 <= 4.2 (h(4.2) = 1.4e-9; continued linearly in the exponent beyond),
// P a degree-9 polynomial fitted with weight h (tools/fit_phi.py; every step rounded to fp32 in the check), and Phi = x < 0 ? h : 1 - h.
// One branch-free path, 9 FMAs + one v_exp_f32: max |error of Phi| 7.3e-8 (1.2 ulp of 1), and no cancellation on the negative side
// (relative error of GELU 4e-6 where |GELU| > 1e-3; 0.5 (1 + erf) loses it there).  libm erff + expf cost ~95 instructions per GELU'.
__device__ __forceinline__ float gg_phi_f32(float x) {
    const float tu = fabsf(x) * 0.70710678118654752f;
    const float t = fminf(tu, 4.2f);
    float p = 1.15539833e-05f;
    p = fmaf(p, t, -0.000152371736f);
    p = fmaf(p, t, 0.000845456321f);
    p = fmaf(p, t, -0.00226795045f);
    p = fmaf(p, t, 7.51803382e-05f);
    p = fmaf(p, t, 0.0277323835f);
    p = fmaf(p, t, -0.1483116f);
    p = fmaf(p, t, -0.918442011f);
    p = fmaf(p, t, -1.6279074f);
    p = fmaf(p, t, -1.0f);
    p = fmaf(tu - t, -12.2f, p);          // beyond the fit range: keep falling at the slope log2 h has at 4.2, so Phi(x) -> 0 (and x Phi(x) -> 0) for x -> -inf
    const float h = __builtin_amdgcn_exp2f(p);
    return x < 0.f ? h : 1.0f - h;
}
__device__ __forceinline__ float gg_gelu_f32(float x) { return x * gg_phi_f32(x); }
__device__ __forceinline__ float gg_gelu_grad_f32(float x) {       // Phi(x) + x * phi(x)
    return fmaf(x * 0.3989422804014327f, __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f), gg_phi_f32(x));
}
__device__ __forceinline__ float gg_act_f32(float x, int act) {
    if (act == 1 /* GG_ACT_GELU */) return gg_gelu_f32(x);
    if (act == 2 /* GG_ACT_QUICK_GELU */) return x / (1.0f + __expf(-1.702f * x));
    return x;
}
__device__ __forceinline__ float gg_act_grad_f32(float x, int act) {
    if (act == 1) return gg_gelu_grad_f32(x);
    if (act == 2) { const float sg = 1.0f / (1.0f + __expf(-1.702f * x)); return sg + 1.702f * x * sg * (1.0f - sg); }
    return 1.0f;
}
// The same fp32 activations four at a time.  The polynomial of Phi runs on float2 values, i.e. as v_pk_fma_f32 (two elements per VALU slot; the
// products are the same fused multiply-adds as in gg_phi_f32, so the results are bit-identical to the scalar forms): the fp32 GELU sites that are
// VALU-bound -- BatchNorm + GELU on the depthwise load, the GEMM prologue / GELU epilogues -- spend ~30 % fewer issue slots per element.
__device__ __forceinline__ f32x2 gg_phi_f32_v2(f32x2 x) {
    const f32x2 tu = (f32x2){fabsf(x.x), fabsf(x.y)} * 0.70710678118654752f;
    const f32x2 t = {fminf(tu.x, 4.2f), fminf(tu.y, 4.2f)};
    f32x2 p = (f32x2)(1.15539833e-05f);
    p = p * t + (f32x2)(-0.000152371736f);
    p = p * t + (f32x2)(0.000845456321f);
    p = p * t + (f32x2)(-0.00226795045f);
    p = p * t + (f32x2)(7.51803382e-05f);
    p = p * t + (f32x2)(0.0277323835f);
    p = p * t + (f32x2)(-0.1483116f);
    p = p * t + (f32x2)(-0.918442011f);
    p = p * t + (f32x2)(-1.6279074f);
    p = p * t + (f32x2)(-1.0f);
    p = (tu - t) * (f32x2)(-12.2f) + p;
    const float h0 = __builtin_amdgcn_exp2f(p.x), h1 = __builtin_amdgcn_exp2f(p.y);
    return (f32x2){x.x < 0.f ? h0 : 1.0f - h0, x.y < 0.f ? h1 : 1.0f - h1};
}
__device__ __forceinline__ f32x2 gg_gelu_grad_f32_v2(f32x2 x) {        // Phi(x) + x * phi(x)
    const f32x2 e = x * x * (f32x2)(-0.72134752044448170f);
    const f32x2 pdf = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
    return (x * (f32x2)(0.3989422804014327f)) * pdf + gg_phi_f32_v2(x);
}
__device__ __forceinline__ f32x4 gg_act_f32_v4(f32x4 z, int act) {
    if (act == 1 /* GG_ACT_GELU */) {
        const f32x2 a = {z[0], z[1]}, b = {z[2], z[3]};
        const f32x2 ga = a * gg_phi_f32_v2(a), gb = b * gg_phi_f32_v2(b);
        return (f32x4){ga.x, ga.y, gb.x, gb.y};
    }
    if (act == 2 /* GG_ACT_QUICK_GELU */) return (f32x4){gg_act_f32(z[0], 2), gg_act_f32(z[1], 2), gg_act_f32(z[2], 2), gg_act_f32(z[3], 2)};
    return z;
}
__device__ __forceinline__ f32x4 gg_act_grad_f32_v4(f32x4 z, int act) {
    if (act == 1) {
        const f32x2 ga = gg_gelu_grad_f32_v2((f32x2){z[0], z[1]}), gb = gg_gelu_grad_f32_v2((f32x2){z[2], z[3]});
        return (f32x4){ga.x, ga.y, gb.x, gb.y};
    }
    if (act == 2) return (f32x4){gg_act_grad_f32(z[0], 2), gg_act_grad_f32(z[1], 2), gg_act_grad_f32(z[2], 2), gg_act_grad_f32(z[3], 2)};
    return (f32x4){1.f, 1.f, 1.f, 1.f};
}
__device__ __forceinline__ float gg_gelu(float x) { return 0.5f * x * (1.0f + gg_erf_sqrt2(x)); }
__device__ __forceinline__ float gg_gelu_grad_exp(float x) {       // Phi-polynomial + exp form (kept for reference / tests)
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return fmaf(x, pdf, 0.5f * (1.0f + gg_erf_sqrt2(x)));
}
__device__ __forceinline__ float gg_quick_gelu(float x) { return x * __frcp_rn(1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float gg_quick_gelu_grad(float x) {
    const float s = __frcp_rn(1.0f + __expf(-1.702f * x));
    return s + 1.702f * x * s * (1.0f - s);
}
// ---- two-at-a-time forms: every multiply-add below is a v_pk_fma_f32 (2 lanes of work per VALU slot) --------------
// The GELU epilogues and BatchNorm(+GELU) passes are VALU-bound once their memory traffic is fused away, so the hot
// element-wise code runs on float2 values.  erf: same polynomial as gg_erf_sqrt2, sign handled by the odd form
// (clamped signed argument, v_med3_f32) instead of abs / copysign.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gg_clamp2(f32x2 x, float a) {
    return (f32x2){__builtin_amdgcn_fmed3f(x.x, -a, a), __builtin_amdgcn_fmed3f(x.y, -a, a)};
}
__device__ __forceinline__ f32x2 gg_erf_sqrt2_v2(f32x2 x) {
    const f32x2 u = x * 0.70710678118654752f;             // (input clamp dropped: see gg_erf_sqrt2)
    const f32x2 t = u * u;
    f32x2 p = (f32x2)(3.912539981e-08f);
    p = p * t + (f32x2)(-1.883036475e-06f);
    p = p * t + (f32x2)(4.008835822e-05f);
    p = p * t + (f32x2)(-5.029218737e-04f);
    p = p * t + (f32x2)(4.196857568e-03f);
    p = p * t + (f32x2)(-2.499890141e-02f);
    p = p * t + (f32x2)(1.109308004e-01f);
    p = p * t + (f32x2)(-3.752196431e-01f);
    p = p * t + (f32x2)(1.128250599e+00f);
    return gg_clamp2(p * u, 1.0f);
}
__device__ __forceinline__ f32x2 gg_gelu_v2(f32x2 x) {
    const f32x2 h = x * 0.5f;
    return h * gg_erf_sqrt2_v2(x) + h;
}
// GELU'(x) = Phi(x) + x*phi(x) = 0.5 + u*Q(s), u = clamp(x, +-4.75), s = 2u^2/4.75^2 - 1  (odd part fitted directly,
// degree 23, Horner in the centred variable; |error| <= 1.2e-5 incl. fp32 evaluation; beyond the clamp GELU' is 0 / 1
// to 2.4e-5).  No exp, no erf: 8 VALU slots per element against 25 for Phi-polynomial + exp.
__device__ __forceinline__ f32x2 gg_gelu_grad_v2(f32x2 x) {
    const f32x2 u = gg_clamp2(x, 4.75f);
    const f32x2 s = (u * u) * 8.864265928e-02f + (f32x2)(-1.0f);
    f32x2 p = (f32x2)(-1.329242953e-02f);
    p = p * s + (f32x2)(2.763605337e-02f);
    p = p * s + (f32x2)(-1.633125265e-02f);
    p = p * s + (f32x2)(2.348791181e-02f);
    p = p * s + (f32x2)(-6.481112773e-02f);
    p = p * s + (f32x2)(8.656523583e-02f);
    p = p * s + (f32x2)(-8.702833417e-02f);
    p = p * s + (f32x2)(8.773912323e-02f);
    p = p * s + (f32x2)(-8.319626930e-02f);
    p = p * s + (f32x2)(7.598317362e-02f);
    p = p * s + (f32x2)(-8.164826059e-02f);
    p = p * s + (f32x2)(1.501617604e-01f);
    return p * u + (f32x2)(0.5f);
}
__device__ __forceinline__ float gg_gelu_grad(float x) { return gg_gelu_grad_v2((f32x2){x, x}).x; }
// act codes shared by GEMM epilogues and norm kernels
enum { GG_ACT_NONE = 0, GG_ACT_GELU = 1, GG_ACT_QUICK_GELU = 2 };
__device__ __forceinline__ float gg_act(float x, int act) {
    return act == GG_ACT_GELU ? gg_gelu(x) : (act == GG_ACT_QUICK_GELU ? gg_quick_gelu(x) : x);
}
__device__ __forceinline__ float gg_act_grad(float x, int act) {
    return act == GG_ACT_GELU ? gg_gelu_grad(x) : (act == GG_ACT_QUICK_GELU ? gg_quick_gelu_grad(x) : 1.0f);
}
// pairwise NONE / GELU only (the BatchNorm paths of TinyViT); `gelu` is wave-uniform
__device__ __forceinline__ f32x2 gg_act_v2(f32x2 x, bool gelu) { return gelu ? gg_gelu_v2(x) : x; }
__device__ __forceinline__ f32x2 gg_act_grad_v2(f32x2 x, bool gelu) { return gelu ? gg_gelu_grad_v2(x) : (f32x2)(1.0f); }

// ---- wave / block reductions (wave = 64 lanes) -----------------------------------
__device__ __forceinline__ float gg_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float gg_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double gg_wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// block-wide sum for blockDim.x == NT (multiple of 64); `red` has NT/64 floats; result on all threads
template <int NT>
__device__ __forceinline__ float gg_block_sum(float v, float* red) {
    v = gg_wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}
template <int NT>
__device__ __forceinline__ float gg_block_max(float v, float* red) {
    v = gg_wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = -INFINITY;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t = fmaxf(t, red[i]);
    return t;
}

// ---- two-stage column reduction of per-block partial rows ---------------------------------------------
// Producers (GEMM / dwconv / BN / LN kernels) leave one partial row per block: part[nparts][W].  With tens of
// thousands of producer blocks a one-thread-per-column finalize is a serial latency chain, so rows are first
// folded to <= GG_REDUCE_SLICES rows by a bandwidth-shaped kernel (coalesced along W, fp64 accumulation).  The folded
// rows are written BEHIND the valid rows of the same buffer, which therefore needs nparts + GG_REDUCE_SLICES rows.
#define GG_REDUCE_SLICES 64
// a finalize kernel shaped by gg_fold_cols2 (64 columns x GG_FOLD_TY row lanes per block) takes this many rows directly
#define GG_REDUCE_DIRECT_MAX 512
#define GG_FOLD_TY 16
static __global__ __launch_bounds__(1024) void gg_reduce_rows_kernel(const float* __restrict__ part, int nparts, int W,
                                                                     float* __restrict__ out, int rows_per_slice) {
    __shared__ double red[GG_FOLD_TY][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + tx;
    const int r0 = blockIdx.y * rows_per_slice, r1 = min(nparts, r0 + rows_per_slice);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (col < W) {
        const float* p = part + col;
        int r = r0 + ty;
        for (; r + 3 * GG_FOLD_TY < r1; r += 4 * GG_FOLD_TY) {          // four independent loads in flight per lane
            const float a = p[(int64_t)r * W], b = p[(int64_t)(r + GG_FOLD_TY) * W], c = p[(int64_t)(r + 2 * GG_FOLD_TY) * W],
                        d = p[(int64_t)(r + 3 * GG_FOLD_TY) * W];
            s0 += (double)a; s1 += (double)b; s2 += (double)c; s3 += (double)d;
        }
        for (; r < r1; r += GG_FOLD_TY) s0 += (double)p[(int64_t)r * W];
    }
    red[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty == 0 && col < W) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < GG_FOLD_TY; ++i) t += red[i][tx];
        out[(int64_t)blockIdx.y * W + col] = (float)t;
    }
}
// returns the rows to finalize from (either the originals or the folded ones) through *rows / *nrows; direct_max = the row count the
// caller's finalize kernel handles without folding
static inline void gg_reduce_rows(float* part, int nparts, int W, hipStream_t st, const float** rows, int* nrows, int direct_max = GG_REDUCE_SLICES) {
    if (nparts <= direct_max) { *rows = part; *nrows = nparts; return; }
    const int rps = (int)((nparts + GG_REDUCE_SLICES - 1) / GG_REDUCE_SLICES);
    const int slices = (nparts + rps - 1) / rps;
    float* out = part + (int64_t)nparts * W;
    hipLaunchKernelGGL(gg_reduce_rows_kernel, dim3((unsigned)((W + 63) / 64), (unsigned)slices), dim3(64 * GG_FOLD_TY), 0, st, part, nparts, W, out, rps);
    *rows = out; *nrows = slices;
}
// Block of 64 x GG_FOLD_TY threads (launch with dim3(64 * GG_FOLD_TY)): fp64 sums over rows [0, nrows) of part[.][W] at columns c0 and c1
// for the block's 64 channels.  The totals are valid in the threads with (threadIdx.x >> 6) == 0; `ok` = this lane's channel exists.
__device__ __forceinline__ void gg_fold_cols2(const float* __restrict__ part, int nrows, int64_t W, int c0, int c1, bool ok, double& s, double& q) {
    __shared__ double red[2][GG_FOLD_TY][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
    if (ok) {
        int r = ty;
        for (; r + GG_FOLD_TY < nrows; r += 2 * GG_FOLD_TY) {
            const float x0 = part[r * W + c0], y0 = part[r * W + c1], x1 = part[(r + GG_FOLD_TY) * W + c0], y1 = part[(r + GG_FOLD_TY) * W + c1];
            a0 += (double)x0; b0 += (double)y0; a1 += (double)x1; b1 += (double)y1;
        }
        if (r < nrows) { a0 += (double)part[r * W + c0]; b0 += (double)part[r * W + c1]; }
    }
    red[0][ty][tx] = a0 + a1;
    red[1][ty][tx] = b0 + b1;
    __syncthreads();
    s = 0.0; q = 0.0;
    if (ty == 0) {
#pragma unroll
        for (int i = 0; i < GG_FOLD_TY; ++i) { s += red[0][i][tx]; q += red[1][i][tx]; }
    }
}

// XCD-aware block remap: consecutive logical ids land on the same XCD (bijective for any n).
__device__ __forceinline__ int gg_xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7, k = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}
