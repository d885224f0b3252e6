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

static inline int64_t gg_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t gg_align(int64_t a, int64_t b) { return gg_cdiv(a, b) * b; }

// ---- device math ---------------------------------------------------------------
// erf by Abramowitz & Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32 round-off level) -- libm's erff costs ~5x more
// VALU and made the GELU epilogues, not HBM, the limiter of the fc1 GEMMs.  The exp(-u^2) term is shared with the
// Gaussian pdf needed by the derivative.
__device__ __forceinline__ void gg_erf_parts(float x, float& erf_v, float& expmu2) {
    const float u = fabsf(x) * 0.70710678118654752f;
    const float t = __frcp_rn(fmaf(0.3275911f, u, 1.0f));
    const float e = __expf(-u * u);
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float er = fmaf(-poly * t, e, 1.0f);
    erf_v = copysignf(er, x);
    expmu2 = e;
}
__device__ __forceinline__ float gg_gelu(float x) {
    float er, e;
    gg_erf_parts(x, er, e);
    return 0.5f * x * (1.0f + er);
}
__device__ __forceinline__ float gg_gelu_grad(float x) {
    float er, e;
    gg_erf_parts(x, er, e);
    return fmaf(x * 0.3989422804014327f, e, 0.5f * (1.0f + er));
}
__device__ __forceinline__ float gg_quick_gelu(float x) { return x * __frcp_rn(1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float gg_quick_gelu_grad(float x) {
    const float s = __frcp_rn(1.0f + __expf(-1.702f * x));
    return s + 1.702f * x * s * (1.0f - s);
}
// act codes shared by GEMM epilogues and norm kernels
enum { GG_ACT_NONE = 0, GG_ACT_GELU = 1, GG_ACT_QUICK_GELU = 2 };
__device__ __forceinline__ float gg_act(float x, int act) {
    return act == GG_ACT_GELU ? gg_gelu(x) : (act == GG_ACT_QUICK_GELU ? gg_quick_gelu(x) : x);
}
__device__ __forceinline__ float gg_act_grad(float x, int act) {
    return act == GG_ACT_GELU ? gg_gelu_grad(x) : (act == GG_ACT_QUICK_GELU ? gg_quick_gelu_grad(x) : 1.0f);
}

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
static __global__ __launch_bounds__(256) void gg_reduce_rows_kernel(const float* __restrict__ part, int nparts, int W,
                                                                    float* __restrict__ out, int rows_per_slice) {
    __shared__ double red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + tx;
    const int r0 = blockIdx.y * rows_per_slice, r1 = min(nparts, r0 + rows_per_slice);
    double s = 0.0;
    if (col < W)
        for (int r = r0 + ty; r < r1; r += 4) s += (double)part[(int64_t)r * W + col];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && col < W) out[(int64_t)blockIdx.y * W + col] = (float)(red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
}
// returns the rows to finalize from (either the originals or the folded ones) through *rows / *nrows
static inline void gg_reduce_rows(float* part, int nparts, int W, hipStream_t st, const float** rows, int* nrows) {
    if (nparts <= GG_REDUCE_SLICES) { *rows = part; *nrows = nparts; return; }
    const int rps = (int)((nparts + GG_REDUCE_SLICES - 1) / GG_REDUCE_SLICES);
    const int slices = (nparts + rps - 1) / rps;
    float* out = part + (int64_t)nparts * W;
    hipLaunchKernelGGL(gg_reduce_rows_kernel, dim3((unsigned)((W + 63) / 64), (unsigned)slices), dim3(256), 0, st, part, nparts, W, out, rps);
    *rows = out; *nrows = slices;
}

// XCD-aware block remap: consecutive logical ids land on the same XCD (bijective for any n).
__device__ __forceinline__ int gg_xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7, k = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}
