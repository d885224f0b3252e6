// Shared device/host helpers for libgg (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define GG_WAVE 64

// ---- error plumbing (api.cpp) -------------------------------------------------
extern "C" const char* gg_last_error(void);
void gg_set_error(const char* fmt, ...);
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
__device__ __forceinline__ float gg_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gg_gelu_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
__device__ __forceinline__ float gg_quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float gg_quick_gelu_grad(float x) {
    const float s = 1.0f / (1.0f + __expf(-1.702f * x));
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

// XCD-aware block remap: consecutive logical ids land on the same XCD (bijective for any n).
__device__ __forceinline__ int gg_xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7, k = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}
