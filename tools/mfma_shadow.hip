// dev microbenchmark: how many independent VALU / LDS / store instructions of the SAME wave fit in the shadow of a v_mfma_f32_16x16x4_f32
// (8 passes = 32 cycles)?  One wave per SIMD (256 workgroups of 256 threads), loop of 16 MFMAs each followed by K filler instructions.
// Companion of valu_under_mfma.hip (another wave's VALU / store instructions do NOT issue while a wave has MFMAs pending).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int K, int KIND>
__global__ __launch_bounds__(256) void shadow(float* out, unsigned long long* cyc, int iters, float a, float b) {
    __shared__ float lds[8192];
    lds[threadIdx.x] = a;
    __syncthreads();
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    f32x4 w = {a, b, a, b};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 pv[4], pw = {b, a};
    for (int i = 0; i < 4; ++i) pv[i] = (f32x2){a + i, b + i};
    const unsigned addr = threadIdx.x * 16;
    float* dst = out + (size_t)blockIdx.x * 256 * 64 + threadIdx.x * 4;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i * K + k) & 7]) : "v"(b), "v"(a));
                else if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pv[(i * K + k) & 3]) : "v"(pw));
                else if (KIND == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(i * K + k) & 7]));
                else if (KIND == 1) { f32x4 r; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(((0 * 16) & 0xFFF))); }
                else asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(dst), "v"(w) : "memory");
            }
        }
        if (KIND == 1 || KIND == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = w[0];
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += pv[i].x + pv[i].y;
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int K, int KIND>
static void run(float* out, unsigned long long* cyc, const char* name) {
    const int iters = 2000;
    hipLaunchKernelGGL((shadow<K, KIND>), dim3(256), dim3(256), 0, 0, out, cyc, iters, 1.0f, 0.5f);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; ++i) s += (double)h[i];
    printf("%-22s x %d per MFMA: %.1f cycles per MFMA (+%d fillers)\n", name, K, s / 256 / iters / 16, K);
}
int main() {
    float* out; hipMalloc(&out, (size_t)256 * 256 * 64 * 4);
    unsigned long long* cyc; hipMalloc(&cyc, 256 * 8);
    run<0, 0>(out, cyc, "v_fma_f32"); run<1, 0>(out, cyc, "v_fma_f32"); run<2, 0>(out, cyc, "v_fma_f32"); run<4, 0>(out, cyc, "v_fma_f32");
    run<6, 0>(out, cyc, "v_fma_f32"); run<8, 0>(out, cyc, "v_fma_f32"); run<12, 0>(out, cyc, "v_fma_f32");
    run<1, 1>(out, cyc, "ds_read_b128"); run<2, 1>(out, cyc, "ds_read_b128"); run<4, 1>(out, cyc, "ds_read_b128");
    run<1, 2>(out, cyc, "global_store_dwordx4"); run<2, 2>(out, cyc, "global_store_dwordx4");
    run<1, 3>(out, cyc, "v_pk_fma_f32"); run<2, 3>(out, cyc, "v_pk_fma_f32"); run<4, 3>(out, cyc, "v_pk_fma_f32");
    run<1, 4>(out, cyc, "v_exp_f32"); run<2, 4>(out, cyc, "v_exp_f32");
    return 0;
}
