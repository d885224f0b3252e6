// dev microbenchmark: sustained v_mfma_f32_16x16x4_f32 / 32x32x2 rate of the whole chip (what "100 %" means for the fp32 GEMMs)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int ACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
    f32x4 acc[ACC];
    for (int i = 0; i < ACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < ACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {256, 512, 768, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            const int iters = 20000;
            hipEventRecord(e0);
            hipLaunchKernelGGL(k16<16>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fl = (double)wgs * 4 * iters * 16 * 2.0 * 16 * 16 * 4;
            printf("16x16x4  wgs %4d: %.2f ms  %.1f TFLOP/s\n", wgs, ms, fl / ms / 1e9);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k32, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            fl = (double)wgs * 4 * iters * 4 * 2.0 * 32 * 32 * 2;
            printf("32x32x2  wgs %4d: %.2f ms  %.1f TFLOP/s\n", wgs, ms, fl / ms / 1e9);
        }
    }
    return 0;
}
