// dev microbenchmark: issue rate of v_mfma_f32_16x16x16_bf16 and v_mfma_f32_16x16x32_bf16, as one dependent accumulation chain and as four independent ones
// (one wave per SIMD, operands in registers).  build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/mr tools/mfma_rate16.hip && /tmp/mr
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int KIND, int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    s4 a4 = {(short)threadIdx.x, 1, 2, 3}, b4 = {3, 2, 1, (short)threadIdx.x};
    b8 a8, b8v;
    for (int j = 0; j < 8; ++j) { a8[j] = (__bf16)(float)(threadIdx.x + j); b8v[j] = (__bf16)(float)(j + 1); }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int c = CHAINS == 1 ? 0 : (u & 3);
            if (KIND == 16) acc[c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[c], 0, 0, 0);
            else acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8v, acc[c], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND, int CHAINS> void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-40s %.1f cycles per MFMA (one wave per SIMD)\n", name, (double)c / (iters * 16.0));
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    run<16, 1>("16x16x16 bf16, one dependent chain", out, cyc);
    run<16, 4>("16x16x16 bf16, four chains", out, cyc);
    run<32, 1>("16x16x32 bf16, one dependent chain", out, cyc);
    run<32, 4>("16x16x32 bf16, four chains", out, cyc);
    return 0;
}
