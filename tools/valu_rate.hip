// dev microbenchmark: issue rate of the vector instructions the split-bf16 conversions use (one wave per SIMD, 8 independent registers, inline asm so that the
// instruction measured is the instruction named).  build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/vr tools/valu_rate.hip && /tmp/vr
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP2(name, text)                                                                             \
    struct name { static __device__ __forceinline__ void op(unsigned& d, unsigned a, unsigned b) { asm volatile(text : "+v"(d) : "v"(a), "v"(b)); } };
OP2(CvtPk, "v_cvt_pk_bf16_f32 %0, %0, %1")
OP2(And, "v_and_b32 %0, %0, %1")
OP2(Perm, "v_perm_b32 %0, %0, %1, %2")
OP2(Sub, "v_sub_f32 %0, %0, %1")
OP2(Lshl, "v_lshlrev_b32 %0, 16, %0")
OP2(Exp, "v_exp_f32 %0, %0")
OP2(Fma, "v_fma_f32 %0, %0, %1, %2")
OP2(AndOr, "v_and_or_b32 %0, %0, %1, %2")
struct PkAdd { static __device__ __forceinline__ void op(unsigned long long& d, unsigned long long a) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d) : "v"(a)); } };
template <typename O>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned long long* cyc) {
    unsigned r[8];
    for (int j = 0; j < 8; ++j) r[j] = threadIdx.x * 7 + j;
    const unsigned a = 0x3f800000u + threadIdx.x, b = 0x07060302u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u) O::op(r[u & 7], a, b);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned s = 0;
    for (int j = 0; j < 8; ++j) s ^= r[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
__global__ __launch_bounds__(256) void kpk(unsigned* out, int iters, unsigned long long* cyc) {
    unsigned long long r[8];
    for (int j = 0; j < 8; ++j) r[j] = threadIdx.x * 7 + j;
    const unsigned long long a = 0x3f8000003f800000ull;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u) PkAdd::op(r[u & 7], a);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long s = 0;
    for (int j = 0; j < 8; ++j) s ^= r[j];
    out[blockIdx.x * 256 + threadIdx.x] = (unsigned)s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <typename O> void run(const char* name, unsigned* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<O>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-24s %.2f cycles per instruction (one wave per SIMD)\n", name, (double)c / (iters * 32.0));
}
int main() {
    unsigned* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    run<CvtPk>("v_cvt_pk_bf16_f32", out, cyc);
    run<And>("v_and_b32", out, cyc);
    run<Perm>("v_perm_b32", out, cyc);
    run<Sub>("v_sub_f32", out, cyc);
    run<Lshl>("v_lshlrev_b32", out, cyc);
    run<Exp>("v_exp_f32", out, cyc);
    run<Fma>("v_fma_f32", out, cyc);
    run<AndOr>("v_and_or_b32", out, cyc);
    hipLaunchKernelGGL(kpk, dim3(256), dim3(256), 0, 0, out, 2000, cyc);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-24s %.2f cycles per instruction (one wave per SIMD)\n", "v_pk_add_f32", (double)c / (2000 * 32.0));
    return 0;
}
