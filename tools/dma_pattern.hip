// dev microbenchmark: L2/HBM -> LDS bandwidth of the fp32 ring GEMM's operand stream WITHOUT the matrix work, for two fragment shapes:
//   SK = 16: a DMA instruction fetches 16 rows x 64 B (what gemm_nt_f32_ring_kernel does), SK = 32: 8 rows x 128 B (whole cache lines).
// Same ring (3 stages x 16 KB, counted vmcnt, one barrier per stage), same tile walk (tn fastest, XCD-contiguous), 3 workgroups per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/dma_pattern tools/dma_pattern.hip ; run: tools/bin/dma_pattern M N K
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int SK>
__global__ __launch_bounds__(256, 3) void dma_kernel(const float* A, const float* B, int M, int N, int K, int tilesN, int tiles, float* sink) {
    constexpr int NST = 3, STAGE = 4096;                 // floats per stage (16 KB)
    constexpr int ROWS = STAGE / SK;                     // rows per stage (A rows + B rows), 256 or 128
    constexpr int RPI = 1024 / (SK * 4);                 // rows per DMA instruction (16 or 8)
    constexpr int IPW = ROWS / RPI / 4;                  // instructions per wave per stage (= 4)
    constexpr int BMR = ROWS / 2;                        // rows of A (= rows of B) per tile
    __shared__ __attribute__((aligned(16))) float smem[NST * STAGE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bid = ((blockIdx.x & 7) * ((tiles + 7) >> 3) + (blockIdx.x >> 3));   // XCD-contiguous
    if (bid >= tiles) return;
    const int tm = bid / tilesN, tn = bid % tilesN;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)tm * BMR * K), 0, BMR * K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)tn * BMR * K), 0, BMR * K * 4, 0x00020000);
    const int lpr = SK / 4;                              // lanes per row
    const int drow = lane / lpr, dch = lane % lpr;
    unsigned voff[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
        const int blk = wave + 4 * j;                    // blocks of RPI rows: first half A, second half B
        const int row = (blk % (BMR / RPI)) * RPI + drow;
        voff[j] = (unsigned)row * (unsigned)K * 4u + dch * 16u;
    }
    const int nk = K / SK;
    auto issue = [&](int st) {
        float* base = smem + (st % NST) * STAGE;
#pragma unroll
        for (int j = 0; j < IPW; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds((wave + 4 * j) < (BMR / RPI) ? rsA : rsB, (__attribute__((address_space(3))) void*)(base + (wave + 4 * j) * 256), 16,
                                                     (int)voff[j], st * SK * 4, 0, 0);
    };
    for (int st = 0; st < NST && st < nk; ++st) issue(st);
    float acc = 0.f;
    for (int s = 0; s < nk; ++s) {
        const int later = min(nk - 1, s + NST - 1) - s;
        if (later >= 2) wait_vmcnt<2 * IPW>(); else if (later == 1) wait_vmcnt<IPW>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        acc += smem[(s % NST) * STAGE + threadIdx.x * 4];
        __builtin_amdgcn_s_barrier();
        if (s + NST < nk) issue(s + NST);
    }
    if (acc == 12345.678f) sink[0] = acc;
}
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 200704, N = argc > 2 ? atoi(argv[2]) : 1152, K = argc > 3 ? atoi(argv[3]) : 384;
    float *A, *B, *sink;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&sink, 4);
    hipMemset(A, 1, (size_t)M * K * 4); hipMemset(B, 1, (size_t)N * K * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int sk : {16, 32}) {
        const int bmr = sk == 16 ? 128 : 64;
        const int tilesM = M / bmr, tilesN = N / bmr, tiles = tilesM * tilesN;
        float ms = 0;
        for (int it = 0; it < 4; ++it) {
            hipEventRecord(e0);
            if (sk == 16) hipLaunchKernelGGL(dma_kernel<16>, dim3(((tiles + 7) / 8) * 8), dim3(256), 0, 0, A, B, M, N, K, tilesN, tiles, sink);
            else hipLaunchKernelGGL(dma_kernel<32>, dim3(((tiles + 7) / 8) * 8), dim3(256), 0, 0, A, B, M, N, K, tilesN, tiles, sink);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        }
        const double bytes = (double)tiles * (K / sk) * 16384.0;
        printf("SK=%d (%d rows x %d B per instr): %d tiles, %.1f us, %.2f TB/s into LDS (%.1f GB/s per CU)\n", sk, 1024 / (sk * 4), sk * 4, tiles, ms * 1e3,
               bytes / ms / 1e9, bytes / ms / 1e6 / 256);
    }
    return 0;
}
