// dev microbenchmark: the operand stream of a bf16 NT GEMM WITHOUT the matrix work -- how fast does LDS-DMA deliver A / B tiles into LDS, as a function of the piece shape
// (rows x bytes per `buffer_load ... lds` instruction), the tile (flop per byte) and the ring depth.  Same tile walk as gemm_nt_dma_kernel (XCD-contiguous, tn fastest),
// counted vmcnt, one barrier per stage.  Reports TB/s into LDS and the bf16 MFMA rate that stream could feed (TB/s x flop per operand byte of the tile).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/dma_bw tools/dma_bw.hip ; run: tools/bin/dma_bw M N K
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// BM x BN tile, SKB bytes of k per row and stage, NST ring stages, NTHR threads
template <int BM, int BN, int SKB, int NST, int NTHR, int OCC>
__global__ __launch_bounds__(NTHR, OCC) void dma_kernel(const char* A, const char* B, int M, int N, int Kb, int tilesN, int tiles, float* sink) {
    constexpr int NW = NTHR / 64;
    constexpr int STAGE = (BM + BN) * SKB;               // bytes per stage
    constexpr int RPI = 1024 / SKB;                      // rows per DMA instruction
    constexpr int PIECES = (BM + BN) / RPI;              // 1 KB pieces per stage
    constexpr int IPW = PIECES / NW;                     // per wave
    constexpr int LPR = SKB / 16;                        // lanes per row
    static_assert(PIECES % NW == 0, "pieces must divide over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = tiles >> 3, r = tiles & 7, x = blockIdx.x & 7, y = blockIdx.x >> 3;
    const int bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    const int tm = bid / tilesN, tn = bid % tilesN;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)tm * BM * Kb), 0, BM * Kb, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)tn * BN * Kb), 0, BN * Kb, 0x00020000);
    unsigned voff[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
        const int pc = wave + NW * j;                    // piece: first BM / RPI are A rows, then B rows
        const int row = (pc < BM / RPI ? pc : pc - BM / RPI) * RPI + lane / LPR;
        voff[j] = (unsigned)row * (unsigned)Kb + (lane % LPR) * 16u;
    }
    const int nk = Kb / SKB;
    auto issue = [&](int st) {
        char* base = smem + (st % NST) * STAGE;
#pragma unroll
        for (int j = 0; j < IPW; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds((wave + NW * j) < BM / RPI ? rsA : rsB, (__attribute__((address_space(3))) void*)(base + (wave + NW * j) * 1024), 16,
                                                     (int)voff[j], st * SKB, 0, 0);
    };
    for (int st = 0; st < NST - 1 && st < nk; ++st) issue(st);
    float acc = 0.f;
    for (int s = 0; s < nk; ++s) {
        // stage s landed (NST - 2 later stages may be in flight), barrier, refill the buffer stage s - 1 used
        const int later = min(nk - 1, s + NST - 2) - s;
        if (later >= 2) wait_vmcnt<2 * IPW>(); else if (later == 1) wait_vmcnt<IPW>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (s + NST - 1 < nk) issue(s + NST - 1);
        acc += *reinterpret_cast<float*>(smem + (s % NST) * STAGE + threadIdx.x * 16);
    }
    if (acc == 12345.678f) sink[0] = acc;
}
template <int BM, int BN, int SKB, int NST, int NTHR, int OCC>
static void run(const char* name, const char* A, const char* B, int M, int N, int K, float* sink) {
    const int Kb = K * 2, tilesM = M / BM, tilesN = N / BN, tiles = tilesM * tilesN;
    const size_t lds = (size_t)NST * (BM + BN) * SKB;
    hipFuncSetAttribute((const void*)dma_kernel<BM, BN, SKB, NST, NTHR, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 5; ++it) {
        float ms;
        hipEventRecord(e0);
        hipLaunchKernelGGL((dma_kernel<BM, BN, SKB, NST, NTHR, OCC>), dim3(tiles), dim3(NTHR), lds, 0, A, B, M, N, Kb, tilesN, tiles, sink);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    const double bytes = (double)tiles * (Kb / SKB) * (BM + BN) * SKB;
    const double fpb = 2.0 * BM * BN / ((BM + BN) * 2.0);        // flop per operand byte
    printf("%-34s tile %3dx%3d  %2d rows x %3d B per piece, ring %d x %3d KB, %d thr x %d/CU: %8.1f us  %6.2f TB/s into LDS (%5.1f GB/s/CU)  feeds %5.0f TF\n", name, BM, BN,
           1024 / SKB, SKB, NST, (BM + BN) * SKB / 1024, NTHR, OCC, best * 1e3, bytes / best / 1e9, bytes / best / 1e6 / 256, bytes / best / 1e9 * fpb);
}
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 51200, N = argc > 2 ? atoi(argv[2]) : 2304, K = argc > 3 ? atoi(argv[3]) : 768;
    char *A, *B; float* sink;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&sink, 4);
    hipMemset(A, 1, (size_t)M * K * 2); hipMemset(B, 1, (size_t)N * K * 2);
    printf("M=%d N=%d K=%d\n", M, N, K);
    run<256, 128, 64, 3, 256, 2>("current (k32 stages)", A, B, M, N, K, sink);
    run<256, 128, 64, 4, 256, 2>("k32 stages, ring 4 (1/CU by LDS)", A, B, M, N, K, sink);
    run<256, 128, 128, 2, 256, 1>("k64 stages, ring 2", A, B, M, N, K, sink);
    run<256, 128, 128, 3, 256, 1>("k64 stages, ring 3", A, B, M, N, K, sink);
    run<256, 256, 128, 2, 512, 1>("256^2 k64 ring 2, 8 waves", A, B, M, N, K, sink);
    run<256, 256, 64, 4, 512, 1>("256^2 k32 ring 4, 8 waves", A, B, M, N, K, sink);
    run<256, 256, 64, 3, 512, 1>("256^2 k32 ring 3, 8 waves", A, B, M, N, K, sink);
    run<192, 128, 128, 2, 256, 2>("192x128 k64 ring 2, 2/CU", A, B, M, N, K, sink);
    run<128, 128, 128, 2, 256, 2>("128^2 k64 ring 2, 2/CU", A, B, M, N, K, sink);
    run<128, 128, 64, 3, 256, 3>("128^2 k32 ring 3, 3/CU", A, B, M, N, K, sink);
    run<128, 256, 128, 2, 256, 1>("128x256 k64 ring 2", A, B, M, N, K, sink);
    return 0;
}
