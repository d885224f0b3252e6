// microbenchmark: store / load patterns of the GEMM epilogue vs a contiguous stream (dev tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void fill_contig(f4* p, long n16) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n16; i += gridDim.x * 256L) p[i] = (f4){1, 2, 3, 4};
}
// tile pattern: C[M][N] bf16, block = 128x128 tile, thread stores 8 x 16B: row = pass*16 + tid/16, col chunk tid%16
__global__ __launch_bounds__(256) void fill_tile(char* C, int M, int N, int tilesN) {
    const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
    const int chunk = threadIdx.x & 15, rr = threadIdx.x >> 4;
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const long row = tm * 128L + pass * 16 + rr;
        *(f4*)(C + (row * N + tn * 128 + chunk * 8) * 2) = (f4){1, 2, 3, 4};
    }
}
// same bytes but a block owns 128 rows x full N (all N tiles), i.e. fully contiguous 128*N*2 bytes
__global__ __launch_bounds__(256) void fill_rowpanel(char* C, int M, int N) {
    const long base = (long)blockIdx.x * 128 * N * 2;
    const long bytes = 128L * N * 2;
    for (long o = threadIdx.x * 16L; o < bytes; o += 256 * 16) *(f4*)(C + base + o) = (f4){1, 2, 3, 4};
}
// tile pattern with the stores of one tile issued by ONE wave per 32 rows... (wave w -> rows w*32..): 4 KiB contiguous? no: 32 rows x 256B
__global__ __launch_bounds__(256) void copy_tile(const char* A, char* C, int M, int N, int tilesN) {
    const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
    const int chunk = threadIdx.x & 15, rr = threadIdx.x >> 4;
    f4 v[8];
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const long row = tm * 128L + pass * 16 + rr;
        v[pass] = *(const f4*)(A + (row * N + tn * 128 + chunk * 8) * 2);
    }
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
        const long row = tm * 128L + pass * 16 + rr;
        *(f4*)(C + (row * N + tn * 128 + chunk * 8) * 2) = v[pass];
    }
}
int main() {
    const int M = 3211264, N = 384;
    const long bytes = (long)M * N * 2;
    char *C, *A;
    hipMalloc(&C, bytes); hipMalloc(&A, bytes);
    hipMemset(A, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    const int tilesN = N / 128, tiles = (M / 128) * tilesN;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fill_contig, dim3(8192), dim3(256), 0, 0, (f4*)C, bytes / 16); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("fill_contig   %8.1f us  %7.0f GB/s\n", ms / 5 * 1e3, bytes / (ms / 5) / 1e6);
        hipEventRecord(e0); for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fill_tile, dim3(tiles), dim3(256), 0, 0, C, M, N, tilesN); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("fill_tile     %8.1f us  %7.0f GB/s\n", ms / 5 * 1e3, bytes / (ms / 5) / 1e6);
        hipEventRecord(e0); for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fill_rowpanel, dim3(M / 128), dim3(256), 0, 0, C, M, N); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("fill_rowpanel %8.1f us  %7.0f GB/s\n", ms / 5 * 1e3, bytes / (ms / 5) / 1e6);
        hipEventRecord(e0); for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(copy_tile, dim3(tiles), dim3(256), 0, 0, A, C, M, N, tilesN); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("copy_tile     %8.1f us  %7.0f GB/s (r+w)\n", ms / 5 * 1e3, 2.0 * bytes / (ms / 5) / 1e6);
    }
    return 0;
}
