// dev microbenchmark: the fp32 GEMM epilogue's store pattern (per wave instruction 16 rows x 64 B out of a 128 x 128 f32 tile, row stride N*4)
// against row-contiguous stores of the same tile (2 rows x 512 B per wave instruction) and a flat fill.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void fill_flat(f4* p, long n16) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n16; i += gridDim.x * 256L) p[i] = (f4){1, 2, 3, 4};
}
// fragment layout of gemm_nt_f32: wave (wm, wn) owns a 64 x 64 quadrant; lane (lr = lane & 15, lg = lane >> 4) stores 16 B at
// row wm*64 + mt*16 + lr, col wn*64 + nt*16 + lg*4 for mt, nt in 0..3
__global__ __launch_bounds__(256) void fill_frag(float* C, int N, int tilesN) {
    const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1, lr = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const long row = tm * 128L + wm * 64 + mt * 16 + lr;
            *(f4*)(C + row * N + tn * 128 + wn * 64 + nt * 16 + lg * 4) = (f4){1, 2, 3, 4};
        }
}
// same tile, row layout: thread t stores 16 B at row t/32 + 8*i, col (t%32)*4: a wave instruction = 2 rows x 512 B
__global__ __launch_bounds__(256) void fill_rows(float* C, int N, int tilesN) {
    const int tm = blockIdx.x / tilesN, tn = blockIdx.x % tilesN;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const long row = tm * 128L + (threadIdx.x >> 5) + 8 * i;
        *(f4*)(C + row * N + tn * 128 + (threadIdx.x & 31) * 4) = (f4){1, 2, 3, 4};
    }
}
int main() {
    const int M = 3211264, N = 384;
    const long bytes = (long)M * N * 4;
    float* C;
    if (hipMalloc(&C, bytes) != hipSuccess) return 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    const int tilesN = N / 128, tiles = (M / 128) * tilesN;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fill_flat, dim3(8192), dim3(256), 0, 0, (f4*)C, bytes / 16); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("fill_flat  %8.1f us  %7.0f GB/s\n", ms / 5 * 1e3, bytes / (ms / 5) / 1e6);
        hipEventRecord(e0); for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fill_frag, dim3(tiles), dim3(256), 0, 0, C, N, tilesN); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("fill_frag  %8.1f us  %7.0f GB/s  (16 rows x 64 B per wave instruction)\n", ms / 5 * 1e3, bytes / (ms / 5) / 1e6);
        hipEventRecord(e0); for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fill_rows, dim3(tiles), dim3(256), 0, 0, C, N, tilesN); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("fill_rows  %8.1f us  %7.0f GB/s  (2 rows x 512 B per wave instruction)\n", ms / 5 * 1e3, bytes / (ms / 5) / 1e6);
    }
    return 0;
}
