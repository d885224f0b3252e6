// dev microbenchmark: what does a VALU / LDS / store instruction of one wave cost while the other waves of its SIMD keep the MFMA pipe busy?
// 4 workgroups of 4 waves per CU (one wave per SIMD each).  Workgroups (blockIdx / 256) % 4 == 0 time NV independent VALU instructions (or LDS
// reads, or 16-byte stores); the other three run v_mfma_f32_16x16x4_f32 back to back for the whole time (nm = 0: they exit at once -> the baseline); prio = the timed
// waves' s_setprio level (the MFMA waves stay at 0).
// Explains the fp32 GEMM epilogue: its ~700 (plain) ... ~1500 (GELU) VALU instructions take 11 / 24 us when co-resident workgroups are in their k-loops.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* cyc, unsigned* hwid, unsigned long long* wall, int nm, int nv, int mode, int prio, int nmw, float a, float b) {
    __shared__ float lds[8192];
    const int role = (blockIdx.x >> 8) & 3;
    const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
    lds[threadIdx.x] = a; lds[threadIdx.x + 256] = b;
    __syncthreads();
    if (role != 0) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < (role <= nmw ? nm : 0); ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        float s = 0;
        for (int i = 0; i < 16; ++i) s += acc[i][0];
        out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
        if ((threadIdx.x & 63) == 0) {
            unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            hwid[blockIdx.x * 4 + (threadIdx.x >> 6)] = hw | (xcc << 28);
            wall[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = w0; wall[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
        }
        return;
    }
    // let the MFMA workgroups get going
    if (prio >= 0) __builtin_amdgcn_s_sleep(100);
    if (prio == 1) __builtin_amdgcn_s_setprio(1);
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    f32x4 w = {a, b, a, b};
    float* dst = out + (size_t)blockIdx.x * 256 * 64 + threadIdx.x * 4;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (mode == 0) {
        for (int it = 0; it < nv / 8; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], b, a);          // 8 independent chains
    } else if (mode == 1) {
        for (int it = 0; it < nv; ++it) { w += *reinterpret_cast<const f32x4*>(lds + ((threadIdx.x * 4 + it * 16) & 1023)); }
    } else if (mode == 2) {
        for (int it = 0; it < nv; ++it) { *reinterpret_cast<f32x4*>(dst + (it & 15) * 1024) = w; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (mode == 3) {          // LDS reads only: immediate offsets, scalar loop control, no VALU at all
        const unsigned addr = threadIdx.x * 16;
        for (int it = 0; it < nv / 8; ++it) {
            f32x4 r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:4096\n ds_read_b128 %2, %8 offset:8192\n ds_read_b128 %3, %8 offset:12288\n"
                         "ds_read_b128 %4, %8 offset:16\n ds_read_b128 %5, %8 offset:4112\n ds_read_b128 %6, %8 offset:8208\n ds_read_b128 %7, %8 offset:12304\n s_waitcnt lgkmcnt(0)"
                         : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr) : "memory");
        }
    } else {                         // stores only: one base address, immediate offsets
        for (int it = 0; it < nv / 4; ++it) {
            asm volatile("global_store_dwordx4 %0, %1, off\n global_store_dwordx4 %0, %1, off offset:1024\n global_store_dwordx4 %0, %1, off offset:2048\n global_store_dwordx4 %0, %1, off offset:3072"
                         :: "v"(dst), "v"(w) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = w[0] + w[1];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        hwid[blockIdx.x * 4 + (threadIdx.x >> 6)] = hw | (xcc << 28);
        wall[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = w0; wall[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = __builtin_amdgcn_s_memrealtime();
    }
}
int main() {
    const int wgs = 1024;
    float* out; hipMalloc(&out, (size_t)wgs * 256 * 64 * 4);
    unsigned long long* cyc; hipMalloc(&cyc, wgs * 4 * 8); hipMemset(cyc, 0, wgs * 4 * 8);
    unsigned* hw; hipMalloc(&hw, wgs * 4 * 4);
    unsigned long long* wall; hipMalloc(&wall, wgs * 4 * 16);
    std::vector<unsigned> hh(wgs * 4); std::vector<unsigned long long> hwall(wgs * 8);
    std::vector<unsigned long long> h(wgs * 4);
    const char* names[5] = {"v_fma_f32 (8 chains)", "ds_read_b128", "global_store_dwordx4", "ds_read_b128, no VALU", "global_store x4, no VALU"};
    for (int mode : {0, 3, 4})
        for (int nv : {2048, 32768})
            for (int nm : {0, 4000}) for (int prio : {0, 3}) for (int nmw : {1, 3}) {
                if (nm == 0 && (prio > 0 || nmw != 3)) continue;      // prio -1: no s_sleep before the timed section
                hipMemset(cyc, 0, wgs * 4 * 8);
                hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 0, 0, out, cyc, hw, wall, nm, nv, mode, prio, nmw, 1.0f, 0.5f);
                hipDeviceSynchronize();
                hipMemcpy(h.data(), cyc, wgs * 4 * 8, hipMemcpyDeviceToHost);
                if (mode == 0 && nv == 512 && nm && prio == 0) {       // one SIMD's timeline: who shared it, from when to when (10 ns ticks)
                    hipMemcpy(hh.data(), hw, wgs * 4 * 4, hipMemcpyDeviceToHost);
                    hipMemcpy(hwall.data(), wall, wgs * 4 * 16, hipMemcpyDeviceToHost);
                    unsigned long long base = ~0ull;
                    for (int i = 0; i < wgs * 4; ++i) base = std::min(base, hwall[2 * i]);
                    const unsigned key = hh[0] & 0xF000FF30u;      // xcc | se | cu | simd of wave 0
                    for (int i = 0; i < wgs * 4; ++i)
                        if ((hh[i] & 0xF000FF30u) == key)
                            printf("    same SIMD as wave 0: block %4d wave %d role %d  hw %08x  %llu .. %llu\n", i / 4, i % 4, (i / 4 >> 8) & 3, hh[i], hwall[2 * i] - base, hwall[2 * i + 1] - base);
                }
                std::vector<double> v;
                for (auto x : h) if (x) v.push_back((double)x);
                std::sort(v.begin(), v.end());
                printf("%-24s n=%5d prio %d, %d %s: median %.0f cycles = %.1f per instruction (p10 %.1f, p90 %.1f), %zu waves\n", names[mode], nv, prio, nm ? nmw : 0,
                       nm ? "co-resident waves per SIMD in MFMA" : "(alone)                           ", v[v.size() / 2], v[v.size() / 2] / nv,
                       v[v.size() / 10] / nv, v[v.size() * 9 / 10] / nv, v.size());
            }
    return 0;
}
