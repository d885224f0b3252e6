// Spatial (3x3) kernels of the reference-precision (fp32) mode: NHWC f32 activations, 16 bytes (4 channels) per lane.
// Same operations as conv.hip (timm ConvNorm convs of PatchEmbed / MBConv / PatchMerging / local_conv, models/tinyvit.py:135), f32
// storage and arithmetic; HBM-bound byte movers, so the structure that matters is the same: lanes run over channels first, every wave
// access is a contiguous run of an NHWC row, and the depthwise kernels walk down the image with the 3x3 input window in registers
// (each input element is fetched from HBM once; the left / right neighbours come out of L1).
#include "common.h"
#include "../../include/gg.h"
#include <string.h>

namespace {

__device__ __forceinline__ float act_exact(float x, int act) { return gg_act_f32(x, act); }
__device__ __forceinline__ float act_grad_exact(float x, int act) { return gg_act_grad_f32(x, act); }

// ---------------------------------------------------------------- im2col (dense 3x3, pad 1)
// x f32 NCHW (B,3,H,W) -> col f32 [B*Ho*Wo, 32]; k = (ky*3+kx)*3 + ci for k < 27, zeros above.  thread = (pixel, 16-byte eighth)
__global__ __launch_bounds__(256) void im2col_nchw3_f32_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int H, int W,
                                                               int Ho, int Wo, int stride) {
    const int64_t total = (int64_t)B * Ho * Wo * 8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned pu = (unsigned)(i >> 3);
        const int q = (int)(i & 7);
        const int ox = (int)(pu % (unsigned)Wo);
        const int oy = (int)((pu / (unsigned)Wo) % (unsigned)Ho);
        const int b = (int)(pu / ((unsigned)Wo * (unsigned)Ho));
        const float* xb = x + (int64_t)b * 3 * H * W;
        f32x4 t;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = q * 4 + j;
            const int tap = (k * 11) >> 5, ci = k - tap * 3;          // k / 3, k % 3 for k < 32
            const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
            const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
            const bool ok = k < 27 && iy >= 0 && iy < H && ix >= 0 && ix < W;
            t[j] = ok ? xb[((int64_t)ci * H + iy) * W + ix] : 0.f;
        }
        *reinterpret_cast<f32x4*>(col + (int64_t)pu * 32 + q * 4) = t;
    }
}

// x f32 NHWC (B,H,W,C) -> col f32 [B*Ho*Wo, 9*C]; k = (ky*3+kx)*C + c.  BN: x is a saved pre-BatchNorm conv output and
// act(gamma*(x-mean)*rstd+beta) is applied to every gathered chunk (padding taps stay zero); scale / shift from an LDS table (C <= 512)
template <bool BN>
__global__ __launch_bounds__(256) void im2col_nhwc_f32_kernel(const float* __restrict__ x, const float* __restrict__ stat,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                              float* __restrict__ col, int B, int H, int W, int C, int Ho, int Wo, int stride) {
    __shared__ __attribute__((aligned(16))) float tab[BN ? 2 * 512 : 4];
    if (BN) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            const float sc = stat[C + c] * gamma[c];
            tab[c] = sc;
            tab[512 + c] = beta[c] - stat[c] * sc;
        }
        __syncthreads();
    }
    const int cg = C >> 2;
    const int per_pix = 9 * cg;
    const int64_t total = (int64_t)B * Ho * Wo * per_pix;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % per_pix);
        const int64_t p = i / per_pix;
        const int tap = ch / cg, g = ch % cg;
        const int ky = tap / 3, kx = tap % 3;
        const unsigned pu = (unsigned)p;
        const int ox = (int)(pu % (unsigned)Wo);
        const int oy = (int)((pu / (unsigned)Wo) % (unsigned)Ho);
        const int b = (int)(pu / ((unsigned)Wo * (unsigned)Ho));
        const int iy = oy * stride + ky - 1, ix = ox * stride + kx - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            v = *reinterpret_cast<const f32x4*>(x + (((int64_t)b * H + iy) * W + ix) * C + g * 4);
            if (BN) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(tab + g * 4), sh = *reinterpret_cast<const f32x4*>(tab + 512 + g * 4);
                v = gg_act_f32_v4(v * sc + sh, act);
            }
        }
        *reinterpret_cast<f32x4*>(col + p * (9 * C) + tap * C + g * 4) = v;
    }
}

// transpose of im2col_nhwc: dcol f32 [B*Ho*Wo, 9*C] -> dx f32 NHWC (gather form, no atomics)
__global__ __launch_bounds__(256) void col2im_nhwc_f32_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B, int H, int W,
                                                              int C, int Ho, int Wo, int stride) {
    const int cg = C >> 2;
    const int64_t total = (int64_t)B * H * W * cg;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const int64_t p = i / cg;
        const unsigned pu = (unsigned)p;
        const int ix = (int)(pu % (unsigned)W);
        const int iy = (int)((pu / (unsigned)W) % (unsigned)H);
        const int b = (int)(pu / ((unsigned)W * (unsigned)H));
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + 1 - ky;
            if (ty < 0 || (ty % stride) != 0) continue;
            const int oy = ty / stride;
            if (oy >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + 1 - kx;
                if (tx < 0 || (tx % stride) != 0) continue;
                const int ox = tx / stride;
                if (ox >= Wo) continue;
                acc += *reinterpret_cast<const f32x4*>(dcol + (((int64_t)b * Ho + oy) * Wo + ox) * (9 * C) + (ky * 3 + kx) * C + g * 4);
            }
        }
        *reinterpret_cast<f32x4*>(dx + p * C + g * 4) = acc;
    }
}

// col2im (stride-2 3x3) + the BatchNorm-backward reduce of the ConvNorm whose output was gathered: da is formed per pixel, multiplied by
// act'(BN(y)) and written as dz; per-channel (sum dz, sum dz*xhat) leave as one partial row per block.  da is never written and the
// stand-alone reduce pass never runs (f32 twin of col2im_nhwc_bnbwd_kernel; PatchEmbed conv1 <- conv2).  Thread = (4 channels, pixel lane).
__global__ __launch_bounds__(256) void col2im_nhwc_bnbwd_f32_kernel(const float* __restrict__ dcol, const float* __restrict__ y,
                                                                    const float* __restrict__ stat, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, int act, float* __restrict__ dz,
                                                                    float* __restrict__ part, int B, int H, int W, int C, int Ho, int Wo, int CG, int PP) {
    extern __shared__ float c2i_red[];          // [PP][2][C]
    const int g = threadIdx.x % CG, pp = threadIdx.x / CG;
    const int c0 = g * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = s;
    if (pp < PP) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(stat + c0), rstd = *reinterpret_cast<const f32x4*>(stat + C + c0);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0), be = *reinterpret_cast<const f32x4*>(beta + c0);
        const int64_t npix = (int64_t)B * H * W;
        const unsigned HW = (unsigned)H * (unsigned)W;
        for (int64_t p = (int64_t)blockIdx.x * PP + pp; p < npix; p += (int64_t)gridDim.x * PP) {
            const unsigned pu = (unsigned)p;
            const unsigned b = pu / HW, rem = pu - b * HW;
            const int iy = (int)(rem / (unsigned)W), ix = (int)(rem - (unsigned)iy * (unsigned)W);
            const f32x4 v = *reinterpret_cast<const f32x4*>(y + p * C + c0);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            // oy = (iy + 1 - ky) / 2 with iy + 1 - ky even -> ky = 1 for even rows, ky in {0, 2} for odd rows; same in x.  Tap order (ky, kx)
            // ascending, as col2im_nhwc_f32_kernel adds them: the sum is bit-identical to the unfused route's da
            const int ky0 = (iy & 1) ? 0 : 1, nky = (iy & 1) ? 2 : 1;
            const int kx0 = (ix & 1) ? 0 : 1, nkx = (ix & 1) ? 2 : 1;
            for (int a = 0; a < nky; ++a) {
                const int ky = ky0 + 2 * a, oy = (iy + 1 - ky) >> 1;
                if (oy < 0 || oy >= Ho) continue;
                for (int c = 0; c < nkx; ++c) {
                    const int kx = kx0 + 2 * c, ox = (ix + 1 - kx) >> 1;
                    if (ox < 0 || ox >= Wo) continue;
                    acc += *reinterpret_cast<const f32x4*>(dcol + (((int64_t)b * Ho + oy) * Wo + ox) * (9 * C) + (ky * 3 + kx) * C + c0);
                }
            }
            const f32x4 xh = (v - mu) * rstd;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = acc[j] * gg_act_grad_f32(fmaf(ga[j], xh[j], be[j]), act);
            s += o;
            q += o * xh;
            *reinterpret_cast<f32x4*>(dz + p * C + c0) = o;
        }
    }
    if (pp < PP) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { c2i_red[(pp * 2 + 0) * C + c0 + j] = s[j]; c2i_red[(pp * 2 + 1) * C + c0 + j] = q[j]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        float t = 0.f;
        for (int k = 0; k < PP; ++k) t += c2i_red[k * 2 * C + i];
        part[(int64_t)blockIdx.x * 2 * C + i] = t;
    }
}

// ---------------------------------------------------------------- column-walking depthwise 3x3
// Thread = (4 channels, one output column); a block covers PX adjacent output columns x all C channels of one image and walks down a
// strip of output rows.  MODE 0: y = conv(x, taps) (+ per-block partial BatchNorm statistics of y); MODE 1: the same with flipped taps
// (stride-1 data gradient); MODE 2: weight gradient -- acc[tap] += window[tap] * dy, one partial row [9][C] per block.
// Fusions (the BatchNorm passes either side of a frozen depthwise ConvNorm ride on its loads / stores; bn_apply / bn_bwd passes and the
// tensors between them disappear -- the f32 twins of gg_dwconv3x3_fwd_fused / gg_dwconv3x3_bwd_data_fused):
//   IN 1: x is the producer ConvNorm's saved PRE-BatchNorm output; act(sc*x + sh) is applied to every in-image load (padding stays 0)
//   IN 2: x = dz, x2 = y of the ConvNorm whose BatchNorm-backward apply step dy = c0*dz + c1*y + c2 is formed on load
//   EPI : (MODE 1) o *= act'(BN(ep_y)) at the output position and the partial rows hold (sum dz, sum dz*xhat) for gg_bn_bwd_finalize
enum { DWM_FWD = 0, DWM_FLIP = 1, DWM_WGRAD = 2 };
struct DwFuse {
    const float* x2;                         // IN 2: second input tensor
    const float* in_a; const float* in_b; const float* in_c;   // IN 1: stat [2][C], gamma, beta;  IN 2: coef [3][C] in in_a
    int in_act;
    const float* ep_y; const float* ep_stat; const float* ep_gamma; const float* ep_beta; int ep_act;
};
template <int S, int MODE, int IN = 0, bool EPI = false>
__global__ __launch_bounds__(256) void dw3x3_walk_f32_kernel(const float* __restrict__ x, const float* __restrict__ taps,
                                                             float* __restrict__ y, const float* __restrict__ dy, int H, int W, int C,
                                                             int Ho, int Wo, int PX, int rows_per_strip, float* __restrict__ part, DwFuse fz) {
    extern __shared__ float sred[];
    const int CG = C >> 2;
    const int cg = threadIdx.x % CG, px = threadIdx.x / CG;
    const int ox = blockIdx.x * PX + px;
    const int b = blockIdx.y;
    const int oy0 = blockIdx.z * rows_per_strip, oy1 = min(Ho, oy0 + rows_per_strip);
    const bool live = px < PX && ox < Wo;
    const float* xb = x + (int64_t)b * H * W * C + cg * 4;
    f32x4 w[9];
    if (MODE != DWM_WGRAD) {
#pragma unroll
        for (int k = 0; k < 9; ++k) w[k] = *reinterpret_cast<const f32x4*>(taps + (MODE == DWM_FLIP ? 8 - k : k) * C + cg * 4);
    }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[9];                    // MODE 2: the 9 tap gradients;  MODE 0: acc[0], acc[1] = sum / sum of squares of y
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = zero;
    const int ix0 = ox * S - 1;
    const bool c0 = ix0 >= 0, c2 = ix0 + 2 < W;           // ix0 + 1 < W always holds for a live column
    // per-channel coefficients of the fused passes (this thread's 4 channels are fixed)
    f32x4 ia = zero, ib = zero, ic = zero;
    if (IN == 1) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(fz.in_a + cg * 4), rs = *reinterpret_cast<const f32x4*>(fz.in_a + C + cg * 4);
        ia = rs * *reinterpret_cast<const f32x4*>(fz.in_b + cg * 4);
        ib = *reinterpret_cast<const f32x4*>(fz.in_c + cg * 4) - mu * ia;
    }
    if (IN == 2) {
        ia = *reinterpret_cast<const f32x4*>(fz.in_a + cg * 4); ib = *reinterpret_cast<const f32x4*>(fz.in_a + C + cg * 4);
        ic = *reinterpret_cast<const f32x4*>(fz.in_a + 2 * C + cg * 4);
    }
    f32x4 esc = zero, esh = zero, ers = zero, enm = zero;
    if (EPI) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(fz.ep_stat + cg * 4);
        ers = *reinterpret_cast<const f32x4*>(fz.ep_stat + C + cg * 4);
        esc = ers * *reinterpret_cast<const f32x4*>(fz.ep_gamma + cg * 4);
        esh = *reinterpret_cast<const f32x4*>(fz.ep_beta + cg * 4) - mu * esc;
        enm = -mu * ers;
    }
    const float* x2b = IN == 2 ? fz.x2 + (int64_t)b * H * W * C + cg * 4 : nullptr;
    auto ld = [&](const float* p, const float* p2) {
        f32x4 v = *reinterpret_cast<const f32x4*>(p);
        if (IN == 1) {
            v = gg_act_f32_v4(v * ia + ib, fz.in_act);
        }
        if (IN == 2) v = ia * v + (ib * *reinterpret_cast<const f32x4*>(p2) + ic);
        return v;
    };
    auto row = [&](int iy, f32x4 (&r)[3]) {
        if (live && iy >= 0 && iy < H) {
            const int64_t o = ((int64_t)iy * W + ix0) * C;
            const float* p = xb + o;
            const float* p2 = IN == 2 ? x2b + o : nullptr;
            r[0] = c0 ? ld(p, p2) : zero;
            r[1] = ld(p + C, IN == 2 ? p2 + C : nullptr);
            r[2] = c2 ? ld(p + 2 * C, IN == 2 ? p2 + 2 * C : nullptr) : zero;
        } else { r[0] = r[1] = r[2] = zero; }
    };
    f32x4 r0[3], r1[3], r2[3];
    if (S == 1) { row(oy0 - 1, r0); row(oy0, r1); }
    else row(2 * oy0 - 1, r0);
    for (int oy = oy0; oy < oy1; ++oy) {
        if (S == 1) row(oy + 1, r2);
        else { row(2 * oy, r1); row(2 * oy + 1, r2); }
        if (MODE == DWM_WGRAD) {
            f32x4 g = zero;
            if (live) g = *reinterpret_cast<const f32x4*>(dy + (((int64_t)b * Ho + oy) * Wo + ox) * C + cg * 4);
#pragma unroll
            for (int k = 0; k < 3; ++k) { acc[k] += r0[k] * g; acc[3 + k] += r1[k] * g; acc[6 + k] += r2[k] * g; }
        } else {
            f32x4 o = r0[0] * w[0];
            o += r0[1] * w[1]; o += r0[2] * w[2];
            o += r1[0] * w[3]; o += r1[1] * w[4]; o += r1[2] * w[5];
            o += r2[0] * w[6]; o += r2[1] * w[7]; o += r2[2] * w[8];
            if (live) {
                const int64_t oo = (((int64_t)b * Ho + oy) * Wo + ox) * C + cg * 4;
                if (EPI) {      // dz = da * act'(gamma*xhat + beta); sums of dz and dz*xhat
                    const f32x4 yv = *reinterpret_cast<const f32x4*>(fz.ep_y + oo);
                    o *= gg_act_grad_f32_v4(yv * esc + esh, fz.ep_act);
                    acc[0] += o; acc[1] += o * (yv * ers + enm);
                } else if (part) { acc[0] += o; acc[1] += o * o; }
                *reinterpret_cast<f32x4*>(y + oo) = o;
            }
        }
        if (S == 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { r0[k] = r1[k]; r1[k] = r2[k]; }
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) r0[k] = r2[k];
        }
    }
    if (!part) return;
    // fold the block's PX columns: sred[px][NR][C] -> part[block][NR][C]
    constexpr int NR = MODE == DWM_WGRAD ? 9 : 2;
    const int nslots = blockDim.x / CG;
#pragma unroll
    for (int k = 0; k < NR; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) sred[(px * NR + k) * C + cg * 4 + j] = acc[k][j];
    __syncthreads();
    const int64_t blk = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    for (int i = threadIdx.x; i < NR * C; i += blockDim.x) {
        float t = 0.f;
        for (int k = 0; k < nslots; ++k) t += sred[k * NR * C + i];
        part[blk * NR * C + i] = t;
    }
}

// Stride 1, FOUR adjacent output columns x 4 channels per thread: 6 loads per 4 results instead of 3 per 1, and whatever rides on the load
// (BatchNorm + GELU of the ConvNorm in front: IN 1; BatchNorm backward's apply step c0*dz + c1*y + c2: IN 2) is evaluated 1.5 instead of 3
// times per input element -- with the exact erf of the fp32 mode that is what the one-column kernel spent its time on (2.1 TB/s against
// 4.2 TB/s unfused).  Row window held in registers, rotated by renaming (the loop body is unrolled 3x).  MODE: DWM_FWD or DWM_FLIP.
template <int MODE, int IN, bool EPI>
__global__ __launch_bounds__(256) void dw3x3_s1_multi_f32_kernel(const float* __restrict__ x, const float* __restrict__ taps, float* __restrict__ y,
                                                                 int H, int W, int C, int PXG, int rows_per_strip, float* __restrict__ part, DwFuse fz) {
    extern __shared__ float sred[];
    constexpr int OX = 4;
    const int CG = C >> 2;
    const int cg = threadIdx.x % CG, pg = threadIdx.x / CG;
    const int ox0 = (blockIdx.x * PXG + pg) * OX;
    const int b = blockIdx.y;
    const int oy0 = blockIdx.z * rows_per_strip, oy1 = min(H, oy0 + rows_per_strip);
    const bool live = pg < PXG && ox0 < W;
    const float* xb = x + (int64_t)b * H * W * C + cg * 4;
    const float* x2b = IN == 2 ? fz.x2 + (int64_t)b * H * W * C + cg * 4 : nullptr;
    f32x4 w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = *reinterpret_cast<const f32x4*>(taps + (MODE == DWM_FLIP ? 8 - k : k) * C + cg * 4);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 ia = zero, ib = zero, ic = zero;
    if (IN == 1) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(fz.in_a + cg * 4), rs = *reinterpret_cast<const f32x4*>(fz.in_a + C + cg * 4);
        ia = rs * *reinterpret_cast<const f32x4*>(fz.in_b + cg * 4);
        ib = *reinterpret_cast<const f32x4*>(fz.in_c + cg * 4) - mu * ia;
    }
    if (IN == 2) {
        ia = *reinterpret_cast<const f32x4*>(fz.in_a + cg * 4); ib = *reinterpret_cast<const f32x4*>(fz.in_a + C + cg * 4);
        ic = *reinterpret_cast<const f32x4*>(fz.in_a + 2 * C + cg * 4);
    }
    f32x4 esc = zero, esh = zero, ers = zero, enm = zero;
    if (EPI) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(fz.ep_stat + cg * 4);
        ers = *reinterpret_cast<const f32x4*>(fz.ep_stat + C + cg * 4);
        esc = ers * *reinterpret_cast<const f32x4*>(fz.ep_gamma + cg * 4);
        esh = *reinterpret_cast<const f32x4*>(fz.ep_beta + cg * 4) - mu * esc;
        enm = -mu * ers;
    }
    bool cok[OX + 2];                                      // input columns ox0-1 .. ox0+4 inside the image
#pragma unroll
    for (int j = 0; j < OX + 2; ++j) cok[j] = live && (ox0 - 1 + j) >= 0 && (ox0 - 1 + j) < W;
    auto row = [&](int iy, f32x4 (&r)[OX + 2]) {
        const bool rok = iy >= 0 && iy < H;
        const int64_t o = ((int64_t)(rok ? iy : 0) * W + (ox0 - 1)) * C;
        f32x4 v[OX + 2], v2[IN == 2 ? OX + 2 : 1];
#pragma unroll
        for (int j = 0; j < OX + 2; ++j) {                 // every load of the row is in flight before the first transform
            v[j] = (rok && cok[j]) ? *reinterpret_cast<const f32x4*>(xb + o + (int64_t)j * C) : zero;
            if (IN == 2) v2[j] = (rok && cok[j]) ? *reinterpret_cast<const f32x4*>(x2b + o + (int64_t)j * C) : zero;
        }
#pragma unroll
        for (int j = 0; j < OX + 2; ++j) {
            f32x4 t = v[j];
            if (IN == 1) {
                t = gg_act_f32_v4(t * ia + ib, fz.in_act);
            }
            if (IN == 2) t = ia * t + (ib * v2[IN == 2 ? j : 0] + ic);
            r[j] = (rok && cok[j]) ? t : zero;             // padding is zero AFTER the transform
        }
    };
    f32x4 s0 = zero, s1 = zero;                            // FWD: sum y, sum y^2;  EPI: sum dz, sum dz*xhat
    auto emit = [&](int oy, const f32x4 (&ra)[OX + 2], const f32x4 (&rb)[OX + 2], const f32x4 (&rc)[OX + 2]) {
        if (!live) return;
        const int64_t obase = (((int64_t)b * H + oy) * W + ox0) * C + cg * 4;
        f32x4 ey[EPI ? OX : 1];
        if (EPI) {
#pragma unroll
            for (int j = 0; j < OX; ++j) ey[j] = (ox0 + j < W) ? *reinterpret_cast<const f32x4*>(fz.ep_y + obase + (int64_t)j * C) : zero;
        }
#pragma unroll
        for (int j = 0; j < OX; ++j) {
            f32x4 o = ra[j] * w[0];
            o += ra[j + 1] * w[1]; o += ra[j + 2] * w[2];
            o += rb[j] * w[3]; o += rb[j + 1] * w[4]; o += rb[j + 2] * w[5];
            o += rc[j] * w[6]; o += rc[j + 1] * w[7]; o += rc[j + 2] * w[8];
            if (ox0 + j < W) {
                if (EPI) {
                    const f32x4 yv = ey[EPI ? j : 0];
                    o *= gg_act_grad_f32_v4(yv * esc + esh, fz.ep_act);
                    s0 += o; s1 += o * (yv * ers + enm);
                } else if (part) { s0 += o; s1 += o * o; }
                *reinterpret_cast<f32x4*>(y + obase + (int64_t)j * C) = o;
            }
        }
    };
    f32x4 r0[OX + 2], r1[OX + 2], r2[OX + 2];
    row(oy0 - 1, r0); row(oy0, r1);
    int oy = oy0;
    while (oy < oy1) {
        row(oy + 1, r2); emit(oy, r0, r1, r2); if (++oy >= oy1) break;
        row(oy + 1, r0); emit(oy, r1, r2, r0); if (++oy >= oy1) break;
        row(oy + 1, r1); emit(oy, r2, r0, r1); ++oy;
    }
    if (!part) return;
    const int nslots = blockDim.x / CG;
#pragma unroll
    for (int q = 0; q < 4; ++q) { sred[(pg * 2 + 0) * C + cg * 4 + q] = s0[q]; sred[(pg * 2 + 1) * C + cg * 4 + q] = s1[q]; }
    __syncthreads();
    const int64_t blk = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        float t = 0.f;
        for (int k = 0; k < nslots; ++k) t += sred[k * 2 * C + i];
        part[blk * 2 * C + i] = t;
    }
}

// stride-2 data gradient, gather form: dx[iy][ix] = sum over the taps (ky,kx) with (iy+1-ky, ix+1-kx) both even of
// w[ky][kx] * dy[(iy+1-ky)/2][(ix+1-kx)/2]   (at most 4 taps per input pixel)
__global__ __launch_bounds__(256) void dw3x3_s2_bwd_data_f32_kernel(const float* __restrict__ dy, const float* __restrict__ taps,
                                                                    float* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo) {
    const int cg = C >> 2;
    const int64_t total = (int64_t)B * H * W * cg;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const int64_t p = i / cg;
        const unsigned pu = (unsigned)p;
        const int ix = (int)(pu % (unsigned)W);
        const int iy = (int)((pu / (unsigned)W) % (unsigned)H);
        const int b = (int)(pu / ((unsigned)W * (unsigned)H));
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + 1 - ky;
            if (ty < 0 || (ty & 1)) continue;
            const int oy = ty >> 1;
            if (oy >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + 1 - kx;
                if (tx < 0 || (tx & 1)) continue;
                const int ox = tx >> 1;
                if (ox >= Wo) continue;
                const f32x4 wv = *reinterpret_cast<const f32x4*>(taps + (ky * 3 + kx) * C + g * 4);
                acc += wv * *reinterpret_cast<const f32x4*>(dy + (((int64_t)b * Ho + oy) * Wo + ox) * C + g * 4);
            }
        }
        *reinterpret_cast<f32x4*>(dx + p * C + g * 4) = acc;
    }
}

// The same gather with the BatchNorm-backward passes of the ConvNorms on both sides riding on it (PatchMerging backward, frozen chain):
//   IN: dy = c0*dz + c1*y2 + c2 formed at every tap from the GEMM-epilogue dz2 and the saved conv2 output (coef [3][C]): no apply pass, no dy tensor;
//   EP: out = dz1 = da1 * act'(BN1(y1)) and one partial row [2][C] per block of (sum dz1, sum dz1 * xhat1): no reduce pass, no da1 tensor.
// A thread keeps ONE group of 4 channels for all its pixels (threads = C/4 channel groups x PX pixels), so the column sums stay in registers.
template <bool IN, bool EP>
__global__ __launch_bounds__(256) void dw3x3_s2_bwd_data_fused_f32_kernel(const float* __restrict__ dz, const float* __restrict__ y2, const float* __restrict__ coef,
                                                                          const float* __restrict__ taps, float* __restrict__ out, int B, int H, int W, int C, int Ho,
                                                                          int Wo, const float* __restrict__ ep_y, const float* __restrict__ ep_stat,
                                                                          const float* __restrict__ ep_gamma, const float* __restrict__ ep_beta, int ep_act,
                                                                          float* __restrict__ part, int64_t quads_per_block) {
    // Work item = one 2 x 2 QUAD of input pixels rooted at (2a, 2b): its four pixels read the same 2 x 2 neighbourhood dy[a..a+1][b..b+1] and split the nine taps
    // 1 + 2 + 2 + 4 between them (stride 2, pad 1: an even row sees only tap row 1 of output row a, an odd row tap row 2 of a and tap row 0 of a + 1; columns likewise),
    // so the 4 (x2 with IN) gradient loads and the 4 ep_y loads of a quad are all in flight before the first use -- the per-pixel gather issued <= 4 dependent,
    // branch-guarded loads per pixel and ran at 2.5 TB/s.
    __shared__ float red[256 * 8];
    const int CG = C >> 2, PX = blockDim.x / CG;
    const int g = threadIdx.x % CG, px = threadIdx.x / CG, c4 = g * 4;
    f32x4 tw[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) tw[k] = *reinterpret_cast<const f32x4*>(taps + k * C + c4);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 c0 = {1.f, 1.f, 1.f, 1.f}, c1 = zero, c2 = zero;
    if (IN) { c0 = *reinterpret_cast<const f32x4*>(coef + c4); c1 = *reinterpret_cast<const f32x4*>(coef + C + c4); c2 = *reinterpret_cast<const f32x4*>(coef + 2 * C + c4); }
    f32x4 sc = zero, sh = zero, rs = zero, nm = zero;
    if (EP) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float mu = ep_stat[c4 + r], rstd = ep_stat[C + c4 + r];
            sc[r] = ep_gamma[c4 + r] * rstd; sh[r] = ep_beta[c4 + r] - mu * sc[r]; rs[r] = rstd; nm[r] = -mu * rstd;
        }
    }
    f32x4 cs = zero, cq = zero;
    const int QH = (H + 1) >> 1, QW = (W + 1) >> 1;                 // quads per image column / row (== Ho, Wo)
    const int64_t total = (int64_t)B * QH * QW;
    const int64_t q0 = blockIdx.x * quads_per_block, q1 = q0 + quads_per_block < total ? q0 + quads_per_block : total;
    for (int64_t q = q0 + px; q < q1; q += PX) {
        const unsigned qu = (unsigned)q;
        const int qb = (int)(qu % (unsigned)QW), qa = (int)((qu / (unsigned)QW) % (unsigned)QH), b = (int)(qu / ((unsigned)QW * (unsigned)QH));
        const bool a1 = qa + 1 < Ho, b1 = qb + 1 < Wo;              // (qa < Ho and qb < Wo always: QH == Ho, QW == Wo)
        const int64_t o00 = (((int64_t)b * Ho + qa) * Wo + qb) * C + c4;
        const int64_t o01 = o00 + C, o10 = o00 + (int64_t)Wo * C, o11 = o10 + C;
        f32x4 d00 = *reinterpret_cast<const f32x4*>(dz + o00);
        f32x4 d01 = b1 ? *reinterpret_cast<const f32x4*>(dz + o01) : zero;
        f32x4 d10 = a1 ? *reinterpret_cast<const f32x4*>(dz + o10) : zero;
        f32x4 d11 = (a1 && b1) ? *reinterpret_cast<const f32x4*>(dz + o11) : zero;
        if (IN) {      // dy = c0 dz + c1 y2 + c2 at the positions that exist (beyond the map the gradient is zero, not c2)
            const f32x4 y00 = *reinterpret_cast<const f32x4*>(y2 + o00);
            const f32x4 y01 = b1 ? *reinterpret_cast<const f32x4*>(y2 + o01) : zero;
            const f32x4 y10 = a1 ? *reinterpret_cast<const f32x4*>(y2 + o10) : zero;
            const f32x4 y11 = (a1 && b1) ? *reinterpret_cast<const f32x4*>(y2 + o11) : zero;
            d00 = c0 * d00 + (c1 * y00 + c2);
            d01 = b1 ? c0 * d01 + (c1 * y01 + c2) : zero;
            d10 = a1 ? c0 * d10 + (c1 * y10 + c2) : zero;
            d11 = (a1 && b1) ? c0 * d11 + (c1 * y11 + c2) : zero;
        }
        const int iy = 2 * qa, ix = 2 * qb;
        const bool r1 = iy + 1 < H, k1 = ix + 1 < W;                // the quad's odd row / column exist (odd maps: the last quad is cut)
        const int64_t p00 = (((int64_t)b * H + iy) * W + ix) * C + c4;
        const int64_t p01 = p00 + C, p10 = p00 + (int64_t)W * C, p11 = p10 + C;
        f32x4 e00 = zero, e01 = zero, e10 = zero, e11 = zero;
        if (EP) {
            e00 = *reinterpret_cast<const f32x4*>(ep_y + p00);
            if (k1) e01 = *reinterpret_cast<const f32x4*>(ep_y + p01);
            if (r1) e10 = *reinterpret_cast<const f32x4*>(ep_y + p10);
            if (r1 && k1) e11 = *reinterpret_cast<const f32x4*>(ep_y + p11);
        }
        // taps [ky*3 + kx]:  x[iy][ix] collects w[ky][kx] dy[oy][ox] with iy = 2 oy + ky - 1, ix = 2 ox + kx - 1
        f32x4 v00 = tw[4] * d00;
        f32x4 v01 = tw[5] * d00 + tw[3] * d01;
        f32x4 v10 = tw[7] * d00 + tw[1] * d10;
        f32x4 v11 = tw[8] * d00 + tw[6] * d01 + (tw[2] * d10 + tw[0] * d11);
        if (EP) {
            auto fin = [&](f32x4& v, const f32x4& yv, bool on) {
                if (!on) return;
                const f32x4 xh = yv * rs + nm, z = yv * sc + sh;
                v *= gg_act_grad_f32_v4(z, ep_act);
                cs += v; cq += v * xh;
            };
            fin(v00, e00, true); fin(v01, e01, k1); fin(v10, e10, r1); fin(v11, e11, r1 && k1);
        }
        *reinterpret_cast<f32x4*>(out + p00) = v00;
        if (k1) *reinterpret_cast<f32x4*>(out + p01) = v01;
        if (r1) *reinterpret_cast<f32x4*>(out + p10) = v10;
        if (r1 && k1) *reinterpret_cast<f32x4*>(out + p11) = v11;
    }
    if (EP) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { red[threadIdx.x * 8 + r] = cs[r]; red[threadIdx.x * 8 + 4 + r] = cq[r]; }
        __syncthreads();
        if (px == 0) {
            for (int j = 1; j < PX; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { cs[r] += red[(j * CG + g) * 8 + r]; cq[r] += red[(j * CG + g) * 8 + 4 + r]; }
            }
            float* row = part + (int64_t)blockIdx.x * 2 * C;
            *reinterpret_cast<f32x4*>(row + c4) = cs;
            *reinterpret_cast<f32x4*>(row + C + c4) = cq;
        }
    }
}

// part [nparts][9][C] -> grad (C,1,3,3) (+)=
__global__ void dw_wgrad_final_f32_kernel(const float* __restrict__ part, int nparts, int C, float* __restrict__ grad, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // over 9*C, part layout [k][c]
    if (i >= 9 * C) return;
    double s = 0.0;
    for (int p = 0; p < nparts; ++p) s += (double)part[(int64_t)p * 9 * C + i];
    const int k = i / C, c = i % C;
    grad[c * 9 + k] = accumulate ? grad[c * 9 + k] + (float)s : (float)s;
}

struct WalkGeom { int PX, threads, nbx, strips, rows_per_strip; };
WalkGeom walk_geom(int B, int Ho, int Wo, int C) {
    WalkGeom g;
    const int CG = C / 4;
    g.PX = std::max(1, std::min(256 / CG, Wo));
    g.threads = CG * g.PX;
    g.nbx = (int)gg_cdiv(Wo, g.PX);
    // enough blocks to fill the chip (>= ~2048), strips of >= 4 rows so the 2-row window warm-up stays cheap
    int strips = (int)gg_cdiv(2048, (int64_t)g.nbx * B);
    strips = std::max(1, std::min(strips, (int)gg_cdiv(Ho, 4)));
    g.rows_per_strip = (int)gg_cdiv(Ho, strips);
    g.strips = (int)gg_cdiv(Ho, g.rows_per_strip);
    return g;
}
// stride-1 multi-column kernel: PX = groups of 4 output columns per block
WalkGeom multi_geom(int B, int H, int W, int C) {
    WalkGeom g;
    const int CG = C / 4, groups = (int)gg_cdiv(W, 4);
    g.PX = std::max(1, std::min(256 / CG, groups));
    g.threads = CG * g.PX;
    g.nbx = (int)gg_cdiv(groups, g.PX);
    int strips = (int)gg_cdiv(1536, (int64_t)g.nbx * B);
    strips = std::max(1, std::min(strips, (int)gg_cdiv(H, 4)));
    g.rows_per_strip = (int)gg_cdiv(H, strips);
    g.strips = (int)gg_cdiv(H, g.rows_per_strip);
    return g;
}
bool dw_use_multi(int stride) {
    static const bool off = gg_dev_env("GG_DW_F32_NO_MULTI") != nullptr;
    return stride == 1 && !off;
}
int grid_for(int64_t n, int cap = 16384) { return (int)std::max<int64_t>(1, std::min<int64_t>(gg_cdiv(n, 256), cap)); }

}  // namespace

// ------------------------------------------------------------------------------------------- host
extern "C" int gg_im2col_nchw3_f32_f32(const float* x, float* col, int B, int H, int W, int stride, void* stream) {
    GG_CHECK(x && col && B > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2), "gg_im2col_nchw3_f32_f32: bad args");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    GG_PROF(GG_CAT_MOVE, 0, 4.0 * B * 3.0 * H * W + 128.0 * B * Ho * Wo, stream);
    hipLaunchKernelGGL(im2col_nchw3_f32_kernel, dim3(grid_for((int64_t)B * Ho * Wo * 8, 65536)), dim3(256), 0, (hipStream_t)stream, x, col, B, H, W,
                       Ho, Wo, stride);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_im2col_nhwc_f32(const float* x, const float* stat, const float* gamma, const float* beta, int act, float* col, int B, int H,
                                  int W, int C, int stride, void* stream) {
    GG_CHECK(x && col && B > 0 && (C & 3) == 0 && (stride == 1 || stride == 2), "gg_im2col_nhwc_f32: bad args (C %% 4)");
    GG_CHECK(!stat || (gamma && beta && C <= 512), "gg_im2col_nhwc_f32: the BatchNorm form needs gamma, beta and C <= 512");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    GG_PROF(GG_CAT_MOVE, 0, 4.0 * B * (double)H * W * C + 36.0 * B * Ho * Wo * C, stream);
    const dim3 grid(grid_for((int64_t)B * Ho * Wo * 9 * (C / 4), 65536));
    if (stat) hipLaunchKernelGGL(im2col_nhwc_f32_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, stat, gamma, beta, act, col, B, H, W, C, Ho, Wo, stride);
    else hipLaunchKernelGGL(im2col_nhwc_f32_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, stat, gamma, beta, act, col, B, H, W, C, Ho, Wo, stride);
    GG_LAUNCH_CHECK();
    return 0;
}
// stride-2 3x3 only (PatchEmbed).  part: [nparts][2][C] with nparts rows as given (<= 65535); follow with gg_bn_bwd_finalize(part, nparts, ...)
extern "C" int gg_col2im_nhwc_bnbwd_f32(const float* dcol, const float* y, const float* stat, const float* gamma, const float* beta, int act,
                                        float* dz, float* part, int nparts, int B, int H, int W, int C, void* stream) {
    GG_CHECK(dcol && y && stat && gamma && beta && dz && part && nparts > 0 && nparts <= 65535 && B > 0 && (C & 3) == 0 && C / 4 <= 256,
             "gg_col2im_nhwc_bnbwd_f32: bad args");
    GG_CHECK(((uintptr_t)dcol & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)dz & 15) == 0 && ((uintptr_t)stat & 15) == 0 &&
             ((uintptr_t)gamma & 15) == 0 && ((uintptr_t)beta & 15) == 0, "gg_col2im_nhwc_bnbwd_f32: operands must be 16-byte aligned");
    GG_CHECK((int64_t)B * H * W < ((int64_t)1 << 32), "gg_col2im_nhwc_bnbwd_f32: tensor too large for 32-bit pixel indexing");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int CG = C / 4, PP = std::max(1, 256 / CG);
    const size_t lds = (size_t)PP * 2 * C * sizeof(float);
    GG_CHECK(lds <= 64 * 1024, "gg_col2im_nhwc_bnbwd_f32: C too large");
    GG_PROF(GG_CAT_MOVE, 0, 12.0 * B * H * W * C + 36.0 * B * Ho * Wo * C, stream);
    hipLaunchKernelGGL(col2im_nhwc_bnbwd_f32_kernel, dim3((unsigned)nparts), dim3(CG * PP), lds, (hipStream_t)stream, dcol, y, stat, gamma, beta, act,
                       dz, part, B, H, W, C, Ho, Wo, CG, PP);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_col2im_nhwc_f32(const float* dcol, float* dx, int B, int H, int W, int C, int stride, void* stream) {
    GG_CHECK(dcol && dx && B > 0 && (C & 3) == 0 && (stride == 1 || stride == 2), "gg_col2im_nhwc_f32: bad args");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    GG_PROF(GG_CAT_MOVE, 0, 4.0 * B * (double)H * W * C + 36.0 * B * Ho * Wo * C, stream);
    hipLaunchKernelGGL(col2im_nhwc_f32_kernel, dim3(grid_for((int64_t)B * H * W * (C / 4), 65536)), dim3(256), 0, (hipStream_t)stream, dcol, dx, B, H,
                       W, C, Ho, Wo, stride);
    GG_LAUNCH_CHECK();
    return 0;
}

// partial rows written by the forward (colstats) and by the fused stride-1 data gradient (ep_partials) for an OUTPUT map of Ho x Wo
extern "C" int gg_dwconv_f32_stat_rows(int B, int Ho, int Wo, int C, int stride) {
    const WalkGeom g = dw_use_multi(stride) ? multi_geom(B, Ho, Wo, C) : walk_geom(B, Ho, Wo, C);
    return g.nbx * B * g.strips;
}
static int dw_wgrad_rows(int B, int Ho, int Wo, int C) {      // the tap-gradient walk always uses the one-column geometry
    const WalkGeom g = walk_geom(B, Ho, Wo, C);
    return g.nbx * B * g.strips;
}
static int dw_walk_launch(int mode, const float* x, const float* taps, float* y, const float* dy, int B, int H, int W, int C, int stride,
                          float* part, void* stream, const DwFuse* fuse = nullptr, int in_mode = 0, bool epi = false) {
    GG_CHECK((C & 3) == 0 && C >= 4 && C <= 1024, "gg_dwconv3x3 f32: C must be a multiple of 4, <= 1024 (got %d)", C);
    GG_CHECK(B <= 65535, "gg_dwconv3x3 f32: batch too large for one launch");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    if (mode != DWM_WGRAD && dw_use_multi(stride)) {
        const WalkGeom g = multi_geom(B, H, W, C);
        const dim3 grid(g.nbx, B, g.strips), block(g.threads);
        const size_t lds = part ? (size_t)g.PX * 2 * C * sizeof(float) : 0;
        DwFuse fz;
        memset(&fz, 0, sizeof(fz));
        if (fuse) fz = *fuse;
#define GG_DW_MULTI(M_, I_, E_) hipLaunchKernelGGL((dw3x3_s1_multi_f32_kernel<M_, I_, E_>), grid, block, lds, (hipStream_t)stream, x, taps, y, H, W, C, g.PX, g.rows_per_strip, part, fz)
        if (mode == DWM_FWD && in_mode == 0 && !epi) GG_DW_MULTI(DWM_FWD, 0, false);
        else if (mode == DWM_FWD && in_mode == 1 && !epi) GG_DW_MULTI(DWM_FWD, 1, false);
        else if (mode == DWM_FLIP && in_mode == 0 && !epi) GG_DW_MULTI(DWM_FLIP, 0, false);
        else if (mode == DWM_FLIP && in_mode == 2 && !epi) GG_DW_MULTI(DWM_FLIP, 2, false);
        else if (mode == DWM_FLIP && in_mode == 0 && epi) GG_DW_MULTI(DWM_FLIP, 0, true);
        else if (mode == DWM_FLIP && in_mode == 2 && epi) GG_DW_MULTI(DWM_FLIP, 2, true);
        else { gg_set_error("gg_dwconv3x3 f32: multi-column variant (mode %d, in %d, epi %d) is not built", mode, in_mode, (int)epi); return -1; }
#undef GG_DW_MULTI
        GG_LAUNCH_CHECK();
        return 0;
    }
    const WalkGeom g = walk_geom(B, Ho, Wo, C);
    const dim3 grid(g.nbx, B, g.strips), block(g.threads);
    const int NR = mode == DWM_WGRAD ? 9 : 2;
    const size_t lds = part ? (size_t)g.PX * NR * C * sizeof(float) : 0;
    GG_CHECK(lds <= 64 * 1024, "gg_dwconv3x3 f32: reduction tile too large");
    hipStream_t st = (hipStream_t)stream;
    DwFuse fz;
    memset(&fz, 0, sizeof(fz));
    if (fuse) fz = *fuse;
#define GG_DW_LAUNCHF(S_, M_, I_, E_) hipLaunchKernelGGL((dw3x3_walk_f32_kernel<S_, M_, I_, E_>), grid, block, lds, st, x, taps, y, dy, H, W, C, Ho, Wo, g.PX, g.rows_per_strip, part, fz)
#define GG_DW_LAUNCH(S_, M_) GG_DW_LAUNCHF(S_, M_, 0, false)
    if (in_mode != 0 || epi) {
        if (mode == DWM_FWD && in_mode == 1 && !epi) { if (stride == 1) GG_DW_LAUNCHF(1, DWM_FWD, 1, false); else GG_DW_LAUNCHF(2, DWM_FWD, 1, false); }
        else if (mode == DWM_FLIP && stride == 1 && in_mode == 2 && epi) GG_DW_LAUNCHF(1, DWM_FLIP, 2, true);
        else if (mode == DWM_FLIP && stride == 1 && in_mode == 2 && !epi) GG_DW_LAUNCHF(1, DWM_FLIP, 2, false);
        else if (mode == DWM_FLIP && stride == 1 && in_mode == 0 && epi) GG_DW_LAUNCHF(1, DWM_FLIP, 0, true);
        else { gg_set_error("gg_dwconv3x3 f32: fused variant (mode %d, stride %d, in %d, epi %d) is not built", mode, stride, in_mode, (int)epi); return -1; }
        GG_LAUNCH_CHECK();
        return 0;
    }
    if (stride == 1) {
        if (mode == DWM_FWD) GG_DW_LAUNCH(1, DWM_FWD);
        else if (mode == DWM_FLIP) GG_DW_LAUNCH(1, DWM_FLIP);
        else GG_DW_LAUNCH(1, DWM_WGRAD);
    } else {
        if (mode == DWM_FWD) GG_DW_LAUNCH(2, DWM_FWD);
        else if (mode == DWM_WGRAD) GG_DW_LAUNCH(2, DWM_WGRAD);
        else { gg_set_error("gg_dwconv3x3 f32: flipped stride-2 walk does not exist"); return -1; }
    }
#undef GG_DW_LAUNCH
#undef GG_DW_LAUNCHF
    GG_LAUNCH_CHECK();
    return 0;
}
// y = depthwise conv3x3(x, taps [9][C]) pad 1; colstats: gg_dwconv_f32_stat_rows(B, Ho, Wo, C, stride) partial rows [2][C] (sum y, sum y^2) or NULL
extern "C" int gg_dwconv3x3_fwd_f32(const float* x, const float* taps, float* y, int B, int H, int W, int C, int stride, float* colstats,
                                    void* stream) {
    GG_CHECK(x && taps && y && B > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2), "gg_dwconv3x3_fwd_f32: bad args");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    GG_PROF(GG_CAT_DWCONV, 18.0 * B * Ho * Wo * C, 4.0 * B * C * ((double)H * W + (double)Ho * Wo), stream);
    return dw_walk_launch(DWM_FWD, x, taps, y, nullptr, B, H, W, C, stride, colstats, stream);
}
// dx = gradient of the conv w.r.t. its input; (H, W) = the conv INPUT map
extern "C" int gg_dwconv3x3_bwd_data_f32(const float* dy, const float* taps, float* dx, int B, int H, int W, int C, int stride, void* stream) {
    GG_CHECK(dy && taps && dx && B > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2) && (C & 3) == 0, "gg_dwconv3x3_bwd_data_f32: bad args");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    GG_PROF(GG_CAT_DWCONV, 18.0 * B * Ho * Wo * C, 4.0 * B * C * ((double)H * W + (double)Ho * Wo), stream);
    if (stride == 1) return dw_walk_launch(DWM_FLIP, dy, taps, dx, nullptr, B, H, W, C, 1, nullptr, stream);
    hipLaunchKernelGGL(dw3x3_s2_bwd_data_f32_kernel, dim3(grid_for((int64_t)B * H * W * (C / 4), 65536)), dim3(256), 0, (hipStream_t)stream, dy, taps,
                       dx, B, H, W, C, Ho, Wo);
    GG_LAUNCH_CHECK();
    return 0;
}
static int dw_s2_fused_blocks(int B, int H, int W, int C) {       // work items are 2 x 2 quads of input pixels
    const int CG = C / 4, PX = std::max(1, 256 / CG);
    const int64_t total = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2);
    return (int)std::max<int64_t>(1, std::min<int64_t>(4096, gg_cdiv(total, (int64_t)PX * 4)));
}
extern "C" int gg_dwconv_f32_s2_fused_stat_rows(int B, int H, int W, int C) { return dw_s2_fused_blocks(B, H, W, C); }
// stride-2 data gradient (H, W = the conv INPUT map) with the BatchNorm-backward passes on both sides riding on it; see the kernel
extern "C" int gg_dwconv3x3_s2_bwd_data_fused_f32(const float* dz_in, const float* y_in, const float* in_coef, const float* taps, float* out, int B, int H,
                                                  int W, int C, const float* ep_y, const float* ep_stat, const float* ep_gamma, const float* ep_beta,
                                                  int ep_act, float* ep_partials, void* stream) {
    GG_CHECK(dz_in && taps && out && B > 0 && H > 0 && W > 0 && (C & 3) == 0 && C <= 1024, "gg_dwconv3x3_s2_bwd_data_fused_f32: bad args (C %% 4, C <= 1024)");
    GG_CHECK(!in_coef || y_in, "gg_dwconv3x3_s2_bwd_data_fused_f32: in_coef needs y_in");
    GG_CHECK(!ep_y || (ep_stat && ep_gamma && ep_beta && ep_partials), "gg_dwconv3x3_s2_bwd_data_fused_f32: the epilogue needs stat, gamma, beta and partials");
    GG_CHECK((int64_t)B * H * W < ((int64_t)1 << 32), "gg_dwconv3x3_s2_bwd_data_fused_f32: too many pixels for 32-bit decoding");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    GG_PROF(GG_CAT_DWCONV, 18.0 * B * Ho * Wo * C, 4.0 * B * C * ((double)H * W * (1.0 + (ep_y != nullptr)) + (double)Ho * Wo * (1.0 + (in_coef != nullptr))), stream);
    const int CG = C / 4, PX = std::max(1, 256 / CG);
    const int blocks = dw_s2_fused_blocks(B, H, W, C);
    const int64_t total = (int64_t)B * Ho * Wo;                       // quads: ceil(H/2) x ceil(W/2) == Ho x Wo
    const int64_t ppb = gg_align(gg_cdiv(total, blocks), PX);
    const dim3 grid((unsigned)gg_cdiv(total, ppb)), block((unsigned)(CG * PX));
    GG_CHECK((int)grid.x <= blocks, "gg_dwconv3x3_s2_bwd_data_fused_f32: internal grid error");
    if (ep_y && (int)grid.x < blocks)       // rows the finalize step will read but no block writes
        GG_HIP(hipMemsetAsync(ep_partials + (int64_t)grid.x * 2 * C, 0, (size_t)(blocks - grid.x) * 2 * C * sizeof(float), (hipStream_t)stream));
#define GG_S2F(I_, E_) hipLaunchKernelGGL((dw3x3_s2_bwd_data_fused_f32_kernel<I_, E_>), grid, block, 0, (hipStream_t)stream, dz_in, y_in, in_coef, taps, out, B, H, W, \
                                          C, Ho, Wo, ep_y, ep_stat, ep_gamma, ep_beta, ep_act, ep_partials, ppb)
    if (in_coef && ep_y) GG_S2F(true, true);
    else if (in_coef) GG_S2F(true, false);
    else if (ep_y) GG_S2F(false, true);
    else GG_S2F(false, false);
#undef GG_S2F
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int64_t gg_dwconv_f32_wgrad_scratch_floats(int B, int H, int W, int C, int stride) {
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    return ((int64_t)dw_wgrad_rows(B, Ho, Wo, C) + GG_REDUCE_SLICES) * 9 * C;
}
// grad (C,1,3,3) (+)= sum_{b,y,x} x[b, y*s+ky-1, x*s+kx-1, c] * dy[b, y, x, c]
extern "C" int gg_dwconv3x3_bwd_weight_f32(const float* x, const float* dy, int B, int H, int W, int C, int stride, float* scratch, float* grad,
                                           int accumulate, void* stream) {
    GG_CHECK(x && dy && scratch && grad && B > 0 && (stride == 1 || stride == 2), "gg_dwconv3x3_bwd_weight_f32: bad args");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    GG_PROF(GG_CAT_DWCONV, 18.0 * B * Ho * Wo * C, 4.0 * B * C * ((double)H * W + (double)Ho * Wo), stream);
    GG_TRY(dw_walk_launch(DWM_WGRAD, x, nullptr, nullptr, dy, B, H, W, C, stride, scratch, stream));
    const float* rows; int nrows;
    gg_reduce_rows(scratch, dw_wgrad_rows(B, Ho, Wo, C), 9 * C, (hipStream_t)stream, &rows, &nrows);
    hipLaunchKernelGGL(dw_wgrad_final_f32_kernel, dim3((unsigned)gg_cdiv(9 * C, 256)), dim3(256), 0, (hipStream_t)stream, rows, nrows, C, grad, accumulate);
    GG_LAUNCH_CHECK();
    return 0;
}

// depthwise ConvNorm whose input is act(BatchNorm(x_prebn)) of the ConvNorm in front, formed while loading (timm MBConv.conv2 /
// PatchMerging.conv2): replaces gg_bn_apply_f32 + gg_dwconv3x3_fwd_f32, the activation tensor is never stored.  in_stat = [mean | rstd][C]
extern "C" int gg_dwconv3x3_fwd_fused_f32(const float* x_prebn, const float* in_stat, const float* in_gamma, const float* in_beta, int in_act,
                                          const float* taps, float* y, int B, int H, int W, int C, int stride, float* colstats, void* stream) {
    GG_CHECK(x_prebn && in_stat && in_gamma && in_beta && taps && y && B > 0 && (stride == 1 || stride == 2), "gg_dwconv3x3_fwd_fused_f32: bad args");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    GG_PROF(GG_CAT_DWCONV, 18.0 * B * Ho * Wo * C, 4.0 * B * C * ((double)H * W + (double)Ho * Wo), stream);
    DwFuse f;
    memset(&f, 0, sizeof(f));
    f.in_a = in_stat; f.in_b = in_gamma; f.in_c = in_beta; f.in_act = in_act;
    return dw_walk_launch(DWM_FWD, x_prebn, taps, y, nullptr, B, H, W, C, stride, colstats, stream, &f, 1, false);
}
// stride-1 data gradient with the BatchNorm-backward passes of the ConvNorms on both sides riding on it:
//   input : in_coef != NULL -> dy = coef0*dz_in + coef1*y_in + coef2 formed on load (apply step of the ConvNorm this conv belongs to)
//   output: ep_y != NULL    -> out = dz = da * ep_act'(BN(ep_y)), ep_partials <- gg_dwconv_f32_stat_rows(B,H,W,C,1) rows of (sum dz, sum dz*xhat)
extern "C" int gg_dwconv3x3_bwd_data_fused_f32(const float* dz_in, const float* y_in, const float* in_coef, const float* taps, float* out, int B, int H,
                                               int W, int C, const float* ep_y, const float* ep_stat, const float* ep_gamma, const float* ep_beta,
                                               int ep_act, float* ep_partials, void* stream) {
    GG_CHECK(dz_in && taps && out && B > 0, "gg_dwconv3x3_bwd_data_fused_f32: bad args");
    GG_CHECK(!in_coef || y_in, "gg_dwconv3x3_bwd_data_fused_f32: in_coef needs y_in");
    GG_CHECK(!ep_y || (ep_stat && ep_gamma && ep_beta && ep_partials), "gg_dwconv3x3_bwd_data_fused_f32: the epilogue needs stat, gamma, beta and partials");
    GG_PROF(GG_CAT_DWCONV, 18.0 * B * H * W * C, 4.0 * B * C * (double)H * W * (2.0 + (in_coef != nullptr) + (ep_y != nullptr)), stream);
    DwFuse f;
    memset(&f, 0, sizeof(f));
    f.x2 = y_in; f.in_a = in_coef; f.ep_y = ep_y; f.ep_stat = ep_stat; f.ep_gamma = ep_gamma; f.ep_beta = ep_beta; f.ep_act = ep_act;
    return dw_walk_launch(DWM_FLIP, dz_in, taps, out, nullptr, B, H, W, C, 1, ep_y ? ep_partials : nullptr, stream, &f, in_coef ? 2 : 0, ep_y != nullptr);
}
