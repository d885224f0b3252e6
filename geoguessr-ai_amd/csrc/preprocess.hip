// Input pipeline and embedding-store kernels on the two sides of the encoder ("next" rows f2 / f3 of SURVEY.md 8f).
//
// gg_preprocess_bilinear: what training's batch loop does to every image batch before the encoder
// (main_coordinator_idun_s3.py:337-381): F.interpolate(size, mode="bilinear", align_corners=False) -> (/255 for uint8) ->
// (x - mean) / std, as ONE pass (HBM-bound: every source texel is read from L2-resident rows, the output written once).
// Index / weight arithmetic follows ATen's area_pixel_compute_source_index + compute_indices_weights for the linear mode.
//
// gg_segment_mean: prototype building (models/proto_refiner.py:461-517): the running mean of member embeddings per cluster,
// summed in member order in fp32 like the reference's sum_cpu.add_() loop, then divided by the count.
#include "common.h"
#include "../../include/gg.h"

struct PreParams {
    const void* src; int src_u8;
    int N, Hs, Ws, Hd, Wd;
    float* dst;
    float mean[3], istd_unused[3], stdv[3];
    int normalize;
    float scale_h, scale_w;
};
__device__ __forceinline__ float pre_fetch(const PreParams& p, int64_t plane, int y, int x) {
    const int64_t i = plane + (int64_t)y * p.Ws + x;
    return p.src_u8 ? (float)reinterpret_cast<const unsigned char*>(p.src)[i] : reinterpret_cast<const float*>(p.src)[i];
}
__global__ __launch_bounds__(256) void preprocess_bilinear_kernel(PreParams p) {
    const int64_t total = (int64_t)p.N * 3 * p.Hd * p.Wd;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % p.Wd), y = (int)((i / p.Wd) % p.Hd);
        const int64_t nc = i / ((int64_t)p.Wd * p.Hd);
        const int c = (int)(nc % 3);
        // ATen: src = scale * (dst + 0.5) - 0.5, clamped at 0; idx0 = (int)src; idx1 = min(idx0 + 1, in - 1); lambda1 = src - idx0
        const float sy = fmaxf(p.scale_h * ((float)y + 0.5f) - 0.5f, 0.f), sx = fmaxf(p.scale_w * ((float)x + 0.5f) - 0.5f, 0.f);
        const int y0 = min((int)sy, p.Hs - 1), x0 = min((int)sx, p.Ws - 1);
        const int y1 = min(y0 + 1, p.Hs - 1), x1 = min(x0 + 1, p.Ws - 1);
        const float ly1 = fminf(fmaxf(sy - (float)y0, 0.f), 1.f), lx1 = fminf(fmaxf(sx - (float)x0, 0.f), 1.f);
        const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const int64_t plane = nc * (int64_t)p.Hs * p.Ws;
        float v;
        if (p.Hs == p.Hd && p.Ws == p.Wd) v = pre_fetch(p, plane, y, x);      // F.interpolate to the same size is the identity
        else
            v = ly0 * (lx0 * pre_fetch(p, plane, y0, x0) + lx1 * pre_fetch(p, plane, y0, x1)) +
                ly1 * (lx0 * pre_fetch(p, plane, y1, x0) + lx1 * pre_fetch(p, plane, y1, x1));
        if (p.src_u8) v = v / 255.0f;
        if (p.normalize) v = (v - p.mean[c]) / p.stdv[c];
        p.dst[i] = v;
    }
}
extern "C" int gg_preprocess_bilinear(const void* src, int src_u8, int N, int Hs, int Ws, float* dst, int Hd, int Wd,
                                      const float* mean3, const float* std3, void* stream) {
    GG_CHECK(src && dst && N > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0, "gg_preprocess_bilinear: bad args");
    GG_CHECK((mean3 == nullptr) == (std3 == nullptr), "gg_preprocess_bilinear: mean and std come together");
    PreParams p;
    p.src = src; p.src_u8 = src_u8; p.N = N; p.Hs = Hs; p.Ws = Ws; p.Hd = Hd; p.Wd = Wd; p.dst = dst;
    p.normalize = mean3 != nullptr;
    for (int c = 0; c < 3; ++c) {
        p.mean[c] = mean3 ? mean3[c] : 0.f; p.stdv[c] = std3 ? std3[c] : 1.f; p.istd_unused[c] = 0.f;
        if (std3) GG_CHECK(std3[c] != 0.f, "gg_preprocess_bilinear: zero std");
    }
    p.scale_h = (float)Hs / (float)Hd; p.scale_w = (float)Ws / (float)Wd;
    const int64_t total = (int64_t)N * 3 * Hd * Wd;
    GG_PROF(GG_CAT_MOVE, 0, 4.0 * total + (src_u8 ? 1.0 : 4.0) * N * 3 * (double)Hs * Ws, stream);
    hipLaunchKernelGGL(preprocess_bilinear_kernel, dim3((unsigned)std::min<int64_t>(gg_cdiv(total, 256), 65536)), dim3(256), 0,
                       (hipStream_t)stream, p);
    GG_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------- gg_preprocess_pil
// The raw-image side of the embedders (pretrain/tinyvit_embedder.py:51-53,67-69: timm's eval transform; pretrain/clip_embedder.py:25,51-55: CLIPProcessor;
// inference.py:74-85: a torchvision Compose): resize with Pillow (bicubic / bilinear, the filter support scaled with the reduction = antialiased), centre
// crop, 1/255, (x - mean) / std.  The resize is Pillow's ImagingResample on 8-bit pixels (src/libImaging/Resample.c, Pillow 12.2), restated integer for
// integer -- the uint8 result is bit-identical to Image.resize (tests/golden/preprocess_pil.npz, produced by running Pillow / transformers):
//   * per axis and output index: the window [xmin, xmin + xmax) and double-precision filter weights, normalised by their sum, in 22-bit fixed point
//     (pil_coeffs_kernel: the same double operations in the same order, fused multiply-adds switched off);
//   * horizontal pass into an 8-BIT intermediate image: clip8((2^21 + sum pixel * k) >> 22); then the vertical pass the same way.
// Only the crop window is computed: its columns in the horizontal pass, its rows in the vertical one.
struct PilAxis { int in_size, out_size, ksize, filter; double scale, filterscale, support; };
__device__ __forceinline__ double pil_filter(double x, int filter) {
#pragma clang fp contract(off)
    if (x < 0.0) x = -x;
    if (filter == 2) return x < 1.0 ? 1.0 - x : 0.0;                       // BILINEAR
    const double a = -0.5;                                                 // BICUBIC
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
__global__ __launch_bounds__(64) void pil_coeffs_kernel(PilAxis ax, int* __restrict__ bounds, int* __restrict__ kk) {
#pragma clang fp contract(off)
    const int xx = blockIdx.x * 64 + threadIdx.x;
    if (xx >= ax.out_size) return;
    const double center = 0.0 + (xx + 0.5) * ax.scale;
    const double ss = 1.0 / ax.filterscale;
    int xmin = (int)(center - ax.support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + ax.support + 0.5);
    if (xmax > ax.in_size) xmax = ax.in_size;
    xmax -= xmin;
    int* k = kk + (int64_t)xx * ax.ksize;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += pil_filter((x + xmin - center + 0.5) * ss, ax.filter);
    for (int x = 0; x < ax.ksize; ++x) {
        int v = 0;
        if (x < xmax) {
            double w = pil_filter((x + xmin - center + 0.5) * ss, ax.filter);
            if (ww != 0.0) w /= ww;
            v = w < 0 ? (int)(-0.5 + w * (double)(1 << 22)) : (int)(0.5 + w * (double)(1 << 22));
        }
        k[x] = v;
    }
    bounds[2 * xx] = xmin; bounds[2 * xx + 1] = xmax;
}
__device__ __forceinline__ unsigned char pil_clip8(int ss) { return (unsigned char)min(max(ss >> 22, 0), 255); }
// tmp[y][x][c] (x over the crop's columns) = horizontal pass of src[y][.][c]; resample == 0: a copy of those columns
__global__ __launch_bounds__(256) void pil_horizontal_kernel(const unsigned char* __restrict__ src, int Hs, int Ws, const int* __restrict__ bounds, const int* __restrict__ kk,
                                                             int ksize, int resample, int left, int Wc, unsigned char* __restrict__ tmp) {
    const int64_t total = (int64_t)Hs * Wc;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wc), y = (int)(i / Wc), xx = x + left;
        const unsigned char* row = src + (int64_t)y * Ws * 3;
        unsigned char r, g, b;
        if (!resample) { r = row[xx * 3]; g = row[xx * 3 + 1]; b = row[xx * 3 + 2]; }
        else {
            const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
            const int* k = kk + (int64_t)xx * ksize;
            int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
            for (int j = 0; j < xmax; ++j) {
                const int w = k[j];
                const unsigned char* px = row + (xmin + j) * 3;
                s0 += px[0] * w; s1 += px[1] * w; s2 += px[2] * w;
            }
            r = pil_clip8(s0); g = pil_clip8(s1); b = pil_clip8(s2);
        }
        unsigned char* o = tmp + i * 3;
        o[0] = r; o[1] = g; o[2] = b;
    }
}
// vertical pass over tmp (Hs rows x Wc columns) for the crop's rows, then 1/255 and (x - mean) / std; dst is CHW float32, dst_u8 (optional) the HWC uint8 crop
__global__ __launch_bounds__(256) void pil_vertical_kernel(const unsigned char* __restrict__ tmp, int Hs, int Wc, const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                           int resample, int top, int Hc, int mul_rescale, float m0, float m1, float m2, float d0, float d1, float d2, int normalize,
                                                           float* __restrict__ dst, unsigned char* __restrict__ dst_u8) {
    const int64_t total = (int64_t)Hc * Wc;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wc), y = (int)(i / Wc), yy = y + top;
        unsigned char px[3];
        if (!resample) { const unsigned char* q = tmp + ((int64_t)yy * Wc + x) * 3; px[0] = q[0]; px[1] = q[1]; px[2] = q[2]; }
        else {
            const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
            const int* k = kk + (int64_t)yy * ksize;
            int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
            for (int j = 0; j < ymax; ++j) {
                const int w = k[j];
                const unsigned char* q = tmp + ((int64_t)(ymin + j) * Wc + x) * 3;
                s0 += q[0] * w; s1 += q[1] * w; s2 += q[2] * w;
            }
            px[0] = pil_clip8(s0); px[1] = pil_clip8(s1); px[2] = pil_clip8(s2);
        }
        const float mean[3] = {m0, m1, m2}, stdv[3] = {d0, d1, d2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = mul_rescale ? (float)px[c] * (1.0f / 255.0f) : (float)px[c] / 255.0f;      // transformers rescales by 1/255, torchvision's ToTensor divides by 255
            if (normalize) v = (v - mean[c]) / stdv[c];
            dst[(int64_t)c * total + i] = v;
            if (dst_u8) dst_u8[i * 3 + c] = px[c];
        }
    }
}
static bool pil_axis(int in_size, int out_size, int filter, PilAxis* ax) {
    // precompute_coeffs: filterscale = scale = (double)(in1 - in0) / outSize with float box coordinates in0 = 0, in1 = inSize; support = filter support * max(scale, 1)
    ax->in_size = in_size; ax->out_size = out_size; ax->filter = filter;
    ax->scale = (double)((float)in_size - 0.0f) / out_size;
    ax->filterscale = ax->scale < 1.0 ? 1.0 : ax->scale;
    ax->support = (filter == 2 ? 1.0 : 2.0) * ax->filterscale;
    const double ks = ceil(ax->support) * 2 + 1;
    if (ks > 1 << 20) return false;
    ax->ksize = (int)ceil(ax->support) * 2 + 1;
    return true;
}
static int64_t pil_align(int64_t b) { return (b + 255) / 256 * 256; }
extern "C" int64_t gg_preprocess_pil_workspace_bytes(int Hs, int Ws, int filter, int Hr, int Wr, int Wc) {
    PilAxis ax, ay;
    if (Hs <= 0 || Ws <= 0 || Hr <= 0 || Wr <= 0 || Wc <= 0 || (filter != 2 && filter != 3) || !pil_axis(Ws, Wr, filter, &ax) || !pil_axis(Hs, Hr, filter, &ay)) return -1;
    return pil_align(8LL * Wr) + pil_align(4LL * Wr * ax.ksize) + pil_align(8LL * Hr) + pil_align(4LL * Hr * ay.ksize) + pil_align(3LL * Hs * Wc);
}
extern "C" int gg_preprocess_pil(const void* src_hwc_u8, int Hs, int Ws, int filter, int Hr, int Wr, int crop_top, int crop_left, int Hc, int Wc, int mul_rescale,
                                 const float* mean3, const float* std3, float* dst_chw, void* dst_u8_hwc, void* workspace, void* stream) {
    GG_CHECK(src_hwc_u8 && dst_chw && workspace && Hs > 0 && Ws > 0 && Hr > 0 && Wr > 0 && Hc > 0 && Wc > 0, "gg_preprocess_pil: bad args");
    GG_CHECK(filter == 2 || filter == 3, "gg_preprocess_pil: filter must be 2 (PIL BILINEAR) or 3 (PIL BICUBIC), got %d", filter);
    GG_CHECK(crop_top >= 0 && crop_left >= 0 && crop_top + Hc <= Hr && crop_left + Wc <= Wr,
             "gg_preprocess_pil: the crop window (%d, %d) + (%d x %d) must lie inside the resized image (%d x %d): the upstream transforms pad here, which is not built",
             crop_top, crop_left, Hc, Wc, Hr, Wr);
    GG_CHECK((mean3 == nullptr) == (std3 == nullptr), "gg_preprocess_pil: mean and std come together");
    PilAxis ax, ay;
    GG_CHECK(pil_axis(Ws, Wr, filter, &ax) && pil_axis(Hs, Hr, filter, &ay), "gg_preprocess_pil: reduction factor too large");
    GG_CHECK((int64_t)Hs * Ws < (1LL << 31) / 3 && (int64_t)Wr * ax.ksize < (1LL << 29) && (int64_t)Hr * ay.ksize < (1LL << 29), "gg_preprocess_pil: image too large");
    if (std3) for (int c = 0; c < 3; ++c) GG_CHECK(std3[c] != 0.f, "gg_preprocess_pil: zero std");
    char* w = (char*)workspace;
    int* bx = (int*)w; w += pil_align(8LL * Wr);
    int* kx = (int*)w; w += pil_align(4LL * Wr * ax.ksize);
    int* by = (int*)w; w += pil_align(8LL * Hr);
    int* ky = (int*)w; w += pil_align(4LL * Hr * ay.ksize);
    unsigned char* tmp = (unsigned char*)w;
    hipStream_t st = (hipStream_t)stream;
    const int rx = Wr != Ws, ry = Hr != Hs;                  // ImagingResample: a pass runs only when that axis changes size
    GG_PROF(GG_CAT_MOVE, 0, 3.0 * Hs * Ws + 3.0 * Hs * Wc * 2 + 12.0 * Hc * Wc, stream);
    if (rx) hipLaunchKernelGGL(pil_coeffs_kernel, dim3((unsigned)gg_cdiv(Wr, 64)), dim3(64), 0, st, ax, bx, kx);
    if (ry) hipLaunchKernelGGL(pil_coeffs_kernel, dim3((unsigned)gg_cdiv(Hr, 64)), dim3(64), 0, st, ay, by, ky);
    hipLaunchKernelGGL(pil_horizontal_kernel, dim3((unsigned)std::min<int64_t>(gg_cdiv((int64_t)Hs * Wc, 256), 65536)), dim3(256), 0, st, (const unsigned char*)src_hwc_u8, Hs, Ws,
                       bx, kx, ax.ksize, rx, crop_left, Wc, tmp);
    hipLaunchKernelGGL(pil_vertical_kernel, dim3((unsigned)std::min<int64_t>(gg_cdiv((int64_t)Hc * Wc, 256), 65536)), dim3(256), 0, st, tmp, Hs, Wc, by, ky, ay.ksize, ry, crop_top,
                       Hc, mul_rescale, mean3 ? mean3[0] : 0.f, mean3 ? mean3[1] : 0.f, mean3 ? mean3[2] : 0.f, std3 ? std3[0] : 1.f, std3 ? std3[1] : 1.f, std3 ? std3[2] : 1.f,
                       mean3 != nullptr, dst_chw, (unsigned char*)dst_u8_hwc);
    GG_LAUNCH_CHECK();
    return 0;
}

// out[k][:] = (sum over members m of cluster k, in list order, of emb[member[m]][:]) / count_k   (zeros for empty clusters)
// One 256-thread block per cluster; thread d strides over the embedding dim, so each column is summed sequentially in member
// order -- the same fp32 additions, in the same order, as the reference's running sum.
__global__ __launch_bounds__(256) void segment_mean_kernel(const float* __restrict__ emb, int64_t ld, const int64_t* __restrict__ ptr,
                                                           const int64_t* __restrict__ member, int D, float* __restrict__ out) {
    const int k = blockIdx.x;
    const int64_t b = ptr[k], e = ptr[k + 1];
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float s = 0.f;
        for (int64_t m = b; m < e; ++m) s += emb[member[m] * ld + d];
        out[(int64_t)k * D + d] = e > b ? s / (float)(e - b) : 0.f;
    }
}
extern "C" int gg_segment_mean(const float* emb, int64_t ld, const int64_t* ptr, const int64_t* member, int num_segments, int D,
                               float* out, void* stream) {
    GG_CHECK(emb && ptr && member && out && num_segments > 0 && D > 0 && ld >= D, "gg_segment_mean: bad args");
    hipLaunchKernelGGL(segment_mean_kernel, dim3(num_segments), dim3(256), 0, (hipStream_t)stream, emb, ld, ptr, member, D, out);
    GG_LAUNCH_CHECK();
    return 0;
}
