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
