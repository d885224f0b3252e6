// AdamW over flat fp32 parameter ranges (torch.optim.AdamW semantics; reference
// main_coordinator_idun_s3.py:286-291, training/train_eval_loop.py:188-190).  HBM-bound: 16 B read + 12 B
// written per parameter.
#include "common.h"
#include "../../include/gg.h"

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float lr, float beta1, float beta2, float eps,
                                                    float wd, float bc1, float bc2_sqrt, float grad_scale) {
    const int64_t n4 = n >> 2;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = gg[j] * grad_scale;
            pp[j] *= (1.f - lr * wd);
            mm[j] = beta1 * mm[j] + (1.f - beta1) * gr;
            vv[j] = beta2 * vv[j] + (1.f - beta2) * gr * gr;
            const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
            pp[j] -= (lr / bc1) * (mm[j] / denom);
        }
        reinterpret_cast<f32x4*>(p)[i] = pp;
        reinterpret_cast<f32x4*>(m)[i] = mm;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    // tail
    const int64_t t = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < n) {
        const float gr = g[t] * grad_scale;
        float pp = p[t] * (1.f - lr * wd);
        const float mm = beta1 * m[t] + (1.f - beta1) * gr;
        const float vv = beta2 * v[t] + (1.f - beta2) * gr * gr;
        pp -= (lr / bc1) * (mm / (sqrtf(vv) / bc2_sqrt + eps));
        p[t] = pp; m[t] = mm; v[t] = vv;
    }
}

__global__ void fill_f32_kernel(float* __restrict__ p, int64_t n, float val) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = val;
}

extern "C" int gg_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int step, float lr,
                             float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
    GG_CHECK(params && grads && exp_avg && exp_avg_sq && n > 0 && step > 0, "gg_adamw_step: bad args");
    GG_CHECK(((uintptr_t)params & 15) == 0 && ((uintptr_t)grads & 15) == 0 && ((uintptr_t)exp_avg & 15) == 0 &&
                 ((uintptr_t)exp_avg_sq & 15) == 0, "gg_adamw_step: ranges must start 16-byte aligned");
    GG_PROF(GG_CAT_OPTIM, 0, 28.0 * n, stream);
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(gg_cdiv(n / 4 + 1, 256), 8192));
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, lr, beta1,
                       beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
    GG_LAUNCH_CHECK();
    return 0;
}
extern "C" int gg_fill_f32(float* p, int64_t n, float value, void* stream) {
    GG_CHECK(p && n > 0, "gg_fill_f32: bad args");
    int blocks = (int)std::min<int64_t>(gg_cdiv(n, 256), 8192);
    hipLaunchKernelGGL(fill_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, n, value);
    GG_LAUNCH_CHECK();
    return 0;
}
