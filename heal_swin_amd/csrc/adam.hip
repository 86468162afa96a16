// hs_adam_step: the optimizer step of the training loop over FLAT parameter / gradient / moment buffers.
//
// The reference's trainer steps torch.optim.Adam / AdamW over the model's ~330 parameter tensors
// (training/optimizer.py:57-66).  Here the gradients already live in a few flat fp32 buckets (parallel.GradBucketAllReduce);
// heal_swin_amd.optim.FlatAdam lays parameters and moments out the same way, so a step is ONE elementwise launch per bucket that
// also writes the bf16 copy of the updated parameters the next forward's GEMMs read (ops.ParamCastCache): 16 B read + 14 B written
// per parameter instead of a multi-tensor Adam (28 B) plus a separate multi-tensor cast (6 B) plus their ~40 launches -- on the
// launch-heavy HEAL-SWIN-T / nside 128 step the two were 0.86 + 0.4 ms of 16.7 ms (profiles/archive_r01_r04/r04_k_T128_summary.txt).
//
// Arithmetic = torch.optim.Adam (amsgrad = False, maximize = False), single-tensor form:
//   g += wd * p (Adam)   |   p *= 1 - lr * wd (AdamW, `decoupled`)
//   m += (g - m) (1 - b1);   v = b2 v + (1 - b2) g^2;   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// with the step count t = *step + 1 read from DEVICE memory (hs_adam_advance increments it after the last bucket), so that a
// captured HIP graph of the whole training step replays with the right bias corrections.
#include "hs_device.h"

namespace hs {
namespace {

struct AdamArgs {
    float* p;
    const float* g;
    float* m;
    float* v;
    uint16_t* lowp;  // bf16 copy of the updated parameters, or null
    int64_t n;
    float lr;
    const float* lr_dev;  // overrides lr when non-null (a device scalar: learning-rate schedules under graph replay)
    float beta1, beta2, eps, weight_decay;
    int decoupled;
    const int64_t* step;
};

__device__ __forceinline__ float adam_one(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps, float wd,
                                          int decoupled, float inv_bc1, float inv_sqrt_bc2) {
    if (wd != 0.f) {
        if (decoupled) p *= 1.f - lr * wd;
        else g += wd * p;
    }
    m += (g - m) * (1.f - b1);
    v = v * b2 + (1.f - b2) * g * g;
    const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
    p -= (lr * inv_bc1) * (m / denom);
    return p;
}

__global__ void __launch_bounds__(256) adam_kernel(const AdamArgs a) {
    __shared__ float corr[2];
    const float lr = a.lr_dev ? *a.lr_dev : a.lr;
    if (threadIdx.x == 0) {  // bias corrections in double, once per workgroup (torch forms them on the host in double)
        const double t = (double)(*a.step + 1);
        corr[0] = (float)(1.0 / (1.0 - pow((double)a.beta1, t)));
        corr[1] = (float)(1.0 / sqrt(1.0 - pow((double)a.beta2, t)));
    }
    __syncthreads();
    const float inv_bc1 = corr[0], inv_sqrt_bc2 = corr[1];
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < a.n) {
        float4 p = *(const float4*)(a.p + i), m = *(const float4*)(a.m + i), v = *(const float4*)(a.v + i);
        const float4 g = *(const float4*)(a.g + i);
        adam_one(p.x, g.x, m.x, v.x, lr, a.beta1, a.beta2, a.eps, a.weight_decay, a.decoupled, inv_bc1, inv_sqrt_bc2);
        adam_one(p.y, g.y, m.y, v.y, lr, a.beta1, a.beta2, a.eps, a.weight_decay, a.decoupled, inv_bc1, inv_sqrt_bc2);
        adam_one(p.z, g.z, m.z, v.z, lr, a.beta1, a.beta2, a.eps, a.weight_decay, a.decoupled, inv_bc1, inv_sqrt_bc2);
        adam_one(p.w, g.w, m.w, v.w, lr, a.beta1, a.beta2, a.eps, a.weight_decay, a.decoupled, inv_bc1, inv_sqrt_bc2);
        *(float4*)(a.p + i) = p;
        *(float4*)(a.m + i) = m;
        *(float4*)(a.v + i) = v;
        if (a.lowp) *(uint2*)(a.lowp + i) = make_uint2(pack_bf16x2(p.x, p.y), pack_bf16x2(p.z, p.w));
    } else {
        for (int64_t j = i; j < a.n; ++j) {
            float p = a.p[j], m = a.m[j], v = a.v[j];
            adam_one(p, a.g[j], m, v, lr, a.beta1, a.beta2, a.eps, a.weight_decay, a.decoupled, inv_bc1, inv_sqrt_bc2);
            a.p[j] = p;
            a.m[j] = m;
            a.v[j] = v;
            if (a.lowp) a.lowp[j] = float_to_bf16(p);
        }
    }
}

__global__ void adam_advance_kernel(int64_t* step) { *step += 1; }

}  // namespace
}  // namespace hs

extern "C" {

int hs_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, const float* lr_dev, float beta1,
                 float beta2, float eps, float weight_decay, int decoupled, const int64_t* step, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(p && g && m && v && step, "hs_adam_step: null pointer");
    HS_CHECK_ARG(n > 0 && n < ((int64_t)1 << 40), "hs_adam_step: bad length");
    HS_CHECK_ARG(((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16 == 0 && (uintptr_t)p_bf16 % 8 == 0,
                 "hs_adam_step: buffers must be 16-byte aligned (bf16 copy: 8)");
    HS_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "hs_adam_step: bad hyper-parameters");
    AdamArgs a{p, g, m, v, (uint16_t*)p_bf16, n, lr, lr_dev, beta1, beta2, eps, weight_decay, decoupled, step};
    const int64_t blocks = (n + 1023) / 1024;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    HS_LAUNCH_CHECK("adam_step");
    return HS_OK;
}

int hs_adam_advance(int64_t* step, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(step, "hs_adam_advance: null pointer");
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step);
    HS_LAUNCH_CHECK("adam_advance");
    return HS_OK;
}

}  // extern "C"
