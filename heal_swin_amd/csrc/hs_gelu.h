// Exact-erf GELU arithmetic shared by the elementwise kernels (gelu.hip) and the GEMM epilogues (gemm_nt.hip).
#pragma once
#include "hs_device.h"

namespace hs {

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, below fp32 round-off of the products it enters); the single
// exponential exp(-x^2/2) it needs is also the Gaussian pdf factor of GELU', so forward and backward cost one v_exp_f32,
// one v_rcp_f32 and a handful of FMAs per element (libm erff made these kernels VALU-bound instead of HBM-bound).
struct GeluParts {
    float cdf, e;  // Phi(x), exp(-x^2/2)
};
__device__ __forceinline__ GeluParts gelu_parts(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    const float e = __expf(-z * z);
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float erf_abs = fmaf(-poly * t, e, 1.f);
    GeluParts r;
    r.cdf = 0.5f * (1.f + copysignf(erf_abs, x));
    r.e = e;
    return r;
}
__device__ __forceinline__ float gelu_f(float x) { return x * gelu_parts(x).cdf; }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const GeluParts g = gelu_parts(x);
    return fmaf(x * 0.3989422804014327f, g.e, g.cdf);
}

}  // namespace hs
