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

// Two elements at a time for the GEMM epilogues, which are VALU-bound (one wave per output block, ~20 VALU slots per
// element): <2 x float> arithmetic selects the packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, two
// lanes' worth per issue), the argument is pre-scaled so that exp(-x^2/2) = exp2(-z'^2) needs no extra multiply, and
//   h = erfc(|x|/sqrt2) / 2 = P(t) t e   (A&S 7.1.26 with the 1/2 folded into the coefficients)
//   GELU(x)  = max(x, 0) - |x| h,     GELU'(x) = 1/2 + copysign(1/2 - h, x) + x e / sqrt(2 pi)
// Same polynomial and therefore the same error bound as gelu_parts above.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_parts2(f32x2 x, f32x2& ax, f32x2& h, f32x2& e) {
    constexpr float kZ = 0.8493218002880191f;              // sqrt(log2(e) / 2):  exp(-x^2/2) = exp2(-(kZ |x|)^2)
    constexpr float kP = 0.3275911f * 0.8325546111576977f;  // A&S p times sqrt(ln 2): p |x| / sqrt2 = kP (kZ |x|)
    ax = f32x2{fabsf(x.x), fabsf(x.y)};
    const f32x2 z = ax * kZ;
    const f32x2 d = z * kP + 1.f;
    const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const f32x2 s = -z * z;
    e = f32x2{__builtin_amdgcn_exp2f(s.x), __builtin_amdgcn_exp2f(s.y)};
    f32x2 poly = t * (0.5f * 1.061405429f) + (0.5f * -1.453152027f);
    poly = poly * t + (0.5f * 1.421413741f);
    poly = poly * t + (0.5f * -0.284496736f);
    poly = poly * t + (0.5f * 0.254829592f);
    h = poly * (t * e);
}
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {
    f32x2 ax, h, e;
    gelu_parts2(x, ax, h, e);
    const f32x2 relu = {fmaxf(x.x, 0.f), fmaxf(x.y, 0.f)};
    return relu - ax * h;
}
__device__ __forceinline__ f32x2 gelu_grad2(f32x2 x) {
    f32x2 ax, h, e;
    gelu_parts2(x, ax, h, e);
    const f32x2 w = 0.5f - h;
    const f32x2 cdf = f32x2{copysignf(w.x, x.x), copysignf(w.y, x.y)} + 0.5f;
    return (x * 0.3989422804014327f) * e + cdf;
}

}  // namespace hs
