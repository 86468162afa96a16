// Exact-erf GELU arithmetic (nn.GELU, swin_hp_transformer.py:31) shared by the elementwise kernels (gelu.hip) and the GEMM
// epilogues (gemm_nt.hip).
//
// Division-free, one exponential per element (round 4):
//     h(a) = Phi(-a) = exp2(p(a)),   p = degree-5 polynomial in a = |x|   (log2 of the Gaussian tail is nearly quadratic)
//     GELU(x)  = max(x, 0) - a h
//     GELU'(x) = 1/2 + copysign(1/2 - h, x) + x exp2(-x^2 log2(e)/2 - log2 sqrt(2 pi))
// p is the weighted minimax fit of log2 Phi(-a) on [0, 6.5] (weight 1 + a, so that both Phi and a Phi are bounded); its leading
// coefficient is negative and p decreases monotonically beyond the fitted range, so the tail underflows to 0 like the function.
// Evaluated in fp32 (Horner, v_exp_f32): |Phi - exact| <= 9.3e-7, |GELU - exact| <= 8.3e-7, |GELU' - exact| <= 9.6e-7 for all x
// (tests/test_gpu_kernels.py::test_gelu_matches_erf_gelu bounds the shipped kernels at 2e-6) -- three orders of magnitude below
// a bf16 rounding of the result.  Rounds 1-3 used Abramowitz & Stegun 7.1.26 (1.5e-7: one v_rcp_f32 + one v_exp_f32 + 7
// multiply-adds per element); this form costs 12 instead of 18 VALU issue slots per element PAIR in the packed forward
// epilogue and half the transcendentals (profiles/archive_r01_r04/r03_gemm_trace_role_split.txt: that epilogue is VALU-bound).
#pragma once
#include "hs_device.h"

namespace hs {

constexpr float kGeluP0 = -1.000002567e+00f, kGeluP1 = -1.151043788e+00f, kGeluP2 = -4.594249196e-01f, kGeluP3 = -5.234669296e-02f,
                kGeluP4 = 7.289804275e-03f, kGeluP5 = -5.021511049e-04f;
constexpr float kGeluE2 = -0.7213475204444817f;  // -log2(e) / 2
constexpr float kGeluE0 = -1.3257480647361595f;  // -log2(sqrt(2 pi))

__device__ __forceinline__ float gelu_tail(float a) {  // Phi(-a), a >= 0
    float p = fmaf(kGeluP5, a, kGeluP4);
    p = fmaf(p, a, kGeluP3);
    p = fmaf(p, a, kGeluP2);
    p = fmaf(p, a, kGeluP1);
    p = fmaf(p, a, kGeluP0);
    return __builtin_amdgcn_exp2f(p);
}
__device__ __forceinline__ float gelu_f(float x) {
    const float a = fabsf(x);
    return fmaf(-a, gelu_tail(a), fmaxf(x, 0.f));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float h = gelu_tail(fabsf(x));
    const float e = __builtin_amdgcn_exp2f(fmaf(x * x, kGeluE2, kGeluE0));  // exp(-x^2/2) / sqrt(2 pi)
    return fmaf(x, e, 0.5f + copysignf(0.5f - h, x));
}

// Two elements at a time for the GEMM epilogues, which are VALU-bound (one wave per output block): <2 x float> arithmetic
// selects the packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, two lanes' worth per issue).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_tail2(f32x2 a) {
    f32x2 p = a * kGeluP5 + kGeluP4;
    p = p * a + kGeluP3;
    p = p * a + kGeluP2;
    p = p * a + kGeluP1;
    p = p * a + kGeluP0;
    return f32x2{__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
}
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {
    const f32x2 a = {fabsf(x.x), fabsf(x.y)};
    const f32x2 relu = {fmaxf(x.x, 0.f), fmaxf(x.y, 0.f)};
    return relu - a * gelu_tail2(a);
}
__device__ __forceinline__ f32x2 gelu_grad2(f32x2 x) {
    const f32x2 a = {fabsf(x.x), fabsf(x.y)};
    const f32x2 w = 0.5f - gelu_tail2(a);
    const f32x2 s = (x * x) * kGeluE2 + kGeluE0;
    const f32x2 e = {__builtin_amdgcn_exp2f(s.x), __builtin_amdgcn_exp2f(s.y)};
    const f32x2 cdf = f32x2{copysignf(w.x, x.x), copysignf(w.y, x.y)} + 0.5f;
    return x * e + cdf;
}

}  // namespace hs
