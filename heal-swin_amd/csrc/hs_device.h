// Device-side helpers shared by the HIP kernels of libhealswin (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include "hs_common.h"

namespace hs {

struct bf16_t {
    uint16_t bits;
};

__device__ __forceinline__ float bf16_to_float(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
// round-to-nearest-even, NaN preserved (same rounding as torch's float -> bfloat16)
__device__ __forceinline__ uint16_t float_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

template <typename T>
struct io;
template <>
struct io<float> {
    static __device__ __forceinline__ float load(const void* p, int64_t i) { return ((const float*)p)[i]; }
    static __device__ __forceinline__ void store(void* p, int64_t i, float v) { ((float*)p)[i] = v; }
};
template <>
struct io<bf16_t> {
    static __device__ __forceinline__ float load(const void* p, int64_t i) { return bf16_to_float(((const uint16_t*)p)[i]); }
    static __device__ __forceinline__ void store(void* p, int64_t i, float v) { ((uint16_t*)p)[i] = float_to_bf16(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

inline int hip_fail(hipError_t e, const char* what) {
    (void)hipGetLastError();
    return fail(HS_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

}  // namespace hs

#define HS_HIP_CHECK(expr)                                   \
    do {                                                     \
        hipError_t e__ = (expr);                             \
        if (e__ != hipSuccess) return hs::hip_fail(e__, #expr); \
    } while (0)

#define HS_LAUNCH_CHECK(name)                                      \
    do {                                                           \
        hipError_t e__ = hipGetLastError();                        \
        if (e__ != hipSuccess) return hs::hip_fail(e__, "launch " name); \
    } while (0)
