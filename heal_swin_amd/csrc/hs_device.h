// Device-side helpers shared by the HIP kernels of libhealswin (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include "hs_common.h"

namespace hs {

struct bf16_t {
    uint16_t bits;
};

__device__ __forceinline__ float bf16_to_float(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
// round-to-nearest-even (same rounding as torch's float -> bfloat16): one v_cvt_pk_bf16_f32 on gfx950 -- a software
// RNE sequence costs ~7 VALU ops per value and dominated the attention kernels' instruction count before
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint16_t float_to_bf16(float f) {
    const __bf16 h = (__bf16)f;
    return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
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

// Counter-based dropout mask shared by the elementwise kernels: the keep decision of element i is a pure function of
// (seed, i), so a backward pass regenerates the forward's mask instead of reading a mask tensor.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {  // "lowbias32" integer finaliser
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
struct ElemRng {
    uint32_t key_lo, key_hi, thresh16;
    float keep_scale;
    __device__ __forceinline__ ElemRng(float p, uint64_t seed) {
        key_lo = (uint32_t)seed;
        key_hi = mix32((uint32_t)(seed >> 32) + 0x85EBCA6Bu);  // pre-mixed once per thread
        thresh16 = p >= 1.f ? 65536u : (uint32_t)(p * 65536.f);  // drop probability in steps of 2^-16
        keep_scale = p >= 1.f ? 0.f : 1.f / (1.f - p);
    }
    // multiplier of element i: 0 (dropped) or 1/(1-p).  One hash serves the element pair (2j, 2j+1), 16 bits each: the kernels
    // process 4 or 8 consecutive elements per lane, so half of the hashes are shared.
    __device__ __forceinline__ float mult(int64_t i) const {
        const uint64_t j = (uint64_t)i >> 1;
        // keyed, two full rounds with one key word entering BETWEEN them: masks of different seeds are then related by a
        // pseudo-random index map, not by an index translation (j + key) or XOR-translation (j ^ key) as with a single
        // keyed round
        const uint32_t h = mix32(mix32((uint32_t)j ^ key_hi ^ ((uint32_t)(j >> 32) * 0x9E3779B9u)) ^ key_lo);
        const uint32_t bits = (i & 1) ? (h >> 16) : (h & 0xffffu);
        return bits >= thresh16 ? keep_scale : 0.f;
    }
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

// csrc/reduce_many.hip: queue the record-wise sum of `slices` partial records (count floats each, stride in_stride; the first n_w
// go to dw, the rest to db) for the next hs_reduce_flush on stream s, instead of launching it now
int reduce_defer(const float* part, int64_t in_stride, int slices, int64_t n_w, int64_t count, float* dw, float* db, int accumulate,
                 hipStream_t s);
// ... or launch it at once (same kernel, same summation order)
int reduce_now(const float* part, int64_t in_stride, int slices, int64_t n_w, int64_t count, float* dw, float* db, int accumulate,
               hipStream_t s);

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
