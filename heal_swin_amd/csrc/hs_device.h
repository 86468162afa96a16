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
// Seeds under HIP-graph replay.  Every stochastic kernel takes its seed by value, so a captured launch would repeat its mask on every
// replay.  hs_set_seed_epoch(ptr) registers a device-resident 64-bit counter: when set, every mask generator adds counter * odd
// constant to its seed at kernel start (two scalar loads), and the owner of the graph advances the counter once per replayed step
// (forward and backward of a step see the same value).  The pointer is a static __device__ variable of each translation unit
// (no relocatable device code in this build), written by the unit's set_seed_epoch_* function.
static __device__ const unsigned long long* g_seed_epoch = nullptr;
__device__ __forceinline__ uint64_t epoch_seed(uint64_t seed) {
    const unsigned long long* e = g_seed_epoch;
    return e ? seed + (uint64_t)(*e) * 0x9E3779B97F4A7C15ull : seed;
}
#define HS_DEFINE_SEED_EPOCH_SETTER(name)                                                                              \
    int name(const void* counter) {                                                                                    \
        return hipMemcpyToSymbol(HIP_SYMBOL(g_seed_epoch), &counter, sizeof(counter)) == hipSuccess ? 0 : 3; \
    }

struct ElemRng {
    uint32_t key_lo, key_hi, thresh16;
    float keep_scale;
    __device__ __forceinline__ ElemRng(float p, uint64_t seed) {
        seed = epoch_seed(seed);
        key_lo = (uint32_t)seed;
        key_hi = mix32((uint32_t)(seed >> 32) + 0x85EBCA6Bu);  // pre-mixed once per thread
        thresh16 = p >= 1.f ? 65536u : (uint32_t)(p * 65536.f);  // drop probability in steps of 2^-16
        keep_scale = p >= 1.f ? 0.f : 1.f / (1.f - p);
    }
    // The mask of element i (16 random bits, dropped when they are below thresh16) is defined in two levels, so that the integer
    // multiplies -- quarter rate on the VALU, and what a stochastic kernel's time goes into -- are shared by 8 elements:
    //   chunk c = i >> 3 : key ck(c) = two full finaliser rounds over the chunk number, one key word entering BETWEEN them (masks of
    //                      different seeds are related by a pseudo-random index map, not by an index (XOR-)translation);
    //   pair p = (i >> 1) & 3 of the chunk: u = ck * M_p (four fixed odd multipliers: (u_p, u_q) is the lattice of an LCG with
    //                      multiplier M_q / M_p, uniform in both coordinates), element 2j takes lo16(u) ^ hi16(u), 2j + 1 hi16(u).
    // 1.5 multiplies per element instead of 4.  mult(i) is the definition; mult8 / mult4 are what kernels with aligned element
    // runs call (same bits).
    static constexpr uint32_t kM0 = 0x9E3779B1u, kM1 = 0x85EBCA6Bu, kM2 = 0xC2B2AE35u, kM3 = 0x27D4EB2Fu;
    __device__ __forceinline__ uint32_t chunk_key(uint64_t c) const {
        return mix32(mix32((uint32_t)c ^ key_hi ^ ((uint32_t)(c >> 32) * 0x9E3779B9u)) ^ key_lo);
    }
    static __device__ __forceinline__ uint32_t pair_bits(uint32_t ck, uint32_t m) {
        const uint32_t u = ck * m;
        return u ^ (u >> 16);
    }
    __device__ __forceinline__ float keep_lo(uint32_t h) const { return (h & 0xffffu) >= thresh16 ? keep_scale : 0.f; }
    __device__ __forceinline__ float keep_hi(uint32_t h) const { return (h >> 16) >= thresh16 ? keep_scale : 0.f; }
    // multiplier of element i: 0 (dropped) or 1/(1-p)
    __device__ __forceinline__ float mult(int64_t i) const {
        const uint32_t ck = chunk_key((uint64_t)i >> 3);
        const uint32_t pr = ((uint32_t)i >> 1) & 3u;
        const uint32_t h = pair_bits(ck, pr == 0 ? kM0 : pr == 1 ? kM1 : pr == 2 ? kM2 : kM3);
        return (i & 1) ? keep_hi(h) : keep_lo(h);
    }
    // elements 8 c ... 8 c + 7
    __device__ __forceinline__ void mult8(uint64_t c, float (&out)[8]) const {
        const uint32_t ck = chunk_key(c);
        const uint32_t h0 = pair_bits(ck, kM0), h1 = pair_bits(ck, kM1), h2 = pair_bits(ck, kM2), h3 = pair_bits(ck, kM3);
        out[0] = keep_lo(h0), out[1] = keep_hi(h0), out[2] = keep_lo(h1), out[3] = keep_hi(h1);
        out[4] = keep_lo(h2), out[5] = keep_hi(h2), out[6] = keep_lo(h3), out[7] = keep_hi(h3);
    }
    // In-order form for an aligned run of V elements starting at e0 (a multiple of V): ck = run_key<V>(e0) once, then
    // run_mult<V>(e0, k, ck, h) for k = 0 ... V - 1 in order inside the kernel's own unrolled element loop (k folds to a constant; h
    // carries a pair's bits from its even to its odd element) -- two live registers instead of V multipliers.
    template <int V>
    __device__ __forceinline__ uint32_t run_key(int64_t e0) const {
        return (V == 8 || V == 4) ? chunk_key((uint64_t)e0 >> 3) : 0u;
    }
    template <int V>
    __device__ __forceinline__ float run_mult(int64_t e0, int k, uint32_t ck, uint32_t& h) const {
        if constexpr (V == 8 || V == 4) {
            if ((k & 1) == 0) {
                const int pr = k >> 1;
                uint32_t m;
                if constexpr (V == 8) {
                    m = pr == 0 ? kM0 : pr == 1 ? kM1 : pr == 2 ? kM2 : kM3;
                } else {
                    const bool second = ((uint32_t)e0 >> 2) & 1u;
                    m = pr == 0 ? (second ? kM2 : kM0) : (second ? kM3 : kM1);
                }
                h = pair_bits(ck, m);
                return keep_lo(h);
            }
            return keep_hi(h);
        } else {
            return mult(e0 + k);
        }
    }
    // elements i0 ... i0 + 3, i0 a multiple of 4 (the first or the second half of a chunk)
    __device__ __forceinline__ void mult4(int64_t i0, float (&out)[4]) const {
        const uint32_t ck = chunk_key((uint64_t)i0 >> 3);
        const bool second = ((uint32_t)i0 >> 2) & 1u;
        const uint32_t ha = pair_bits(ck, second ? kM2 : kM0), hb = pair_bits(ck, second ? kM3 : kM1);
        out[0] = keep_lo(ha), out[1] = keep_hi(ha), out[2] = keep_lo(hb), out[3] = keep_hi(hb);
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
