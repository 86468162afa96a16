// y = dropout(GELU(x)) and its backward, elementwise over the MLP hidden tensor (exact erf GELU: nn.GELU default, reference
// Mlp.forward, models_torch/swin_hp_transformer.py:39-41: fc1 -> act -> drop).  HBM-bound: one read + one write forward,
// two reads + one write backward, 16-byte vectors; the dropout mask (train mode, drop_rate > 0) is a pure function of
// (seed, element index) and is regenerated in the backward, so it costs no extra pass and no mask tensor.
#include "hs_gelu.h"

namespace hs {
HS_DEFINE_SEED_EPOCH_SETTER(set_seed_epoch_gelu)
namespace {

template <typename T>
struct vec;
template <>
struct vec<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const void* p, int64_t i, float* v) {
        const uint4 t = *(const uint4*)((const uint16_t*)p + i);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = __uint_as_float(w[k] << 16);
            v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(void* p, int64_t i, const float* v) {
        *(uint4*)((uint16_t*)p + i) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                                 pack_bf16x2(v[6], v[7]));
    }
};
template <>
struct vec<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const void* p, int64_t i, float* v) {
        const float4 t = *(const float4*)((const float*)p + i);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(void* p, int64_t i, const float* v) {
        *(float4*)((float*)p + i) = make_float4(v[0], v[1], v[2], v[3]);
    }
};

template <typename T, bool DROP>
__global__ void __launch_bounds__(256) gelu_fwd_kernel(const void* __restrict__ x, void* __restrict__ y, int64_t n, float p,
                                                       uint64_t seed) {
    constexpr int V = vec<T>::N;
    const ElemRng rng(p, seed);
    const int64_t nvec = n / V;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nvec; c += (int64_t)gridDim.x * blockDim.x) {
        float v[V];
        vec<T>::load(x, c * V, v);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            v[k] = gelu_f(v[k]);
            if (DROP) v[k] *= rng.mult(c * V + k);
        }
        vec<T>::store(y, c * V, v);
    }
    if (blockIdx.x == 0 && threadIdx.x < n - nvec * V) {  // tail (n not a multiple of the vector width)
        const int64_t i = nvec * V + threadIdx.x;
        float g = gelu_f(io<T>::load(x, i));
        if (DROP) g *= rng.mult(i);
        io<T>::store(y, i, g);
    }
}

template <typename T, bool DROP>
__global__ void __launch_bounds__(256) gelu_bwd_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                       void* __restrict__ dx, int64_t n, float p, uint64_t seed) {
    constexpr int V = vec<T>::N;
    const ElemRng rng(p, seed);
    const int64_t nvec = n / V;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nvec; c += (int64_t)gridDim.x * blockDim.x) {
        float v[V], g[V];
        vec<T>::load(x, c * V, v);
        vec<T>::load(dy, c * V, g);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float d = g[k] * gelu_grad_f(v[k]);
            if (DROP) d *= rng.mult(c * V + k);
            v[k] = d;
        }
        vec<T>::store(dx, c * V, v);
    }
    if (blockIdx.x == 0 && threadIdx.x < n - nvec * V) {
        const int64_t i = nvec * V + threadIdx.x;
        float d = io<T>::load(dy, i) * gelu_grad_f(io<T>::load(x, i));
        if (DROP) d *= rng.mult(i);
        io<T>::store(dx, i, d);
    }
}

// out = x + rs * drop(t)  (the standalone form of a block's residual branch -- end of a stage, foreign norm layers: dropout on
// the branch output, swin_hp_transformer.py:43 / :173, and DropPath, :334-338) and its backward dt = dy * rs * mask.  rs is
// the per-sample DropPath factor row_scale[i / elems_per_sample] (NULL = 1); x == NULL gives out = rs * drop(t).
template <typename T>
__global__ void __launch_bounds__(256) residual_drop_kernel(const void* __restrict__ x, const void* __restrict__ t, void* __restrict__ out,
                                                            const float* __restrict__ row_scale, int64_t elems_per_sample, int64_t n,
                                                            float p, uint64_t seed) {
    constexpr int V = vec<T>::N;
    const ElemRng rng(p, seed);
    const bool dropping = p > 0.f;
    const int64_t nvec = n / V;  // elems_per_sample is a multiple of V (checked by the host)
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nvec; c += (int64_t)gridDim.x * blockDim.x) {
        float a[V], b[V];
        vec<T>::load(t, c * V, b);
        const float rs = row_scale ? row_scale[(c * V) / elems_per_sample] : 1.f;
        if (x) vec<T>::load(x, c * V, a);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float v = b[k] * rs;
            if (dropping) v *= rng.mult(c * V + k);
            b[k] = x ? a[k] + v : v;
        }
        vec<T>::store(out, c * V, b);
    }
}

// fp32 runs with bf16 x 3 products (split3.hip): the GELU output (forward) and the hidden gradient (backward) are read by
// NOTHING but such products -- fc2 and its weight gradient; fc1's input and weight gradients -- so these forms write the
// [hi | hi | lo] bf16 operand directly instead of an fp32 tensor that a split pass would read again: 4 (+ 4) bytes in and 6 out per
// element instead of 4 (+ 4) in, 4 out, 4 in, 6 out.  x [rows, k] fp32 (forward: the pre-activation; backward: dy and it).
template <bool BWD, bool DROP>
__global__ void __launch_bounds__(256) gelu_split3_kernel(const float* __restrict__ dy, const float* __restrict__ x, uint16_t* __restrict__ out,
                                                          int64_t rows, int k, float p, uint64_t seed) {
    const ElemRng rng(p, seed);
    const int kq = k >> 2;
    const int64_t total = rows * kq;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / kq;
        const int c = (int)(i - row * kq) * 4;
        const int64_t e = row * k + c;
        const float4 xv = *(const float4*)(x + e);
        float v[4] = {xv.x, xv.y, xv.z, xv.w};
        if (BWD) {
            const float4 g = *(const float4*)(dy + e);
            v[0] = g.x * gelu_grad_f(v[0]); v[1] = g.y * gelu_grad_f(v[1]); v[2] = g.z * gelu_grad_f(v[2]); v[3] = g.w * gelu_grad_f(v[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = gelu_f(v[j]);
        }
        if (DROP) {
            float mk[4];
            rng.mult4(e, mk);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= mk[j];
        }
        const uint32_t h0 = pack_bf16x2(v[0], v[1]), h1 = pack_bf16x2(v[2], v[3]);
        const uint2 hi = make_uint2(h0, h1);
        const uint2 lo = make_uint2(pack_bf16x2(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u)),
                                    pack_bf16x2(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u)));
        uint16_t* o = out + row * 3 * (int64_t)k + c;
        *(uint2*)o = hi;
        *(uint2*)(o + k) = hi;
        *(uint2*)(o + 2 * k) = lo;
    }
}

inline unsigned grid_for(int64_t n, int v) {
    int64_t b = (n / v + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace
}  // namespace hs

extern "C" {

int hs_gelu_fwd(const void* x, void* y, int64_t n, float drop_p, uint64_t seed, int dtype, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(x && y && n >= 0, "bad arguments");
    HS_CHECK_ARG(drop_p >= 0.f && drop_p <= 1.f, "drop_p must be in [0, 1]");
    HS_CHECK_ARG(dtype == HS_F32 || dtype == HS_BF16, "dtype must be HS_F32 or HS_BF16");
    HS_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0), "buffers must be 16-byte aligned");
    if (n == 0) return HS_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool drop = drop_p > 0.f;
    if (dtype == HS_BF16) {
        if (drop) hipLaunchKernelGGL((gelu_fwd_kernel<bf16_t, true>), dim3(grid_for(n, 8)), dim3(256), 0, s, x, y, n, drop_p, seed);
        else hipLaunchKernelGGL((gelu_fwd_kernel<bf16_t, false>), dim3(grid_for(n, 8)), dim3(256), 0, s, x, y, n, drop_p, seed);
    } else {
        if (drop) hipLaunchKernelGGL((gelu_fwd_kernel<float, true>), dim3(grid_for(n, 4)), dim3(256), 0, s, x, y, n, drop_p, seed);
        else hipLaunchKernelGGL((gelu_fwd_kernel<float, false>), dim3(grid_for(n, 4)), dim3(256), 0, s, x, y, n, drop_p, seed);
    }
    HS_LAUNCH_CHECK("gelu_fwd");
    return HS_OK;
}

int hs_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, float drop_p, uint64_t seed, int dtype, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(dy && x && dx && n >= 0, "bad arguments");
    HS_CHECK_ARG(drop_p >= 0.f && drop_p <= 1.f, "drop_p must be in [0, 1]");
    HS_CHECK_ARG(dtype == HS_F32 || dtype == HS_BF16, "dtype must be HS_F32 or HS_BF16");
    HS_CHECK_ARG(((uintptr_t)x % 16 == 0) && ((uintptr_t)dy % 16 == 0) && ((uintptr_t)dx % 16 == 0), "buffers must be 16-byte aligned");
    if (n == 0) return HS_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool drop = drop_p > 0.f;
    if (dtype == HS_BF16) {
        if (drop) hipLaunchKernelGGL((gelu_bwd_kernel<bf16_t, true>), dim3(grid_for(n, 8)), dim3(256), 0, s, dy, x, dx, n, drop_p, seed);
        else hipLaunchKernelGGL((gelu_bwd_kernel<bf16_t, false>), dim3(grid_for(n, 8)), dim3(256), 0, s, dy, x, dx, n, drop_p, seed);
    } else {
        if (drop) hipLaunchKernelGGL((gelu_bwd_kernel<float, true>), dim3(grid_for(n, 4)), dim3(256), 0, s, dy, x, dx, n, drop_p, seed);
        else hipLaunchKernelGGL((gelu_bwd_kernel<float, false>), dim3(grid_for(n, 4)), dim3(256), 0, s, dy, x, dx, n, drop_p, seed);
    }
    HS_LAUNCH_CHECK("gelu_bwd");
    return HS_OK;
}

int hs_gelu_split3(const float* dy, const float* x, void* out3, int64_t rows, int k, float drop_p, uint64_t seed, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(x && out3 && rows > 0 && k > 0 && k % 4 == 0, "hs_gelu_split3: null pointer, or k not a positive multiple of 4");
    HS_CHECK_ARG(drop_p >= 0.f && drop_p <= 1.f, "drop_p must be in [0, 1]");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(grid_for(rows * k, 4)), block(256);
    const bool drop = drop_p > 0.f;
    uint16_t* o = (uint16_t*)out3;
    if (dy) {
        if (drop) hipLaunchKernelGGL((gelu_split3_kernel<true, true>), grid, block, 0, s, dy, x, o, rows, k, drop_p, seed);
        else hipLaunchKernelGGL((gelu_split3_kernel<true, false>), grid, block, 0, s, dy, x, o, rows, k, drop_p, seed);
    } else {
        if (drop) hipLaunchKernelGGL((gelu_split3_kernel<false, true>), grid, block, 0, s, dy, x, o, rows, k, drop_p, seed);
        else hipLaunchKernelGGL((gelu_split3_kernel<false, false>), grid, block, 0, s, dy, x, o, rows, k, drop_p, seed);
    }
    HS_LAUNCH_CHECK("gelu_split3");
    return HS_OK;
}

int hs_residual_drop(const void* x, const void* t, void* out, const float* row_scale, int64_t elems_per_sample, int64_t n,
                     float drop_p, uint64_t seed, int dtype, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(t && out && n >= 0, "bad arguments");
    HS_CHECK_ARG(drop_p >= 0.f && drop_p <= 1.f, "drop_p must be in [0, 1]");
    HS_CHECK_ARG(dtype == HS_F32 || dtype == HS_BF16, "dtype must be HS_F32 or HS_BF16");
    const int v = dtype == HS_BF16 ? 8 : 4;
    HS_CHECK_ARG(n % v == 0 && (!row_scale || (elems_per_sample > 0 && elems_per_sample % v == 0 && n % elems_per_sample == 0)),
                 "n and elems_per_sample must be multiples of the 16-byte vector width");
    HS_CHECK_ARG(((uintptr_t)t % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)x % 16 == 0), "buffers must be 16-byte aligned");
    if (n == 0) return HS_OK;
    if (dtype == HS_BF16)
        hipLaunchKernelGGL(residual_drop_kernel<bf16_t>, dim3(grid_for(n, 8)), dim3(256), 0, (hipStream_t)stream, x, t, out, row_scale,
                           elems_per_sample, n, drop_p, seed);
    else
        hipLaunchKernelGGL(residual_drop_kernel<float>, dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, x, t, out, row_scale,
                           elems_per_sample, n, drop_p, seed);
    HS_LAUNCH_CHECK("residual_drop");
    return HS_OK;
}

}  // extern "C"
