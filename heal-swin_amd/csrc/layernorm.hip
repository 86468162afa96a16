// Row LayerNorm (eps 1e-5, affine) over contiguous rows, optional fused residual add; forward and backward.
// HBM-bound: one wavefront per row, 16-byte vector loads along the row, the row cached in registers
// between the statistics pass and the normalise pass (each element is read once and written once).
// Statistics and accumulations are fp32 for both fp32 and bf16 activations.
#include <type_traits>

#include "hs_device.h"

namespace hs {
namespace {

constexpr float kLnEps = 1e-5f;  // torch.nn.LayerNorm default, used by every norm in the reference
constexpr int kBwdMaxBlocks = 512;

template <typename T, int VEC>
struct vec_io;
template <>
struct vec_io<float, 4> {
    static __device__ __forceinline__ void load(const void* p, int64_t i, float* v) {
        const float4 t = *(const float4*)((const float*)p + i);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(void* p, int64_t i, const float* v) {
        *(float4*)((float*)p + i) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <>
struct vec_io<bf16_t, 8> {
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
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (uint32_t)float_to_bf16(v[2 * k]) | ((uint32_t)float_to_bf16(v[2 * k + 1]) << 16);
        *(uint4*)((uint16_t*)p + i) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <typename T>
struct vec_io<T, 1> {
    static __device__ __forceinline__ void load(const void* p, int64_t i, float* v) { v[0] = io<T>::load(p, i); }
    static __device__ __forceinline__ void store(void* p, int64_t i, const float* v) { io<T>::store(p, i, v[0]); }
};

// lane `lane` owns the VEC-wide chunks lane, lane+64, ... of the row
template <typename T, int VEC, int ITERS>
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(const void* __restrict__ x, const void* __restrict__ residual,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            void* __restrict__ y, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, int64_t rows, int width) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int nchunk = width / VEC;
    for (int64_t row = wave; row < rows; row += nwaves) {
        const int64_t base = row * width;
        float v[ITERS][VEC];
        float sum = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = lane + 64 * it;
            if (c < nchunk) {
                vec_io<T, VEC>::load(x, base + (int64_t)c * VEC, v[it]);
#pragma unroll
                for (int k = 0; k < VEC; ++k) sum += v[it][k];
            }
        }
        const float mean = wave_sum(sum) / (float)width;
        float sq = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = lane + 64 * it;
            if (c < nchunk) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const float d = v[it][k] - mean;
                    sq = fmaf(d, d, sq);
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) / (float)width + kLnEps);
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = lane + 64 * it;
            if (c < nchunk) {
                float g[VEC], b[VEC], r[VEC], o[VEC];
                vec_io<float, VEC == 8 ? 4 : VEC>::load(gamma, (int64_t)c * VEC, g);
                vec_io<float, VEC == 8 ? 4 : VEC>::load(beta, (int64_t)c * VEC, b);
                if constexpr (VEC == 8) {
                    vec_io<float, 4>::load(gamma, (int64_t)c * VEC + 4, g + 4);
                    vec_io<float, 4>::load(beta, (int64_t)c * VEC + 4, b + 4);
                }
                if (residual) vec_io<T, VEC>::load(residual, base + (int64_t)c * VEC, r);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    o[k] = fmaf((v[it][k] - mean) * rstd, g[k], b[k]);
                    if (residual) o[k] += r[k];
                }
                vec_io<T, VEC>::store(y, base + (int64_t)c * VEC, o);
            }
        }
        if (lane == 0 && mean_out) {
            mean_out[row] = mean;
            rstd_out[row] = rstd;
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma.  Each workgroup also accumulates
// sum(dy * xhat) and sum(dy) over its rows and writes one partial row pair to `partials`
// ([gridDim.x][2][width]); layernorm_param_reduce_kernel sums them.
template <typename T, int VEC, int ITERS>
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, void* __restrict__ dx,
                                                            float* __restrict__ partials, int64_t rows, int width) {
    extern __shared__ __attribute__((aligned(16))) float red[];  // [3 waves][2][width]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * nw + wid;
    const int64_t nwaves = (int64_t)gridDim.x * nw;
    const int nchunk = width / VEC;
    float dg[ITERS][VEC], db[ITERS][VEC], gm[ITERS][VEC];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = lane + 64 * it;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            dg[it][k] = 0.f;
            db[it][k] = 0.f;
            gm[it][k] = (c < nchunk) ? gamma[c * VEC + k] : 0.f;
        }
    }
    for (int64_t row = wave; row < rows; row += nwaves) {
        const int64_t base = row * width;
        const float mean = mean_in[row], rstd = rstd_in[row];
        float xh[ITERS][VEC], g[ITERS][VEC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = lane + 64 * it;
            if (c < nchunk) {
                float dyv[VEC];
                vec_io<T, VEC>::load(x, base + (int64_t)c * VEC, xh[it]);
                vec_io<T, VEC>::load(dy, base + (int64_t)c * VEC, dyv);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    xh[it][k] = (xh[it][k] - mean) * rstd;
                    g[it][k] = dyv[k] * gm[it][k];
                    s1 += g[it][k];
                    s2 = fmaf(g[it][k], xh[it][k], s2);
                    dg[it][k] = fmaf(dyv[k], xh[it][k], dg[it][k]);
                    db[it][k] += dyv[k];
                }
            }
        }
        const float m1 = wave_sum(s1) / (float)width, m2 = wave_sum(s2) / (float)width;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = lane + 64 * it;
            if (c < nchunk) {
                float o[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) o[k] = rstd * (g[it][k] - m1 - xh[it][k] * m2);
                vec_io<T, VEC>::store(dx, base + (int64_t)c * VEC, o);
            }
        }
    }
    // combine the workgroup's waves: waves 1.. park their sums in LDS, wave 0 adds and writes the partial
    if (wid > 0) {
        float* mine = red + (size_t)(wid - 1) * 2 * width;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = lane + 64 * it;
            if (c < nchunk)
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    mine[c * VEC + k] = dg[it][k];
                    mine[width + c * VEC + k] = db[it][k];
                }
        }
    }
    __syncthreads();
    if (wid == 0) {
        float* outp = partials + (size_t)blockIdx.x * 2 * width;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = lane + 64 * it;
            if (c < nchunk)
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float a = dg[it][k], b = db[it][k];
                    for (int w = 0; w < nw - 1; ++w) {
                        a += red[(size_t)w * 2 * width + c * VEC + k];
                        b += red[(size_t)w * 2 * width + width + c * VEC + k];
                    }
                    outp[c * VEC + k] = a;
                    outp[width + c * VEC + k] = b;
                }
        }
    }
}

__global__ void layernorm_param_reduce_kernel(const float* __restrict__ partials, float* __restrict__ dgamma,
                                              float* __restrict__ dbeta, int nblocks, int width) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;  // over 2*width
    if (col >= 2 * width) return;
    float acc = 0.f;
    for (int b = 0; b < nblocks; ++b) acc += partials[(size_t)b * 2 * width + col];
    if (col < width) dgamma[col] = acc;
    else dbeta[col - width] = acc;
}

int bwd_blocks(int64_t rows) {
    const int64_t want = (rows + 3) / 4;
    return (int)(want < kBwdMaxBlocks ? (want < 1 ? 1 : want) : kBwdMaxBlocks);
}

template <typename T, int VEC, int ITERS>
int run_fwd(const void* x, const void* res, const float* g, const float* b, void* y, float* mean, float* rstd, int64_t rows,
            int width, hipStream_t s) {
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((layernorm_fwd_kernel<T, VEC, ITERS>), dim3((unsigned)blocks), dim3(256), 0, s, x, res, g, b, y, mean,
                       rstd, rows, width);
    HS_LAUNCH_CHECK("layernorm_fwd");
    return HS_OK;
}

template <typename T, int VEC, int ITERS>
int run_bwd(const void* dy, const void* x, const float* g, const float* mean, const float* rstd, void* dx, float* dgamma,
            float* dbeta, float* ws, int64_t rows, int width, hipStream_t s) {
    const int blocks = bwd_blocks(rows);
    const size_t smem = (size_t)3 * 2 * width * sizeof(float);
    auto kern = layernorm_bwd_kernel<T, VEC, ITERS>;
    if (smem > 48 * 1024) HS_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), smem, s, dy, x, g, mean, rstd, dx, ws, rows, width);
    HS_LAUNCH_CHECK("layernorm_bwd");
    hipLaunchKernelGGL(layernorm_param_reduce_kernel, dim3((2 * width + 255) / 256), dim3(256), 0, s, ws, dgamma, dbeta, blocks,
                       width);
    HS_LAUNCH_CHECK("layernorm_param_reduce");
    return HS_OK;
}

// picks the vector width and the per-lane register tile for a row width
template <typename T, int VEC, typename F>
int with_iters(int width, F&& f) {
    const int chunks = width / VEC;
    const int iters = (chunks + 63) / 64;
    if (iters <= 1) return f(std::integral_constant<int, 1>{});
    if (iters <= 2) return f(std::integral_constant<int, 2>{});
    if (iters <= 4) return f(std::integral_constant<int, 4>{});
    if (iters <= 8) return f(std::integral_constant<int, 8>{});
    if (iters <= 16 && VEC == 1) return f(std::integral_constant<int, 16>{});
    return fail(HS_ERR_UNSUPPORTED, "layernorm width %d too large", width);
}

}  // namespace
}  // namespace hs

extern "C" {

int hs_layernorm_fwd(const void* x, const void* residual, const float* gamma, const float* beta, void* y, float* mean,
                     float* rstd, int64_t rows, int width, int dtype, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(x && gamma && beta && y, "null pointer");
    HS_CHECK_ARG((mean == nullptr) == (rstd == nullptr), "mean and rstd must both be given or both be null");
    HS_CHECK_ARG(rows >= 0 && width > 0, "bad shape");
    HS_CHECK_ARG(dtype == HS_F32 || dtype == HS_BF16, "dtype must be HS_F32 or HS_BF16");
    if (rows == 0) return HS_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == HS_BF16) {
        if (width % 8 == 0)
            return with_iters<bf16_t, 8>(width, [&](auto it) { return run_fwd<bf16_t, 8, decltype(it)::value>(x, residual, gamma, beta, y, mean, rstd, rows, width, s); });
        return with_iters<bf16_t, 1>(width, [&](auto it) { return run_fwd<bf16_t, 1, decltype(it)::value>(x, residual, gamma, beta, y, mean, rstd, rows, width, s); });
    }
    if (width % 4 == 0)
        return with_iters<float, 4>(width, [&](auto it) { return run_fwd<float, 4, decltype(it)::value>(x, residual, gamma, beta, y, mean, rstd, rows, width, s); });
    return with_iters<float, 1>(width, [&](auto it) { return run_fwd<float, 1, decltype(it)::value>(x, residual, gamma, beta, y, mean, rstd, rows, width, s); });
}

int64_t hs_layernorm_bwd_workspace(int64_t rows, int width) { return (int64_t)hs::bwd_blocks(rows) * 2 * width; }

int hs_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                     float* dgamma, float* dbeta, float* workspace, int64_t rows, int width, int dtype, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && workspace, "null pointer");
    HS_CHECK_ARG(rows > 0 && width > 0, "bad shape");
    HS_CHECK_ARG(dtype == HS_F32 || dtype == HS_BF16, "dtype must be HS_F32 or HS_BF16");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == HS_BF16) {
        if (width % 8 == 0)
            return with_iters<bf16_t, 8>(width, [&](auto it) { return run_bwd<bf16_t, 8, decltype(it)::value>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, width, s); });
        return with_iters<bf16_t, 1>(width, [&](auto it) { return run_bwd<bf16_t, 1, decltype(it)::value>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, width, s); });
    }
    if (width % 4 == 0)
        return with_iters<float, 4>(width, [&](auto it) { return run_bwd<float, 4, decltype(it)::value>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, width, s); });
    return with_iters<float, 1>(width, [&](auto it) { return run_bwd<float, 1, decltype(it)::value>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, width, s); });
}

}  // extern "C"
