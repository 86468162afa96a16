// Row LayerNorm (eps 1e-5, affine) over contiguous rows, optional fused residual add; forward and backward.
// HBM-bound: one wavefront per row, 16-byte vector loads along the row, the row cached in registers
// between the statistics pass and the normalise pass (each element is read once and written once).
// Statistics and accumulations are fp32 for both fp32 and bf16 activations.
#include <cstdlib>
#include <type_traits>

#include "hs_device.h"

namespace hs {
HS_DEFINE_SEED_EPOCH_SETTER(set_seed_epoch_layernorm)

// Sample (image) of a row for the per-sample DropPath factor: row / rows_per_sample without the 64-bit division (~150 instructions
// per row and lane -- more than the rest of a stochastic LayerNorm row costs).  Exact: below 2^24 rows the float quotient is off by
// at most one, which the two comparisons repair; larger tensors keep the division.
__device__ __forceinline__ int64_t sample_of(int64_t row, int64_t rows_per_sample) {
    if (row >= (1 << 24)) return row / rows_per_sample;
    const uint32_t r = (uint32_t)row, d = (uint32_t)rows_per_sample;
    uint32_t q = (uint32_t)((float)r * __builtin_amdgcn_rcpf((float)d));
    if (q * d > r) --q;
    if ((q + 1) * d <= r) ++q;
    return (int64_t)q;
}
namespace {

constexpr float kLnEps = 1e-5f;  // torch.nn.LayerNorm default, used by every norm in the reference
constexpr int kBwdMaxBlocks = 2048;

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
    static __device__ __forceinline__ void decode(const uint4& t, float* v) {
        v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y); v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
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
        for (int k = 0; k < 4; ++k) w[k] = pack_bf16x2(v[2 * k], v[2 * k + 1]);
        *(uint4*)((uint16_t*)p + i) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    static __device__ __forceinline__ void decode(const uint4& t, float* v) {
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = __uint_as_float(w[k] << 16);
            v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
        }
    }
};
// (non-temporal stores for the residual-stream sum were measured here: +1-3 % on the kernel, nothing on the step --
// profiles/archive_r01_r04/r03_ln_store_ab.txt; the plain store stays)
template <typename T, int VEC>
__device__ __forceinline__ void store_stream(void* p, int64_t i, const float* v) {
    vec_io<T, VEC>::store(p, i, v);
}

template <typename T>
struct vec_io<T, 1> {
    static __device__ __forceinline__ void load(const void* p, int64_t i, float* v) { v[0] = io<T>::load(p, i); }
    static __device__ __forceinline__ void store(void* p, int64_t i, const float* v) { io<T>::store(p, i, v[0]); }
    static __device__ __forceinline__ void decode(const uint4&, float*) {}
};

// A wavefront normalises 64/LPR rows at a time: LPR lanes per row (a power of two >= the row's 16-byte chunk
// count, capped at 64), lane `sub` of a row owning chunks sub, sub+LPR, ...  Narrow rows (C = 96..256 in bf16 are
// 12..32 chunks) therefore still keep all 64 lanes busy and every global access is a 16-byte vector.
template <typename T>
__device__ __forceinline__ float round_to(float v);
template <>
__device__ __forceinline__ float round_to<float>(float v) { return v; }
template <>
__device__ __forceinline__ float round_to<bf16_t>(float v) { return bf16_to_float(float_to_bf16(v)); }

template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <typename T, int VEC, int LPR, int ITERS>
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(const void* __restrict__ x, const void* __restrict__ residual,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            void* __restrict__ y, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out, int64_t rows, int width,
                                                            const void* __restrict__ add_in, void* __restrict__ sum_out,
                                                            const float* __restrict__ row_scale, int64_t rows_per_sample,
                                                            float drop_p, uint64_t seed, const void* __restrict__ lo_in,
                                                            void* __restrict__ lo_out) {
    // Stochastic extras (train mode).  With add_in (v1):  s = x + rs * drop(add_in),  y = LN(s).
    // Without (v2 / plain):                               y = [residual +] rs * LN(drop(x)).
    // rs = row_scale[sample_of(row, rows_per_sample)] is the per-sample DropPath factor, drop() the counter-based dropout mask.
    // Compensated residual stream (lo_in / lo_out, optional, activation dtype): the residual stream of a stage is the sum of
    // up to 36 branch outputs; stored in bf16 every add rounds it (2^-9 relative), which is a third of the bf16 logit error of
    // HEAL-SWIN-B (tests/experiments/bf16_error_budget.py).  With lo_out the stream operand is hi + lo: the new sum is
    // formed in fp32, `hi` = its rounding goes to sum_out / y as before and `lo` = the rounding remainder to lo_out; the
    // next add reads both, so the stream carries 16 mantissa bits at 4 bytes per element while every other consumer (GEMMs,
    // the backward) keeps reading the plain bf16 `hi` tensor.
    const bool dropping = drop_p > 0.f;
    const ElemRng rng(drop_p, seed);
    constexpr int RPW = 64 / LPR;  // rows per wave
    const int lane = threadIdx.x & 63, sub = lane % LPR, rsub = lane / LPR;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int nchunk = width / VEC;
    const float inv_w = 1.f / (float)width;
    // Software prefetch (16-byte chunks, one chunk per lane: rows up to 1024 bytes): the x (and add_in) chunks of the wave's NEXT rows are
    // requested before the current rows are reduced.  One row group per wave kept only 1 KB per wave (32 KB per CU) in
    // flight: the plain norm streamed at 3.0 TB/s where the fused add + norm, with two loads per lane, reached 4.4
    // (tools/bench_ln.py on 400 MB tensors).
    constexpr bool PF = (VEC * (int)sizeof(T) == 16) && ITERS == 1;  // (ITERS == 2 would drop the kernel from 4 to 3 waves per SIMD)
    uint4 px[ITERS], pa[ITERS];
    auto fetch = [&](int64_t r0) {
        const int64_t row_l = r0 + rsub;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = sub + LPR * it;
            if (row_l < rows && c < nchunk) {
                const int64_t off = (row_l * width + (int64_t)c * VEC) * (int64_t)sizeof(T);
                px[it] = *(const uint4*)((const char*)x + off);
                if (add_in) pa[it] = *(const uint4*)((const char*)add_in + off);
            }
        }
    };
    if constexpr (PF) {
        if (wave * RPW < rows) fetch(wave * RPW);
    }
    for (int64_t row0 = wave * RPW; row0 < rows; row0 += nwaves * RPW) {
        const int64_t row = row0 + rsub;
        const bool live = row < rows;
        const int64_t base = row * width;
        const float rs = (row_scale && live) ? row_scale[sample_of(row, rows_per_sample)] : 1.f;
        float v[ITERS][VEC];
        uint4 cx[ITERS], ca[ITERS];
        if constexpr (PF) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                cx[it] = px[it];
                ca[it] = pa[it];
            }
            if (row0 + nwaves * RPW < rows) fetch(row0 + nwaves * RPW);
        }
        float sum = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = sub + LPR * it;
            if (live && c < nchunk) {
                if constexpr (PF) vec_io<T, VEC>::decode(cx[it], v[it]);
                else vec_io<T, VEC>::load(x, base + (int64_t)c * VEC, v[it]);
                if (add_in) {  // s = x + rs*drop(add_in), rounded to the activation dtype exactly as a separate add would store it
                    float a2[VEC], l2[VEC], hi[VEC];
                    if constexpr (PF) vec_io<T, VEC>::decode(ca[it], a2);
                    else vec_io<T, VEC>::load(add_in, base + (int64_t)c * VEC, a2);
                    if (lo_in) vec_io<T, VEC>::load(lo_in, base + (int64_t)c * VEC, l2);
                    const int64_t e0 = base + (int64_t)c * VEC;
                    uint32_t hp = 0;
                    const uint32_t ck = dropping ? rng.template run_key<VEC>(e0) : 0u;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        float add = a2[k] * rs;
                        if (dropping) add *= rng.template run_mult<VEC>(e0, k, ck, hp);
                        const float full = v[it][k] + (lo_in ? l2[k] : 0.f) + add;
                        hi[k] = round_to<T>(full);  // the stream as every other consumer sees it
                        l2[k] = full - hi[k];
                        // without compensation LN sees the stored (rounded) sum, exactly as after a separate add; with it, the
                        // un-rounded one (the backward re-normalises the stored tensor: a 2^-9 relative difference in xhat)
                        v[it][k] = lo_out ? full : hi[k];
                    }
                    store_stream<T, VEC>(sum_out, base + (int64_t)c * VEC, hi);
                    if (lo_out) vec_io<T, VEC>::store(lo_out, base + (int64_t)c * VEC, l2);
                } else if (dropping) {
                    const int64_t e0 = base + (int64_t)c * VEC;
                    uint32_t hp = 0;
                    const uint32_t ck = rng.template run_key<VEC>(e0);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) v[it][k] *= rng.template run_mult<VEC>(e0, k, ck, hp);
                }
#pragma unroll
                for (int k = 0; k < VEC; ++k) sum += v[it][k];
            }
        }
        const float mean = row_sum<LPR>(sum) * inv_w;
        float sq = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = sub + LPR * it;
            if (live && c < nchunk) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const float d = v[it][k] - mean;
                    sq = fmaf(d, d, sq);
                }
            }
        }
        const float rstd = rsqrtf(row_sum<LPR>(sq) * inv_w + kLnEps);
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = sub + LPR * it;
            if (live && c < nchunk) {
                float g[VEC], b[VEC], r[VEC], o[VEC];
                vec_io<float, VEC == 8 ? 4 : VEC>::load(gamma, (int64_t)c * VEC, g);
                vec_io<float, VEC == 8 ? 4 : VEC>::load(beta, (int64_t)c * VEC, b);
                if constexpr (VEC == 8) {
                    vec_io<float, 4>::load(gamma, (int64_t)c * VEC + 4, g + 4);
                    vec_io<float, 4>::load(beta, (int64_t)c * VEC + 4, b + 4);
                }
                if (residual) vec_io<T, VEC>::load(residual, base + (int64_t)c * VEC, r);
                if (residual && lo_in && !add_in) {  // v2 placement: the residual operand is the compensated stream
                    float rl[VEC];
                    vec_io<T, VEC>::load(lo_in, base + (int64_t)c * VEC, rl);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) r[k] += rl[k];
                }
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    o[k] = fmaf((v[it][k] - mean) * rstd, g[k], b[k]);
                    if (!add_in) o[k] *= rs;  // v2: DropPath scales the normalised branch
                    if (residual) o[k] += r[k];
                }
                if (lo_out && !add_in) {  // v2 stream (y = residual + LN) or a plain LN whose consumer wants hi + lo: y = hi, remainder to lo_out
                    float ol[VEC];
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        const float hi = round_to<T>(o[k]);
                        ol[k] = o[k] - hi;
                        o[k] = hi;
                    }
                    vec_io<T, VEC>::store(lo_out, base + (int64_t)c * VEC, ol);
                }
                vec_io<T, VEC>::store(y, base + (int64_t)c * VEC, o);
            }
        }
        if (live && sub == 0 && mean_out) {
            mean_out[row] = mean;
            rstd_out[row] = rstd;
        }
    }
}

// element k of a packed 16-byte chunk, and the chunk of VEC fp32 values rounded to the activation dtype
template <typename T>
__device__ __forceinline__ float chunk_elem(const uint4& w, int k);
template <>
__device__ __forceinline__ float chunk_elem<bf16_t>(const uint4& w, int k) {
    const uint32_t d = (k >> 1) == 0 ? w.x : (k >> 1) == 1 ? w.y : (k >> 1) == 2 ? w.z : w.w;
    return (k & 1) ? __uint_as_float(d & 0xffff0000u) : __uint_as_float(d << 16);
}
template <>
__device__ __forceinline__ float chunk_elem<float>(const uint4& w, int k) {
    return __uint_as_float(k == 0 ? w.x : k == 1 ? w.y : k == 2 ? w.z : w.w);
}
template <typename T>
__device__ __forceinline__ uint4 chunk_pack(const float* v);
template <>
__device__ __forceinline__ uint4 chunk_pack<bf16_t>(const float* v) {
    return make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}
template <>
__device__ __forceinline__ uint4 chunk_pack<float>(const float* v) {
    return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
}

// The training default of the forward, specialised like layernorm_bwd_fast_kernel below (round 4): rows in 16-byte chunks (bf16 or fp32),
// y = [residual +] LN(x) or (sum_out = x + add_in, y = LN(sum_out)); no dropout, DropPath scale or compensated stream.  Next
// rows' chunks requested packed before the current rows are reduced, 32-bit byte offsets from uniform bases, launch sized to
// one resident round.
// EX (round 5): the train-mode extras of the v2 / plain form -- y = [residual +] rs * LN(drop(x)), the paper's drop rates -- on the
// same kernel (the general kernel carried them at half the rate: 202 vs ~100 us at T @ 256 stage 0).  Same mask as everywhere
// (ElemRng keyed by the element index), so the general and the specialised kernels are interchangeable per call.
template <typename T, int LPR, int ITERS, bool EX = false>
__global__ void __launch_bounds__(256, (ITERS == 1 ? 6 : ITERS == 2 ? 5 : ITERS == 4 ? 3 : 1)) layernorm_fwd_fast_kernel(
    const void* __restrict__ x, const void* __restrict__ add_in, const float* __restrict__ gamma, const float* __restrict__ beta,
    void* __restrict__ y, void* __restrict__ sum_out, float* __restrict__ mean_out, float* __restrict__ rstd_out, int64_t rows,
    int width, const void* __restrict__ residual, const float* __restrict__ row_scale = nullptr, int64_t rows_per_sample = 1,
    float drop_p = 0.f, uint64_t seed = 0) {
    const ElemRng rng(drop_p, seed);
    const bool dropping = EX && drop_p > 0.f;
    // (add_in and residual are exclusive: the second packed stream `pa` carries whichever is given)
    constexpr int RPW = 64 / LPR, VEC = 16 / (int)sizeof(T), ES = (int)sizeof(T);
    constexpr bool PF = ITERS <= 2;
    const int lane = threadIdx.x & 63, sub = lane % LPR, rsub = lane / LPR;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 6) * RPW;
    const int nchunk = width / VEC;
    const float inv_w = 1.f / (float)width;
    const bool adding = add_in != nullptr, with_res = residual != nullptr;
    const void* second = adding ? add_in : residual;
    uint4 px[ITERS], pa[ITERS];
    auto fetch = [&](int64_t r0) {
        const int64_t row = r0 + rsub;
        if (row < rows) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int c = sub + LPR * it;
                if (c < nchunk) {
                    const uint32_t e = ((uint32_t)row * (uint32_t)width + (uint32_t)(c * VEC)) * (uint32_t)ES;
                    px[it] = *(const uint4*)((const char*)x + e);
                    if (second) pa[it] = *(const uint4*)((const char*)second + e);
                }
            }
        }
    };
    int64_t row0 = wave * RPW;
    if (PF && row0 < rows) fetch(row0);
    for (; row0 < rows; row0 += stride) {
        const int64_t row = row0 + rsub;
        const bool live = row < rows;
        if (!PF) fetch(row0);
        uint4 cx[ITERS], ca[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            cx[it] = px[it];
            ca[it] = pa[it];
        }
        if (PF && row0 + stride < rows) fetch(row0 + stride);
        float v[ITERS][VEC];
        float sum = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = sub + LPR * it;
            if (live && c < nchunk) {
                vec_io<T, VEC>::decode(cx[it], v[it]);
                if constexpr (EX) {
                    if (dropping) {
                        const int64_t e0 = row * width + (int64_t)c * VEC;
                        uint32_t hp = 0;
                        const uint32_t ck = rng.template run_key<VEC>(e0);
#pragma unroll
                        for (int k = 0; k < VEC; ++k) v[it][k] *= rng.template run_mult<VEC>(e0, k, ck, hp);
                    }
                }
                if (adding) {  // the stream as every other consumer sees it: rounded to bf16 exactly as a separate add would store it
                    float a2[VEC];
                    vec_io<T, VEC>::decode(ca[it], a2);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) v[it][k] += a2[k];
                    const uint4 w = chunk_pack<T>(v[it]);
                    vec_io<T, VEC>::decode(w, v[it]);
                    const uint32_t e = ((uint32_t)row * (uint32_t)width + (uint32_t)(c * VEC)) * (uint32_t)ES;
                    *(uint4*)((char*)sum_out + e) = w;
                }
#pragma unroll
                for (int k = 0; k < VEC; ++k) sum += v[it][k];
            }
        }
        const float mean = row_sum<LPR>(sum) * inv_w;
        float sq = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = sub + LPR * it;
            if (live && c < nchunk) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const float d = v[it][k] - mean;
                    sq = fmaf(d, d, sq);
                }
            }
        }
        const float rstd = rsqrtf(row_sum<LPR>(sq) * inv_w + kLnEps);
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = sub + LPR * it;
            if (live && c < nchunk) {
                float g[VEC], b[VEC], o[VEC];
#pragma unroll
                for (int q = 0; q < VEC / 4; ++q) {
                    const float4 g4 = *(const float4*)(gamma + c * VEC + 4 * q), b4 = *(const float4*)(beta + c * VEC + 4 * q);
                    g[4 * q] = g4.x; g[4 * q + 1] = g4.y; g[4 * q + 2] = g4.z; g[4 * q + 3] = g4.w;
                    b[4 * q] = b4.x; b[4 * q + 1] = b4.y; b[4 * q + 2] = b4.z; b[4 * q + 3] = b4.w;
                }
#pragma unroll
                for (int k = 0; k < VEC; ++k) o[k] = fmaf((v[it][k] - mean) * rstd, g[k], b[k]);
                if constexpr (EX) {  // DropPath factor of this row's sample
                    const float rs = row_scale ? row_scale[sample_of(row, rows_per_sample)] : 1.f;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) o[k] *= rs;
                }
                if (with_res) {  // v2 placement (ref :334-335): y = residual + LN(x), one rounding
#pragma unroll
                    for (int k = 0; k < VEC; ++k) o[k] += chunk_elem<T>(ca[it], k);
                }
                const uint32_t e = ((uint32_t)row * (uint32_t)width + (uint32_t)(c * VEC)) * (uint32_t)ES;
                *(uint4*)((char*)y + e) = chunk_pack<T>(o);
            }
        }
        if (live && sub == 0 && mean_out) {
            mean_out[row] = mean;
            rstd_out[row] = rstd;
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma.  Each lane also accumulates sum(dy * xhat) and
// sum(dy) for its columns; they are combined across the wave's row groups by shuffles, across the workgroup's waves
// through LDS, and written as one partial row pair per workgroup ([gridDim.x][2][width]) for the final reduce.
template <typename T, int VEC, int LPR, int ITERS>
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, void* __restrict__ dx,
                                                            float* __restrict__ partials, int64_t rows, int width,
                                                            const void* __restrict__ dres_in, void* __restrict__ dadd_out,
                                                            const float* __restrict__ row_scale, int64_t rows_per_sample,
                                                            float drop_p, uint64_t seed, int v1_mode) {
    // v1_mode (fused add + LN, `x` is the saved sum s):  g = LN_bwd(dy) + dres_in;  dx = g;  dadd_out = rs * mask * g.
    // otherwise (`x` is the raw input, LN saw u = mask * x, the output was rs * LN(u)):
    //            dy_eff = rs * dy;  dx = mask * LN_bwd(dy_eff)  (dgamma / dbeta use dy_eff).
    const bool dropping = drop_p > 0.f;
    const ElemRng rng(drop_p, seed);
    extern __shared__ __attribute__((aligned(16))) float red[];  // [3 waves][2][width]
    constexpr int RPW = 64 / LPR;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int sub = lane % LPR, rsub = lane / LPR;
    const int64_t wave = (int64_t)blockIdx.x * nw + wid;
    const int64_t nwaves = (int64_t)gridDim.x * nw;
    const int nchunk = width / VEC;
    const float inv_w = 1.f / (float)width;
    float dg[ITERS][VEC], db[ITERS][VEC], gm[ITERS][VEC];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = sub + LPR * it;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            dg[it][k] = 0.f;
            db[it][k] = 0.f;
            gm[it][k] = (c < nchunk) ? gamma[c * VEC + k] : 0.f;
        }
    }
    for (int64_t row0 = wave * RPW; row0 < rows; row0 += nwaves * RPW) {
        const int64_t row = row0 + rsub;
        const bool live = row < rows;
        const int64_t base = row * width;
        const float mean = live ? mean_in[row] : 0.f, rstd = live ? rstd_in[row] : 0.f;
        const float rs = (row_scale && live) ? row_scale[sample_of(row, rows_per_sample)] : 1.f;
        float xh[ITERS][VEC], g[ITERS][VEC];
        float d2[ITERS][VEC];  // gradient arriving through the residual path of the fused add: requested WITH x and dy (it
                               // depends on nothing; loaded behind the row reductions it was a second serial round trip per row)
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = sub + LPR * it;
            if (live && c < nchunk) {
                float dyv[VEC];
                if (dres_in) vec_io<T, VEC>::load(dres_in, base + (int64_t)c * VEC, d2[it]);
                // (a software prefetch of the next rows' chunks, as in the forward kernel, was measured to change nothing here:
                // two loads per lane are in flight already, 4.7-5.1 TB/s)
                vec_io<T, VEC>::load(x, base + (int64_t)c * VEC, xh[it]);
                vec_io<T, VEC>::load(dy, base + (int64_t)c * VEC, dyv);
                const int64_t e0 = base + (int64_t)c * VEC;
                uint32_t hp = 0;
                const uint32_t ck = (!v1_mode && dropping) ? rng.template run_key<VEC>(e0) : 0u;
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    if (!v1_mode) {
                        dyv[k] *= rs;
                        if (dropping) xh[it][k] *= rng.template run_mult<VEC>(e0, k, ck, hp);
                    }
                    xh[it][k] = (xh[it][k] - mean) * rstd;
                    g[it][k] = dyv[k] * gm[it][k];
                    s1 += g[it][k];
                    s2 = fmaf(g[it][k], xh[it][k], s2);
                    dg[it][k] = fmaf(dyv[k], xh[it][k], dg[it][k]);
                    db[it][k] += dyv[k];
                }
            }
        }
        const float m1 = row_sum<LPR>(s1) * inv_w, m2 = row_sum<LPR>(s2) * inv_w;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = sub + LPR * it;
            if (live && c < nchunk) {
                float o[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) o[k] = rstd * (g[it][k] - m1 - xh[it][k] * m2);
                if (dres_in) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) o[k] += d2[it][k];
                }
                if (v1_mode) {
                    vec_io<T, VEC>::store(dx, base + (int64_t)c * VEC, o);
                    if (dadd_out) {  // gradient of the added operand: through DropPath scale and dropout mask
                        float o2[VEC];
                        const int64_t e0 = base + (int64_t)c * VEC;
                        uint32_t hp = 0;
                        const uint32_t ck = dropping ? rng.template run_key<VEC>(e0) : 0u;
#pragma unroll
                        for (int k = 0; k < VEC; ++k) {
                            o2[k] = o[k] * rs;
                            if (dropping) o2[k] *= rng.template run_mult<VEC>(e0, k, ck, hp);
                        }
                        vec_io<T, VEC>::store(dadd_out, base + (int64_t)c * VEC, o2);
                    }
                } else {
                    if (dropping) {
                        const int64_t e0 = base + (int64_t)c * VEC;
                        uint32_t hp = 0;
                        const uint32_t ck = rng.template run_key<VEC>(e0);
#pragma unroll
                        for (int k = 0; k < VEC; ++k) o[k] *= rng.template run_mult<VEC>(e0, k, ck, hp);
                    }
                    vec_io<T, VEC>::store(dx, base + (int64_t)c * VEC, o);
                }
            }
        }
    }
    // fold the wave's row groups (lanes sub, sub+LPR, ... hold the same columns)
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
#pragma unroll
            for (int off = LPR; off < 64; off <<= 1) {
                dg[it][k] += __shfl_xor(dg[it][k], off, 64);
                db[it][k] += __shfl_xor(db[it][k], off, 64);
            }
        }
    if (wid > 0 && rsub == 0) {
        float* mine = red + (size_t)(wid - 1) * 2 * width;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = sub + LPR * it;
            if (c < nchunk)
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    mine[c * VEC + k] = dg[it][k];
                    mine[width + c * VEC + k] = db[it][k];
                }
        }
    }
    __syncthreads();
    if (wid == 0 && rsub == 0) {
        float* outp = partials + (size_t)blockIdx.x * 2 * width;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = sub + LPR * it;
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

// The training default, specialised (round 4): bf16 rows in 16-byte chunks, no dropout, no DropPath scale, one gradient out.
// Same arithmetic and partial-sum layout as the general kernel above.  What differs is what is in flight: the general kernel
// needs 110 registers at 512 columns (4 waves per SIMD, one row of x, dy [, dsum] per wave requested at a time: 32-48 KB per CU,
// 3.2-3.9 TB/s on the 98 304 x 512 rows of stage 2; 162 registers and 1.6-2.0 TB/s at 1024 columns).  Here the chunks of the
// wave's NEXT rows are requested -- and held packed, 4 registers per 16 bytes -- before the current rows are reduced, and the
// launch is sized to ONE resident round of workgroups (run_bwd_fast).
// EX: the backward of y = rs * LN(drop(x)) (non-v1 form of the general kernel): dy_eff = rs dy, xhat from mask * x, dx = mask * LN_bwd.
template <typename T, int LPR, int ITERS, bool EX = false>
__global__ void __launch_bounds__(256, (ITERS == 1 ? (EX ? 4 : 5) : ITERS == 2 ? 3 : 1)) layernorm_bwd_fast_kernel(const void* __restrict__ dy, const void* __restrict__ x,
                                                                 const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                                 const float* __restrict__ rstd_in, void* __restrict__ dx,
                                                                 float* __restrict__ partials, int64_t rows, int width,
                                                                 const void* __restrict__ dres_in, const float* __restrict__ row_scale = nullptr,
                                                                 int64_t rows_per_sample = 1, float drop_p = 0.f, uint64_t seed = 0) {
    const ElemRng rng(drop_p, seed);
    const bool dropping = EX && drop_p > 0.f;
    extern __shared__ __attribute__((aligned(16))) float red[];  // [3 waves][2][width]
    constexpr int RPW = 64 / LPR, VEC = 16 / (int)sizeof(T), ES = (int)sizeof(T);
    constexpr bool PF = ITERS <= 2;  // (wider rows have >= 4 chunks per lane and tensor in flight already)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int sub = lane % LPR, rsub = lane / LPR;
    const int64_t wave = (int64_t)blockIdx.x * nw + wid;
    const int64_t stride = (int64_t)gridDim.x * nw * RPW;
    const int nchunk = width / VEC;
    const float inv_w = 1.f / (float)width;
    const bool has_res = dres_in != nullptr;
    float dg[ITERS][VEC], db[ITERS][VEC], gm[ITERS][VEC];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = sub + LPR * it;
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) {
            const float4 g4 = c < nchunk ? *(const float4*)(gamma + c * VEC + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            gm[it][4 * q] = g4.x; gm[it][4 * q + 1] = g4.y; gm[it][4 * q + 2] = g4.z; gm[it][4 * q + 3] = g4.w;
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            dg[it][k] = 0.f;
            db[it][k] = 0.f;
        }
    }
    // every tensor is addressed as (uniform base) + (32-bit byte offset): one offset register per chunk instead of a 64-bit
    // pointer per tensor and chunk (the launcher keeps rows * width * 2 below 4 GiB for this kernel)
    uint4 px[ITERS], pdy[ITERS], pd2[ITERS];
    float pmean = 0.f, prstd = 0.f;
    auto fetch = [&](int64_t r0) {
        const int64_t row = r0 + rsub;
        if (row < rows) {
            const uint32_t ro = (uint32_t)row * 4u;
            pmean = *(const float*)((const char*)mean_in + ro);
            prstd = *(const float*)((const char*)rstd_in + ro);
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int c = sub + LPR * it;
                if (c < nchunk) {
                    const uint32_t e = ((uint32_t)row * (uint32_t)width + (uint32_t)(c * VEC)) * (uint32_t)ES;
                    px[it] = *(const uint4*)((const char*)x + e);
                    pdy[it] = *(const uint4*)((const char*)dy + e);
                    if (has_res) pd2[it] = *(const uint4*)((const char*)dres_in + e);
                }
            }
        }
    };
    int64_t row0 = wave * RPW;
    if (PF && row0 < rows) fetch(row0);
    for (; row0 < rows; row0 += stride) {
        const int64_t row = row0 + rsub;
        const bool live = row < rows;
        if (!PF) fetch(row0);
        uint4 cx[ITERS], cdy[ITERS], cd2[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            cx[it] = px[it];
            cdy[it] = pdy[it];
            cd2[it] = pd2[it];
        }
        const float mean = pmean, rstd = prstd;
        if (PF && row0 + stride < rows) fetch(row0 + stride);
        // x and dy stay PACKED across the row reductions and are decoded a second time behind them (xhat and dy * gamma in fp32
        // would be 16 registers per chunk over the two shuffle trees: 114 registers, 4 waves per SIMD; this form needs 80: 6)
        float s1 = 0.f, s2 = 0.f;
        float rs = 1.f;
        if constexpr (EX) rs = (row_scale && live) ? row_scale[sample_of(row, rows_per_sample)] : 1.f;
        uint32_t keep[ITERS];  // EX: the chunk's keep decisions, one bit per element -- the hash (two integer-multiply rounds per element
                               // pair: the dominant cost of this variant) is evaluated ONCE and reused by the second decode
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            keep[it] = 0u;
            const int c = sub + LPR * it;
            if (live && c < nchunk) {
                const int64_t e0 = row * width + (int64_t)c * VEC;
                uint32_t hp = 0, ck = 0;
                if constexpr (EX) {
                    if (dropping) ck = rng.template run_key<VEC>(e0);
                }
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float xv = chunk_elem<T>(cx[it], k), dv = chunk_elem<T>(cdy[it], k);
                    if constexpr (EX) {
                        dv *= rs;
                        if (dropping) {
                            const float mk = rng.template run_mult<VEC>(e0, k, ck, hp);
                            keep[it] |= (mk != 0.f ? 1u : 0u) << k;
                            xv *= mk;
                        }
                    }
                    const float xh = (xv - mean) * rstd, g = dv * gm[it][k];
                    s1 += g;
                    s2 = fmaf(g, xh, s2);
                    dg[it][k] = fmaf(dv, xh, dg[it][k]);
                    db[it][k] += dv;
                }
            }
        }
        const float m1 = row_sum<LPR>(s1) * inv_w, m2 = row_sum<LPR>(s2) * inv_w;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {  // (opaque copies: the second decode must not be merged with the first)
            asm volatile("" : "+v"(cx[it].x), "+v"(cx[it].y), "+v"(cx[it].z), "+v"(cx[it].w));
            asm volatile("" : "+v"(cdy[it].x), "+v"(cdy[it].y), "+v"(cdy[it].z), "+v"(cdy[it].w));
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = sub + LPR * it;
            if (live && c < nchunk) {
                float o[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float xv = chunk_elem<T>(cx[it], k), dv = chunk_elem<T>(cdy[it], k);
                    float mk = 1.f;
                    if constexpr (EX) {
                        dv *= rs;
                        if (dropping) {
                            mk = ((keep[it] >> k) & 1u) ? rng.keep_scale : 0.f;
                            xv *= mk;
                        }
                    }
                    const float xh = (xv - mean) * rstd, g = dv * gm[it][k];
                    o[k] = rstd * (g - m1 - xh * m2);
                    if (has_res) o[k] += chunk_elem<T>(cd2[it], k);
                    if constexpr (EX) o[k] *= mk;
                }
                const uint32_t e = ((uint32_t)row * (uint32_t)width + (uint32_t)(c * VEC)) * (uint32_t)ES;
                *(uint4*)((char*)dx + e) = chunk_pack<T>(o);
            }
        }
    }
    // fold the wave's row groups (lanes sub, sub+LPR, ... hold the same columns), then the workgroup's waves through LDS
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
#pragma unroll
            for (int off = LPR; off < 64; off <<= 1) {
                dg[it][k] += __shfl_xor(dg[it][k], off, 64);
                db[it][k] += __shfl_xor(db[it][k], off, 64);
            }
        }
    if (wid > 0 && rsub == 0) {
        float* mine = red + (size_t)(wid - 1) * 2 * width;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = sub + LPR * it;
            if (c < nchunk)
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    mine[c * VEC + k] = dg[it][k];
                    mine[width + c * VEC + k] = db[it][k];
                }
        }
    }
    __syncthreads();
    if (wid == 0 && rsub == 0) {
        float* outp = partials + (size_t)blockIdx.x * 2 * width;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = sub + LPR * it;
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

// Sum of the per-workgroup partial rows [nblocks][2 * width] into dgamma | dbeta (overwrite, or add when accumulate != 0) in
// ONE launch: a workgroup owns 16 columns, its 16 waves x 4 row phases fold every 64th row (8 loads in flight), the phases are
// combined by shuffles and the 16 wave sums in a fixed order through LDS.  Deterministic.  (Two dependent launches -- a 16-group
// level and a final level -- cost 10.4 us per LayerNorm backward, 100 times per step; 64 columns per workgroup 9.8 us in the step
// at 512 columns -- 16 workgroups on 256 CUs; this form ~5 us.)
constexpr int kReduceGroups = 16;  // (still sizes the workspace returned by hs_layernorm_bwd_workspace)
__global__ void __launch_bounds__(1024) layernorm_param_reduce_kernel(const float* __restrict__ partials, float* __restrict__ dgamma,
                                                                      float* __restrict__ dbeta, int nblocks, int width,
                                                                      int accumulate) {
    // a workgroup owns 16 columns (64 workgroups at 512 columns instead of 16: the 5 MB of partial rows are L2-resident and the
    // launch was latency-bound on 16 CUs); lane -> (column lane % 16, row phase lane / 16), wave w folds rows 4 w + phase, + 64, ...
    __shared__ float part[16][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int width2 = 2 * width, col = blockIdx.x * 16 + (lane & 15);
    float acc = 0.f;
    if (col < width2) {
#pragma unroll 8
        for (int b = wave * 4 + (lane >> 4); b < nblocks; b += 64) acc += partials[(size_t)b * width2 + col];
    }
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    if (lane < 16) part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && lane < 16 && col < width2) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += part[w][lane];
        float* dst = col < width ? dgamma + col : dbeta + (col - width);
        *dst = accumulate ? *dst + tot : tot;
    }
}

// workgroups of the backward: enough to fill the chip for long inputs, few enough that the partial rows stay cheap
int bwd_blocks(int64_t rows) {
    // rows per workgroup: 48 (was 128) keeps 8 workgroups per CU busy on the 98 304-row stage as well (89 -> 83.5 us; the
    // partial rows of the parameter reduce grow with it: 6.7 -> 9.8 us)
    constexpr int div = 48;
    int64_t want = rows / div;
    if (want < 64) want = 64;
    if (want > kBwdMaxBlocks) want = kBwdMaxBlocks;
    const int64_t by_rows = (rows + 15) / 16;
    if (want > by_rows) want = by_rows;
    return (int)(want < 1 ? 1 : want);
}

struct LnExtra {  // stochastic extras and the compensated-stream operands, all optional
    const float* row_scale = nullptr;
    int64_t rows_per_sample = 1;
    float drop_p = 0.f;
    uint64_t seed = 0;
    const void* lo_in = nullptr;
    void* lo_out = nullptr;
};

template <typename T, int VEC, int LPR, int ITERS>
int run_fwd(const void* x, const void* res, const float* g, const float* b, void* y, float* mean, float* rstd, int64_t rows,
            int width, hipStream_t s, const void* add_in, void* sum_out, const LnExtra& ex) {
    constexpr int rows_per_block = 4 * (64 / LPR);
    int64_t blocks = (rows + rows_per_block - 1) / rows_per_block;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((layernorm_fwd_kernel<T, VEC, LPR, ITERS>), dim3((unsigned)blocks), dim3(256), 0, s, x, res, g, b, y,
                       mean, rstd, rows, width, add_in, sum_out, ex.row_scale, ex.rows_per_sample, ex.drop_p, ex.seed, ex.lo_in,
                       ex.lo_out);
    HS_LAUNCH_CHECK("layernorm_fwd");
    return HS_OK;
}

template <typename T, int VEC, int LPR, int ITERS>
int run_bwd(const void* dy, const void* x, const float* g, const float* mean, const float* rstd, void* dx, float* dgamma,
            float* dbeta, float* ws, int64_t rows, int width, hipStream_t s, const void* dres_in, void* dadd_out,
            const LnExtra& ex, int v1_mode, int accumulate) {
    const int blocks = bwd_blocks(rows);
    const size_t smem = (size_t)3 * 2 * width * sizeof(float);
    auto kern = layernorm_bwd_kernel<T, VEC, LPR, ITERS>;
    if (smem > 48 * 1024) HS_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), smem, s, dy, x, g, mean, rstd, dx, ws, rows, width, dres_in, dadd_out,
                       ex.row_scale, ex.rows_per_sample, ex.drop_p, ex.seed, v1_mode);
    HS_LAUNCH_CHECK("layernorm_bwd");
    if ((accumulate & HS_ACC_DEFER) && width % 4 == 0)  // the parameter reduce joins the stream's deferred queue (csrc/reduce_many.hip)
        return reduce_defer(ws, 2 * width, blocks, width, 2 * width, dgamma, dbeta, accumulate & 1, s);
    if (width % 4 == 0) return reduce_now(ws, 2 * width, blocks, width, 2 * width, dgamma, dbeta, accumulate & 1, s);
    accumulate &= 1;
    hipLaunchKernelGGL(layernorm_param_reduce_kernel, dim3((2 * width + 15) / 16), dim3(1024), 0, s, ws, dgamma, dbeta, blocks, width,
                       accumulate);
    HS_LAUNCH_CHECK("layernorm_param_reduce");
    return HS_OK;
}

template <typename T, int LPR, int ITERS, bool EX = false>
int run_fwd_fast(const void* x, const void* add_in, const float* g, const float* b, void* y, void* sum_out, float* mean, float* rstd,
                 int64_t rows, int width, hipStream_t s, const void* residual, const LnExtra& ex = LnExtra{}) {
    auto kern = layernorm_fwd_fast_kernel<T, LPR, ITERS, EX>;
    static int resident = 0;
    if (resident == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)kern, 256, 0) != hipSuccess || n < 1) n = 4;
        resident = n;
    }
    constexpr int rows_per_pass = 4 * (64 / LPR);
    int64_t blocks = (int64_t)usable_cus() * resident;
    const int64_t by_rows = (rows + rows_per_pass - 1) / rows_per_pass;
    if (blocks > by_rows) blocks = by_rows;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, s, x, add_in, g, b, y, sum_out, mean, rstd, rows, width, residual,
                       ex.row_scale, ex.rows_per_sample, ex.drop_p, ex.seed);
    HS_LAUNCH_CHECK("layernorm_fwd_fast");
    return HS_OK;
}

// One resident round: as many workgroups as the chip holds at this instantiation's register / LDS footprint (every workgroup
// then sees the same number of rows and there is no second, partly filled round), at most kBwdMaxBlocks partial rows, at
// least one row group per wave.
template <typename T, int LPR, int ITERS, bool EX = false>
int run_bwd_fast(const void* dy, const void* x, const float* g, const float* mean, const float* rstd, void* dx, float* dgamma,
                 float* dbeta, float* ws, int64_t rows, int width, hipStream_t s, const void* dres_in, int accumulate,
                 const LnExtra& ex = LnExtra{}) {
    auto kern = layernorm_bwd_fast_kernel<T, LPR, ITERS, EX>;
    const size_t smem = (size_t)3 * 2 * width * sizeof(float);
    if (smem > 48 * 1024) HS_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    static int resident_width = 0, resident = 0;  // per instantiation; the LDS footprint follows the width
    if (resident_width != width) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)kern, 256, smem) != hipSuccess || n < 1) n = 4;
        resident = n;
        resident_width = width;
    }
    constexpr int rows_per_pass = 4 * (64 / LPR);
    int64_t blocks = (int64_t)usable_cus() * resident;
    if (blocks > kBwdMaxBlocks) blocks = kBwdMaxBlocks;
    const int64_t by_rows = (rows + rows_per_pass - 1) / rows_per_pass;
    if (blocks > by_rows) blocks = by_rows;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), smem, s, dy, x, g, mean, rstd, dx, ws, rows, width, dres_in, ex.row_scale,
                       ex.rows_per_sample, ex.drop_p, ex.seed);
    HS_LAUNCH_CHECK("layernorm_bwd_fast");
    if ((accumulate & HS_ACC_DEFER) && width % 4 == 0)
        return reduce_defer(ws, 2 * width, (int)blocks, width, 2 * width, dgamma, dbeta, accumulate & 1, s);
    if (width % 4 == 0) return reduce_now(ws, 2 * width, (int)blocks, width, 2 * width, dgamma, dbeta, accumulate & 1, s);
    accumulate &= 1;
    hipLaunchKernelGGL(layernorm_param_reduce_kernel, dim3((2 * width + 15) / 16), dim3(1024), 0, s, ws, dgamma, dbeta, (int)blocks,
                       width, accumulate);
    HS_LAUNCH_CHECK("layernorm_param_reduce");
    return HS_OK;
}

// picks lanes-per-row and the per-lane register tile for a row of `width` elements in VEC-wide chunks
template <typename T, int VEC, typename F>
int with_shape(int width, F&& f) {
    const int chunks = width / VEC;
    using std::integral_constant;
    if (chunks <= 2) return f(integral_constant<int, 2>{}, integral_constant<int, 1>{});
    if (chunks <= 4) return f(integral_constant<int, 4>{}, integral_constant<int, 1>{});
    if (chunks <= 8) return f(integral_constant<int, 8>{}, integral_constant<int, 1>{});
    if (chunks <= 16) return f(integral_constant<int, 16>{}, integral_constant<int, 1>{});
    if (chunks <= 32) return f(integral_constant<int, 32>{}, integral_constant<int, 1>{});
    if (chunks <= 64) return f(integral_constant<int, 64>{}, integral_constant<int, 1>{});
    if (chunks <= 128) return f(integral_constant<int, 64>{}, integral_constant<int, 2>{});
    if (chunks <= 256) return f(integral_constant<int, 64>{}, integral_constant<int, 4>{});
    if (chunks <= 512) return f(integral_constant<int, 64>{}, integral_constant<int, 8>{});
    if (chunks <= 1024 && VEC == 1) return f(integral_constant<int, 64>{}, integral_constant<int, 16>{});
    return fail(HS_ERR_UNSUPPORTED, "layernorm width %d too large", width);
}

}  // namespace
}  // namespace hs

namespace {

int check_extra(const hs::LnExtra& ex, int64_t rows) {
    HS_CHECK_ARG(ex.drop_p >= 0.f && ex.drop_p <= 1.f, "drop_p must be in [0, 1]");
    HS_CHECK_ARG(!ex.row_scale || (ex.rows_per_sample > 0 && rows % ex.rows_per_sample == 0), "rows must be a multiple of rows_per_sample");
    return HS_OK;
}

int ln_fwd_impl(const void* x, const void* residual, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                int64_t rows, int width, int dtype, void* stream, const void* add_in, void* sum_out, const hs::LnExtra& ex) {
    using namespace hs;
    HS_CHECK_ARG(x && gamma && beta && y, "null pointer");
    HS_CHECK_ARG((mean == nullptr) == (rstd == nullptr), "mean and rstd must both be given or both be null");
    HS_CHECK_ARG((add_in == nullptr) == (sum_out == nullptr), "add_in and sum_out go together");
    HS_CHECK_ARG(rows >= 0 && width > 0, "bad shape");
    HS_CHECK_ARG(dtype == HS_F32 || dtype == HS_BF16, "dtype must be HS_F32 or HS_BF16");
    if (int st = check_extra(ex, rows)) return st;
    if (rows == 0) return HS_OK;
    hipStream_t s = (hipStream_t)stream;
    const bool plain = !(residual && add_in) && !ex.row_scale && ex.drop_p == 0.f && !ex.lo_in && !ex.lo_out;
    // the v2 / plain form with dropout and / or a DropPath factor (y = [residual +] rs LN(drop(x))): the specialised kernel's EX variant
    const bool extras = !add_in && (ex.row_scale || ex.drop_p > 0.f) && !ex.lo_in && !ex.lo_out;
    if (dtype == HS_BF16 && extras && width % 8 == 0 && width <= 4096 && rows * width * 2 < (1ll << 32))
        return with_shape<bf16_t, 8>(width, [&](auto lpr, auto it) { return run_fwd_fast<bf16_t, decltype(lpr)::value, decltype(it)::value, true>(x, add_in, gamma, beta, y, sum_out, mean, rstd, rows, width, s, residual, ex); });
    if (dtype == HS_BF16) {
        if (plain && width % 8 == 0 && width <= 4096 && rows * width * 2 < (1ll << 32))
            return with_shape<bf16_t, 8>(width, [&](auto lpr, auto it) { return run_fwd_fast<bf16_t, decltype(lpr)::value, decltype(it)::value>(x, add_in, gamma, beta, y, sum_out, mean, rstd, rows, width, s, residual); });
        if (width % 8 == 0)
            return with_shape<bf16_t, 8>(width, [&](auto lpr, auto it) { return run_fwd<bf16_t, 8, decltype(lpr)::value, decltype(it)::value>(x, residual, gamma, beta, y, mean, rstd, rows, width, s, add_in, sum_out, ex); });
        return with_shape<bf16_t, 1>(width, [&](auto lpr, auto it) { return run_fwd<bf16_t, 1, decltype(lpr)::value, decltype(it)::value>(x, residual, gamma, beta, y, mean, rstd, rows, width, s, add_in, sum_out, ex); });
    }
    if (plain && width % 4 == 0 && width <= 2048 && rows * width * 4 < (1ll << 32))
        return with_shape<float, 4>(width, [&](auto lpr, auto it) { return run_fwd_fast<float, decltype(lpr)::value, decltype(it)::value>(x, add_in, gamma, beta, y, sum_out, mean, rstd, rows, width, s, residual); });
    if (width % 4 == 0)
        return with_shape<float, 4>(width, [&](auto lpr, auto it) { return run_fwd<float, 4, decltype(lpr)::value, decltype(it)::value>(x, residual, gamma, beta, y, mean, rstd, rows, width, s, add_in, sum_out, ex); });
    return with_shape<float, 1>(width, [&](auto lpr, auto it) { return run_fwd<float, 1, decltype(lpr)::value, decltype(it)::value>(x, residual, gamma, beta, y, mean, rstd, rows, width, s, add_in, sum_out, ex); });
}

int ln_bwd_impl(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx, float* dgamma,
                float* dbeta, float* workspace, int64_t rows, int width, int dtype, void* stream, const void* dres_in,
                void* dadd_out, const hs::LnExtra& ex, int v1_mode, int accumulate) {
    using namespace hs;
    HS_CHECK_ARG(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && workspace, "null pointer");
    HS_CHECK_ARG(rows > 0 && width > 0, "bad shape");
    HS_CHECK_ARG(dtype == HS_F32 || dtype == HS_BF16, "dtype must be HS_F32 or HS_BF16");
    if (int st = check_extra(ex, rows)) return st;
    hipStream_t s = (hipStream_t)stream;
    const bool plain = !ex.row_scale && ex.drop_p == 0.f && !dadd_out;
    if (dtype == HS_BF16 && !plain && !v1_mode && !dadd_out && !dres_in && width % 8 == 0 && width <= 4096 && rows * width * 2 < (1ll << 32))
        return with_shape<bf16_t, 8>(width, [&](auto lpr, auto it) { return run_bwd_fast<bf16_t, decltype(lpr)::value, decltype(it)::value, true>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, width, s, dres_in, accumulate, ex); });
    if (dtype == HS_F32 && plain && width % 4 == 0 && width <= 2048 && rows * width * 4 < (1ll << 32))
        return with_shape<float, 4>(width, [&](auto lpr, auto it) { return run_bwd_fast<float, decltype(lpr)::value, decltype(it)::value>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, width, s, dres_in, accumulate); });
    if (dtype == HS_BF16) {
        if (plain && width % 8 == 0 && width <= 4096 && rows * width * 2 < (1ll << 32))
            return with_shape<bf16_t, 8>(width, [&](auto lpr, auto it) { return run_bwd_fast<bf16_t, decltype(lpr)::value, decltype(it)::value>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, width, s, dres_in, accumulate); });
        if (width % 8 == 0)
            return with_shape<bf16_t, 8>(width, [&](auto lpr, auto it) { return run_bwd<bf16_t, 8, decltype(lpr)::value, decltype(it)::value>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, width, s, dres_in, dadd_out, ex, v1_mode, accumulate); });
        return with_shape<bf16_t, 1>(width, [&](auto lpr, auto it) { return run_bwd<bf16_t, 1, decltype(lpr)::value, decltype(it)::value>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, width, s, dres_in, dadd_out, ex, v1_mode, accumulate); });
    }
    if (width % 4 == 0)
        return with_shape<float, 4>(width, [&](auto lpr, auto it) { return run_bwd<float, 4, decltype(lpr)::value, decltype(it)::value>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, width, s, dres_in, dadd_out, ex, v1_mode, accumulate); });
    return with_shape<float, 1>(width, [&](auto lpr, auto it) { return run_bwd<float, 1, decltype(lpr)::value, decltype(it)::value>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, width, s, dres_in, dadd_out, ex, v1_mode, accumulate); });
}

hs::LnExtra make_extra(const float* row_scale, int64_t rows_per_sample, float drop_p, uint64_t seed) {
    hs::LnExtra ex;
    ex.row_scale = row_scale;
    ex.rows_per_sample = rows_per_sample > 0 ? rows_per_sample : 1;
    ex.drop_p = drop_p;
    ex.seed = seed;
    return ex;
}

}  // namespace

extern "C" {

int hs_layernorm_fwd(const void* x, const void* residual, const float* gamma, const float* beta, void* y, float* mean,
                     float* rstd, int64_t rows, int width, int dtype, void* stream) {
    return ln_fwd_impl(x, residual, gamma, beta, y, mean, rstd, rows, width, dtype, stream, nullptr, nullptr, hs::LnExtra{});
}

int hs_add_layernorm_fwd(const void* a, const void* b, const float* gamma, const float* beta, void* sum_out, void* y,
                         float* mean, float* rstd, int64_t rows, int width, int dtype, void* stream) {
    HS_CHECK_ARG(b && sum_out, "null pointer");
    return ln_fwd_impl(a, nullptr, gamma, beta, y, mean, rstd, rows, width, dtype, stream, b, sum_out, hs::LnExtra{});
}

int64_t hs_layernorm_bwd_workspace(int64_t rows, int width) {
    (void)rows;  // the bf16 kernel sizes its launch by the chip, not by the rows: room for the largest launch of either kernel
    return (int64_t)(hs::kBwdMaxBlocks + hs::kReduceGroups) * 2 * width;
}

int hs_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                     float* dgamma, float* dbeta, float* workspace, int accumulate, int64_t rows, int width, int dtype,
                     void* stream) {
    return ln_bwd_impl(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, width, dtype, stream, nullptr, nullptr,
                       hs::LnExtra{}, 0, accumulate);
}

int hs_add_layernorm_bwd(const void* dy, const void* dsum, const void* sum, const float* gamma, const float* mean,
                         const float* rstd, void* dx, float* dgamma, float* dbeta, float* workspace, int accumulate, int64_t rows,
                         int width, int dtype, void* stream) {
    return ln_bwd_impl(dy, sum, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, width, dtype, stream, dsum, nullptr,
                       hs::LnExtra{}, 1, accumulate);
}

/* the general forward: every optional operand of the kernels above in one entry point.  Exactly one of `residual` (v2 placement:
 * y = residual + rs LN(drop(x))) and `add_in` (v1: sum_out = x + rs drop(add_in), y = LN(sum_out)) may be given; lo_in / lo_out are the
 * compensated-stream remainders of the stream operand (x for v1, residual for v2) and of the new stream (sum_out for v1, y for v2). */
int hs_layernorm_fwd_ex(const void* x, const void* residual, const void* add_in, const void* lo_in, const float* gamma, const float* beta,
                        void* y, void* sum_out, void* lo_out, float* mean, float* rstd, const float* row_scale, int64_t rows_per_sample,
                        float drop_p, uint64_t seed, int64_t rows, int width, int dtype, void* stream) {
    HS_CHECK_ARG(!(residual && add_in), "hs_layernorm_fwd_ex: residual and add_in are exclusive");
    HS_CHECK_ARG(!lo_in || residual || add_in, "hs_layernorm_fwd_ex: lo_in needs a stream operand (residual or add_in)");
    hs::LnExtra ex = make_extra(row_scale, rows_per_sample, drop_p, seed);
    ex.lo_in = lo_in;
    ex.lo_out = lo_out;
    return ln_fwd_impl(x, residual, gamma, beta, y, mean, rstd, rows, width, dtype, stream, add_in, sum_out, ex);
}

/* train-mode variants: dropout (drop_p, seed) and per-sample DropPath scale (row_scale[rows / rows_per_sample]) fused in */
int hs_layernorm_drop_fwd(const void* x, const void* residual, const float* gamma, const float* beta, void* y, float* mean,
                          float* rstd, const float* row_scale, int64_t rows_per_sample, float drop_p, uint64_t seed,
                          int64_t rows, int width, int dtype, void* stream) {
    return ln_fwd_impl(x, residual, gamma, beta, y, mean, rstd, rows, width, dtype, stream, nullptr, nullptr,
                       make_extra(row_scale, rows_per_sample, drop_p, seed));
}

int hs_layernorm_drop_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, void* dx,
                          float* dgamma, float* dbeta, float* workspace, int accumulate, const float* row_scale,
                          int64_t rows_per_sample, float drop_p, uint64_t seed, int64_t rows, int width, int dtype, void* stream) {
    return ln_bwd_impl(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, rows, width, dtype, stream, nullptr, nullptr,
                       make_extra(row_scale, rows_per_sample, drop_p, seed), 0, accumulate);
}

int hs_add_layernorm_drop_fwd(const void* a, const void* b, const float* gamma, const float* beta, void* sum_out, void* y,
                              float* mean, float* rstd, const float* row_scale, int64_t rows_per_sample, float drop_p,
                              uint64_t seed, int64_t rows, int width, int dtype, void* stream) {
    HS_CHECK_ARG(b && sum_out, "null pointer");
    return ln_fwd_impl(a, nullptr, gamma, beta, y, mean, rstd, rows, width, dtype, stream, b, sum_out,
                       make_extra(row_scale, rows_per_sample, drop_p, seed));
}

int hs_add_layernorm_drop_bwd(const void* dy, const void* dsum, const void* sum, const float* gamma, const float* mean,
                              const float* rstd, void* da, void* db, float* dgamma, float* dbeta, float* workspace,
                              int accumulate, const float* row_scale, int64_t rows_per_sample, float drop_p, uint64_t seed,
                              int64_t rows, int width, int dtype, void* stream) {
    HS_CHECK_ARG(db, "null pointer");
    return ln_bwd_impl(dy, sum, gamma, mean, rstd, da, dgamma, dbeta, workspace, rows, width, dtype, stream, dsum, db,
                       make_extra(row_scale, rows_per_sample, drop_p, seed), 1, accumulate);
}

}  // extern "C"
