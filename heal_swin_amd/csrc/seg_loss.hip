// Class-weighted cross-entropy of the segmentation caller (reference: nn.CrossEntropyLoss(weight) on logits [B, K, Npix]
// and labels [B, Npix], heal_swin/models_lightning/segmentation/model_lightning_swin_hp.py:39-45, :104-111), SURVEY 8a row L:
//     loss = sum_i w[y_i] * (logsumexp_c z_i[c] - z_i[y_i]) / sum_i w[y_i]          (pixels with y_i == ignore_index skipped)
//     dz_i[c] = scale * w[y_i] * (softmax(z_i)[c] - [c == y_i]),   scale = upstream gradient / sum_i w[y_i]
// One thread per pixel; the K logits of a pixel are read through explicit element strides, so both the model's native
// output (K contiguous per pixel, then viewed as [B, K, Npix]) and a class-major [B, K, Npix] tensor are read coalesced and
// no transposed / fp32 copy of the logits is made (the torch composition costs five full passes over them).  HBM-bound:
// forward reads the logits once, backward reads them once (softmax recomputed) and writes the gradient once.
// Deterministic: per-workgroup partial sums in a fixed grid, summed by the caller.
#include <type_traits>

#include "hs_device.h"

namespace hs {
namespace {

constexpr int kMaxClasses = 64;
constexpr int kCeBlocks = 2048;

struct CeArgs {
    const void* logits;
    const void* labels;
    const float* weights;  // [K] or null (all ones)
    int64_t batch, npix;
    int K;
    int64_t sb, sk, sp;  // element strides of logits over batch, class, pixel
    int label_bytes;     // 1 (uint8), 4 (int32) or 8 (int64)
    int64_t ignore_index;
};

__device__ __forceinline__ int64_t load_label(const CeArgs& a, int64_t i) {
    if (a.label_bytes == 1) return ((const uint8_t*)a.labels)[i];
    if (a.label_bytes == 4) return ((const int32_t*)a.labels)[i];
    return ((const int64_t*)a.labels)[i];
}

// log-sum-exp of one pixel's logits (two passes over K values that sit in L1 after the first touch)
template <typename T>
__device__ __forceinline__ float pixel_lse(const CeArgs& a, int64_t base) {
    float m = -INFINITY;
    for (int c = 0; c < a.K; ++c) m = fmaxf(m, io<T>::load(a.logits, base + c * a.sk));
    float s = 0.f;
    for (int c = 0; c < a.K; ++c) s += expf(io<T>::load(a.logits, base + c * a.sk) - m);
    return m + logf(s);
}

template <typename T>
__global__ void __launch_bounds__(256) seg_ce_fwd_kernel(CeArgs a, float* __restrict__ partials) {
    const int64_t total = a.batch * a.npix;
    float num = 0.f, den = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t y = load_label(a, i);
        if (y == a.ignore_index || (uint64_t)y >= (uint64_t)a.K) continue;  // out-of-range labels contribute nothing
        const int64_t b = i / a.npix, px = i - b * a.npix;
        const int64_t base = b * a.sb + px * a.sp;
        const float w = a.weights ? a.weights[y] : 1.f;
        const float lse = pixel_lse<T>(a, base);
        num = fmaf(w, lse - io<T>::load(a.logits, base + y * a.sk), num);
        den += w;
    }
    __shared__ float red[2][4];
    num = wave_sum(num);
    den = wave_sum(den);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][wave] = num;
        red[1][wave] = den;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partials[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) seg_ce_bwd_kernel(CeArgs a, const float* __restrict__ scale_p, void* __restrict__ dlogits,
                                                         int64_t db, int64_t dk, int64_t dp) {
    const int64_t total = a.batch * a.npix;
    const float scale = scale_p[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t y = load_label(a, i);
        const int64_t b = i / a.npix, px = i - b * a.npix;
        const int64_t base = b * a.sb + px * a.sp, dbase = b * db + px * dp;
        if (y == a.ignore_index || (uint64_t)y >= (uint64_t)a.K) {
            for (int c = 0; c < a.K; ++c) io<T>::store(dlogits, dbase + c * dk, 0.f);
            continue;
        }
        const float g = scale * (a.weights ? a.weights[y] : 1.f);
        const float lse = pixel_lse<T>(a, base);
        for (int c = 0; c < a.K; ++c) {
            const float p = expf(io<T>::load(a.logits, base + c * a.sk) - lse);
            io<T>::store(dlogits, dbase + c * dk, g * (p - (c == y ? 1.f : 0.f)));
        }
    }
}

// ---- fast path: bf16 logits with the K classes of a pixel contiguous inside a 16-element (32-byte) row -- the model's own
// output layout (the head is padded to 16 classes).  The generic kernels above issue 3 K two-byte loads per pixel and run
// at ~1 TB/s; here a pixel is two 16-byte loads, the K logits stay in registers, and the gradient leaves as whole vectors.
constexpr int kRow = 16;

template <typename T>
__device__ __forceinline__ void load_row16(const void* logits, int64_t base, float* z);
template <>
__device__ __forceinline__ void load_row16<bf16_t>(const void* logits, int64_t base, float* z) {
    const uint4 a = *(const uint4*)((const uint16_t*)logits + base), b = *(const uint4*)((const uint16_t*)logits + base + 8);
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        z[2 * i] = __uint_as_float(w[i] << 16);
        z[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
template <>
__device__ __forceinline__ void load_row16<float>(const void* logits, int64_t base, float* z) {  // fp32 logits rows of 16 (64 B)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = *(const float4*)((const float*)logits + base + 4 * q);
        z[4 * q] = v.x; z[4 * q + 1] = v.y; z[4 * q + 2] = v.z; z[4 * q + 3] = v.w;
    }
}
__device__ __forceinline__ float row_lse(const float* z, int K, float* zy, int64_t y) {
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < kRow; ++c)
        if (c < K) m = fmaxf(m, z[c]);
    float s = 0.f, pick = 0.f;
#pragma unroll
    for (int c = 0; c < kRow; ++c)
        if (c < K) {
            s += expf(z[c] - m);
            if (c == y) pick = z[c];
        }
    *zy = pick;
    return m + logf(s);
}

template <typename T>
__global__ void __launch_bounds__(256) seg_ce_fwd_row16_kernel(CeArgs a, float* __restrict__ partials) {
    const int64_t total = a.batch * a.npix;
    float num = 0.f, den = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t y = load_label(a, i);
        if (y == a.ignore_index || (uint64_t)y >= (uint64_t)a.K) continue;
        const int64_t b = i / a.npix, px = i - b * a.npix;
        float z[kRow], zy;
        load_row16<T>(a.logits, b * a.sb + px * kRow, z);
        const float w = a.weights ? a.weights[y] : 1.f;
        const float lse = row_lse(z, a.K, &zy, y);
        num = fmaf(w, lse - zy, num);
        den += w;
    }
    __shared__ float red[2][4];
    num = wave_sum(num);
    den = wave_sum(den);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][wave] = num;
        red[1][wave] = den;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        partials[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// writes columns 0 .. K-1 only (K a multiple of 4: 8-byte groups), so a destination whose rows interleave with other data
// is left alone exactly as by the generic kernel
template <typename T>
__global__ void __launch_bounds__(256) seg_ce_bwd_row16_kernel(CeArgs a, const float* __restrict__ scale_p, void* __restrict__ dlogits,
                                                               int64_t db) {
    const int64_t total = a.batch * a.npix;
    const float scale = scale_p[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t y = load_label(a, i);
        const int64_t b = i / a.npix, px = i - b * a.npix;
        float d[kRow];
#pragma unroll
        for (int c = 0; c < kRow; ++c) d[c] = 0.f;
        if (!(y == a.ignore_index || (uint64_t)y >= (uint64_t)a.K)) {
            float z[kRow], zy;
            load_row16<T>(a.logits, b * a.sb + px * kRow, z);
            const float g = scale * (a.weights ? a.weights[y] : 1.f);
            const float lse = row_lse(z, a.K, &zy, y);
#pragma unroll
            for (int c = 0; c < kRow; ++c)
                if (c < a.K) d[c] = g * (expf(z[c] - lse) - (c == y ? 1.f : 0.f));
        }
        if constexpr (std::is_same<T, float>::value) {
            float* o = (float*)dlogits + b * db + px * kRow;
#pragma unroll
            for (int q = 0; q < kRow / 4; ++q)
                if (4 * q < a.K) *(float4*)(o + 4 * q) = make_float4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
        } else {
            uint16_t* o = (uint16_t*)dlogits + b * db + px * kRow;
#pragma unroll
            for (int q = 0; q < kRow / 4; ++q)
                if (4 * q < a.K) *(uint2*)(o + 4 * q) = make_uint2(pack_bf16x2(d[4 * q], d[4 * q + 1]), pack_bf16x2(d[4 * q + 2], d[4 * q + 3]));
        }
    }
}

// The fast path reads the WHOLE 32-byte row of a pixel (two 16-byte loads): only for K > 8 (the second load then overlaps the
// K used columns) and a pointer at the start of a 32-byte row -- a caller view such as buf[..., 8:16] (sp = 16, K = 8) would
// otherwise read 16 bytes past the storage on its last pixel.
bool row16_layout(const void* ptr, int64_t sb, int64_t sk, int64_t sp, int K, int dtype) {
    const uintptr_t row_bytes = dtype == HS_BF16 ? 32 : 64;
    return sk == 1 && sp == kRow && K > 8 && K <= kRow && K % 4 == 0 && sb % kRow == 0 && ((uintptr_t)ptr & (row_bytes - 1)) == 0;
}

int check_args(const CeArgs& a, int dtype) {
    HS_CHECK_ARG(a.logits && a.labels, "null pointer");
    HS_CHECK_ARG(a.batch > 0 && a.npix > 0, "bad shape");
    HS_CHECK_ARG(a.K >= 1 && a.K <= kMaxClasses, "n_classes must be in [1, 64]");
    HS_CHECK_ARG(a.label_bytes == 1 || a.label_bytes == 4 || a.label_bytes == 8, "labels must be uint8, int32 or int64");
    HS_CHECK_ARG(dtype == HS_F32 || dtype == HS_BF16, "dtype must be HS_F32 or HS_BF16");
    return HS_OK;
}

unsigned ce_grid(int64_t total) {
    int64_t b = (total + 255) / 256;
    if (b > kCeBlocks) b = kCeBlocks;
    return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace
}  // namespace hs

extern "C" {

int64_t hs_seg_ce_partials(int64_t batch, int64_t npix) { return batch > 0 && npix > 0 ? (int64_t)hs::ce_grid(batch * npix) : 0; }

int hs_seg_ce_fwd(const void* logits, const void* labels, const float* class_weights, float* partials, int64_t batch, int64_t npix,
                  int n_classes, int64_t stride_b, int64_t stride_k, int64_t stride_p, int label_bytes, int64_t ignore_index,
                  int dtype, void* stream) {
    using namespace hs;
    const CeArgs a{logits, labels, class_weights, batch, npix, n_classes, stride_b, stride_k, stride_p, label_bytes, ignore_index};
    if (int st = check_args(a, dtype)) return st;
    HS_CHECK_ARG(partials, "null pointer");
    const unsigned grid = ce_grid(batch * npix);
    if (row16_layout(logits, stride_b, stride_k, stride_p, n_classes, dtype)) {
        if (dtype == HS_BF16) hipLaunchKernelGGL(seg_ce_fwd_row16_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, partials);
        else hipLaunchKernelGGL(seg_ce_fwd_row16_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, partials);
    } else if (dtype == HS_BF16) hipLaunchKernelGGL(seg_ce_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, partials);
    else hipLaunchKernelGGL(seg_ce_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, partials);
    HS_LAUNCH_CHECK("seg_ce_fwd");
    return HS_OK;
}

int hs_seg_ce_bwd(const void* logits, const void* labels, const float* class_weights, const float* scale, void* dlogits,
                  int64_t batch, int64_t npix, int n_classes, int64_t stride_b, int64_t stride_k, int64_t stride_p,
                  int64_t dstride_b, int64_t dstride_k, int64_t dstride_p, int label_bytes, int64_t ignore_index, int dtype,
                  void* stream) {
    using namespace hs;
    const CeArgs a{logits, labels, class_weights, batch, npix, n_classes, stride_b, stride_k, stride_p, label_bytes, ignore_index};
    if (int st = check_args(a, dtype)) return st;
    HS_CHECK_ARG(scale && dlogits, "null pointer");
    const unsigned grid = ce_grid(batch * npix);
    if (row16_layout(logits, stride_b, stride_k, stride_p, n_classes, dtype) &&
        row16_layout(dlogits, dstride_b, dstride_k, dstride_p, n_classes, dtype)) {
        if (dtype == HS_BF16)
            hipLaunchKernelGGL(seg_ce_bwd_row16_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, scale, dlogits, dstride_b);
        else
            hipLaunchKernelGGL(seg_ce_bwd_row16_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, scale, dlogits, dstride_b);
    } else if (dtype == HS_BF16)
        hipLaunchKernelGGL(seg_ce_bwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, scale, dlogits, dstride_b,
                           dstride_k, dstride_p);
    else
        hipLaunchKernelGGL(seg_ce_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, scale, dlogits, dstride_b,
                           dstride_k, dstride_p);
    HS_LAUNCH_CHECK("seg_ce_bwd");
    return HS_OK;
}

}  // extern "C"
