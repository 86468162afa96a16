// Relative-position bias gather  bias[h,i,j] = table[rel_idx[i,j], h]  and its gradient
// (swin_hp_transformer.py:152-159).  Tiny (nH x Ws x Ws elements); one launch per attention call.
#include "hs_device.h"

namespace {

__global__ void rel_bias_gather_kernel(const float* __restrict__ table, const int32_t* __restrict__ rel_idx,
                                       float* __restrict__ bias, int nH, int ws2) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nH * ws2) return;
    const int h = e / ws2, ij = e - h * ws2;
    bias[e] = table[(int64_t)rel_idx[ij] * nH + h];
}

// one workgroup per table row t: sums the dbias entries whose index is t (deterministic, no atomics)
__global__ void rel_bias_scatter_grad_kernel(const float* __restrict__ dbias, const int32_t* __restrict__ rel_idx,
                                             float* __restrict__ dtable, int nH, int ws2) {
    const int t = blockIdx.x, h = blockIdx.y;
    float acc = 0.f;
    for (int ij = threadIdx.x; ij < ws2; ij += blockDim.x)
        if (rel_idx[ij] == t) acc += dbias[(int64_t)h * ws2 + ij];
    acc = hs::wave_sum(acc);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += part[w];
        dtable[(int64_t)t * nH + h] = tot;
    }
}

// the same sum from the entries grouped by table row (order = stable argsort of rel_idx, offsets = its run boundaries): 16
// lanes per (t, h), lane s adds entries s, s + 16, ... of the row's run (independent loads in flight), then a fixed
// xor-tree over the 16 lanes -- deterministic, and T * nH * 16 threads instead of T * nH workgroups that each scan the
// whole index (15.8 us per attention backward before)
__global__ void __launch_bounds__(256) rel_bias_scatter_grad_sorted_kernel(const float* __restrict__ dbias, const int32_t* __restrict__ order,
                                                                           const int32_t* __restrict__ offsets, float* __restrict__ dtable,
                                                                           int rows, int nH, int ws2, int accumulate) {
    const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, sub = threadIdx.x & 15;
    const bool live = g < rows * nH;
    const int t = live ? g / nH : 0, h = live ? g - t * nH : 0;
    const int k0 = offsets[t], k1 = live ? offsets[t + 1] : k0;
    float acc = 0.f;
#pragma unroll 4
    for (int k = k0 + sub; k < k1; k += 16) acc += dbias[(int64_t)h * ws2 + order[k]];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (live && sub == 0) dtable[g] = accumulate ? dtable[g] + acc : acc;
}

// Cosine attention's per-head score scale (ref swin_hp_transformer.py:144-147): scale = exp(min(logit_scale, ln 100)), and its
// backward d logit_scale = d scale * scale * [logit_scale <= ln 100] -- one launch each instead of torch's clamp / exp / mul /
// compare / where kernels (seven launches of a few microseconds per block and step; 24 blocks in HEAL-SWIN-T)
constexpr float kLogitMax = 4.605170185988092f;  // ln(1 / 0.01)
__global__ void __launch_bounds__(64) cos_scale_kernel(const float* __restrict__ ls, const float* __restrict__ dscale, float* __restrict__ out,
                                                       int n, int accumulate) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float l = ls[i], s = __expf(fminf(l, kLogitMax));
    if (!dscale) out[i] = s;
    else {
        const float d = l <= kLogitMax ? dscale[i] * s : 0.f;
        out[i] = accumulate ? out[i] + d : d;
    }
}

// ---- all attention blocks of a model in one launch (T @ nside 128 is bound by its ~600 launches per step: 2 x 22 of them were these)
constexpr int kManyMax = 48;
struct GatherJobs {
    const float* table[kManyMax];
    int heads[kManyMax], first_head[kManyMax];  // bias of job j = base + first_head[j] * ws2
};
__global__ void rel_bias_gather_many_kernel(GatherJobs jobs, const int32_t* __restrict__ rel_idx, float* __restrict__ base, int ws2) {
    const int j = blockIdx.y, nH = jobs.heads[j];
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nH * ws2) return;
    const int h = e / ws2, ij = e - h * ws2;
    base[(int64_t)jobs.first_head[j] * ws2 + e] = jobs.table[j][(int64_t)rel_idx[ij] * nH + h];
}
struct ScatterJobs {
    const float* dbias[kManyMax];
    float* dtable[kManyMax];
    int heads[kManyMax], accumulate[kManyMax];
};
__global__ void __launch_bounds__(256) rel_bias_scatter_many_kernel(ScatterJobs jobs, const int32_t* __restrict__ order,
                                                                    const int32_t* __restrict__ offsets, int rows, int ws2) {
    const int j = blockIdx.y, nH = jobs.heads[j];
    const int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, sub = threadIdx.x & 15;
    if (((blockIdx.x * blockDim.x) >> 4) >= rows * nH) return;  // (whole workgroup beyond this job's table)
    const bool live = g < rows * nH;
    const int t = live ? g / nH : 0, h = live ? g - t * nH : 0;
    const int k0 = offsets[t], k1 = live ? offsets[t + 1] : k0;
    const float* dbias = jobs.dbias[j];
    float acc = 0.f;
#pragma unroll 4
    for (int k = k0 + sub; k < k1; k += 16) acc += dbias[(int64_t)h * ws2 + order[k]];
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    float* dtable = jobs.dtable[j];
    if (live && sub == 0) dtable[g] = jobs.accumulate[j] ? dtable[g] + acc : acc;
}
struct ScaleJobs {
    const float* ls[kManyMax];
    const float* dscale[kManyMax];
    float* out[kManyMax];
    int heads[kManyMax], accumulate[kManyMax];
};
__global__ void __launch_bounds__(64) cos_scale_many_kernel(ScaleJobs jobs) {
    const int j = blockIdx.x, i = threadIdx.x;
    if (i >= jobs.heads[j]) return;
    const float l = jobs.ls[j][i], s = __expf(fminf(l, kLogitMax));
    float* out = jobs.out[j];
    if (!jobs.dscale[j]) out[i] = s;
    else {
        const float d = l <= kLogitMax ? jobs.dscale[j][i] * s : 0.f;
        out[i] = jobs.accumulate[j] ? out[i] + d : d;
    }
}

}  // namespace

extern "C" {

int hs_rel_bias_gather_many(const void* const* tables, const int* heads, int count, const int32_t* rel_idx, float* bias_base,
                            int table_rows, int window_size, void* stream) {
    HS_CHECK_ARG(tables && heads && rel_idx && bias_base && count > 0 && table_rows > 0 && window_size > 0, "hs_rel_bias_gather_many: bad arguments");
    const int ws2 = window_size * window_size;
    int first = 0;
    for (int j0 = 0; j0 < count; j0 += kManyMax) {
        GatherJobs jobs{};
        const int n = count - j0 < kManyMax ? count - j0 : kManyMax;
        int max_heads = 0;
        for (int j = 0; j < n; ++j) {
            HS_CHECK_ARG(tables[j0 + j] && heads[j0 + j] > 0, "hs_rel_bias_gather_many: null table or no heads (job %d)", j0 + j);
            jobs.table[j] = (const float*)tables[j0 + j];
            jobs.heads[j] = heads[j0 + j];
            jobs.first_head[j] = first;
            first += heads[j0 + j];
            max_heads = heads[j0 + j] > max_heads ? heads[j0 + j] : max_heads;
        }
        hipLaunchKernelGGL(rel_bias_gather_many_kernel, dim3((max_heads * ws2 + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, jobs, rel_idx,
                           bias_base, ws2);
        HS_LAUNCH_CHECK("rel_bias_gather_many");
    }
    return HS_OK;
}

int hs_rel_bias_scatter_grad_sorted_many(const void* const* dbias, void* const* dtables, const int* heads, const int* accumulate, int count,
                                         const int32_t* order, const int32_t* offsets, int table_rows, int window_size, void* stream) {
    HS_CHECK_ARG(dbias && dtables && heads && accumulate && order && offsets && count > 0 && table_rows > 0 && window_size > 0,
                 "hs_rel_bias_scatter_grad_sorted_many: bad arguments");
    for (int j0 = 0; j0 < count; j0 += kManyMax) {
        ScatterJobs jobs{};
        const int n = count - j0 < kManyMax ? count - j0 : kManyMax;
        int max_heads = 0;
        for (int j = 0; j < n; ++j) {
            HS_CHECK_ARG(dbias[j0 + j] && dtables[j0 + j] && heads[j0 + j] > 0, "hs_rel_bias_scatter_grad_sorted_many: null pointer or no heads (job %d)", j0 + j);
            jobs.dbias[j] = (const float*)dbias[j0 + j];
            jobs.dtable[j] = (float*)dtables[j0 + j];
            jobs.heads[j] = heads[j0 + j];
            jobs.accumulate[j] = accumulate[j0 + j];
            max_heads = heads[j0 + j] > max_heads ? heads[j0 + j] : max_heads;
        }
        hipLaunchKernelGGL(rel_bias_scatter_many_kernel, dim3((table_rows * max_heads * 16 + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, jobs,
                           order, offsets, table_rows, window_size * window_size);
        HS_LAUNCH_CHECK("rel_bias_scatter_grad_sorted_many");
    }
    return HS_OK;
}

int hs_cos_head_scale_many(const void* const* logit_scale, const void* const* dscale, void* const* out, const int* heads, const int* accumulate,
                           int count, void* stream) {
    HS_CHECK_ARG(logit_scale && out && heads && count > 0, "hs_cos_head_scale_many: bad arguments");
    for (int j0 = 0; j0 < count; j0 += kManyMax) {
        ScaleJobs jobs{};
        const int n = count - j0 < kManyMax ? count - j0 : kManyMax;
        for (int j = 0; j < n; ++j) {
            HS_CHECK_ARG(logit_scale[j0 + j] && out[j0 + j] && heads[j0 + j] > 0 && heads[j0 + j] <= 64,
                         "hs_cos_head_scale_many: null pointer or head count outside [1, 64] (job %d)", j0 + j);
            jobs.ls[j] = (const float*)logit_scale[j0 + j];
            jobs.dscale[j] = dscale ? (const float*)dscale[j0 + j] : nullptr;
            jobs.out[j] = (float*)out[j0 + j];
            jobs.heads[j] = heads[j0 + j];
            jobs.accumulate[j] = accumulate ? accumulate[j0 + j] : 0;
        }
        hipLaunchKernelGGL(cos_scale_many_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, jobs);
        HS_LAUNCH_CHECK("cos_head_scale_many");
    }
    return HS_OK;
}

int hs_rel_bias_gather(const float* table, const int32_t* rel_idx, float* bias, int table_rows, int num_heads,
                       int window_size, void* stream) {
    HS_CHECK_ARG(table && rel_idx && bias && table_rows > 0 && num_heads > 0 && window_size > 0, "bad arguments");
    const int ws2 = window_size * window_size, n = num_heads * ws2;
    hipLaunchKernelGGL(rel_bias_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, table, rel_idx, bias,
                       num_heads, ws2);
    HS_LAUNCH_CHECK("rel_bias_gather");
    return HS_OK;
}

int hs_rel_bias_scatter_grad(const float* dbias, const int32_t* rel_idx, float* dtable, int table_rows, int num_heads,
                             int window_size, void* stream) {
    HS_CHECK_ARG(dbias && rel_idx && dtable && table_rows > 0 && num_heads > 0 && window_size > 0, "bad arguments");
    const int ws2 = window_size * window_size;
    const int nthr = ws2 >= 256 ? 256 : 64;
    hipLaunchKernelGGL(rel_bias_scatter_grad_kernel, dim3(table_rows, num_heads), dim3(nthr), 0, (hipStream_t)stream, dbias,
                       rel_idx, dtable, num_heads, ws2);
    HS_LAUNCH_CHECK("rel_bias_scatter_grad");
    return HS_OK;
}

int hs_rel_bias_scatter_grad_sorted(const float* dbias, const int32_t* order, const int32_t* offsets, float* dtable,
                                    int table_rows, int num_heads, int window_size, void* stream) {
    HS_CHECK_ARG(dbias && order && offsets && dtable && table_rows > 0 && num_heads > 0 && window_size > 0, "bad arguments");
    const int n = table_rows * num_heads * 16;
    hipLaunchKernelGGL(rel_bias_scatter_grad_sorted_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dbias, order,
                       offsets, dtable, table_rows, num_heads, window_size * window_size, 0);
    HS_LAUNCH_CHECK("rel_bias_scatter_grad_sorted");
    return HS_OK;
}

int hs_rel_bias_scatter_grad_sorted_add(const float* dbias, const int32_t* order, const int32_t* offsets, float* dtable,
                                        int table_rows, int num_heads, int window_size, void* stream) {
    HS_CHECK_ARG(dbias && order && offsets && dtable && table_rows > 0 && num_heads > 0 && window_size > 0, "bad arguments");
    const int n = table_rows * num_heads * 16;
    hipLaunchKernelGGL(rel_bias_scatter_grad_sorted_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dbias, order,
                       offsets, dtable, table_rows, num_heads, window_size * window_size, 1);
    HS_LAUNCH_CHECK("rel_bias_scatter_grad_sorted_add");
    return HS_OK;
}

int hs_cos_head_scale_fwd(const float* logit_scale, float* scale, int num_heads, void* stream) {
    HS_CHECK_ARG(logit_scale && scale && num_heads > 0, "hs_cos_head_scale_fwd: bad arguments");
    hipLaunchKernelGGL(cos_scale_kernel, dim3((num_heads + 63) / 64), dim3(64), 0, (hipStream_t)stream, logit_scale, (const float*)nullptr,
                       scale, num_heads, 0);
    HS_LAUNCH_CHECK("cos_head_scale_fwd");
    return HS_OK;
}

int hs_cos_head_scale_bwd(const float* logit_scale, const float* dscale, float* dlogit_scale, int num_heads, int accumulate, void* stream) {
    HS_CHECK_ARG(logit_scale && dscale && dlogit_scale && num_heads > 0, "hs_cos_head_scale_bwd: bad arguments");
    hipLaunchKernelGGL(cos_scale_kernel, dim3((num_heads + 63) / 64), dim3(64), 0, (hipStream_t)stream, logit_scale, dscale, dlogit_scale,
                       num_heads, accumulate);
    HS_LAUNCH_CHECK("cos_head_scale_bwd");
    return HS_OK;
}

}  // extern "C"
