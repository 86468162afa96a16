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

}  // namespace

extern "C" {

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

}  // extern "C"
