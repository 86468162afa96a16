// Batched 2-D transposes of 16-bit matrices in ONE launch: the [in, out] copies of the Linear weights that the input-gradient
// products of hs_gemm_nt take as their B operand are re-made after every optimizer step (ops.ParamCastCache.get_t); one
// aten copy_ per weight was ~90 launches of a few microseconds each per step (1.9 ms of GPU time per HEAL-SWIN-B step with the
// other casts, 1.3 ms of host time per step on the launch-bound HEAL-SWIN-T / nside 128 workload).
#include "hs_device.h"

namespace hs {
namespace {

struct TransposeJob {
    const uint16_t* src;  // [rows, cols] row-major
    uint16_t* dst;        // [cols, rows] row-major
    int64_t rows, cols;
};

__global__ void __launch_bounds__(256) transpose_many_kernel(const TransposeJob* __restrict__ jobs) {
    __shared__ uint16_t tile[32][33];
    const TransposeJob j = jobs[blockIdx.y];
    const int64_t tiles_c = (j.cols + 31) / 32, tiles = ((j.rows + 31) / 32) * tiles_c;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8 threads, 4 rows each
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int64_t r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t r = r0 + ty + 8 * i, c = c0 + tx;
            if (r < j.rows && c < j.cols) tile[ty + 8 * i][tx] = j.src[r * j.cols + c];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t c = c0 + ty + 8 * i, r = r0 + tx;
            if (r < j.rows && c < j.cols) j.dst[c * j.rows + r] = tile[tx][ty + 8 * i];
        }
        __syncthreads();
    }
}

}  // namespace
}  // namespace hs

extern "C" {

int hs_transpose_many_16(const void* jobs, int count, int blocks_per_job, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(jobs && count > 0 && count <= 65535 && blocks_per_job > 0, "hs_transpose_many_16: bad arguments");
    hipLaunchKernelGGL(transpose_many_kernel, dim3((unsigned)blocks_per_job, (unsigned)count), dim3(256), 0, (hipStream_t)stream,
                       (const TransposeJob*)jobs);
    HS_LAUNCH_CHECK("transpose_many");
    return HS_OK;
}

}  // extern "C"
