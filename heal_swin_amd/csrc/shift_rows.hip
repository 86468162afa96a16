// Standalone row gather along the nested pixel axis:  out[b, j, :] = x[b, idx[j], :]
// (or the roll  out[b, j, :] = x[b, (j + roll) mod N, :]  when idx is NULL).
//
// Replaces shifter.shift / shift_back when they are called on their own
// (models_torch/hp_shifting.py:69-73 torch.roll, :302-306 and :400-404 x[:, idcs].contiguous()).
// Inside the model the permutation is fused into the window-attention kernel and this kernel is not on
// the path; it exists for the reference's shifter API and for the gather/scatter bandwidth measurement
// of BASELINE config 4.  Pure HBM traffic: every row is one contiguous run of C*elt bytes, moved with
// 16-byte accesses; a wavefront moves (64*16)/row_bytes rows per instruction.
#include "hs_device.h"

namespace {

__global__ void __launch_bounds__(256) gather_rows_kernel(const uint4* __restrict__ x, uint4* __restrict__ out,
                                                          const int32_t* __restrict__ idx, int64_t roll, int64_t n_tokens,
                                                          int64_t total_rows, int vec_per_row) {
    const int64_t nvec = total_rows * vec_per_row;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = v / vec_per_row;
        const int c = (int)(v - row * vec_per_row);
        const int64_t b = row / n_tokens, j = row - b * n_tokens;
        int64_t s;
        if (idx) s = idx[j];
        else {
            s = j + roll;
            if (s >= n_tokens) s -= n_tokens;
        }
        out[v] = x[(b * n_tokens + s) * vec_per_row + c];
    }
}

__global__ void gather_rows_bytes_kernel(const uint8_t* __restrict__ x, uint8_t* __restrict__ out,
                                         const int32_t* __restrict__ idx, int64_t roll, int64_t n_tokens, int64_t total_rows,
                                         int row_bytes) {
    const int64_t n = total_rows * row_bytes;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n; v += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = v / row_bytes;
        const int c = (int)(v - row * row_bytes);
        const int64_t b = row / n_tokens, j = row - b * n_tokens;
        int64_t s = idx ? (int64_t)idx[j] : (j + roll) % n_tokens;
        out[v] = x[(b * n_tokens + s) * row_bytes + c];
    }
}

}  // namespace

extern "C" int hs_gather_rows(const void* x, void* out, const int32_t* idx, int64_t roll, int batch, int64_t n_tokens,
                              int64_t row_bytes, void* stream) {
    HS_CHECK_ARG(x && out && x != out, "x and out must be distinct non-null buffers");
    HS_CHECK_ARG(batch > 0 && n_tokens > 0 && row_bytes > 0 && row_bytes < (1ll << 30), "bad shape");
    HS_CHECK_ARG(roll >= 0 && roll < n_tokens, "roll must be in [0, n_tokens)");
    const int64_t rows = (int64_t)batch * n_tokens;
    hipStream_t s = (hipStream_t)stream;
    if (row_bytes % 16 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0)) {
        const int vpr = (int)(row_bytes / 16);
        int64_t blocks = (rows * vpr + 255) / 256;
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const uint4*)x, (uint4*)out, idx, roll,
                           n_tokens, rows, vpr);
    } else {
        int64_t blocks = (rows * row_bytes + 255) / 256;
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(gather_rows_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const uint8_t*)x, (uint8_t*)out,
                           idx, roll, n_tokens, rows, (int)row_bytes);
    }
    HS_LAUNCH_CHECK("gather_rows");
    return HS_OK;
}
