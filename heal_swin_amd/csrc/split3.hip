// bf16 x 3 products for the fp32 path (the reference's own precision, training/train_config.py:95 `precision: int = 32`).
// gfx950 has no reduced-precision fast path for fp32 operands (no xf32 MFMA): v_mfma_f32_32x32x2_f32 runs at the fp32 vector
// rate, 1/16 of the bf16 MFMA rate.  An fp32 value is hi + lo + r with hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-17 |x|, so
//     a . b  ~=  a_hi . b_hi + a_hi . b_lo + a_lo . b_hi          (dropped: a_lo . b_lo and the residuals, ~1e-5 relative)
// is ONE bf16 product of three-fold depth over the concatenated operands
//     A' = [a_hi | a_hi | a_lo]   (mode 0, activations)        B' = [b_hi | b_lo | b_hi]   (mode 1, weights)
// accumulated in fp32: 3/16 of the fp32-MFMA time.  This kernel writes A' / B' in one pass (16-byte reads, 8-byte writes).
#include "hs_device.h"

namespace hs {
namespace {

__global__ void __launch_bounds__(256) split_bf16x3_kernel(const float* __restrict__ x, uint16_t* __restrict__ out, int64_t rows, int k,
                                                           int mode) {
    const int kq = k >> 2;  // float4 groups per row
    const int64_t total = rows * kq;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / kq;
        const int c = (int)(i - row * kq) * 4;
        const float4 v = *(const float4*)(x + row * k + c);
        const uint32_t h0 = pack_bf16x2(v.x, v.y), h1 = pack_bf16x2(v.z, v.w);
        const float rx = v.x - __uint_as_float(h0 << 16), ry = v.y - __uint_as_float(h0 & 0xffff0000u);
        const float rz = v.z - __uint_as_float(h1 << 16), rw = v.w - __uint_as_float(h1 & 0xffff0000u);
        const uint2 hi = make_uint2(h0, h1), lo = make_uint2(pack_bf16x2(rx, ry), pack_bf16x2(rz, rw));
        uint16_t* o = out + row * 3 * (int64_t)k + c;
        *(uint2*)o = hi;
        *(uint2*)(o + k) = mode == 0 ? hi : lo;
        *(uint2*)(o + 2 * k) = mode == 0 ? lo : hi;
    }
}

}  // namespace
}  // namespace hs

extern "C" {

int hs_split_bf16x3(const float* x, void* out, int64_t rows, int k, int mode, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(x && out, "hs_split_bf16x3: null pointer");
    HS_CHECK_ARG(rows > 0 && k > 0 && k % 4 == 0, "hs_split_bf16x3: k must be a positive multiple of 4");
    HS_CHECK_ARG(mode == 0 || mode == 1, "hs_split_bf16x3: mode 0 ([hi|hi|lo]) or 1 ([hi|lo|hi])");
    const int64_t total = rows * (k / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)out, rows, k, mode);
    HS_LAUNCH_CHECK("split_bf16x3");
    return HS_OK;
}

}  // extern "C"
