// Fisheye image -> HEALPix sampling, the per-image part of the reference's data preparation
// (heal_swin/data/segmentation/project_on_s2.py:344-372):
//     hp_img  = sample_bilinear(img, v, u).astype(np.uint8)      (:38-73)
//     hp_mask = sample_mask(mask, v, u, s2_bkgd_class)           (:76-80)
// for every image of a camera, against ONE coordinate table (u, v) per calibration (the reference caches it, :141-183).
// One thread per HEALPix pixel: the coordinate arithmetic is done once and applied to all batch x channel planes.
//
// Bit-exact by construction: float64 throughout, the reference's operation order
//     fx1 = (i1 - r) s00 + (r - i0) s10;  fx2 = (i1 - r) s01 + (r - i0) s11;  out = (j1 - q) fx1 + (q - j0) fx2
// with i0 / i1 = floor / ceil (so BOTH weights vanish at integer coordinates and the sample is 0, as in the reference),
// out-of-image neighbours contributing 0, truncation to uint8 -- and no FMA contraction (a fused multiply-add rounds once
// where numpy rounds twice; in constant image regions that decides between c and c - 1 after the truncation).
// HBM-bound gather: 2 x 8 B of coordinates + batch x channels x (4 neighbour bytes from L2, 1 byte out) per pixel.
#include "hs_device.h"

#pragma clang fp contract(off)

namespace {

struct Corner {
    bool ok;
    int64_t off;
};
__device__ __forceinline__ Corner corner(double fx, double fy, int h, int w) {
    // numpy's bounds test on the integer casts; doubles compare the same way, NaN and +-inf fail it
    const bool ok = fx >= 0.0 && fx < (double)h && fy >= 0.0 && fy < (double)w;
    return {ok, ok ? (int64_t)fx * w + (int64_t)fy : 0};
}

__global__ void __launch_bounds__(256) sample_bilinear_u8_kernel(const uint8_t* __restrict__ img, int planes, int h, int w,
                                                                 const double* __restrict__ rx, const double* __restrict__ ry,
                                                                 int64_t n, uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = rx[i], y = ry[i];
    const double x0 = floor(x), x1 = ceil(x), y0 = floor(y), y1 = ceil(y);
    const Corner c00 = corner(x0, y0, h, w), c10 = corner(x1, y0, h, w), c01 = corner(x0, y1, h, w), c11 = corner(x1, y1, h, w);
    const double wx0 = x1 - x, wx1 = x - x0, wy0 = y1 - y, wy1 = y - y0;
    const bool finite = (x - x == 0.0) && (y - y == 0.0);  // NaN / inf coordinates: the reference's NaN result casts to 0
    const int64_t plane = (int64_t)h * w;
    for (int p = 0; p < planes; ++p) {
        const uint8_t* s = img + p * plane;
        const double s00 = c00.ok ? (double)s[c00.off] : 0.0, s10 = c10.ok ? (double)s[c10.off] : 0.0;
        const double s01 = c01.ok ? (double)s[c01.off] : 0.0, s11 = c11.ok ? (double)s[c11.off] : 0.0;
        const double fx1 = wx0 * s00 + wx1 * s10;
        const double fx2 = wx0 * s01 + wx1 * s11;
        const double r = wy0 * fx1 + wy1 * fx2;
        out[p * n + i] = finite ? (uint8_t)(int)r : (uint8_t)0;
    }
}

__global__ void __launch_bounds__(256) sample_mask_u8_kernel(const uint8_t* __restrict__ mask, int planes, int h, int w,
                                                             const double* __restrict__ rx, const double* __restrict__ ry,
                                                             int64_t n, int background, uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Corner c = corner(rint(rx[i]), rint(ry[i]), h, w);  // np.around: round half to even
    const int64_t plane = (int64_t)h * w;
    for (int p = 0; p < planes; ++p) out[p * n + i] = c.ok ? mask[p * plane + c.off] : (uint8_t)background;
}

}  // namespace

extern "C" {

int hs_sample_bilinear_u8(const void* img, int batch, int channels, int height, int width, const double* rx, const double* ry,
                          int64_t n, void* out, void* stream) {
    HS_CHECK_ARG(batch > 0 && channels > 0 && height > 0 && width > 0 && n >= 0, "bad shape");
    HS_CHECK_ARG((int64_t)batch * channels < (1 << 20), "too many image planes");
    if (n == 0) return HS_OK;
    HS_CHECK_ARG(img && rx && ry && out, "null pointer");
    hipLaunchKernelGGL(sample_bilinear_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)img, batch * channels, height, width, rx, ry, n, (uint8_t*)out);
    HS_LAUNCH_CHECK("sample_bilinear_u8");
    return HS_OK;
}

int hs_sample_mask_u8(const void* mask, int batch, int height, int width, const double* rx, const double* ry, int64_t n,
                      int background, void* out, void* stream) {
    HS_CHECK_ARG(batch > 0 && height > 0 && width > 0 && n >= 0, "bad shape");
    HS_CHECK_ARG(background >= 0 && background <= 255, "background class must fit uint8");
    if (n == 0) return HS_OK;
    HS_CHECK_ARG(mask && rx && ry && out, "null pointer");
    hipLaunchKernelGGL(sample_mask_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)mask, batch, height, width, rx, ry, n, background, (uint8_t*)out);
    HS_LAUNCH_CHECK("sample_mask_u8");
    return HS_OK;
}

}  // extern "C"
