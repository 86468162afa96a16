// Decoder tail: the LayerNorm of FinalPatchExpand_X4 and the 1x1 class head in ONE pass over the expanded
// rows, forward and backward, so that the normalised tensor [B, 4 N0, C] never exists in HBM.
// Reference: models_torch/swin_hp_transformer.py:448-452 (`self.norm(x)` of FinalPatchExpand_X4) and :785-788
// (`self.output(x)`, Conv1d(C, f_out, 1, bias=False)); both run on every pixel row (6.3 M rows at nside 256, batch 8).
//
//   forward   logits[row, k] = sum_c xhat[row, c] (gamma_c W[k, c]) + sum_c beta_c W[k, c],   xhat = (y - mean) rstd
//   backward  g[row, c] = sum_k dlogits[row, k] gamma_c W[k, c]          (= dL/dLN_out * gamma)
//             dy = rstd (g - mean_c g - xhat mean_c(g xhat))              (LayerNorm input gradient)
//             D'[row, k] = dlogits[row, k] rstd[row]   (bf16, written),   u[k] = sum_rows dlogits,  t[k] = sum_rows D' mean
//   The parameter gradients follow from ONE weight-gradient product over the raw rows, X[k, c] = sum_rows dlogits xhat =
//   hs_linear_wgrad(D', y)[k, c] - t[k]  (the host side, ops.LnHeadFn):  dW = gamma X + beta u,  dgamma_c = sum_k W X,
//   dbeta_c = sum_k W u.
//
// Layout: a wavefront owns 32 rows per step; lane (l31, half) holds row l31's 16-byte chunks 16 s + 8 half of its C-wide
// row -- which IS the B operand of v_mfma_f32_32x32x16_bf16 with the row index on the accumulator's lane axis.  Row
// statistics are therefore lane-local sums plus one exchange with lane ^ 32, the normalised chunks feed the MFMA straight
// from registers (A = the folded head weight, resident in registers), and the accumulator holds the row's classes
// (forward) or the row's g values in the same chunk order as the lane's y registers (backward: the rows of the A operand
// are permuted to make it so).  HBM-bound: forward reads C x 2 B and writes 32 B per row, backward reads C x 2 + 32 + 8 B
// and writes C x 2 + 32 B.
#include "hs_device.h"

namespace hs {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr float kEps = 1e-5f;  // nn.LayerNorm default, as everywhere in the reference
constexpr int kKP = 16;        // class columns of the padded logits row

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(w[i] << 16);
        f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}

// wfold [32][C] bf16: row k = gamma * W[k, :] (rows >= f_out zero); bvec [32] f32: sum_c beta_c W[k, c]
// F32OUT: the logits leave as fp32 rows of 16 (64 B) instead of bf16 (32 B).  The decoder tail is where bf16 rounding is NOT
// averaged away by anything downstream: the four roundings norm_up -> expand -> xhat -> logits account for 6.4e-3 of the 7.7e-3
// logit error of HEAL-SWIN-B (tests/experiments/bf16_error_budget.py), the whole rest of the network for 2.9e-3.  So here
// the logits keep their fp32 accumulator value and xhat enters the head product as hi + lo (two MFMAs per k-step instead of
// one: the kernel is HBM-bound, the second MFMA is free).
template <int NB, bool F32OUT>
__global__ void __launch_bounds__(256) ln_head_fwd_kernel(const uint16_t* __restrict__ y, const uint16_t* __restrict__ wfold,
                                                          const float* __restrict__ bvec, void* __restrict__ logits_v,
                                                          float* __restrict__ mean_out, float* __restrict__ rstd_out, int64_t rows) {
    constexpr int C = NB * 32, NS = NB * 2;
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    bf16x8 wa[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) wa[s] = *(const bf16x8*)(wfold + l31 * C + 16 * s + 8 * half);
    float bk[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) bk[r] = bvec[4 * half + (r & 3) + 8 * (r >> 2)];
    const float inv_c = 1.f / (float)C;
    for (int64_t row0 = wave * 32; row0 < rows; row0 += nwaves * 32) {
        const int64_t row = row0 + l31;
        const bool live = row < rows;
        float x[NS][8];
        float sum = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const uint4 v = live ? *(const uint4*)(y + row * C + 16 * s + 8 * half) : make_uint4(0, 0, 0, 0);
            unpack8(v, x[s]);
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += x[s][j];
        }
        sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * inv_c;
        float sq = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                x[s][j] -= mean;
                sq = fmaf(x[s][j], x[s][j], sq);
            }
        sq += __shfl_xor(sq, 32, 64);
        const float rstd = rsqrtf(sq * inv_c + kEps);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[s][j] *= rstd;
            const uint4 xb = pack8(x[s]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[s], __builtin_bit_cast(bf16x8, xb), acc, 0, 0, 0);
            if constexpr (F32OUT) {  // the rounding remainder of xhat as a second operand
                float hi[8], lo[8];
                unpack8(xb, hi);
#pragma unroll
                for (int j = 0; j < 8; ++j) lo[j] = x[s][j] - hi[j];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[s], __builtin_bit_cast(bf16x8, pack8(lo)), acc, 0, 0, 0);
            }
        }
        if (live) {
            // accumulator register r = class 4 half + (r & 3) + 8 (r >> 2) of this lane's row: classes 0..15 are r = 0..7
            if constexpr (F32OUT) {
                float* logits = (float*)logits_v;
                *(float4*)(logits + row * kKP + 4 * half) = make_float4(acc[0] + bk[0], acc[1] + bk[1], acc[2] + bk[2], acc[3] + bk[3]);
                *(float4*)(logits + row * kKP + 8 + 4 * half) = make_float4(acc[4] + bk[4], acc[5] + bk[5], acc[6] + bk[6], acc[7] + bk[7]);
            } else {
                uint16_t* logits = (uint16_t*)logits_v;
                uint2 o0 = make_uint2(pack_bf16x2(acc[0] + bk[0], acc[1] + bk[1]), pack_bf16x2(acc[2] + bk[2], acc[3] + bk[3]));
                uint2 o1 = make_uint2(pack_bf16x2(acc[4] + bk[4], acc[5] + bk[5]), pack_bf16x2(acc[6] + bk[6], acc[7] + bk[7]));
                *(uint2*)(logits + row * kKP + 4 * half) = o0;
                *(uint2*)(logits + row * kKP + 8 + 4 * half) = o1;
            }
            if (half == 0) {
                mean_out[row] = mean;
                rstd_out[row] = rstd;
            }
        }
    }
}

// afold [C][16] bf16: afold[c][k] = gamma_c W[k, c] (columns >= f_out zero); part [nwaves][32] f32: u[0..15], t[0..15]
template <int NB, bool F32IN>
__global__ void __launch_bounds__(256) ln_head_bwd_kernel(const uint16_t* __restrict__ y, const float* __restrict__ mean_in,
                                                          const float* __restrict__ rstd_in, const void* __restrict__ dlog_v,
                                                          const uint16_t* __restrict__ afold, uint16_t* __restrict__ dy,
                                                          uint16_t* __restrict__ dprime, float* __restrict__ part, int64_t rows) {
    constexpr int C = NB * 32, NS = NB * 2;
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    // A operand of block i: lane l31 = accumulator row rho; accumulator register r of lane half h is rho = 4 h + (r & 3) +
    // 8 (r >> 2) and has to be element r % 8 of the lane's chunk 2 i + r / 8, i.e. column c = 32 i + 16 (r / 8) + 8 h + r % 8
    bf16x8 aa[NB];
    {
        const int hh = (l31 >> 2) & 1, j4 = l31 & 3, q = l31 >> 3;
        const int c_in = 16 * (q >> 1) + 8 * hh + j4 + 4 * (q & 1);
#pragma unroll
        for (int i = 0; i < NB; ++i) aa[i] = *(const bf16x8*)(afold + (32 * i + c_in) * kKP + 8 * half);
    }
    float uacc[8], tacc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) uacc[j] = tacc[j] = 0.f;
    const float inv_c = 1.f / (float)C;
    for (int64_t row0 = wave * 32; row0 < rows; row0 += nwaves * 32) {
        const int64_t row = row0 + l31;
        const bool live = row < rows;
        const float mean = live ? mean_in[row] : 0.f, rstd = live ? rstd_in[row] : 0.f;
        uint4 dl = make_uint4(0, 0, 0, 0);
        float d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // this lane's 8 classes of the row's dlogits, un-rounded
        if (live) {
            if constexpr (F32IN) {
                const float4 a = *(const float4*)((const float*)dlog_v + row * kKP + 8 * half);
                const float4 b = *(const float4*)((const float*)dlog_v + row * kKP + 8 * half + 4);
                d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
                dl = pack8(d);
            } else {
                dl = *(const uint4*)((const uint16_t*)dlog_v + row * kKP + 8 * half);
                unpack8(dl, d);
            }
        }
        uint4 v[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) v[s] = live ? *(const uint4*)(y + row * C + 16 * s + 8 * half) : make_uint4(0, 0, 0, 0);
        f32x16 g[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) g[i][r] = 0.f;
            g[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[i], __builtin_bit_cast(bf16x8, dl), g[i], 0, 0, 0);
        }
        // D' = dlogits * rstd (bf16) and the two class sums, on this lane's 8 classes
        {
            float dp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) dp[j] = d[j] * rstd;
            const uint4 pk = pack8(dp);
            float dr[8];
            unpack8(pk, dr);  // the rounded values, as the weight-gradient kernel will read them
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uacc[j] += d[j];
                tacc[j] = fmaf(dr[j], mean, tacc[j]);
            }
            if (live) *(uint4*)(dprime + row * kKP + 8 * half) = pk;
        }
        float s1 = 0.f, s2 = 0.f;
        float xh[NS][8];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            unpack8(v[s], xh[s]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xh[s][j] = (xh[s][j] - mean) * rstd;
                const float gv = g[s >> 1][8 * (s & 1) + j];
                s1 += gv;
                s2 = fmaf(gv, xh[s][j], s2);
            }
        }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        const float m1 = s1 * inv_c, m2 = s2 * inv_c;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = rstd * (g[s >> 1][8 * (s & 1) + j] - m1 - xh[s][j] * m2);
            if (live) *(uint4*)(dy + row * C + 16 * s + 8 * half) = pack8(o);
        }
    }
    // class sums over the wave's rows (the 32 lanes of a half hold the same 8 classes)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            uacc[j] += __shfl_xor(uacc[j], off, 64);
            tacc[j] += __shfl_xor(tacc[j], off, 64);
        }
    }
    if (l31 == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            part[wave * 32 + 8 * half + j] = uacc[j];
            part[wave * 32 + 16 + 8 * half + j] = tacc[j];
        }
    }
}

// The backward with the loss fused in: instead of reading a [rows, 16] dlogits tensor the kernel recomputes the row's
// logits from the saved expanded rows (xhat = (y - mean) rstd as hi + lo through the folded head weight, 3 MFMAs per 16 channels
// in an HBM-bound kernel), takes softmax and the weighted cross-entropy gradient
//     dlogits[row, k] = scale w[y] (softmax_k - [k == y]),   scale = dloss / sum_rows w[y]
// in registers, and continues exactly as ln_head_bwd_kernel.  The rows of the folded weight arrive PERMUTED (blocks 4..7 and
// 8..11 exchanged, ops._fold_head_ce) so that accumulator register r < 8 of lane half h is class 8 h + r: the 8 contiguous
// classes of a lane are then directly the B operand of the g = dlogits (gamma W) product and the 16 bytes of D'.
template <int NB>
__global__ void __launch_bounds__(256) ln_head_ce_bwd_kernel(const uint16_t* __restrict__ y, const float* __restrict__ mean_in,
                                                             const float* __restrict__ rstd_in, const uint8_t* __restrict__ labels,
                                                             const float* __restrict__ class_w, const float* __restrict__ scale_ptr,
                                                             int n_classes, const uint16_t* __restrict__ wfold,
                                                             const float* __restrict__ bvec, const uint16_t* __restrict__ afold,
                                                             uint16_t* __restrict__ dy, uint16_t* __restrict__ dprime,
                                                             float* __restrict__ part, int64_t rows) {
    constexpr int C = NB * 32, NS = NB * 2;
    constexpr float kLog2e = 1.4426950408889634f;
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    bf16x8 wa[NS], wl[NS];  // folded head weight, hi and lo, rows permuted (see above)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        wa[s] = *(const bf16x8*)(wfold + l31 * C + 16 * s + 8 * half);
        wl[s] = *(const bf16x8*)(wfold + (32 + l31) * C + 16 * s + 8 * half);
    }
    float bk[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) bk[r] = bvec[4 * half + (r & 3) + 8 * (r >> 2)];
    bf16x8 aa[NB];
    {
        const int hh = (l31 >> 2) & 1, j4 = l31 & 3, q = l31 >> 3;
        const int c_in = 16 * (q >> 1) + 8 * hh + j4 + 4 * (q & 1);
#pragma unroll
        for (int i = 0; i < NB; ++i) aa[i] = *(const bf16x8*)(afold + (32 * i + c_in) * kKP + 8 * half);
    }
    const float scale = scale_ptr[0];
    float uacc[8], tacc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) uacc[j] = tacc[j] = 0.f;
    const float inv_c = 1.f / (float)C;
    for (int64_t row0 = wave * 32; row0 < rows; row0 += nwaves * 32) {
        const int64_t row = row0 + l31;
        const bool live = row < rows;
        const float mean = live ? mean_in[row] : 0.f, rstd = live ? rstd_in[row] : 0.f;
        const int yl = live ? (int)labels[row] : 255;
        uint4 v[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) v[s] = live ? *(const uint4*)(y + row * C + 16 * s + 8 * half) : make_uint4(0, 0, 0, 0);
        // ---- the row's logits again: xhat (hi + lo) through the folded head weight (hi + lo)
        float xh[NS][8];
        f32x16 lg;
#pragma unroll
        for (int r = 0; r < 16; ++r) lg[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            unpack8(v[s], xh[s]);
#pragma unroll
            for (int j = 0; j < 8; ++j) xh[s][j] = (xh[s][j] - mean) * rstd;
            const uint4 hb = pack8(xh[s]);
            float hi[8], lo[8];
            unpack8(hb, hi);
#pragma unroll
            for (int j = 0; j < 8; ++j) lo[j] = xh[s][j] - hi[j];
            lg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[s], __builtin_bit_cast(bf16x8, hb), lg, 0, 0, 0);
            lg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[s], __builtin_bit_cast(bf16x8, pack8(lo)), lg, 0, 0, 0);
            lg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[s], __builtin_bit_cast(bf16x8, hb), lg, 0, 0, 0);
        }
        // ---- softmax and the cross-entropy gradient on this lane's classes 8 half .. 8 half + 7
        float d[8];
        {
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                d[r] = lg[r] + bk[r];
                if (8 * half + r < n_classes) m = fmaxf(m, d[r]);
            }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float ssum = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                d[r] = 8 * half + r < n_classes ? __builtin_amdgcn_exp2f((d[r] - m) * kLog2e) : 0.f;
                ssum += d[r];
            }
            ssum += __shfl_xor(ssum, 32, 64);
            const float wy = yl < n_classes ? (class_w ? class_w[yl] : 1.f) : 0.f;
            const float coef = scale * wy, pinv = 1.f / ssum;
#pragma unroll
            for (int r = 0; r < 8; ++r) d[r] = coef * (d[r] * pinv - (8 * half + r == yl ? 1.f : 0.f));
        }
        const uint4 dl = pack8(d);
        f32x16 g[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) g[i][r] = 0.f;
            g[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[i], __builtin_bit_cast(bf16x8, dl), g[i], 0, 0, 0);
        }
        {
            float dp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) dp[j] = d[j] * rstd;
            const uint4 pk = pack8(dp);
            float dr[8];
            unpack8(pk, dr);  // the rounded values, as the weight-gradient kernel will read them
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uacc[j] += d[j];
                tacc[j] = fmaf(dr[j], mean, tacc[j]);
            }
            if (live) *(uint4*)(dprime + row * kKP + 8 * half) = pk;
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float gv = g[s >> 1][8 * (s & 1) + j];
                s1 += gv;
                s2 = fmaf(gv, xh[s][j], s2);
            }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        const float m1 = s1 * inv_c, m2 = s2 * inv_c;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = rstd * (g[s >> 1][8 * (s & 1) + j] - m1 - xh[s][j] * m2);
            if (live) *(uint4*)(dy + row * C + 16 * s + 8 * half) = pack8(o);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            uacc[j] += __shfl_xor(uacc[j], off, 64);
            tacc[j] += __shfl_xor(tacc[j], off, 64);
        }
    }
    if (l31 == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            part[wave * 32 + 8 * half + j] = uacc[j];
            part[wave * 32 + 16 + 8 * half + j] = tacc[j];
        }
    }
}

int grid_for(int64_t rows) {
    int64_t b = (rows + 127) / 128;  // 4 waves x 32 rows per workgroup and step
    if (b > 256 * 8) b = 256 * 8;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace
}  // namespace hs

extern "C" {

int hs_ln_head_supported(int width, int n_classes, int dtype) {
    return dtype == HS_BF16 && width % 32 == 0 && width >= 64 && width <= 256 && n_classes >= 1 && n_classes <= 16;
}

int64_t hs_ln_head_partials(int64_t rows) { return rows > 0 ? (int64_t)hs::grid_for(rows) * 4 : 0; }

int hs_ln_head_fwd(const void* y, const void* wfold, const float* bvec, void* logits, float* mean, float* rstd, int64_t rows,
                   int width, int dtype, int logits_dtype, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(y && wfold && bvec && logits && mean && rstd, "null pointer");
    HS_CHECK_ARG(rows > 0, "bad shape");
    HS_CHECK_ARG(logits_dtype == HS_BF16 || logits_dtype == HS_F32, "logits_dtype must be HS_BF16 or HS_F32");
    if (!hs_ln_head_supported(width, 1, dtype)) return fail(HS_ERR_UNSUPPORTED, "hs_ln_head: bf16 rows of 64..256 (multiple of 32) columns only");
    const dim3 grid(grid_for(rows)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define HS_LNH_FWD(NB)                                                                                                                   \
    case NB:                                                                                                                             \
        if (logits_dtype == HS_F32)                                                                                                      \
            hipLaunchKernelGGL((ln_head_fwd_kernel<NB, true>), grid, block, 0, s, (const uint16_t*)y, (const uint16_t*)wfold, bvec, logits, mean, rstd, rows); \
        else                                                                                                                             \
            hipLaunchKernelGGL((ln_head_fwd_kernel<NB, false>), grid, block, 0, s, (const uint16_t*)y, (const uint16_t*)wfold, bvec, logits, mean, rstd, rows); \
        break;
    switch (width / 32) {
        HS_LNH_FWD(2) HS_LNH_FWD(3) HS_LNH_FWD(4) HS_LNH_FWD(5) HS_LNH_FWD(6) HS_LNH_FWD(7) HS_LNH_FWD(8)
    }
#undef HS_LNH_FWD
    HS_LAUNCH_CHECK("ln_head_fwd");
    return HS_OK;
}

int hs_ln_head_bwd(const void* y, const float* mean, const float* rstd, const void* dlogits, const void* afold, void* dy,
                   void* dprime, float* partials, int64_t rows, int width, int dtype, int logits_dtype, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(y && mean && rstd && dlogits && afold && dy && dprime && partials, "null pointer");
    HS_CHECK_ARG(rows > 0, "bad shape");
    HS_CHECK_ARG(logits_dtype == HS_BF16 || logits_dtype == HS_F32, "logits_dtype must be HS_BF16 or HS_F32");
    if (!hs_ln_head_supported(width, 1, dtype)) return fail(HS_ERR_UNSUPPORTED, "hs_ln_head: bf16 rows of 64..256 (multiple of 32) columns only");
    const dim3 grid(grid_for(rows)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define HS_LNH_BWD(NB)                                                                                                                   \
    case NB:                                                                                                                             \
        if (logits_dtype == HS_F32)                                                                                                      \
            hipLaunchKernelGGL((ln_head_bwd_kernel<NB, true>), grid, block, 0, s, (const uint16_t*)y, mean, rstd, dlogits, (const uint16_t*)afold, (uint16_t*)dy, (uint16_t*)dprime, partials, rows); \
        else                                                                                                                             \
            hipLaunchKernelGGL((ln_head_bwd_kernel<NB, false>), grid, block, 0, s, (const uint16_t*)y, mean, rstd, dlogits, (const uint16_t*)afold, (uint16_t*)dy, (uint16_t*)dprime, partials, rows); \
        break;
    switch (width / 32) {
        HS_LNH_BWD(2) HS_LNH_BWD(3) HS_LNH_BWD(4) HS_LNH_BWD(5) HS_LNH_BWD(6) HS_LNH_BWD(7) HS_LNH_BWD(8)
    }
#undef HS_LNH_BWD
    HS_LAUNCH_CHECK("ln_head_bwd");
    return HS_OK;
}

int hs_ln_head_ce_bwd(const void* y, const float* mean, const float* rstd, const uint8_t* labels, const float* class_weights,
                      const float* scale, int n_classes, const void* wfold, const float* bvec, const void* afold, void* dy, void* dprime,
                      float* partials, int64_t rows, int width, int dtype, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(y && mean && rstd && labels && scale && wfold && bvec && afold && dy && dprime && partials, "null pointer");
    HS_CHECK_ARG(rows > 0, "bad shape");
    HS_CHECK_ARG(n_classes >= 1 && n_classes <= 16, "hs_ln_head_ce_bwd: 1..16 classes");
    if (!hs_ln_head_supported(width, n_classes, dtype) || width > 128)
        return fail(HS_ERR_UNSUPPORTED, "hs_ln_head_ce_bwd: bf16 rows of 64..128 (multiple of 32) columns only");
    const dim3 grid(grid_for(rows)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define HS_LNH_CE(NB)                                                                                                                   \
    case NB:                                                                                                                            \
        hipLaunchKernelGGL((ln_head_ce_bwd_kernel<NB>), grid, block, 0, s, (const uint16_t*)y, mean, rstd, labels, class_weights, scale, \
                           n_classes, (const uint16_t*)wfold, bvec, (const uint16_t*)afold, (uint16_t*)dy, (uint16_t*)dprime, partials, rows); \
        break;
    switch (width / 32) {
        HS_LNH_CE(2) HS_LNH_CE(3) HS_LNH_CE(4)
    }
#undef HS_LNH_CE
    HS_LAUNCH_CHECK("ln_head_ce_bwd");
    return HS_OK;
}

}  // extern "C"
