// Decoder tail in ONE forward kernel: FinalPatchExpand_X4's Linear(C -> 4C), the 'b n (p c) -> b (n p) c' view, its
// LayerNorm(C) over every pixel row and the 1x1 class head:
//
//     logits[(n, p), k] = sum_c LN_c( sum_j xn[n, j] Wexp[p C + c, j] ) (gamma_c Whead[k, c]) + sum_c beta_c Whead[k, c]
//
// Reference: models_torch/swin_hp_transformer.py:442-452 (`self.expand`, rearrange, `self.norm`) and :785-788 (`self.output`).
// Why fused: (1) parity -- the four bf16 roundings of the tail (norm_up output, expand output, xhat, logits) are not averaged by
// anything downstream and make up 6.4e-3 of the 7.7e-3 logit error of HEAL-SWIN-B, the whole rest of the network 2.9e-3
// (tests/experiments/bf16_error_budget.py); here LayerNorm sees the fp32 accumulators of the expand product, xhat enters the
// head as hi + lo and the logits leave in fp32; with xn_lo (the rounding remainder of the norm_up output) not even that of
// the input remains; (2) traffic -- the [B, 4 N0, C] tensor
// (1.6 GB at nside 256, batch 8) is written at most once (training: the backward's LayerNorm input) and never read by the
// forward; without a gradient it does not exist at all.
//
// Workgroup = 4 wavefronts, persistent, one per CU: Wexp (4C x C bf16, 128 KB at C = 128) stays in LDS for the whole launch.
// A wavefront owns 32 tokens per step: lane (l31, half) holds the 16-byte chunks 2 s + half of token l31's row -- the B operand
// of v_mfma_f32_32x32x16_bf16 with the token on the accumulator's lane axis -- so per child p the product D = Wexp_p xn^T
// (A = weight rows from LDS) leaves the child's C channels of a token in ONE lane pair: LayerNorm statistics are in-register
// sums plus one lane^32 exchange, the normalised registers are (after packing) the B operand of the head product, and the
// logits of the row land in registers 0..7 of the same lane pair.  32 C / 16 + 4 C / 16 MFMAs per 32 pixel rows.
#include "hs_device.h"

namespace hs {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr float kEps = 1e-5f;
constexpr int kKP = 16;     // class columns of the padded logits row
constexpr int kP = 4;       // children per token (patch_size 4: every BASELINE config)
constexpr int kRowB = 256;  // bytes per weight row in LDS (C <= 128 bf16; rows of C = 96 / 64 are padded)
constexpr int kPatchRow = 128;

__device__ __forceinline__ uint4 pack8f(const float* f) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// Optional fused loss (the caller's CrossEntropyLoss(weight), models_lightning/segmentation/model_lightning_swin_hp.py:
// 39-45, 104-111): the 16 logits of a pixel row sit in one lane pair, so log-sum-exp, the label's logit and the row's weighted
// term are eight registers + one lane^32 exchange away; with `logits == nullptr` the [rows, 16] fp32 tensor is never written.
struct TailCe {
    const uint8_t* labels;  // [rows] class ids in pixel order (rows = (token, child)); ids >= n_classes are ignored (weight 0)
    const float* class_w;   // [n_classes] or null (all ones)
    float* loss_part;       // [4 * gridDim.x][2]: per-wave sums of w (lse - logit_y) and of w
    int n_classes;
};

template <int NB>
__global__ void __launch_bounds__(256, 1) expand_ln_head_fwd_kernel(const uint16_t* __restrict__ xn, const uint16_t* __restrict__ xn_lo,
                                                                    const uint16_t* __restrict__ wexp,
                                                                    const uint16_t* __restrict__ wfold, const float* __restrict__ bvec,
                                                                    uint16_t* __restrict__ y, float* __restrict__ logits,
                                                                    float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                    int64_t tokens, TailCe ce) {
    constexpr int C = 32 * NB, KS = 2 * NB, NCH = C / 8;  // channels (= input width), 16-deep k-steps, 16-byte chunks per row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wl = smem;                                  // [kP * C][kRowB], 16-byte chunk ^ (row & 15)
    unsigned char* patch = smem + kP * C * kRowB + (threadIdx.x >> 6) * (32 * kPatchRow);  // per wave: [32 tokens][64 channels] bf16
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;

    // ---------------------------------------------------------------- one-off: Wexp -> LDS
    for (int q = tid; q < kP * C * NCH; q += 256) {
        const int row = q / NCH, ch = q % NCH;
        *(uint4*)(wl + row * kRowB + ((ch ^ (row & 15)) << 4)) = *(const uint4*)(wexp + (int64_t)row * C + ch * 8);
    }
    // folded head weight (gamma * Whead) as A operands: accumulator register 8 j + i of channel tile ct is channel
    // 32 ct + 16 j + 8 (i >> 2) + 4 half + (i & 3): two 8-byte pieces per fragment
    // wfold holds 64 rows: 0..31 the bf16 rounding of gamma * Whead, 32..63 its rounding remainder (hi + lo = the fp32 product
    // to 16 bits: the head weights are the last rounding between norm_up and the logits)
    bf16x8 wfa[NB][2], wfl[NB][2];
#pragma unroll
    for (int ct = 0; ct < NB; ++ct)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint16_t* src = wfold + l31 * C + 32 * ct + 16 * j + 4 * half;
            const uint2 a = *(const uint2*)src, b = *(const uint2*)(src + 8);
            wfa[ct][j] = __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y));
            const uint2 al = *(const uint2*)(src + 32 * C), bl = *(const uint2*)(src + 32 * C + 8);
            wfl[ct][j] = __builtin_bit_cast(bf16x8, make_uint4(al.x, al.y, bl.x, bl.y));
        }
    float bk[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) bk[r] = bvec[4 * half + (r & 3) + 8 * (r >> 2)];
    __syncthreads();

    const int sx = l31 & 15;
    const float inv_c = 1.f / (float)C;
    float ce_num = 0.f, ce_den = 0.f;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (tid >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t tok0 = wave * 32; tok0 < tokens; tok0 += nwaves * 32) {
        const int64_t tok = tok0 + l31;
        const bool live = tok < tokens;
        bf16x8 xb[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uint4 v = live ? *(const uint4*)(xn + tok * C + 16 * ks + 8 * half) : make_uint4(0, 0, 0, 0);
            xb[ks] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll 1
        for (int p = 0; p < kP; ++p) {
            // ------------------------------------------------------------ the child's C channels of 32 tokens: D = Wexp_p xn^T
            f32x16 acc[NB];
#pragma unroll
            for (int ct = 0; ct < NB; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
            const unsigned char* wrow = wl + (p * C + l31) * kRowB;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int co = ((2 * ks + half) ^ sx) << 4;
#pragma unroll
                for (int ct = 0; ct < NB; ++ct) {
                    const bf16x8 a = *(const bf16x8*)(wrow + ct * 32 * kRowB + co);
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xb[ks], acc[ct], 0, 0, 0);
                }
            }
            if (xn_lo) {  // the rounding remainder of the norm_up output as a second operand (re-fetched per child: L1 / L2 hits)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const uint4 v = live ? *(const uint4*)(xn_lo + tok * C + 16 * ks + 8 * half) : make_uint4(0, 0, 0, 0);
                    const int co = ((2 * ks + half) ^ sx) << 4;
#pragma unroll
                    for (int ct = 0; ct < NB; ++ct) {
                        const bf16x8 a = *(const bf16x8*)(wrow + ct * 32 * kRowB + co);
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, v), acc[ct], 0, 0, 0);
                    }
                }
            }
            // ------------------------------------------------------------ the expanded rows, once, for the backward (training)
            const int64_t orow = tok * kP + p;
            if (y) {
#pragma unroll
                for (int pr = 0; pr < (NB + 1) / 2; ++pr) {  // 64 channels (one 128-byte line per row) at a time
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int ct = 2 * pr + t;
                        if (ct < NB) {
#pragma unroll
                            for (int g = 0; g < 4; ++g) {  // registers 4 g .. 4 g + 3 = channels 32 ct + 8 g + 4 half + 0..3
                                const int chunk = 4 * t + g;  // 16-byte chunk inside the 128-byte patch row
                                *(uint2*)(patch + l31 * kPatchRow + ((chunk ^ (l31 & 7)) << 4) + 8 * half) =
                                    make_uint2(pack_bf16x2(acc[ct][4 * g], acc[ct][4 * g + 1]), pack_bf16x2(acc[ct][4 * g + 2], acc[ct][4 * g + 3]));
                            }
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    constexpr int kChunksLast = (NB % 2) ? 4 : 8;
                    const int nchunks = (2 * pr + 1 < NB) ? 8 : kChunksLast;  // valid 16-byte chunks of this pass
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int q = lane + 64 * i, row = q >> 3, chunk = q & 7;
                        const uint4 v = *(const uint4*)(patch + row * kPatchRow + ((chunk ^ (row & 7)) << 4));
                        if (chunk < nchunks && tok0 + row < tokens)
                            *(uint4*)(y + ((tok0 + row) * kP + p) * C + 64 * pr + chunk * 8) = v;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            // ------------------------------------------------------------ LayerNorm statistics of each row (lane pair)
            float sum = 0.f;
#pragma unroll
            for (int ct = 0; ct < NB; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[ct][r];
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * inv_c;
            float sq = 0.f;
#pragma unroll
            for (int ct = 0; ct < NB; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[ct][r] -= mean;
                    sq = fmaf(acc[ct][r], acc[ct][r], sq);
                }
            sq += __shfl_xor(sq, 32, 64);
            const float rstd = rsqrtf(sq * inv_c + kEps);
            // ------------------------------------------------------------ head: logits^T = (gamma W) xhat^T, xhat = hi + lo
            f32x16 lg;
#pragma unroll
            for (int r = 0; r < 16; ++r) lg[r] = 0.f;
#pragma unroll
            for (int ct = 0; ct < NB; ++ct)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float xh[8], lo[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) xh[i] = acc[ct][8 * j + i] * rstd;
                    const uint4 hb = pack8f(xh);
                    const uint32_t hw[4] = {hb.x, hb.y, hb.z, hb.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        lo[2 * i] = xh[2 * i] - bf_lo(hw[i]);
                        lo[2 * i + 1] = xh[2 * i + 1] - bf_hi(hw[i]);
                    }
                    lg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfa[ct][j], __builtin_bit_cast(bf16x8, hb), lg, 0, 0, 0);
                    lg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfa[ct][j], __builtin_bit_cast(bf16x8, pack8f(lo)), lg, 0, 0, 0);
                    lg = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfl[ct][j], __builtin_bit_cast(bf16x8, hb), lg, 0, 0, 0);
                }
            if (ce.labels) {  // weighted cross-entropy of the row, from the fp32 logits in registers
                constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
                float v[8];
                float m = -INFINITY;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    v[r] = lg[r] + bk[r];
                    if (4 * half + (r & 3) + 8 * (r >> 2) < ce.n_classes) m = fmaxf(m, v[r]);
                }
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                const int yl = live ? (int)ce.labels[orow] : 255;
                float ssum = 0.f, pick = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int cls = 4 * half + (r & 3) + 8 * (r >> 2);
                    if (cls < ce.n_classes) ssum += __builtin_amdgcn_exp2f((v[r] - m) * kLog2e);
                    pick = cls == yl ? v[r] : pick;
                }
                ssum += __shfl_xor(ssum, 32, 64);
                pick += __shfl_xor(pick, 32, 64);  // (the other half holds 0)
                const float wy = yl < ce.n_classes ? (ce.class_w ? ce.class_w[yl] : 1.f) : 0.f;
                if (half == 0) {
                    ce_num = fmaf(wy, m + __builtin_amdgcn_logf(ssum) * kLn2 - pick, ce_num);
                    ce_den += wy;
                }
            }
            if (live) {
                // accumulator register r = class 4 half + (r & 3) + 8 (r >> 2) of this lane's row: classes 0..15 are r = 0..7
                if (logits) {
                    *(float4*)(logits + orow * kKP + 4 * half) = make_float4(lg[0] + bk[0], lg[1] + bk[1], lg[2] + bk[2], lg[3] + bk[3]);
                    *(float4*)(logits + orow * kKP + 8 + 4 * half) = make_float4(lg[4] + bk[4], lg[5] + bk[5], lg[6] + bk[6], lg[7] + bk[7]);
                }
                if (half == 0 && mean_out) {
                    mean_out[orow] = mean;
                    rstd_out[orow] = rstd;
                }
            }
        }
    }
    if (ce.labels) {  // every wave writes its pair (zeros if it owned no rows): the host sums the array
        const float n = wave_sum(ce_num), d = wave_sum(ce_den);
        if (lane == 0) {
            ce.loss_part[2 * wave] = n;
            ce.loss_part[2 * wave + 1] = d;
        }
    }
}

}  // namespace
}  // namespace hs

extern "C" {

int hs_expand_ln_head_supported(int width, int children, int n_classes, int dtype) {
    return dtype == HS_BF16 && children == hs::kP && width % 32 == 0 && width >= 64 && width <= 128 && n_classes >= 1 && n_classes <= 16;
}

namespace {
int launch_expand_ln_head(const void* xn, const void* xn_lo, const void* wexp, const void* wfold, const float* bvec, void* y, float* logits,
                          float* mean, float* rstd, int64_t tokens, int width, int children, int dtype, void* stream, hs::TailCe ce,
                          const char* who) {
    using namespace hs;
    HS_CHECK_ARG(xn && wexp && wfold && bvec, "%s: null pointer", who);
    HS_CHECK_ARG(tokens > 0, "%s: bad shape", who);
    HS_CHECK_ARG((y == nullptr) == (mean == nullptr) && (mean == nullptr) == (rstd == nullptr),
                 "%s: y, mean and rstd (what the backward needs) go together", who);
    if (!hs_expand_ln_head_supported(width, children, 1, dtype))
        return fail(HS_ERR_UNSUPPORTED, "%s: bf16, 4 children, C in {64, 96, 128} (got C = %d, children %d): the expand "
                    "weight must fit the LDS", who, width, children);
    const int nb = width / 32;
    const size_t smem = (size_t)kP * width * kRowB + 4 * 32 * kPatchRow;
    const dim3 grid((unsigned)hs_expand_ln_head_blocks(tokens)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define HS_ELH(NB)                                                                                                                  \
    case NB: {                                                                                                                      \
        auto kern = expand_ln_head_fwd_kernel<NB>;                                                                                 \
        static bool configured = false;                                                                                             \
        if (!configured) {                                                                                                          \
            HS_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));           \
            configured = true;                                                                                                      \
        }                                                                                                                           \
        hipLaunchKernelGGL(kern, grid, block, smem, s, (const uint16_t*)xn, (const uint16_t*)xn_lo, (const uint16_t*)wexp, (const uint16_t*)wfold, bvec, \
                           (uint16_t*)y, logits, mean, rstd, tokens, ce);                                                          \
    } break;
    switch (nb) {
        HS_ELH(2) HS_ELH(3) HS_ELH(4)
    }
#undef HS_ELH
    HS_LAUNCH_CHECK("expand_ln_head_fwd");
    return HS_OK;
}
}  // namespace

int64_t hs_expand_ln_head_blocks(int64_t tokens) {
    int64_t blocks = (tokens + 127) / 128;  // 4 waves x 32 tokens per workgroup and step
    const int cus = hs::usable_cus();
    if (blocks > cus) blocks = cus;
    return blocks < 1 ? 1 : blocks;
}

int hs_expand_ln_head_fwd(const void* xn, const void* xn_lo, const void* wexp, const void* wfold, const float* bvec, void* y, float* logits,
                          float* mean, float* rstd, int64_t tokens, int width, int children, int dtype, void* stream) {
    HS_CHECK_ARG(logits, "hs_expand_ln_head_fwd: null pointer");
    return launch_expand_ln_head(xn, xn_lo, wexp, wfold, bvec, y, logits, mean, rstd, tokens, width, children, dtype, stream,
                                 hs::TailCe{nullptr, nullptr, nullptr, 0}, "hs_expand_ln_head_fwd");
}

int hs_expand_ln_head_ce_fwd(const void* xn, const void* xn_lo, const void* wexp, const void* wfold, const float* bvec, const uint8_t* labels,
                             const float* class_weights, int n_classes, void* y, float* logits, float* mean, float* rstd,
                             float* loss_partials, int64_t tokens, int width, int children, int dtype, void* stream) {
    HS_CHECK_ARG(labels && loss_partials, "hs_expand_ln_head_ce_fwd: null pointer");
    HS_CHECK_ARG(n_classes >= 1 && n_classes <= 16, "hs_expand_ln_head_ce_fwd: 1..16 classes");
    return launch_expand_ln_head(xn, xn_lo, wexp, wfold, bvec, y, logits, mean, rstd, tokens, width, children, dtype, stream,
                                 hs::TailCe{labels, class_weights, loss_partials, n_classes}, "hs_expand_ln_head_ce_fwd");
}

}  // extern "C"
