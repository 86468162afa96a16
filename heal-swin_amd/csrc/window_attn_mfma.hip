// MFMA path of the fused  shift -> window_partition -> attention -> window_reverse -> shift_back  op for the
// production shape: window = 64 tokens (8x8 nested block), head_dim = 32, bf16 activations (gfx950 / CDNA4).
//
// Work decomposition
//   workgroup = HG wavefronts = HG consecutive heads ("head group") of one window at a time; wave g owns head g.
//   HG*64 B of every token row's q (and k, v) slice are contiguous, so the workgroup's cooperative loads and
//   stores move whole 128-B lines (HG = 2/4) with 16 B per lane.  Workgroups are persistent: a fixed head group,
//   grid-striding over the B*nW windows, which keeps the head's relative-position bias (64x64 fp32, pre-multiplied
//   by log2 e) in REGISTERS for the whole launch, in exactly the accumulator layout of the score tile.
//
// Gather/scatter: window w, row i is token  idx[w*64+i]  (or (w*64+i+roll) mod N) of the UNshifted qkv tensor;
// the output row goes back to the same token, so shift and shift_back cost nothing beyond an index load.
//
// Per (window, head), all on one wave:
//   S^T = K Q^T          8 x v_mfma_f32_32x32x16_bf16.  Computing the TRANSPOSED scores puts a whole query row
//                        in one lane pair (lane l and l^32 hold the 64 keys of query l&31), so the row max / sum
//                        are in-register reductions plus one cross-half exchange -- no LDS, no 64-lane butterflies.
//   softmax              t = S^T * (scale*log2 e) + bias*log2 e (+ mask) -> exp2(t - max) / sum, fp32.
//   O = P V              8 x MFMA.  The accumulator registers of S^T are, after bf16 packing, directly the A operand
//                        (lane = query row, 8 key slots); the key order of those slots is mirrored on the V side.
// LDS images per head: Q and K row-major [64][32] bf16 with a 16-B-chunk XOR swizzle (conflict-free ds_read_b128
// for the MFMA A/B fragments); V TRANSPOSED [32 d][64 keys] (B fragments need 8 consecutive keys of one feature)
// with feature rows permuted and padded so both the 2-byte transposing writes and the 8-byte reads are conflict-free.
// Cosine attention: k rows are L2-normalised while being staged; the query norm and the head's logit scale are one
// per-lane factor applied to the fp32 scores.
#include "window_attn.h"

namespace hs {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kWs = 64, kHd = 32;
constexpr int kTileBytes = kWs * kHd * 2;  // 4096: [64][32] bf16, rows of 64 B = 4 chunks of 16 B
constexpr int kVtLd = 136;                 // bytes per feature row of the transposed tile (64 keys * 2 B + 8 pad)
constexpr int kVtStride = kHd * kVtLd + 32;  // 4384: (stride/4) % 32 == 8 spreads the heads over the banks
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kNormEps = 1e-12f;
constexpr float kMaskLog2 = -100.f * kLog2e;

__device__ __forceinline__ int swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }
// physical row of feature d inside the transposed tile
__device__ __forceinline__ int vt_row(int d) { return ((d & 7) << 2) + (d >> 3); }

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    return (uint32_t)float_to_bf16(a) | ((uint32_t)float_to_bf16(b) << 16);
}

struct LdsLayout {
    // [HG] Q tiles | [HG] K tiles | [HG] transposed V tiles | qinv[HG][64] | labels[64] | flags
    int q, k, vt, qinv, lab, flag, total;
    __host__ __device__ explicit LdsLayout(int hg) {
        q = 0;
        k = q + hg * kTileBytes;
        vt = k + hg * kTileBytes;
        qinv = vt + hg * kVtStride;
        lab = qinv + hg * kWs * 4;
        flag = lab + kWs;
        total = flag + 16;
    }
};

template <int HG>
__global__ void __launch_bounds__(64 * HG, 2) attn_fwd_mfma_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LdsLayout L(HG);
    const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6;  // g: head inside the group (wave-uniform)
    const int half = lane >> 5, l31 = lane & 31;
    const int h = blockIdx.y * HG + g;
    const int C = p.C;
    const int64_t N = p.N;
    const int nW = (int)(N / kWs);
    const int64_t total_windows = (int64_t)p.B * nW;
    const bool cosine = (p.flags & HS_ATTN_COSINE) != 0;
    const float hscale = p.head_scale[h];
    const uint16_t* qkv = (const uint16_t*)p.qkv;
    uint16_t* out = (uint16_t*)p.out;

    unsigned char* q_tile = smem + L.q + g * kTileBytes;
    unsigned char* k_tile = smem + L.k + g * kTileBytes;
    unsigned char* vt_tile = smem + L.vt + g * kVtStride;
    float* qinv_s = (float*)(smem + L.qinv);
    unsigned char* lab_s = smem + L.lab;

    // relative-position bias of this head, in the S^T accumulator layout: tile (kt, qt), register r holds
    // query qt*32 + l31, key kt*32 + (r&3) + 8*(r>>2) + 4*half
    float biasr[2][2][16];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, qq = qt * 32 + l31;
                biasr[kt][qt][r] = p.bias ? p.bias[((int64_t)h * kWs + qq) * kWs + key] * kLog2e : 0.f;
            }

    // staging geometry: 12 steps = 3 parts (q, k, v) x 4 row blocks of 16 rows; a step moves 16 rows x HG*64 B
    const int srow = tid / (4 * HG);   // 0..15
    const int sc = tid % (4 * HG);     // 16-B chunk inside the row's HG*64-B segment
    const int sg = sc >> 2, scc = sc & 3;
    const int64_t col0 = (int64_t)blockIdx.y * HG * kHd + sc * 8;  // element column inside a C-wide part

    for (int64_t wi = blockIdx.x; wi < total_windows; wi += gridDim.x) {
        const int b = (int)(wi / nW);
        const int w = (int)(wi - (int64_t)b * nW);
        const int64_t j0 = (int64_t)w * kWs;

        // ------------------------------------------------------------ stage q, k, v of this window into LDS
        int64_t tok[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) tok[rb] = (int64_t)b * N + shifted_source(p, j0 + rb * 16 + srow);
        uint4 ld[3][4];
#pragma unroll
        for (int part = 0; part < 3; ++part)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
                ld[part][rb] = *(const uint4*)(qkv + tok[rb] * 3 * C + (int64_t)part * C + col0);
        if (p.labels && tid < kWs) lab_s[tid] = p.labels[j0 + tid];

#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const int row = rb * 16 + srow;
            // --- q: raw; its inverse norm (cosine) becomes a per-row factor of the scores
            {
                const uint4 v = ld[0][rb];
                if (cosine) {
                    const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
                    float ss = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) ss += bf_lo(wd[i]) * bf_lo(wd[i]) + bf_hi(wd[i]) * bf_hi(wd[i]);
                    ss += __shfl_xor(ss, 1, 64);
                    ss += __shfl_xor(ss, 2, 64);
                    if (scc == 0) qinv_s[sg * kWs + row] = 1.f / fmaxf(sqrtf(ss), kNormEps);
                }
                *(uint4*)(smem + L.q + sg * kTileBytes + swz(row, scc)) = v;
            }
            // --- k: L2-normalised rows for cosine attention
            {
                uint4 v = ld[1][rb];
                if (cosine) {
                    uint32_t wd[4] = {v.x, v.y, v.z, v.w};
                    float ss = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) ss += bf_lo(wd[i]) * bf_lo(wd[i]) + bf_hi(wd[i]) * bf_hi(wd[i]);
                    ss += __shfl_xor(ss, 1, 64);
                    ss += __shfl_xor(ss, 2, 64);
                    const float kinv = 1.f / fmaxf(sqrtf(ss), kNormEps);
#pragma unroll
                    for (int i = 0; i < 4; ++i) wd[i] = pack_bf16(bf_lo(wd[i]) * kinv, bf_hi(wd[i]) * kinv);
                    v = make_uint4(wd[0], wd[1], wd[2], wd[3]);
                }
                *(uint4*)(smem + L.k + sg * kTileBytes + swz(row, scc)) = v;
            }
            // --- v: transposed, feature d = scc*8 + i lands at [vt_row(d)][row]
            {
                const uint4 v = ld[2][rb];
                const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
                unsigned char* base = smem + L.vt + sg * kVtStride + row * 2;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    *(uint16_t*)(base + vt_row(scc * 8 + 2 * i) * kVtLd) = (uint16_t)(wd[i] & 0xffffu);
                    *(uint16_t*)(base + vt_row(scc * 8 + 2 * i + 1) * kVtLd) = (uint16_t)(wd[i] >> 16);
                }
            }
        }
        __syncthreads();

        bool mixed = false;  // does this window contain more than one region label?
        if (p.labels) {
            const uint32_t* lw = (const uint32_t*)lab_s;
            const uint32_t first = lab_s[0] * 0x01010101u;
            bool diff = false;
#pragma unroll
            for (int i = 0; i < 16; ++i) diff |= lw[i] != first;
            mixed = diff;  // every lane reads the same 64 bytes: wave-uniform
        }

        // ------------------------------------------------------------ S^T = K Q^T
        f32x16 acc[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[kt][qt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 kf[2], qf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = t * 32 + l31, chunk = ks * 2 + half;
                kf[t] = *(const bf16x8*)(k_tile + swz(row, chunk));
                qf[t] = *(const bf16x8*)(q_tile + swz(row, chunk));
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                    acc[kt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt], qf[qt], acc[kt][qt], 0, 0, 0);
        }

        // ------------------------------------------------------------ softmax over the keys of each query (log2 domain)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int qq = qt * 32 + l31;
            const float fq = hscale * kLog2e * (cosine ? qinv_s[g * kWs + qq] : 1.f);
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float t = fmaf(acc[kt][qt][r], fq, biasr[kt][qt][r]);
                    acc[kt][qt][r] = t;
                    m = fmaxf(m, t);
                }
            if (mixed) {  // rare: windows cut by the shift boundary
                const int my = lab_s[qq];
                m = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        float t = acc[kt][qt][r];
                        if (lab_s[key] != my) t += kMaskLog2;
                        acc[kt][qt][r] = t;
                        m = fmaxf(m, t);
                    }
            }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float l = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(acc[kt][qt][r] - m);
                    acc[kt][qt][r] = e;
                    l += e;
                }
            l += __shfl_xor(l, 32, 64);
            const float linv = 1.f / l;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[kt][qt][r] *= linv;
            if (p.lse && half == 0) p.lse[((int64_t)b * p.nH + h) * N + j0 + qq] = (m + __builtin_amdgcn_logf(l)) * kLn2;
        }

        // ------------------------------------------------------------ O = P V
        f32x16 o[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qt][r] = 0.f;
        const unsigned char* vrow = vt_tile + vt_row(l31) * kVtLd;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int kt = ks >> 1, c = ks & 1;
            const int kbase = kt * 32 + c * 16 + 4 * half;  // slots 0..3 -> keys kbase.., slots 4..7 -> kbase+8..
            const uint2 lo = *(const uint2*)(vrow + kbase * 2);
            const uint2 hi = *(const uint2*)(vrow + (kbase + 8) * 2);
            const uint4 vw = make_uint4(lo.x, lo.y, hi.x, hi.y);
            const bf16x8 vf = __builtin_bit_cast(bf16x8, vw);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                bf16x8 pf;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) pf[jj] = (__bf16)acc[kt][qt][8 * c + jj];
                o[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf, vf, o[qt], 0, 0, 0);
            }
        }

        // ------------------------------------------------------------ O -> LDS (this head's q tile is free now) -> global
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qq = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                *(uint16_t*)(q_tile + qq * 64 + l31 * 2) = float_to_bf16(o[qt][r]);
            }
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const int row = rb * 16 + srow;
            const uint4 v = *(const uint4*)(smem + L.q + sg * kTileBytes + row * 64 + scc * 16);
            *(uint4*)(out + tok[rb] * C + col0) = v;
        }
        __syncthreads();
    }
}

int pick_head_group(int nH) {
    if (nH % 4 == 0) return 4;
    if (nH % 3 == 0) return 3;
    if (nH % 2 == 0) return 2;
    return 1;
}

template <int HG>
int launch_fwd(const AttnParams& p, hipStream_t stream) {
    const LdsLayout L(HG);
    auto kern = attn_fwd_mfma_kernel<HG>;
    HS_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L.total));
    const int groups = p.nH / HG;
    const int64_t windows = (int64_t)p.B * (p.N / kWs);
    int64_t slots = (256 * 3 + groups - 1) / groups;  // ~3 resident workgroups per CU over the whole chip
    if (slots > windows) slots = windows;
    if (slots < 1) slots = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)slots, (unsigned)groups), dim3(64 * HG), L.total, stream, p);
    HS_LAUNCH_CHECK("attn_fwd_mfma");
    return HS_OK;
}

}  // namespace

bool attn_mfma_supported(const AttnParams& p, int dtype) {
    // 16-byte vector access needs 8-element aligned columns: C % 8 == 0 holds since C = 32 * nH
    return dtype == HS_BF16 && p.Ws == kWs && p.hd == kHd && p.dout == nullptr;
}

int launch_attn_fwd_mfma(const AttnParams& p, hipStream_t stream) {
    switch (pick_head_group(p.nH)) {
        case 4: return launch_fwd<4>(p, stream);
        case 3: return launch_fwd<3>(p, stream);
        case 2: return launch_fwd<2>(p, stream);
        default: return launch_fwd<1>(p, stream);
    }
}

int launch_attn_bwd_mfma(const AttnParams&, hipStream_t) { return fail(HS_ERR_UNSUPPORTED, "mfma backward not built"); }

}  // namespace hs
