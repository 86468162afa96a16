// fp32 MFMA path of the fused shift + window attention (Ws = 64, head_dim = 32, fp32 activations): the exact-fp32 route of
// the reference's precision = 32 training (depth regression, BASELINE config 5) on v_mfma_f32_32x32x2_f32 instead of the
// generic VALU kernels.  One wavefront owns one (window, head) at a time and walks the windows of its head persistently;
// no inter-wave synchronisation at all.
//
// v_mfma_f32_32x32x2_f32 contracts only 2 indices per instruction (lane half h supplies k = h), so a 32-deep contraction
// is 16 steps; step j of lane half h may use ANY contraction index as long as both operands agree:
//   * contraction over FEATURES (S = K^ Q^T, dP = V dO^T): feature 2j + h, both operands read from the row-major LDS tiles;
//   * contraction over KEYS or QUERIES with the probabilities as one operand: index kappa(j, h) = (j & 3) + 8 (j >> 2) + 4 h,
//     which is exactly the row that accumulator register j of lane half h holds -- so an accumulator tile is fed back as
//     the B operand register by register, in fp32, with no conversion, shuffle or LDS round trip.
// Forward:   S^T = K^ Q^T (lane = query, registers = keys) -> softmax -> O^T = V^T P^T  (lane = query, registers = features)
// Backward:  phase A, lane = query:  S^T, dP^T -> P^T, dS'^T -> d bias, d scale, dQ^T = K^^T dS'^T
//            phase B, lane = key:    S, dP recomputed in the transposed layout -> P, dS' -> dV^T = dO^T P, dK^^T = Q^T dS'
//            (448 MFMAs instead of the 320 a transposing LDS round trip would need, but no 16 KB fp32 scratch per wave, which
//            would halve the wavefronts per CU).
// LDS tiles: row-major [64 tokens][32 features] fp32, row stride 36 floats (16-byte aligned rows; feature-indexed reads
// of 32 consecutive rows are at most 2-way conflicted, token-row reads are conflict-free).
#include "window_attn.h"

namespace hs {
HS_DEFINE_SEED_EPOCH_SETTER(set_seed_epoch_attn_mfma_f32)
namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kWs = 64, kHd = 32;
constexpr int kLd = 36;                 // floats per staged row
constexpr int kTile = kWs * kLd;        // floats per tile
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kNormEps = 1e-12f;
constexpr float kMaskLog2 = -100.f * kLog2e;

__device__ __forceinline__ int kappa(int j, int half) { return (j & 3) + 8 * (j >> 2) + 4 * half; }
__device__ __forceinline__ f32x16 mfma2(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void zero(f32x16& v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = 0.f;
}
__device__ __forceinline__ float sum8(float v) {  // over the 8 lanes that share a staged row
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

// out[kt][qt] (+)= A-tile(kt) . B-tile(qt)^T over the 32 features: A rows kt*32 + l31, B rows qt*32 + l31
__device__ __forceinline__ void feature_product(const float* a_tile, const float* b_tile, int l31, int half, f32x16 (&out)[2][2]) {
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
        const int col = 2 * j + half;
        const float a0 = a_tile[l31 * kLd + col], a1 = a_tile[(32 + l31) * kLd + col];
        const float b0 = b_tile[l31 * kLd + col], b1 = b_tile[(32 + l31) * kLd + col];
        out[0][0] = mfma2(a0, b0, out[0][0]);
        out[0][1] = mfma2(a0, b1, out[0][1]);
        out[1][0] = mfma2(a1, b0, out[1][0]);
        out[1][1] = mfma2(a1, b1, out[1][1]);
    }
}

// out[ct] (+)= sum over the 64 rows i of tile[i][feature l31] * w[it][ct][.]: w tiles hold row index kappa in their registers
// (it = 32-row block of the contraction index, ct = 32-column block of the lane index)
__device__ __forceinline__ void row_product(const float* tile, const f32x16 (&w)[2][2], int l31, int half, f32x16 (&out)[2]) {
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float a = tile[(it * 32 + kappa(j, half)) * kLd + l31];
            out[0] = mfma2(a, w[it][0][j], out[0]);
            out[1] = mfma2(a, w[it][1][j], out[1]);
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // keep the compiler from hoisting all 32 LDS reads
        }
}

struct Stage {  // lane -> (row within an 8-row pass, float4 chunk)
    int r8, ch;
    __device__ __forceinline__ explicit Stage(int lane) : r8(lane >> 3), ch(lane & 7) {}
};

// ================================================================================================ forward
template <bool DROP>
__global__ void __launch_bounds__(64, 1) attn_fwd_f32_kernel(AttnParams p) {
    if constexpr (DROP) apply_seed_epoch(p);
    __shared__ __attribute__((aligned(16))) float smem[3 * kTile + 2 * kWs];
    float* q_t = smem;
    float* k_t = smem + kTile;
    float* v_t = smem + 2 * kTile;
    float* qinv_s = smem + 3 * kTile;
    unsigned char* lab_s = (unsigned char*)(smem + 3 * kTile + kWs);
    const int lane = threadIdx.x, half = lane >> 5, l31 = lane & 31;
    const int h = blockIdx.y;
    const int C = p.C;
    const int64_t N = p.N;
    const int nW = (int)(N / kWs);
    const int64_t total_windows = (int64_t)p.B * nW;
    const bool cosine = (p.flags & HS_ATTN_COSINE) != 0;
    const float hscale = p.head_scale[h];
    const float* qkv = (const float*)p.qkv;
    float* out = (float*)p.out;
    const Stage st(lane);

    // relative-position bias of this head in the S^T accumulator layout (times log2 e), resident across the windows
    float biasr[2][2][16];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.bias) b4 = *(const float4*)(p.bias + ((int64_t)h * kWs + qt * 32 + l31) * kWs + kt * 32 + 8 * rg + 4 * half);
                biasr[kt][qt][4 * rg] = b4.x * kLog2e;
                biasr[kt][qt][4 * rg + 1] = b4.y * kLog2e;
                biasr[kt][qt][4 * rg + 2] = b4.z * kLog2e;
                biasr[kt][qt][4 * rg + 3] = b4.w * kLog2e;
            }

    for (int64_t wi = blockIdx.x; wi < total_windows; wi += gridDim.x) {
        int lane_o = threadIdx.x;  // opaque copy: keeps per-lane addresses from being hoisted out of the loop and pinned
        asm volatile("" : "+v"(lane_o));
        const int lane = lane_o, half = lane_o >> 5, l31 = lane_o & 31;
        const Stage st(lane_o);
        const int b = (int)(wi / nW);
        const int64_t j0 = (wi - (int64_t)b * nW) * kWs;
        // ------------------------------------------------------------ stage q, k^, v
        lab_s[lane] = p.labels ? p.labels[j0 + lane] : 0;
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int row = ps * 8 + st.r8;
            const float* src = qkv + ((int64_t)b * N + shifted_source(p, j0 + row)) * 3 * C + h * kHd + st.ch * 4;
            const float4 vq = *(const float4*)src;
            float4 vk = *(const float4*)(src + C);
            const float4 vv = *(const float4*)(src + 2 * (int64_t)C);
            if (cosine) {
                const float sq = sum8(vq.x * vq.x + vq.y * vq.y + vq.z * vq.z + vq.w * vq.w);
                const float sk = sum8(vk.x * vk.x + vk.y * vk.y + vk.z * vk.z + vk.w * vk.w);
                const float kinv = 1.f / fmaxf(sqrtf(sk), kNormEps);
                vk = make_float4(vk.x * kinv, vk.y * kinv, vk.z * kinv, vk.w * kinv);
                if (st.ch == 0) qinv_s[row] = 1.f / fmaxf(sqrtf(sq), kNormEps);
            }
            *(float4*)(q_t + row * kLd + st.ch * 4) = vq;
            *(float4*)(k_t + row * kLd + st.ch * 4) = vk;
            *(float4*)(v_t + row * kLd + st.ch * 4) = vv;
        }
        bool mixed = false;
        if (p.labels) {
            const uint32_t* lw = (const uint32_t*)lab_s;
            const uint32_t first = lab_s[0] * 0x01010101u;
#pragma unroll
            for (int i = 0; i < 16; ++i) mixed |= lw[i] != first;
        }
        // ------------------------------------------------------------ S^T = K^ Q^T
        f32x16 acc[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) zero(acc[kt][qt]);
        feature_product(k_t, q_t, l31, half, acc);
        // ------------------------------------------------------------ softmax over the keys of each query (log2 domain)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int qq = qt * 32 + l31;
            const float fq = hscale * kLog2e * (cosine ? qinv_s[qq] : 1.f);
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
                        float t = acc[kt][qt][r];
                        if (lab_s[kt * 32 + kappa(r, half)] != my) t += kMaskLog2;
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
                    const float e = exp2f(acc[kt][qt][r] - m);
                    acc[kt][qt][r] = e;
                    l += e;
                }
            l += __shfl_xor(l, 32, 64);
            const float linv = 1.f / l;
            if constexpr (DROP) {
                const DropRng rng(p, ((int64_t)b * p.nH + h) * N + j0 + qq);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[kt][qt][r] *= linv * rng.mult_half(kt * 32 + kappa(r, 0), half);
            } else {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[kt][qt][r] *= linv;
            }
            if (p.lse && half == 0) p.lse[((int64_t)b * p.nH + h) * N + j0 + qq] = (m + log2f(l)) * kLn2;
        }
        // ------------------------------------------------------------ O^T = V^T P^T, rows of O straight to global
        f32x16 o[2];
        zero(o[0]);
        zero(o[1]);
        row_product(v_t, acc, l31, half, o);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float* dst = out + ((int64_t)b * N + shifted_source(p, j0 + qt * 32 + l31)) * C + h * kHd + 4 * half;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                *(float4*)(dst + 8 * rg) = make_float4(o[qt][4 * rg], o[qt][4 * rg + 1], o[qt][4 * rg + 2], o[qt][4 * rg + 3]);
        }
    }
}

// ================================================================================================ backward
template <bool DROP, bool COS>
__global__ void __launch_bounds__(64, 1) attn_bwd_f32_kernel(AttnParams p, float* __restrict__ dbias_part,
                                                             float* __restrict__ dscale_part) {
    if constexpr (DROP) apply_seed_epoch(p);
    __shared__ __attribute__((aligned(16))) float smem[4 * kTile + 5 * kWs];
    float* q_t = smem;
    float* k_t = smem + kTile;
    float* v_t = smem + 2 * kTile;
    float* do_t = smem + 3 * kTile;
    float* qinv_s = smem + 4 * kTile;
    float* kinv_s = qinv_s + kWs;
    float* dsum_s = kinv_s + kWs;
    float* lse2_s = dsum_s + kWs;
    unsigned char* lab_s = (unsigned char*)(lse2_s + kWs);
    const int lane = threadIdx.x, half = lane >> 5, l31 = lane & 31;
    const int h = blockIdx.y;
    const int C = p.C;
    const int64_t N = p.N;
    const int nW = (int)(N / kWs);
    const int64_t total_windows = (int64_t)p.B * nW;
    const float hscale = p.head_scale[h];
    const float* qkv = (const float*)p.qkv;
    const float* fo = (const float*)p.out;
    const float* dout = (const float*)p.dout;
    float* dqkv = (float*)p.dqkv;
    const Stage st(lane);

    f32x16 dbacc[2][2];  // d bias [query = qt*32 + l31][key = kt*32 + kappa(r)], summed over this wave's windows
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) zero(dbacc[kt][qt]);
    float dscale_acc = 0.f;

    for (int64_t wi = blockIdx.x; wi < total_windows; wi += gridDim.x) {
        // Per-lane addresses (bias rows, LDS rows) are loop-invariant; hoisted out of the window loop they would pin well
        // over a hundred VGPRs and spill.  An opaque copy of the lane id makes the compiler re-derive them per window.
        int lane_o = threadIdx.x;
        asm volatile("" : "+v"(lane_o));
        const int lane = lane_o, half = lane_o >> 5, l31 = lane_o & 31;
        const Stage st(lane_o);
        const int b = (int)(wi / nW);
        const int64_t j0 = (wi - (int64_t)b * nW) * kWs;
        const int64_t lrow = ((int64_t)b * p.nH + h) * N + j0;
        // ------------------------------------------------------------ stage q, k^, v, dO; D = dO . O; norms; lse
        lab_s[lane] = p.labels ? p.labels[j0 + lane] : 0;
        lse2_s[lane] = p.lse[lrow + lane] * kLog2e;
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int row = ps * 8 + st.r8;
            const int64_t tok = (int64_t)b * N + shifted_source(p, j0 + row);
            const float* src = qkv + tok * 3 * C + h * kHd + st.ch * 4;
            const float4 vq = *(const float4*)src;
            float4 vk = *(const float4*)(src + C);
            const float4 vv = *(const float4*)(src + 2 * (int64_t)C);
            const float4 vd = *(const float4*)(dout + tok * C + h * kHd + st.ch * 4);
            const float4 vo = *(const float4*)(fo + tok * C + h * kHd + st.ch * 4);
            const float ds = sum8(vd.x * vo.x + vd.y * vo.y + vd.z * vo.z + vd.w * vo.w);
            if constexpr (COS) {
                const float sq = sum8(vq.x * vq.x + vq.y * vq.y + vq.z * vq.z + vq.w * vq.w);
                const float sk = sum8(vk.x * vk.x + vk.y * vk.y + vk.z * vk.z + vk.w * vk.w);
                const float kinv = 1.f / fmaxf(sqrtf(sk), kNormEps);
                vk = make_float4(vk.x * kinv, vk.y * kinv, vk.z * kinv, vk.w * kinv);
                if (st.ch == 0) {
                    qinv_s[row] = 1.f / fmaxf(sqrtf(sq), kNormEps);
                    kinv_s[row] = kinv;
                }
            }
            if (st.ch == 0) dsum_s[row] = ds;
            *(float4*)(q_t + row * kLd + st.ch * 4) = vq;
            *(float4*)(k_t + row * kLd + st.ch * 4) = vk;
            *(float4*)(v_t + row * kLd + st.ch * 4) = vv;
            *(float4*)(do_t + row * kLd + st.ch * 4) = vd;
        }
        bool mixed = false;
        if (p.labels) {
            const uint32_t* lw = (const uint32_t*)lab_s;
            const uint32_t first = lab_s[0] * 0x01010101u;
#pragma unroll
            for (int i = 0; i < 16; ++i) mixed |= lw[i] != first;
        }

        // ============================================================ phase A: lane = query
        {
            f32x16 accS[2][2], accP[2][2];  // [kt][qt]
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    zero(accS[kt][qt]);
                    zero(accP[kt][qt]);
                }
            feature_product(k_t, q_t, l31, half, accS);   // S^T
            feature_product(v_t, do_t, l31, half, accP);  // dP^T
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int qq = qt * 32 + l31;
                const float qinv = COS ? qinv_s[qq] : 1.f;
                const float fqn = hscale * qinv, fq2 = fqn * kLog2e;
                const float lse2 = lse2_s[qq], dsum = dsum_s[qq];
                const DropRng rng(p, lrow + qq);
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    float brow[16];  // bias (and, in cut windows, the mask) of this lane's query against the 16 keys, log2 domain
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (p.bias) b4 = *(const float4*)(p.bias + ((int64_t)h * kWs + qq) * kWs + kt * 32 + 8 * rg + 4 * half);
                        brow[4 * rg] = b4.x * kLog2e;
                        brow[4 * rg + 1] = b4.y * kLog2e;
                        brow[4 * rg + 2] = b4.z * kLog2e;
                        brow[4 * rg + 3] = b4.w * kLog2e;
                    }
                    if (mixed) {
                        const int my = lab_s[qq];
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (lab_s[kt * 32 + kappa(r, half)] != my) brow[r] += kMaskLog2;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float sraw = accS[kt][qt][r];
                        const float t = fmaf(sraw, fq2, brow[r]);
                        const float pr = exp2f(t - lse2);
                        float dpv = accP[kt][qt][r];
                        if constexpr (DROP) dpv *= rng.mult_half(kt * 32 + kappa(r, 0), half);
                        const float dsv = pr * (dpv - dsum);
                        dbacc[kt][qt][r] += dsv;
                        if constexpr (COS) dscale_acc = fmaf(dsv * qinv, sraw, dscale_acc);
                        accS[kt][qt][r] = dsv * fqn;  // dS'^T
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // dQ^T = K^^T dS'^T  (cosine: X - q (q.X) / |q|^2), rows straight to global
            f32x16 dq[2];
            zero(dq[0]);
            zero(dq[1]);
            row_product(k_t, accS, l31, half, dq);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const int qq = qt * 32 + l31;
                if constexpr (COS) {
                    float pq = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) pq = fmaf(dq[qt][r], q_t[qq * kLd + kappa(r, half)], pq);
                    pq += __shfl_xor(pq, 32, 64);
                    const float qinv = qinv_s[qq];
                    pq *= qinv * qinv;
#pragma unroll
                    for (int r = 0; r < 16; ++r) dq[qt][r] = fmaf(-q_t[qq * kLd + kappa(r, half)], pq, dq[qt][r]);
                }
                float* dst = dqkv + ((int64_t)b * N + shifted_source(p, j0 + qq)) * 3 * C + h * kHd + 4 * half;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    *(float4*)(dst + 8 * rg) = make_float4(dq[qt][4 * rg], dq[qt][4 * rg + 1], dq[qt][4 * rg + 2], dq[qt][4 * rg + 3]);
            }
        }

        // ============================================================ phase B: lane = key
        {
            f32x16 accS[2][2], accP[2][2];  // [qt][kt]: register r <-> query qt*32 + kappa(r), lane <-> key kt*32 + l31
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    zero(accS[qt][kt]);
                    zero(accP[qt][kt]);
                }
            feature_product(q_t, k_t, l31, half, accS);   // S
            feature_product(do_t, v_t, l31, half, accP);  // dP
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qq = qt * 32 + kappa(r, half);  // uniform over a lane half: the LDS reads below are broadcasts
                    const float qinv = COS ? qinv_s[qq] : 1.f;
                    const float fqn = hscale * qinv, fq2 = fqn * kLog2e;
                    const float lse2 = lse2_s[qq], dsum = dsum_s[qq];
                    const int qlab = lab_s[qq];
                    const DropRng rng(p, lrow + qq);
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) {
                        const int key = kt * 32 + l31;
                        const float sraw = accS[qt][kt][r];
                        float bl = p.bias ? p.bias[((int64_t)h * kWs + qq) * kWs + key] * kLog2e : 0.f;
                        if (mixed && lab_s[key] != qlab) bl += kMaskLog2;
                        const float pr = exp2f(fmaf(sraw, fq2, bl) - lse2);
                        float dpv = accP[qt][kt][r], prd = pr;
                        if constexpr (DROP) {
                            const float mlt = rng.mult(key);
                            dpv *= mlt;
                            prd *= mlt;
                        }
                        accP[qt][kt][r] = prd;                        // (dropped) P
                        accS[qt][kt][r] = pr * (dpv - dsum) * fqn;    // dS'
                    }
                    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // bound the hoisting of the bias / LDS loads
                }
            f32x16 dv[2], dk[2];
            zero(dv[0]);
            zero(dv[1]);
            zero(dk[0]);
            zero(dk[1]);
            row_product(do_t, accP, l31, half, dv);  // dV^T = dO^T P
            row_product(q_t, accS, l31, half, dk);   // dK^^T = Q^T dS'
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const int key = kt * 32 + l31;
                if constexpr (COS) {  // gradient through k / |k|
                    float pk = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) pk = fmaf(dk[kt][r], k_t[key * kLd + kappa(r, half)], pk);
                    pk += __shfl_xor(pk, 32, 64);
                    const float kinv = kinv_s[key];
#pragma unroll
                    for (int r = 0; r < 16; ++r) dk[kt][r] = fmaf(-k_t[key * kLd + kappa(r, half)], pk, dk[kt][r]) * kinv;
                }
                float* dst = dqkv + ((int64_t)b * N + shifted_source(p, j0 + key)) * 3 * C + C + h * kHd + 4 * half;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    *(float4*)(dst + 8 * rg) = make_float4(dk[kt][4 * rg], dk[kt][4 * rg + 1], dk[kt][4 * rg + 2], dk[kt][4 * rg + 3]);
                    *(float4*)(dst + C + 8 * rg) = make_float4(dv[kt][4 * rg], dv[kt][4 * rg + 1], dv[kt][4 * rg + 2], dv[kt][4 * rg + 3]);
                }
            }
        }
    }

    if (dbias_part) {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float* dst = dbias_part + ((int64_t)blockIdx.x * p.nH + h) * kWs * kWs + (int64_t)(qt * 32 + l31) * kWs;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    *(float4*)(dst + kt * 32 + 8 * rg + 4 * half) = make_float4(dbacc[kt][qt][4 * rg], dbacc[kt][qt][4 * rg + 1],
                                                                                dbacc[kt][qt][4 * rg + 2], dbacc[kt][qt][4 * rg + 3]);
        }
    }
    if (dscale_part) {
        const float tot = wave_sum(dscale_acc);
        if (lane == 0) dscale_part[(int64_t)blockIdx.x * p.nH + h] = tot;
    }
}

// dst[e] += sum over parts of src[part][e]   (fixed order: deterministic)
__global__ void reduce_parts_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int parts, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float acc = 0.f;
    for (int s = 0; s < parts; ++s) acc += src[(int64_t)s * n + e];
    dst[e] += acc;
}

int f32_slots(const AttnParams& p, int waves_per_cu) {
    const int64_t windows = (int64_t)p.B * (p.N / kWs);
    int64_t slots = ((int64_t)usable_cus() * waves_per_cu) / p.nH;  // rounded DOWN: a workgroup beyond the resident set would double the run time
    if (slots > windows) slots = windows;
    return (int)(slots < 1 ? 1 : slots);
}

}  // namespace

bool attn_mfma_f32_supported(const AttnParams& p, int dtype) { return dtype == HS_F32 && p.Ws == kWs && p.hd == kHd; }

int64_t attn_bwd_mfma_f32_workspace_floats(const AttnParams& p) { return (int64_t)f32_slots(p, 4) * p.nH * (kWs * kWs + 1); }

int launch_attn_fwd_mfma_f32(const AttnParams& p, hipStream_t stream) {
    // 4 wavefronts per CU (one per SIMD): a fifth would fit the LDS but put two on one SIMD and the slowest SIMD sets the pace
    const dim3 grid((unsigned)f32_slots(p, 4), (unsigned)p.nH);
    if (p.drop_p > 0.f) hipLaunchKernelGGL(attn_fwd_f32_kernel<true>, grid, dim3(64), 0, stream, p);
    else hipLaunchKernelGGL(attn_fwd_f32_kernel<false>, grid, dim3(64), 0, stream, p);
    HS_LAUNCH_CHECK("attn_fwd_mfma_f32");
    return HS_OK;
}

int launch_attn_bwd_mfma_f32(const AttnParams& p, float* workspace, hipStream_t stream) {
    if (!workspace) return fail(HS_ERR_INVALID_ARG, "the MFMA backward needs a workspace (hs_window_attn_bwd_workspace)");
    const int slots = f32_slots(p, 4);
    float* dbias_part = p.dbias ? workspace : nullptr;
    float* dscale_part = p.dhead_scale ? workspace + (int64_t)slots * p.nH * kWs * kWs : nullptr;
    const dim3 grid((unsigned)slots, (unsigned)p.nH);
    const bool drop = p.drop_p > 0.f, cos = (p.flags & HS_ATTN_COSINE) != 0;
    if (cos) {
        if (drop) hipLaunchKernelGGL((attn_bwd_f32_kernel<true, true>), grid, dim3(64), 0, stream, p, dbias_part, dscale_part);
        else hipLaunchKernelGGL((attn_bwd_f32_kernel<false, true>), grid, dim3(64), 0, stream, p, dbias_part, dscale_part);
    } else {
        if (drop) hipLaunchKernelGGL((attn_bwd_f32_kernel<true, false>), grid, dim3(64), 0, stream, p, dbias_part, dscale_part);
        else hipLaunchKernelGGL((attn_bwd_f32_kernel<false, false>), grid, dim3(64), 0, stream, p, dbias_part, dscale_part);
    }
    HS_LAUNCH_CHECK("attn_bwd_mfma_f32");
    if (dbias_part) {
        const int64_t n = (int64_t)p.nH * kWs * kWs;
        hipLaunchKernelGGL(reduce_parts_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dbias_part, p.dbias, slots, n);
        HS_LAUNCH_CHECK("reduce dbias partials");
    }
    if (dscale_part) {
        hipLaunchKernelGGL(reduce_parts_f32_kernel, dim3(1), dim3(256), 0, stream, dscale_part, p.dhead_scale, slots, (int64_t)p.nH);
        HS_LAUNCH_CHECK("reduce dscale partials");
    }
    return HS_OK;
}

}  // namespace hs
