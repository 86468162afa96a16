// Shared declarations of the fused shift + window-attention kernels.
#pragma once
#include "hs_device.h"

namespace hs {

struct AttnParams {
    // forward
    const void* qkv;          // [B, N, 3C] natural order
    void* out;                // [B, N, C]  natural order (fwd: written; bwd: saved forward output, read)
    float* lse;               // [B, nH, N] shifted order (fwd: written if non-null; bwd: read)
    const float* bias;        // [nH, Ws, Ws] or null
    const float* head_scale;  // [nH]
    const int32_t* idx;       // [N] or null
    int64_t roll;             // used when idx == null
    const uint8_t* labels;    // [N] shifted order, or null
    int B;
    int64_t N;
    int C;
    int nH;
    int Ws;
    int hd;
    unsigned flags;
    // attention dropout (swin_hp_transformer.py:169): probabilities are zeroed with probability drop_p and the survivors
    // scaled by 1/(1-drop_p); the keep decision of element (image, head, shifted query row, key) is a pure function of
    // (seed, element index), so the backward regenerates exactly the forward's mask
    float drop_p;
    uint32_t seed_lo, seed_hi;
    // backward only
    const void* dout;    // [B, N, C]
    void* dqkv;          // [B, N, 3C]
    float* dbias;        // [nH, Ws, Ws] accumulate, or null
    float* dhead_scale;  // [nH] accumulate, or null
};

// natural-order token row read by shifted position j of an image
__device__ __forceinline__ int64_t shifted_source(const AttnParams& p, int64_t j) {
    if (p.idx) return (int64_t)p.idx[j];
    int64_t s = j + p.roll;
    return s >= p.N ? s - p.N : s;
}

__host__ __device__ constexpr __forceinline__ uint32_t hash32(uint32_t x) {  // "lowbias32" integer finaliser
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
// Called once at the top of every attention kernel (on its by-value copy of the parameters): the replay counter of hs_device.h enters
// the seed words here, not in DropRng -- the scalar loads and their wait stay out of the window loop and its counted LDS waits.
__device__ __forceinline__ void apply_seed_epoch(AttnParams& p) {
    if (p.drop_p > 0.f) {
        const uint64_t seed = epoch_seed(((uint64_t)p.seed_hi << 32) | p.seed_lo);
        p.seed_lo = (uint32_t)seed;
        p.seed_hi = (uint32_t)(seed >> 32);
    }
}

struct DropRng {
    uint32_t row_key, row_key2, thresh16;
    float keep_scale;
    // row = ((b * nH + h) * N + shifted_row): one per (image, head, query)
    __device__ __forceinline__ DropRng(const AttnParams& p, int64_t row) {
        const uint64_t base = (uint64_t)row * 256u;  // up to 256 keys per window
        // keyed with both seed words, one of them entering between the two rounds (once per row): rows of different seeds
        // get unrelated key streams (no row translation between seeds)
        row_key = hash32(hash32((uint32_t)base ^ hash32(p.seed_hi + (uint32_t)(base >> 32) * 0x9E3779B9u)) ^ p.seed_lo);
        row_key2 = hash32(row_key ^ 0x5BD1E995u);  // second key of the row: the keys whose bit 2 is set
        const float pd = p.drop_p;
        thresh16 = pd >= 1.f ? 65536u : (uint32_t)(pd * 65536.f);  // drop probability in steps of 2^-16
        keep_scale = pd >= 1.f ? 0.f : 1.f / (1.f - pd);
    }
    // The 16 mask bits of probability (row, key): one multiply per key PAIR.  u = K * M_q with K the row's first or second key
    // (bit 2 of the key: the MFMA accumulator layouts put keys k and k + 4 in the same register of lanes l and l + 32, so a lane
    // needs ONE of the two for all of its keys) and M_q an odd constant of the pair index with that bit removed -- a compile-time
    // constant wherever the key's register is (mult_half: every MFMA kernel), so the mask costs one `v_mul_lo_u32 v, v, literal`
    // -- quarter rate on the VALU -- per two probabilities.  Key 2q takes lo16(u) ^ hi16(u), key 2q + 1 hi16(u); dropped when
    // below thresh16.  The row keys are uniform over rows and seeds (finaliser rounds above), so (u_q, u_q') is the lattice of an
    // LCG with multiplier M_q' / M_q: uniform in both coordinates.
    static __host__ __device__ constexpr uint32_t pair_mult(uint32_t q) { return hash32(q * 0x9E3779B9u + 0x7F4A7C15u) | 1u; }
    __device__ __forceinline__ float keep_of(uint32_t k, uint32_t m, int odd) const {
        const uint32_t u = k * m, h = u ^ (u >> 16);
        return (odd ? (h >> 16) : (h & 0xffffu)) >= thresh16 ? keep_scale : 0.f;
    }
    // multiplier of probability (row, key): 0 or 1/(1-p)
    __device__ __forceinline__ float mult(int key) const {
        return keep_of((key & 4) ? row_key2 : row_key, pair_mult(((uint32_t)key >> 1) & ~2u), key & 1);
    }
    // the same for key = kc + 4 * half with kc a compile-time constant whose bit 2 is clear
    __device__ __forceinline__ float mult_half(int kc, int half) const {
        return keep_of(half ? row_key2 : row_key, pair_mult((uint32_t)kc >> 1), kc & 1);
    }
};

// fp32-VALU path: any power-of-two Ws in [4, 256], head_dim <= 128; fp32 or bf16 I/O
int launch_attn_fwd_generic(const AttnParams& p, int dtype, hipStream_t stream);
int launch_attn_bwd_generic(const AttnParams& p, int dtype, hipStream_t stream);

// MFMA path: Ws == 64, head_dim == 32, bf16 I/O
bool attn_mfma_supported(const AttnParams& p, int dtype);
int launch_attn_fwd_mfma(const AttnParams& p, hipStream_t stream);
int64_t attn_bwd_mfma_workspace_floats(const AttnParams& p);
int launch_attn_bwd_mfma(const AttnParams& p, float* workspace, hipStream_t stream);

// fp32 MFMA path: Ws == 64, head_dim == 32, fp32 I/O (v_mfma_f32_32x32x2_f32)
bool attn_mfma_f32_supported(const AttnParams& p, int dtype);
int launch_attn_fwd_mfma_f32(const AttnParams& p, hipStream_t stream);
int64_t attn_bwd_mfma_f32_workspace_floats(const AttnParams& p);
int launch_attn_bwd_mfma_f32(const AttnParams& p, float* workspace, hipStream_t stream);

}  // namespace hs
