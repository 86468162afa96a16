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

// fp32-VALU path: any Ws in {4,16,64,256}, head_dim <= 128 (fwd) / <= 64 (bwd); fp32 or bf16 I/O
int launch_attn_fwd_generic(const AttnParams& p, int dtype, hipStream_t stream);
int launch_attn_bwd_generic(const AttnParams& p, int dtype, hipStream_t stream);

// MFMA path: Ws == 64, head_dim == 32, bf16 I/O
bool attn_mfma_supported(const AttnParams& p, int dtype);
int launch_attn_fwd_mfma(const AttnParams& p, hipStream_t stream);
int64_t attn_bwd_mfma_workspace_floats(const AttnParams& p);
int launch_attn_bwd_mfma(const AttnParams& p, float* workspace, hipStream_t stream);

}  // namespace hs
