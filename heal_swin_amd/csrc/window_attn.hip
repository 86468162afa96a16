// C-ABI entry points of the fused shift + window attention op: argument validation and dispatch
// between the MFMA paths (Ws = 64, head_dim = 32; bf16 and fp32) and the fp32-VALU path (everything else).
#include "window_attn.h"

namespace {

int fill_params(hs::AttnParams& p, const void* qkv, void* out, float* lse, const float* bias, const float* head_scale,
                const int32_t* idx, int64_t roll, const uint8_t* labels, int batch, int64_t n_tokens, int channels,
                int num_heads, int window_size, unsigned flags, float attn_drop, uint64_t seed, int dtype) {
    HS_CHECK_ARG(qkv && out && head_scale, "qkv, out and head_scale must not be null");
    HS_CHECK_ARG(dtype == HS_F32 || dtype == HS_BF16, "dtype must be HS_F32 or HS_BF16");
    HS_CHECK_ARG(batch > 0 && n_tokens > 0 && channels > 0 && num_heads > 0, "non-positive size");
    HS_CHECK_ARG(n_tokens < (1ll << 31), "n_tokens must fit int32 (gather table is int32)");
    HS_CHECK_ARG((int64_t)batch * n_tokens < (1ll << 31), "batch * n_tokens must fit int32 (token rows are carried as 32-bit indices)");
    HS_CHECK_ARG(channels % num_heads == 0, "channels %d not divisible by num_heads %d", channels, num_heads);
    // hp_windowing.py:16 asserts a power of two.  Square nested blocks (4^k) are only needed by the relative-position index
    // and the grid shift, which are validated where those tables are built; the kernels take any power of two
    // (e.g. a last stage clamped to 8 * 4^k tokens with 8 base pixels, swin_hp_transformer.py:243-246)
    HS_CHECK_ARG((window_size & (window_size - 1)) == 0, "window_size must be a power of two, got %d", window_size);
    HS_CHECK_ARG(window_size <= 256, "window_size %d > 256 is not supported", window_size);
    HS_CHECK_ARG(n_tokens % window_size == 0, "n_tokens %lld not divisible by window_size %d", (long long)n_tokens, window_size);
    HS_CHECK_ARG(window_size >= 4, "window_size must be at least 4");
    HS_CHECK_ARG(roll >= 0 && roll < n_tokens, "roll must be in [0, n_tokens)");
    HS_CHECK_ARG(attn_drop >= 0.f && attn_drop <= 1.f, "attn_drop must be in [0, 1]");
    p = hs::AttnParams{};
    p.drop_p = attn_drop;
    p.seed_lo = (uint32_t)seed;
    p.seed_hi = (uint32_t)(seed >> 32);
    p.qkv = qkv;
    p.out = out;
    p.lse = lse;
    p.bias = bias;
    p.head_scale = head_scale;
    p.idx = idx;
    p.roll = idx ? 0 : roll;
    p.labels = labels;
    p.B = batch;
    p.N = n_tokens;
    p.C = channels;
    p.nH = num_heads;
    p.Ws = window_size;
    p.hd = channels / num_heads;
    p.flags = flags;
    return HS_OK;
}

}  // namespace

extern "C" {

int hs_window_attn_fwd(const void* qkv, void* out, float* lse, const float* bias, const float* head_scale,
                       const int32_t* idx, int64_t roll, const uint8_t* labels, int batch, int64_t n_tokens,
                       int channels, int num_heads, int window_size, unsigned flags, float attn_drop, uint64_t seed, int dtype,
                       void* stream) {
    hs::AttnParams p;
    if (int st = fill_params(p, qkv, out, lse, bias, head_scale, idx, roll, labels, batch, n_tokens, channels, num_heads,
                             window_size, flags, attn_drop, seed, dtype))
        return st;
    hipStream_t s = (hipStream_t)stream;
    const bool valu = (flags & HS_ATTN_FORCE_VALU) != 0;
    if (!valu && hs::attn_mfma_supported(p, dtype)) return hs::launch_attn_fwd_mfma(p, s);
    if (!valu && hs::attn_mfma_f32_supported(p, dtype)) return hs::launch_attn_fwd_mfma_f32(p, s);
    return hs::launch_attn_fwd_generic(p, dtype, s);
}

int64_t hs_window_attn_bwd_workspace(int batch, int64_t n_tokens, int channels, int num_heads, int window_size, int dtype) {
    hs::AttnParams p{};
    if (batch <= 0 || n_tokens <= 0 || channels <= 0 || num_heads <= 0 || window_size <= 0 || channels % num_heads) return 0;
    p.B = batch;
    p.N = n_tokens;
    p.C = channels;
    p.nH = num_heads;
    p.Ws = window_size;
    p.hd = channels / num_heads;
    if (hs::attn_mfma_supported(p, dtype)) return hs::attn_bwd_mfma_workspace_floats(p);
    return hs::attn_mfma_f32_supported(p, dtype) ? hs::attn_bwd_mfma_f32_workspace_floats(p) : 0;
}

int hs_window_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dbias,
                       float* dhead_scale, float* workspace, const float* bias, const float* head_scale, const int32_t* idx,
                       int64_t roll, const uint8_t* labels, int batch, int64_t n_tokens, int channels, int num_heads,
                       int window_size, unsigned flags, float attn_drop, uint64_t seed, int dtype, void* stream) {
    hs::AttnParams p;
    if (int st = fill_params(p, qkv, const_cast<void*>(out), const_cast<float*>(lse), bias, head_scale, idx, roll, labels,
                             batch, n_tokens, channels, num_heads, window_size, flags, attn_drop, seed, dtype))
        return st;
    HS_CHECK_ARG(dout && dqkv && lse, "dout, dqkv and lse must not be null");
    HS_CHECK_ARG(!bias || dbias, "dbias must be given when bias is");
    HS_CHECK_ARG(!(flags & HS_ATTN_COSINE) || dhead_scale, "dhead_scale must be given for cosine attention");
    p.dout = dout;
    p.dqkv = dqkv;
    p.dbias = bias ? dbias : nullptr;
    p.dhead_scale = (flags & HS_ATTN_COSINE) ? dhead_scale : nullptr;
    hipStream_t s = (hipStream_t)stream;
    const bool valu = (flags & HS_ATTN_FORCE_VALU) != 0;
    if (!valu && hs::attn_mfma_supported(p, dtype)) return hs::launch_attn_bwd_mfma(p, workspace, s);
    // HS_ATTN_OVERWRITE_GRADS is a contract of the ENTRY POINT: the bf16 MFMA path writes dbias / dhead_scale, every other path
    // accumulates -- so a caller that set the flag (and handed over uninitialised buffers) gets them zeroed here first
    if (flags & HS_ATTN_OVERWRITE_GRADS) {
        if (p.dbias) HS_HIP_CHECK(hipMemsetAsync(p.dbias, 0, sizeof(float) * (size_t)num_heads * window_size * window_size, s));
        if (p.dhead_scale) HS_HIP_CHECK(hipMemsetAsync(p.dhead_scale, 0, sizeof(float) * (size_t)num_heads, s));
    }
    if (!valu && hs::attn_mfma_f32_supported(p, dtype)) return hs::launch_attn_bwd_mfma_f32(p, workspace, s);
    return hs::launch_attn_bwd_generic(p, dtype, s);
}

}  // extern "C"
