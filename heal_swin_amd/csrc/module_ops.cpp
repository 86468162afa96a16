// Backward of the block's attention half as ONE C-ABI call (the module-level hs_window_attn_bwd(x, qkv_w, ...)):
//     out = [x +] proj( window_attention( qkv( [LayerNorm](x) ) ) )          forward: hs_window_attn_module_fwd_train
// The forward's kernel is one launch; its backward is NOT one kernel (the fused form needs the 96 KB of Wqkv next to a
// backward that already fills the LDS, and dqkv stays in HBM for the weight gradients either way).  This entry point chains the
// kernels the Python mirror records as autograd nodes (ops.window_attn_module_train) on the caller's stream, in their order:
//     proj   : dW_p += dout^T O (hs_linear_wgrad),  dO   = dout W_p        (hs_gemm_nt on the transposed weight copy)
//     core   : dqkv, dbias, dhead_scale                                     (hs_window_attn_bwd on the saved qkv / lse)
//     qkv    : dW_q += dqkv^T xn (hs_linear_wgrad), dxn  = dqkv W_q         (hs_gemm_nt)
//     norm1  : dx = LayerNorm_bwd(dxn) + dout  (the residual's gradient), dgamma, dbeta      (hs_add_layernorm_bwd)
// so that an operator-level integration binds one symbol per direction.  bf16, the shapes hs_window_attn_module_supported accepts.
#include "hs_common.h"

namespace {
inline int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }
inline int64_t round4(int64_t v) { return (v + 3) & ~(int64_t)3; }
}  // namespace

extern "C" {

/* floats: three bf16 activation buffers (dO [M, C], dqkv [M, 3C], dxn [M, C]: 5 M C / 2 floats) followed by the largest of the
 * chained kernels' own workspaces */
int64_t hs_window_attn_module_bwd_chain_workspace(int batch, int64_t n_tokens, int channels, int num_heads, int window_size) {
    const int64_t rows = (int64_t)batch * n_tokens;
    int64_t ws = hs_window_attn_bwd_workspace(batch, n_tokens, channels, num_heads, window_size, HS_BF16);
    ws = max64(ws, hs_linear_wgrad_workspace(rows, 3 * channels, channels));
    ws = max64(ws, hs_linear_wgrad_workspace(rows, channels, channels));
    ws = max64(ws, hs_layernorm_bwd_workspace(rows, channels));
    return round4(rows * channels * 5 / 2 + 4) + ws;
}

int hs_window_attn_module_bwd_chain(const void* dout, const void* x, const void* xn, const float* mean, const float* rstd, const void* qkv,
                              const void* attn_out, const float* lse, const void* qkv_w_t, const void* proj_w_t, const float* ln_gamma,
                              const float* bias, const float* head_scale, const int32_t* idx, int64_t roll, const uint8_t* labels,
                              void* dx, float* dqkv_w, float* dqkv_b, float* dproj_w, float* dproj_b, float* dln_gamma, float* dln_beta,
                              float* dbias, float* dhead_scale, float* workspace, int accumulate, int batch, int64_t n_tokens, int channels,
                              int num_heads, int window_size, unsigned flags, int dtype, void* stream) {
    using namespace hs;
    HS_CHECK_ARG(dout && qkv && attn_out && lse && qkv_w_t && proj_w_t && head_scale && dx && dqkv_w && dproj_w && dhead_scale && workspace,
                 "hs_window_attn_module_bwd_chain: null operand");
    HS_CHECK_ARG((ln_gamma != nullptr) == (x && xn && mean && rstd && dln_gamma && dln_beta),
                 "hs_window_attn_module_bwd_chain: with a LayerNorm in front pass x, xn, mean, rstd, dln_gamma, dln_beta; without it none of them");
    HS_CHECK_ARG((bias == nullptr) == (dbias == nullptr), "hs_window_attn_module_bwd_chain: bias and dbias go together");
    if (!hs_window_attn_module_supported(channels, num_heads, window_size, dtype))
        return fail(HS_ERR_UNSUPPORTED, "hs_window_attn_module_bwd_chain: bf16, window 64, head_dim 32 and C = 96 or 128 only");
    const int C = channels;
    const int64_t rows = (int64_t)batch * n_tokens;
    // activation scratch in front of the kernels' workspace
    uint16_t* d_o = (uint16_t*)workspace;
    uint16_t* dqkv = d_o + rows * C;
    uint16_t* dxn = dqkv + rows * 3 * C;
    float* ws = workspace + round4(rows * C * 5 / 2 + 4);
    const bool v1 = ln_gamma != nullptr;
    // ---- proj (reference :172)
    if (int st = hs_linear_wgrad(dout, attn_out, dproj_w, dproj_b, ws, rows, C, C, accumulate, dtype, stream)) return st;
    if (int st = hs_gemm_nt(dout, C, proj_w_t, C, C, nullptr, 0, nullptr, 0, 0, nullptr, d_o, nullptr, rows, C, HS_EPI_BIAS, 0.f, 0, dtype, stream))
        return st;
    // ---- attention core (:136-171 with :319-330 around it).  Its parameter gradients are overwritten or added to as asked.
    const unsigned cflags = (flags & HS_ATTN_COSINE) | (accumulate ? 0u : HS_ATTN_OVERWRITE_GRADS);
    if (int st = hs_window_attn_bwd(qkv, attn_out, d_o, lse, dqkv, dbias, dhead_scale, ws, bias, head_scale, idx, roll, labels, batch, n_tokens, C,
                                    num_heads, window_size, cflags, 0.f, 0, dtype, stream))
        return st;
    // ---- qkv (:136); its input is LayerNorm(x) (v1 placement) or x itself
    const void* qkv_in = v1 ? xn : x;
    if (!v1 && !x) return fail(HS_ERR_INVALID_ARG, "hs_window_attn_module_bwd_chain: x (the qkv Linear's input) is needed for its weight gradient");
    if (int st = hs_linear_wgrad(dqkv, qkv_in, dqkv_w, dqkv_b, ws, rows, 3 * C, C, accumulate, dtype, stream)) return st;
    if (!v1)  // dx = dqkv W_q: the caller adds the block's own residual path (v2 placement: x + norm(branch))
        return hs_gemm_nt(dqkv, 3 * C, qkv_w_t, 3 * C, 3 * C, nullptr, 0, nullptr, 0, 0, nullptr, dx, nullptr, rows, C, HS_EPI_BIAS, 0.f, 0, dtype,
                          stream);
    if (int st = hs_gemm_nt(dqkv, 3 * C, qkv_w_t, 3 * C, 3 * C, nullptr, 0, nullptr, 0, 0, nullptr, dxn, nullptr, rows, C, HS_EPI_BIAS, 0.f, 0, dtype,
                            stream))
        return st;
    // ---- norm1 + the residual add's gradient (:315-316): dx = LayerNorm_bwd(dxn) + dout
    const void* dres = (flags & HS_ATTN_RESIDUAL) ? dout : nullptr;
    if (dres) return hs_add_layernorm_bwd(dxn, dres, x, ln_gamma, mean, rstd, dx, dln_gamma, dln_beta, ws, accumulate, rows, C, dtype, stream);
    return hs_layernorm_bwd(dxn, x, ln_gamma, mean, rstd, dx, dln_gamma, dln_beta, ws, accumulate, rows, C, dtype, stream);
}

}  // extern "C"
