// PatchMerging / PatchExpand / FinalPatchExpand_X4 as single C-ABI operators (hs_patch_merge_*, hs_patch_expand_*).
//
// In nested HEALPix order the reference's data movement is free: the four strided slices + cat of PatchMerging
// (models_torch/swin_hp_transformer.py:385-390) are the view [B, N, C] -> [B, N/4, 4C], and PatchExpand's
// 'b n (p c) -> b (n p) c' (:427, :449) is the view [B, N, p*c] -> [B, N*p, c].  What remains is a row LayerNorm and a
// bias-free Linear, and both already exist as hand-written gfx950 kernels: hs_layernorm_* (layernorm.hip), hs_gemm_nt
// (gemm_nt.hip: forward and input-gradient products) and hs_linear_wgrad (linear_wgrad.hip).  The entry points below chain
// them on the caller's stream for one module call each way, so that an operator-level integration binds ONE symbol per
// module and direction.  (The nn.Module mirror in models_torch/ issues the same kernels itself because it chooses per shape
// between hs_gemm_nt and the library GEMM, see ops.own_gemm_ok.)  bf16 activations only: fp32 runs have no own GEMM.
#include "hs_common.h"

namespace {

int check_common(const char* who, int64_t rows, int dtype) {
    if (dtype != HS_BF16) return hs::fail(HS_ERR_UNSUPPORTED, "%s: bf16 activations only (fp32 runs compose hs_layernorm_* with the library GEMM)", who);
    HS_CHECK_ARG(rows > 0, "%s: empty input", who);
    return HS_OK;
}

inline int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }

}  // namespace

extern "C" {

int hs_patch_merge_fwd(const void* x, const float* gamma, const float* beta, const void* w, void* normed, float* mean, float* rstd,
                       void* out, int64_t rows, int dim, int dim_out, int dtype, void* stream) {
    if (int st = check_common("hs_patch_merge_fwd", rows, dtype)) return st;
    HS_CHECK_ARG(x && gamma && beta && w && normed && out, "hs_patch_merge_fwd: null operand");
    HS_CHECK_ARG(dim > 0 && dim % 2 == 0 && dim_out > 0 && dim_out % 4 == 0, "hs_patch_merge_fwd: dim must be even, dim_out a multiple of 4");
    const int width = 4 * dim;  // the merged row of 4 sibling pixels (:385-391)
    if (int st = hs_layernorm_fwd(x, nullptr, gamma, beta, normed, mean, rstd, rows, width, dtype, stream)) return st;  // :391
    return hs_gemm_nt(normed, width, w, width, width, nullptr, 0, nullptr, 0, 0, nullptr, out, nullptr, rows, dim_out, HS_EPI_BIAS, 0.f,
                      0, dtype, stream);  // :392 (reduction, bias=False)
}

int64_t hs_patch_merge_bwd_workspace(int64_t rows, int dim, int dim_out) {
    return max64(hs_layernorm_bwd_workspace(rows, 4 * dim), hs_linear_wgrad_workspace(rows, dim_out, 4 * dim));
}

int hs_patch_merge_bwd(const void* dout, const void* x, const void* normed, const float* gamma, const float* mean, const float* rstd,
                       const void* w_t, void* dnormed, void* dx, float* dw, float* dgamma, float* dbeta, float* workspace,
                       int accumulate, int64_t rows, int dim, int dim_out, int dtype, void* stream) {
    if (int st = check_common("hs_patch_merge_bwd", rows, dtype)) return st;
    HS_CHECK_ARG(dout && x && normed && gamma && mean && rstd && w_t && dnormed && dx && dw && dgamma && dbeta && workspace,
                 "hs_patch_merge_bwd: null operand");
    const int width = 4 * dim;
    // dW[n, k] = sum_rows dout[row, n] * LN(x)[row, k]
    if (int st = hs_linear_wgrad(dout, normed, dw, nullptr, workspace, rows, dim_out, width, accumulate, dtype, stream)) return st;
    // d LN(x) = dout @ W   (NT product on the transposed weight copy w_t [4 dim, dim_out])
    if (int st = hs_gemm_nt(dout, dim_out, w_t, dim_out, dim_out, nullptr, 0, nullptr, 0, 0, nullptr, dnormed, nullptr, rows, width,
                            HS_EPI_BIAS, 0.f, 0, dtype, stream))
        return st;
    return hs_layernorm_bwd(dnormed, x, gamma, mean, rstd, dx, dgamma, dbeta, workspace, accumulate, rows, width, dtype, stream);
}

int hs_patch_expand_fwd(const void* x, const void* w, const float* gamma, const float* beta, void* expanded, float* mean, float* rstd,
                        void* out, int64_t rows, int dim, int dim_exp, int children, int dtype, void* stream) {
    if (int st = check_common("hs_patch_expand_fwd", rows, dtype)) return st;
    HS_CHECK_ARG(x && w && gamma && beta && expanded && out, "hs_patch_expand_fwd: null operand");
    HS_CHECK_ARG(children > 0 && dim_exp % children == 0 && dim % 8 == 0 && dim_exp % 4 == 0,
                 "hs_patch_expand_fwd: dim_exp must split into `children` rows; dim a multiple of 8");
    // :425 / :447 (expand, bias=False), then LayerNorm over each child row of the 'b n (p c) -> b (n p) c' view (:427-428, :449-450)
    if (int st = hs_gemm_nt(x, dim, w, dim, dim, nullptr, 0, nullptr, 0, 0, nullptr, expanded, nullptr, rows, dim_exp, HS_EPI_BIAS, 0.f, 0,
                            dtype, stream))
        return st;
    return hs_layernorm_fwd(expanded, nullptr, gamma, beta, out, mean, rstd, rows * children, dim_exp / children, dtype, stream);
}

int64_t hs_patch_expand_bwd_workspace(int64_t rows, int dim, int dim_exp, int children) {
    if (children <= 0) return 0;
    return max64(hs_layernorm_bwd_workspace(rows * children, dim_exp / children), hs_linear_wgrad_workspace(rows, dim_exp, dim));
}

int hs_patch_expand_bwd(const void* dout, const void* x, const void* expanded, const float* gamma, const float* mean, const float* rstd,
                        const void* w_t, void* dexpanded, void* dx, float* dw, float* dgamma, float* dbeta, float* workspace,
                        int accumulate, int64_t rows, int dim, int dim_exp, int children, int dtype, void* stream) {
    if (int st = check_common("hs_patch_expand_bwd", rows, dtype)) return st;
    HS_CHECK_ARG(dout && x && expanded && gamma && mean && rstd && w_t && dexpanded && dx && dw && dgamma && dbeta && workspace,
                 "hs_patch_expand_bwd: null operand");
    HS_CHECK_ARG(children > 0 && dim_exp % children == 0, "hs_patch_expand_bwd: dim_exp must split into `children` rows");
    if (int st = hs_layernorm_bwd(dout, expanded, gamma, mean, rstd, dexpanded, dgamma, dbeta, workspace, accumulate, rows * children,
                                  dim_exp / children, dtype, stream))
        return st;
    if (int st = hs_linear_wgrad(dexpanded, x, dw, nullptr, workspace, rows, dim_exp, dim, accumulate, dtype, stream)) return st;
    // dx = dexpanded @ W   (w_t = the transposed weight copy [dim, dim_exp])
    return hs_gemm_nt(dexpanded, dim_exp, w_t, dim_exp, dim_exp, nullptr, 0, nullptr, 0, 0, nullptr, dx, nullptr, rows, dim, HS_EPI_BIAS, 0.f,
                      0, dtype, stream);
}

}  // extern "C"
