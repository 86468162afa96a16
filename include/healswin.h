/*
 * healswin.h -- C ABI of the MI355X-native HEAL-SWIN hot path (libhealswin.so).
 *
 * The reference (JanEGerken/HEAL-SWIN) is pure Python/PyTorch and has no FFI of its own; these are the
 * entry points a maintainer would bind (ctypes) to replace the stock-op call
 * sites of heal_swin/models_torch/{hp_windowing,hp_shifting,swin_hp_transformer}.py.  Each entry
 * point cites the reference interface it replaces as `file:line` relative to /root/reference/heal_swin/.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch types.  Device pointers are marked [dev], host [host].
 *   - every function returns an hs_status (0 = ok); nothing throws across the ABI.
 *   - kernels never allocate or free; all buffers (outputs, workspaces) are owned by the caller.
 *   - device functions are stream-ordered on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) and re-entrant (activation-checkpoint recompute calls them twice).
 *   - activations are row-major [B, N, C] ("token rows"), dtype HS_F32 or HS_BF16; statistics,
 *     softmax, bias and all accumulations are fp32 in both modes.
 *   - "natural order" = the reference's nested HEALPix pixel order; "shifted order" = after
 *     shifter.shift().  Window w of an image covers shifted positions [w*Ws, (w+1)*Ws).
 */
#ifndef HEALSWIN_H
#define HEALSWIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    HS_OK = 0,
    HS_ERR_INVALID_ARG = 1,   /* reference: bare `assert` at construction / call time */
    HS_ERR_UNSUPPORTED = 2,   /* shape/dtype outside what the kernels implement */
    HS_ERR_HIP = 3,           /* a HIP runtime call failed (no device, launch failure); see hs_last_error */
    HS_ERR_NOT_PERMUTATION = 4 /* reference: _validate_shift_result, models_torch/hp_shifting.py:96-99,385-388 */
} hs_status;

typedef enum { HS_F32 = 0, HS_BF16 = 1 } hs_dtype;

/* flags of hs_window_attn_* */
#define HS_ATTN_COSINE 1u      /* cosine attention: L2-normalise q,k (eps 1e-12), per-head scale */
#define HS_ATTN_FORCE_VALU 2u  /* run the generic fp32-VALU kernels even where an MFMA kernel exists (cross-checks, A/B) */
#define HS_ATTN_RESIDUAL 4u    /* hs_window_attn_module_fwd: out = x + module(x) (the block's residual add, :316) */
#define HS_ATTN_OVERWRITE_GRADS 8u /* hs_window_attn_bwd, bf16 MFMA path (window 64, head_dim 32): dbias / dhead_scale are WRITTEN instead of
                                    added into (no zero fill needed); the fp32 and VALU paths ignore the flag and accumulate */

const char* hs_version(void);
/* human-readable message of the last failing call on this thread ("" if none) */
const char* hs_last_error(void);
const char* hs_status_string(int status);
/* number of visible HIP devices (0 on a CPU-only host); never fails */
int hs_device_count(void);

/* `count` 2-D transposes of 16-bit matrices in one launch.  jobs [dev]: `count` records {const void* src [rows, cols] row-major;
 * void* dst [cols, rows] row-major; int64 rows; int64 cols} (32 bytes each).  The per-step [in, out] copies of the Linear weights
 * (B operand of the input-gradient products of hs_gemm_nt; autograd of nn.Linear, models_torch/swin_hp_transformer.py:33-35). */
int hs_transpose_many_16(const void* jobs, int count, int blocks_per_job, void* stream);

/* Compute units that the persistent / one-resident-round launches of this library leave FREE (a multiple of 8 in [0, 128]: the same
 * number on each of the 8 XCDs; default 0, or the environment variable HS_RESERVED_CUS).  Data-parallel training (the reference's
 * Lightning DDP, train.py:182-189) runs RCCL's all-reduce kernels on their own stream DURING the backward; a kernel whose grid
 * is sized to fill every CU for its whole duration either delays them to its end or, if they were resident first, runs its
 * last workgroups as a second round.  heal_swin_amd.parallel.GradBucketAllReduce sets this when world_size > 1.
 * Affects hs_linear_wgrad (+ _workspace), hs_gemm_nt, hs_window_attn_* (+ _workspace), hs_window_attn_module_fwd(_train). */
int hs_set_reserved_cus(int n);
int hs_get_reserved_cus(void);
/* HIP-graph replay with dropout (graphs.GraphedTrainStep; replaces nothing in the reference -- its trainer draws torch's Philox
 * stream per call, `nn.Dropout` at models_torch/swin_hp_transformer.py:36, :122, :126).  Every stochastic entry point takes its seed by
 * value, so a captured launch would repeat its mask on every replay.  hs_set_seed_epoch(counter) registers a [dev] uint64 counter for
 * this process (NULL: off, the default): every mask generator then adds counter * odd constant to its seed at kernel start, and the owner
 * of the graph advances the counter once per replayed step (forward and backward of a step read the same value). */
int hs_set_seed_epoch(const void* counter);
const void* hs_get_seed_epoch(void);
/* Diagnostic: `n_workgroups` workgroups of `threads` threads and `lds_bytes` of LDS each that stay resident for `microseconds`
 * on `stream` -- a stand-in for a communication library's long-lived ring kernels (tools/cu_contention.py). */
int hs_debug_occupy_cus(int n_workgroups, int threads, int lds_bytes, double microseconds, void* stream);
/* Diagnostic (tests): one wavefront loads dword i at vector offset 4 i + scalar offset `soffset` through a raw buffer descriptor
 * over `bytes` bytes of src -- into registers (out128[0..63]) and by LDS-DMA (out128[64..127]).  Pins the hardware rule the
 * role-separated operand DMA of hs_gemm_nt relies on: the scalar offset takes part in the range check (out-of-range dwords
 * read 0), so edge tiles never fetch memory behind an operand. */
int hs_debug_buffer_soffset_probe(const void* src, int bytes, int soffset, void* out128, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Host-side HEALPix index tables, built once per model on the host (plain C++, no GPU needed).
 * ---------------------------------------------------------------------------------------------- */

/* healpy.pixelfunc.nest2ring / ring2nest as called at models_torch/hp_shifting.py:333 / :329.
 * nside must be a power of two; indices in [0, 12*nside^2).  in/out [host], n entries. */
int hs_nest2ring(int nside, const int64_t* in, int64_t* out, int64_t n);
int hs_ring2nest(int nside, const int64_t* in, int64_t* out, int64_t n);

/* get_nest_win_idcs, models_torch/hp_windowing.py:43-62.  out [host] int64[side*side], row-major. */
int hs_nest_win_idcs(int window_size, int64_t* out);

/* relative_position_index of WindowAttention.__init__, models_torch/swin_hp_transformer.py:98-114.
 * out [host] int64[Ws*Ws]. */
int hs_rel_pos_index(int window_size, int64_t* out);

/* Shifters.  All three fill
 *   idx    int32[N]  shift(x)[:, j]      = x[:, idx[j]]     (.shift)
 *   inv    int32[N]  shift_back(y)[:, i] = y[:, inv[i]]     (.shift_back), inv = idx^-1
 *   labels uint8[N]  region label of shifted position j; attention inside a window is masked with
 *                    -100 where labels differ (get_attn_mask_from_mask, models_torch/hp_shifting.py:10-28)
 * any of the three output pointers may be NULL.  N = n_pix (roll) or base_pix*nside^2 (grid, ring). */

/* NestRollShift, models_torch/hp_shifting.py:42-73 (valid for any base_pix). */
int hs_build_nest_roll_shift(int64_t n_pix, int window_size, int shift_size,
                             int32_t* idx, int32_t* inv, uint8_t* labels);
/* NestGridShift, models_torch/hp_shifting.py:76-306 (base_pix must be 8, as the reference asserts :78). */
int hs_build_nest_grid_shift(int nside, int base_pix, int window_size,
                             int32_t* idx, int32_t* inv, uint8_t* labels);
/* RingShift, models_torch/hp_shifting.py:309-404 (only valid for base_pix == 8, like the reference). */
int hs_build_ring_shift(int nside, int base_pix, int window_size, int shift_size,
                        int32_t* idx, int32_t* inv, uint8_t* labels);

/* get_attn_mask_from_mask, models_torch/hp_shifting.py:10-28: out [host] float[nW*Ws*Ws] in {0,-100}.
 * Only needed to emit the reference's `attn_mask` state-dict buffer; the kernels read `labels`. */
int hs_attn_mask_from_labels(const uint8_t* labels, int64_t n, int window_size, float* out);

/* ------------------------------------------------------------------------------------------------
 * Relative-position bias: table[rel_idx] gather and its gradient.
 * Replaces models_torch/swin_hp_transformer.py:152-159 (relative_position_bias_table[index].permute(2,0,1)).
 * ---------------------------------------------------------------------------------------------- */
/* bias[h, i, j] = table[rel_idx[i, j], h].  table [dev] f32[T, nH]; rel_idx [dev] int32[Ws*Ws];
 * bias [dev] f32[nH, Ws, Ws]. */
int hs_rel_bias_gather(const float* table, const int32_t* rel_idx, float* bias,
                       int table_rows, int num_heads, int window_size, void* stream);
/* dtable[t, h] = sum over (i,j) with rel_idx[i,j] == t of dbias[h, i, j]   (dtable is overwritten). */
int hs_rel_bias_scatter_grad(const float* dbias, const int32_t* rel_idx, float* dtable,
                             int table_rows, int num_heads, int window_size, void* stream);
/* The same with the index pre-grouped by table row: order [dev] int32[Ws*Ws] = stable argsort of rel_idx, offsets [dev]
 * int32[T + 1] = start of each row's run in `order`.  One thread per (t, h), entries added in ascending (i, j) order. */
int hs_rel_bias_scatter_grad_sorted(const float* dbias, const int32_t* order, const int32_t* offsets, float* dtable,
                                    int table_rows, int num_heads, int window_size, void* stream);
/* ... ADDED to dtable (an fp32 gradient buffer that already holds other contributions). */
int hs_rel_bias_scatter_grad_sorted_add(const float* dbias, const int32_t* order, const int32_t* offsets, float* dtable,
                                        int table_rows, int num_heads, int window_size, void* stream);
/* Cosine attention's per-head score scale, models_torch/swin_hp_transformer.py:144-147:
 *   scale[h] = exp(min(logit_scale[h], ln 100));   d logit_scale[h] = d scale[h] * scale[h] * [logit_scale[h] <= ln 100]
 * (overwritten, or added to when accumulate != 0).  All [dev] f32[num_heads]. */
int hs_cos_head_scale_fwd(const float* logit_scale, float* scale, int num_heads, void* stream);
int hs_cos_head_scale_bwd(const float* logit_scale, const float* dscale, float* dlogit_scale, int num_heads, int accumulate, void* stream);
/* The same three operations for ALL attention blocks of a model in one launch each (`count` jobs; host arrays of device pointers and
 * per-job head counts; every job shares rel_idx / order / offsets, i.e. the window size):
 *   hs_rel_bias_gather_many: bias of job j = bias_base + (heads[0] + ... + heads[j-1]) * Ws*Ws, f32[heads[j], Ws, Ws];
 *   hs_rel_bias_scatter_grad_sorted_many: dtables[j] f32[T, heads[j]] overwritten, or added to where accumulate[j] != 0;
 *   hs_cos_head_scale_many: dscale == NULL: out[j][h] = exp(min(logit_scale[j][h], ln 100)); else out[j][h] (+)= the gradient above
 *   (heads[j] <= 64). */
int hs_rel_bias_gather_many(const void* const* tables, const int* heads, int count, const int32_t* rel_idx, float* bias_base,
                            int table_rows, int window_size, void* stream);
int hs_rel_bias_scatter_grad_sorted_many(const void* const* dbias, void* const* dtables, const int* heads, const int* accumulate, int count,
                                         const int32_t* order, const int32_t* offsets, int table_rows, int window_size, void* stream);
int hs_cos_head_scale_many(const void* const* logit_scale, const void* const* dscale, void* const* out, const int* heads, const int* accumulate,
                           int count, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused  shift -> window_partition -> attention core -> window_reverse -> shift_back.
 *
 * Replaces, in one kernel, models_torch/swin_hp_transformer.py:319 (shifter.shift), :322
 * (window_partition, hp_windowing.py:6-21), :136-171 of WindowAttention.forward (q/k/v split, cosine
 * or scaled QK^T, + relative position bias, + shift mask, softmax, attn @ v, head merge), :327
 * (window_reverse, hp_windowing.py:24-40) and :330 (shifter.shift_back).  Row permutations commute
 * with the row-wise qkv / proj Linear layers, so the kernel gathers rows of the *unshifted* qkv
 * tensor through `idx` and scatters its output rows back through the same `idx`.
 *
 *   qkv    [dev] dtype[B, N, 3C]  output of the qkv Linear in natural order, columns [q|k|v][head][hd]
 *   out    [dev] dtype[B, N, C]   attention output in natural order, columns [head][hd] (input of proj)
 *   lse    [dev] f32[B, nH, N]    log-sum-exp of every score row, indexed by SHIFTED position; saved
 *                                 for the backward pass (may be NULL for inference)
 *   bias   [dev] f32[nH, Ws, Ws]  or NULL (rel_pos_bias is None)
 *   head_scale [dev] f32[nH]      cosine: exp(min(logit_scale, ln 100)) (:144-147); else qk scale (:81,:149)
 *   idx    [dev] int32[N] gather table of the shifter, or NULL: then shifted position j reads token
 *                         (j + roll) mod N   (roll = 0: NoShift; roll = shift_size: NestRollShift)
 *   labels [dev] uint8[N] region labels in shifted order, or NULL (no mask; unshifted blocks)
 *   flags  HS_ATTN_COSINE and/or HS_ATTN_FORCE_VALU, or 0
 *   attn_drop, seed  attention dropout of :169 (self.attn_drop on the probabilities): each probability is zeroed with
 *                    probability attn_drop, survivors scaled by 1/(1-attn_drop); the mask is a pure function of
 *                    (seed, image, head, query, key), so hs_window_attn_bwd called with the same seed reproduces it.
 *                    attn_drop = 0 (eval mode / p = 0) disables it.
 * Supported: Ws in {4,16,64,256}, Ws <= N, N % Ws == 0, head_dim in {1,2,4,8,16,32,64,128}.
 * (Ws = 64, head_dim = 32, bf16 takes the MFMA path; everything else the fp32-VALU path.)
 */
int hs_window_attn_fwd(const void* qkv, void* out, float* lse,
                       const float* bias, const float* head_scale,
                       const int32_t* idx, int64_t roll, const uint8_t* labels,
                       int batch, int64_t n_tokens, int channels, int num_heads, int window_size,
                       unsigned flags, float attn_drop, uint64_t seed, int dtype, void* stream);

/* Backward of the above.
 *   dout   [dev] dtype[B, N, C]   gradient w.r.t. `out`
 *   dqkv   [dev] dtype[B, N, 3C]  gradient w.r.t. `qkv` (every row is written exactly once)
 *   dbias  [dev] f32[nH, Ws, Ws]  ACCUMULATED into (caller zeroes it), or NULL when bias is NULL
 *   dhead_scale [dev] f32[nH]     ACCUMULATED into (caller zeroes it); only written for HS_ATTN_COSINE
 *                                 (the scaled variant's scale is a constant), may be NULL otherwise
 *   workspace [dev] f32[hs_window_attn_bwd_workspace(...)]  per-workgroup partial bias/scale gradients of the
 *                                 MFMA path (reduced deterministically, no atomics); may be NULL when the query
 *                                 returns 0 (fp32-VALU path)
 */
int64_t hs_window_attn_bwd_workspace(int batch, int64_t n_tokens, int channels, int num_heads, int window_size, int dtype);
int hs_window_attn_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                       void* dqkv, float* dbias, float* dhead_scale, float* workspace,
                       const float* bias, const float* head_scale,
                       const int32_t* idx, int64_t roll, const uint8_t* labels,
                       int batch, int64_t n_tokens, int channels, int num_heads, int window_size,
                       unsigned flags, float attn_drop, uint64_t seed, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Standalone shift along the nested pixel axis: out[b, j, :] = x[b, idx[j], :]  (idx NULL: roll,
 * out[b, j, :] = x[b, (j + roll) mod N, :]).  Replaces shifter.shift / shift_back when called on their
 * own: models_torch/hp_shifting.py:69-73 (torch.roll), :302-306 and :400-404 (x[:, idcs].contiguous()).
 * Pass `inv` as idx for shift_back.  Not on the model's path (the permutation is fused into
 * hs_window_attn_*); used by the shifter API and by the gather/scatter bandwidth measurement.
 *   x, out [dev] distinct buffers of batch * n_tokens rows of row_bytes bytes each.
 * ---------------------------------------------------------------------------------------------- */
int hs_gather_rows(const void* x, void* out, const int32_t* idx, int64_t roll,
                   int batch, int64_t n_tokens, int64_t row_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Row LayerNorm over contiguous rows (eps = 1e-5, affine), optionally fused with a residual add.
 * This is the normalisation half of PatchMerging (LayerNorm over the 4C row formed by 4 sibling
 * pixels, models_torch/swin_hp_transformer.py:385-392: the slicing + cat is a free view in nested
 * order), of PatchExpand / FinalPatchExpand_X4 (LayerNorm over each C/p child row after the
 * 'b n (p c) -> b (n p) c' view, :427-428, :449-450) and of the block norms (:316, :334-338, :945, :781).
 *
 *   y = LN(x) * gamma + beta               (residual == NULL)
 *   y = residual + LN(x) * gamma + beta    (v2 norm placement, :334-335)
 *   x, y, residual [dev] dtype[rows, width]; gamma, beta [dev] f32[width];
 *   mean, rstd [dev] f32[rows] saved for backward (may be NULL for inference).
 */
int hs_layernorm_fwd(const void* x, const void* residual, const float* gamma, const float* beta,
                     void* y, float* mean, float* rstd,
                     int64_t rows, int width, int dtype, void* stream);
/* dx [dev] dtype[rows, width]; dgamma, dbeta [dev] f32[width]: sum over all rows, OVERWRITTEN, or ADDED to
 * the existing contents when accumulate != 0 (a parameter's fp32 .grad buffer).
 * The residual branch's gradient is dy itself (identity) and is not produced here.
 * workspace [dev] f32[hs_layernorm_bwd_workspace(rows, width)] */
int64_t hs_layernorm_bwd_workspace(int64_t rows, int width);
int hs_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                     void* dx, float* dgamma, float* dbeta, float* workspace, int accumulate,
                     int64_t rows, int width, int dtype, void* stream);

/* Fused residual add + LayerNorm (v1 norm placement, models_torch/swin_hp_transformer.py:337-338 and :316 of the
 * following block):   sum = a + b (stored in the activation dtype),  y = LN(sum) * gamma + beta.
 * Backward: dx = LN_bwd(dy) + dsum, which is the gradient of BOTH a and b (dsum may be NULL: no other consumer of sum);
 * `sum` is the tensor saved by the forward; workspace as for hs_layernorm_bwd. */
int hs_add_layernorm_fwd(const void* a, const void* b, const float* gamma, const float* beta,
                         void* sum_out, void* y, float* mean, float* rstd,
                         int64_t rows, int width, int dtype, void* stream);
int hs_add_layernorm_bwd(const void* dy, const void* dsum, const void* sum, const float* gamma,
                         const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta, float* workspace,
                         int accumulate, int64_t rows, int width, int dtype, void* stream);

/* Train-mode variants with the block's stochastic regularisers fused in (models_torch/swin_hp_transformer.py:173 proj_drop,
 * :43 Mlp output dropout, :334-338 DropPath): drop() is a counter-based dropout mask (drop_p, seed; regenerated by the
 * backward called with the same seed), rs = row_scale[row / rows_per_sample] the per-sample DropPath factor
 * (0 or 1/keep; row_scale [dev] f32[rows / rows_per_sample], may be NULL = 1).
 *   hs_layernorm_drop_*      (v2 placement):  y = [residual +] rs * LN(drop(x));        dx = mask * LN_bwd(rs * dy)
 *   hs_add_layernorm_drop_*  (v1 placement):  sum = a + rs * drop(b),  y = LN(sum);     da = g, db = rs * mask * g
 *                                             with g = LN_bwd(dy) + dsum */
int hs_layernorm_drop_fwd(const void* x, const void* residual, const float* gamma, const float* beta, void* y,
                          float* mean, float* rstd, const float* row_scale, int64_t rows_per_sample, float drop_p,
                          uint64_t seed, int64_t rows, int width, int dtype, void* stream);
int hs_layernorm_drop_bwd(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd,
                          void* dx, float* dgamma, float* dbeta, float* workspace, int accumulate, const float* row_scale,
                          int64_t rows_per_sample, float drop_p, uint64_t seed, int64_t rows, int width, int dtype,
                          void* stream);
int hs_add_layernorm_drop_fwd(const void* a, const void* b, const float* gamma, const float* beta, void* sum_out, void* y,
                              float* mean, float* rstd, const float* row_scale, int64_t rows_per_sample, float drop_p,
                              uint64_t seed, int64_t rows, int width, int dtype, void* stream);
/* General forward with every optional operand, incl. the COMPENSATED RESIDUAL STREAM (replaces the reference's two plain adds per
 * block, swin_hp_transformer.py:316 / :338 (v1) and :334-335 (v2), whose results it keeps to 16 instead of 8 mantissa bits in bf16):
 * exactly one of `residual` (v2: y = residual + rs LN(drop(x))) and `add_in` (v1: sum_out = x + rs drop(add_in), y = LN(sum_out)).
 * lo_in (optional) is the rounding remainder of the stream operand (x for v1, residual for v2) left by the previous call, lo_out
 * (optional) receives the remainder of the new stream (sum_out for v1, y for v2; for a plain LayerNorm -- neither residual nor add_in
 * -- the rounding remainder of y, for a consumer that takes its operand as hi + lo: hs_expand_ln_head_fwd): stream = hi + lo with hi
 * the plain activation tensor every other kernel reads.  Backward: the plain hs_*layernorm*_bwd entry points on the hi tensors. */
int hs_layernorm_fwd_ex(const void* x, const void* residual, const void* add_in, const void* lo_in, const float* gamma,
                        const float* beta, void* y, void* sum_out, void* lo_out, float* mean, float* rstd,
                        const float* row_scale, int64_t rows_per_sample, float drop_p, uint64_t seed, int64_t rows, int width,
                        int dtype, void* stream);
int hs_add_layernorm_drop_bwd(const void* dy, const void* dsum, const void* sum, const float* gamma, const float* mean,
                              const float* rstd, void* da, void* db, float* dgamma, float* dbeta, float* workspace, int accumulate,
                              const float* row_scale, int64_t rows_per_sample, float drop_p, uint64_t seed,
                              int64_t rows, int width, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * y = dropout(GELU(x)) (exact erf GELU) and its backward dx = dy * mask/(1-p) * GELU'(x): the activation and the
 * dropout behind it in Mlp.forward (models_torch/swin_hp_transformer.py:39-41) in one pass.  drop_p = 0 (eval) is plain
 * GELU.  The mask is a pure function of (seed, element index): pass the forward's seed to the backward.
 * x, y, dy, dx [dev] dtype[n], 16-byte aligned.
 * ---------------------------------------------------------------------------------------------- */
int hs_gelu_fwd(const void* x, void* y, int64_t n, float drop_p, uint64_t seed, int dtype, void* stream);
int hs_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, float drop_p, uint64_t seed, int dtype, void* stream);

/* out = x + rs * drop(t): a residual branch added in its standalone form (end of a stage; dropout on the branch output
 * models_torch/swin_hp_transformer.py:43, :173 and DropPath :334-338 in one pass).  rs = row_scale[i / elems_per_sample]
 * (row_scale [dev] f32[n / elems_per_sample], NULL = 1); x == NULL gives out = rs * drop(t), which is also the backward:
 * dt = hs_residual_drop(NULL, dy, ...) with the forward's seed (dx = dy). */
int hs_residual_drop(const void* x, const void* t, void* out, const float* row_scale, int64_t elems_per_sample, int64_t n,
                     float drop_p, uint64_t seed, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Class-weighted cross-entropy of the segmentation caller (reference: nn.CrossEntropyLoss(weight)(logits[B,K,Npix],
 * labels.long()[B,Npix]), models_lightning/segmentation/model_lightning_swin_hp.py:39-45, :104-111):
 *     loss = sum_i w[y_i] (logsumexp_c z_i[c] - z_i[y_i]) / sum_i w[y_i]
 * logits [dev] dtype, element (b, c, pixel) at b*stride_b + c*stride_k + pixel*stride_p (so the model's native
 * [B, Npix, K] output viewed as [B, K, Npix] needs no copy); labels [dev] uint8 / int32 / int64 (label_bytes 1/4/8)
 * [batch, npix] contiguous; class_weights [dev] f32[n_classes] or NULL; pixels labelled ignore_index (or out of range)
 * are skipped.  n_classes <= 64.
 *   hs_seg_ce_fwd: partials [dev] f32[hs_seg_ce_partials(batch, npix)][2] = per-workgroup (numerator, denominator);
 *                  the caller sums them (fixed order) and divides.
 *   hs_seg_ce_bwd: dlogits (own strides) = scale[0] * w[y] * (softmax(z) - onehot(y)), scale [dev] f32[1] =
 *                  upstream gradient / denominator (a device scalar: no host synchronisation).
 * ---------------------------------------------------------------------------------------------- */
int64_t hs_seg_ce_partials(int64_t batch, int64_t npix);
int hs_seg_ce_fwd(const void* logits, const void* labels, const float* class_weights, float* partials,
                  int64_t batch, int64_t npix, int n_classes, int64_t stride_b, int64_t stride_k, int64_t stride_p,
                  int label_bytes, int64_t ignore_index, int dtype, void* stream);
int hs_seg_ce_bwd(const void* logits, const void* labels, const float* class_weights, const float* scale, void* dlogits,
                  int64_t batch, int64_t npix, int n_classes, int64_t stride_b, int64_t stride_k, int64_t stride_p,
                  int64_t dstride_b, int64_t dstride_k, int64_t dstride_p, int label_bytes, int64_t ignore_index,
                  int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Weight / bias gradient of the path's Linear layers (autograd of nn.Linear at
 * models_torch/swin_hp_transformer.py:33,:35 (Mlp), :116,:118 (qkv, proj), :375 (PatchMerging.reduction), :415-416
 * (PatchExpand.expand), :438, :717 (concat_back_dim)):
 *     dw[n, k] = sum_m dy[m, n] * x[m, k]        dbias[n] = sum_m dy[m, n]
 *   dy [dev] dtype[rows, n_out], x [dev] dtype[rows, k_in] (row-major token rows); dw [dev] f32[n_out, k_in] and
 *   dbias [dev] f32[n_out] (may be NULL) are overwritten (accumulate == 0) or added to (accumulate != 0: the
 *   caller's .grad buffers); workspace [dev] f32[hs_linear_wgrad_workspace(...)].
 * dtype HS_BF16: dy, x bf16; k_in a multiple of 8, n_out a multiple of 4 (of 8 when a token slice exceeds 2 GiB).
 * dtype HS_F32: dy, x f32 (v_mfma_f32_32x32x2_f32); n_out and k_in multiples of 4.
 * Split over the token axis, deterministic.
 * ---------------------------------------------------------------------------------------------- */
int64_t hs_linear_wgrad_workspace(int64_t rows, int n_out, int k_in);
int hs_linear_wgrad(const void* dy, const void* x, float* dw, float* dbias, float* workspace,
                    int64_t rows, int n_out, int k_in, int accumulate, int dtype, void* stream);
/* dw[n, k] = sum_m dy[m, n] * gelu(h[m, k]): the weight gradient of a Linear whose input is gelu(h), taken from the saved
 * pre-activation h (fc2 of reference Mlp.forward :38-44) -- the fused Mlp block then keeps h only.  Exact-erf GELU as everywhere
 * (csrc/hs_gelu.h), applied to the MFMA operand fragments; bf16 and the shapes of hs_linear_wgrad_gelu_supported (the 128 x 128
 * tile: C <= 128-class layers, whose launches are HBM-bound and have the VALU slots). */
int hs_linear_wgrad_gelu_supported(int64_t rows, int n_out, int k_in, int dtype);
int hs_linear_wgrad_gelu(const void* dy, const void* h, float* dw, float* dbias, float* workspace, int64_t rows, int n_out, int k_in,
                         int accumulate, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Deferred parameter-gradient reductions.  hs_linear_wgrad / hs_linear_wgrad_ld and the four LayerNorm backward entry points
 * (hs_layernorm_bwd, hs_add_layernorm_bwd, hs_layernorm_drop_bwd, hs_add_layernorm_drop_bwd) end in a small "sum the
 * per-workgroup partial records into dw / dbias (dgamma / dbeta)" launch.  With HS_ACC_DEFER or-ed into `accumulate` that sum is
 * QUEUED on the call's stream instead of launched; hs_reduce_flush(stream) folds every queued sum of that stream in ONE launch
 * (deterministic order; the queue also flushes itself when it is full).  Contract of a deferring caller: the call's `workspace`
 * stays allocated and untouched until the flush, and nobody reads the gradient buffers before it.  The sums of one flush run
 * side by side without atomics, so two PENDING sums never share a destination element: a call whose dw / dbias range overlaps
 * a pending one (a parameter used twice in a backward, the bf16x3 products of an fp32 weight gradient, micro-batches without a
 * flush in between) flushes the queue first, and so does an immediate (non-deferred) sum into such a range -- issue order is
 * stream order either way.  Destination buffers must outlive the flush: flush before freeing gradient buckets.  The reference has no
 * counterpart (autograd of nn.Linear / nn.LayerNorm, models_torch/swin_hp_transformer.py:33-35, :116-118, :256-262): this is
 * how 340 launches per HEAL-SWIN-B training step become about 10.
 * ---------------------------------------------------------------------------------------------- */
#define HS_ACC_DEFER 2 /* or-ed into `accumulate` (bit 0: add to the existing contents) */
int hs_reduce_pending(void* stream); /* queued sums of that stream */
int hs_reduce_flush(void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused Mlp block of the HBM-bound stages (C = 96 / 128, hidden = 4 C; csrc/mlp_fused.hip): replaces, in ONE launch per direction,
 *   forward   the block's second residual branch  out = x + fc2(gelu(fc1(LayerNorm(x))))
 *             (models_torch/swin_hp_transformer.py:337-338 with Mlp.forward :38-44 and norm2 :262; v1 norm placement)
 *   backward  the input-gradient half of Mlp's autograd:  dh = (dy W2) * gelu'(h),  dn = dh W1
 * hs_mlp_fused_fwd: x, out [dev] bf16[rows, C]; ln_gamma, ln_beta [dev] f32[C] (both NULL: no LayerNorm in front);
 *   w1 [dev] bf16[4C, C], b1 f32[4C] | NULL, w2 [dev] bf16[C, 4C], b2 f32[C] | NULL (nn.Linear layouts);
 *   saved for the backward, each may be NULL (not kept): n_out bf16[rows, C] = LayerNorm(x), mean_out / rstd_out f32[rows],
 *   h_out bf16[rows, 4C] = fc1 output, act_out bf16[rows, 4C] = gelu(h);  flags: HS_ATTN_RESIDUAL adds x to the result.
 * hs_mlp_fused_bwd: dy [dev] bf16[rows, C], h [dev] bf16[rows, 4C] (saved), w2_t [dev] bf16[4C, C] and w1_t [dev] bf16[C, 4C] = the
 *   TRANSPOSED weights; writes dh [dev] bf16[rows, 4C] (operand of fc1's weight gradient) and dn [dev] bf16[rows, C] = dh W1 (+ dres
 *   [dev] bf16[rows, C] when non-NULL: the residual path's gradient of the v2 placement, added in the epilogue; v1: the gradient of
 *   LayerNorm(x)).  Weight / bias / LayerNorm gradients: hs_linear_wgrad(dy, gelu(h)), hs_linear_wgrad(dh, n), hs_add_layernorm_bwd.
 * rows: a multiple of 32.  HS_ERR_UNSUPPORTED outside hs_mlp_fused_supported (the weights live in registers: C <= 128).
 * ---------------------------------------------------------------------------------------------- */
#define HS_MLP_NORM_AFTER 16u /* hs_mlp_fused_fwd flags: v2 norm placement (:334-335), out = x + LayerNorm(fc2(gelu(fc1(x)))): ln_gamma / ln_beta
                                 apply BEHIND the Mlp, n_out receives the un-normalised rows mlp(x) (the LayerNorm backward's input) and
                                 mean_out / rstd_out their statistics */
int hs_mlp_fused_supported(int channels, int hidden, int dtype);
int hs_mlp_fused_fwd(const void* x, const float* ln_gamma, const float* ln_beta, const void* w1, const float* b1, const void* w2,
                     const float* b2, void* n_out, float* mean_out, float* rstd_out, void* h_out, void* act_out, void* out, int64_t rows,
                     int channels, int hidden, unsigned flags, int dtype, void* stream);
int hs_mlp_fused_bwd(const void* dy, const void* h, const void* w2_t, const void* w1_t, const void* dres, void* dh, void* dn, int64_t rows,
                     int channels, int hidden, int dtype, void* stream);
/* Train-mode form of the v2 placement with the branch's regularisers inside the launch (flags must carry HS_MLP_NORM_AFTER):
 *   out = x + rs * LayerNorm(drop_o(fc2(drop_h(gelu(fc1(x))))))       Mlp.drop behind the activation and behind fc2 (:41, :43), DropPath (:335)
 * drop_h / drop_o are the counter-based masks of hs_gemm_nt's GELU epilogue (seed_hidden, element index in [rows, 4C]) and of
 * hs_layernorm_drop_fwd (seed_out, [rows, C]); rs = row_scale[row / rows_per_sample] (row_scale [dev] f32 or NULL; rows_per_sample a multiple
 * of 32).  m_out = fc2's output BEFORE drop_o (hs_layernorm_drop_bwd regenerates the mask), mean_out / rstd_out the statistics of the dropped
 * rows, act_out = the dropped activation (operand of fc2's weight gradient).  Backward: hs_layernorm_drop_bwd(dout, m_out, ...) -> dm, then
 * hs_mlp_fused_drop_bwd(dm, h, ..., dres = dout): dh = (dm W2) * mask_h * gelu'(h), dn = dh W1 + dres. */
int hs_mlp_fused_drop_fwd(const void* x, const float* ln_gamma, const float* ln_beta, const void* w1, const float* b1, const void* w2,
                          const float* b2, void* m_out, float* mean_out, float* rstd_out, void* h_out, void* act_out, void* out,
                          const float* row_scale, int64_t rows_per_sample, float drop_p, uint64_t seed_hidden, uint64_t seed_out,
                          int64_t rows, int channels, int hidden, unsigned flags, int dtype, void* stream);
int hs_mlp_fused_drop_bwd(const void* dy, const void* h, const void* w2_t, const void* w1_t, const void* dres, void* dh, void* dn,
                          float drop_p, uint64_t seed_hidden, int64_t rows, int channels, int hidden, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer step over flat buffers: torch.optim.Adam / AdamW (the reference's training/optimizer.py:57-66; amsgrad = False,
 * maximize = False) on n consecutive fp32 parameters p with gradients g and moments m, v, all [dev] f32[n], 16-byte aligned:
 *     g += wd p (decoupled == 0)  |  p *= 1 - lr wd (decoupled != 0: AdamW)
 *     m += (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g^2;  p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 *   t = *step + 1 with step [dev] int64 (the steps taken so far; hs_adam_advance adds 1 after the last buffer of a step);
 *   lr_dev [dev] f32[1] overrides lr when non-NULL; p_bf16 [dev] bf16[n] (may be NULL) receives the updated parameters rounded to
 *   bf16 -- the copy the next forward's GEMMs read, written from the registers that hold the new value.
 * ---------------------------------------------------------------------------------------------- */
int hs_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, const float* lr_dev, float beta1,
                 float beta2, float eps, float weight_decay, int decoupled, const int64_t* step, void* stream);
int hs_adam_advance(int64_t* step, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused WindowAttention MODULE forward (inference form here, training form and the module backward below): the whole of WindowAttention.forward,
 * models_torch/swin_hp_transformer.py:124-174 -- qkv Linear, head split, (cosine | scaled) scores, relative-position bias,
 * shift mask, softmax, P V, head merge, proj Linear -- with the shift / window partition / reverse / shift back of
 * SwinTransformerBlock.forward (:319-330) and optionally the block's norm1 in front (:315) and residual add behind (:316):
 *     out[b, t, :] = [x[b, t, :] +] proj( attention( qkv( [LayerNorm](x) ) ) )[b, t, :]
 * in ONE launch: x is read once, out written once; qkv [B, N, 3C] and the attention output never exist in HBM
 * (SURVEY 8b's proposed entry point, 8d's fused-module roofline).
 *   x, out      [dev] bf16 [batch, n_tokens, channels], natural order; out may not alias x
 *   qkv_w       [dev] bf16 [3 * channels, channels] (nn.Linear layout, rows [q | k | v][head][32]); qkv_b [dev] f32 or NULL
 *   proj_w      [dev] bf16 [channels, channels]; proj_b [dev] f32 [channels] or NULL
 *   ln_gamma/ln_beta [dev] f32 [channels] or both NULL (eps 1e-5)
 *   bias, head_scale, idx, roll, labels: as hs_window_attn_fwd.  flags: HS_ATTN_COSINE, HS_ATTN_RESIDUAL.
 * Supported where the qkv weights fit the LDS: dtype HS_BF16, window_size 64, head_dim 32, channels 96 or 128 (stage 0 of
 * HEAL-SWIN-T / -B); everything else returns HS_ERR_UNSUPPORTED (hs_window_attn_module_supported tells beforehand) and the
 * caller composes hs_gemm_nt / hs_window_attn_fwd / hs_gemm_nt.  No dropout (inference).
 * ---------------------------------------------------------------------------------------------- */
int hs_window_attn_module_supported(int channels, int num_heads, int window_size, int dtype);
int hs_window_attn_module_fwd(const void* x, void* out, const void* qkv_w, const float* qkv_b, const void* proj_w,
                              const float* proj_b, const float* ln_gamma, const float* ln_beta, const float* bias,
                              const float* head_scale, const int32_t* idx, int64_t roll, const uint8_t* labels, int batch,
                              int64_t n_tokens, int channels, int num_heads, int window_size, unsigned flags, int dtype,
                              void* stream);

/* TRAINING form of the module forward (reference :315-316 around :124-174): the same single launch,
 *     out = [x +] proj( attention( qkv( [LayerNorm](x) ) ) )        (flags: HS_ATTN_RESIDUAL as above; v1 norm placement: LayerNorm +
 *                                                                     residual, v2 placement (:334-335): neither -- xn_out, mean_out,
 *                                                                     rstd_out NULL then, the qkv Linear's input is x itself)
 * which ALSO writes, in natural token order, everything the backward of the four reference modules reads -- so that
 * hs_add_layernorm_bwd (norm1), hs_linear_wgrad + hs_gemm_nt (qkv, proj) and hs_window_attn_bwd run on them unchanged:
 *   xn_out   [dev] bf16 [batch, n_tokens, channels]      LayerNorm(x): the qkv Linear's input
 *   mean_out, rstd_out [dev] f32 [batch * n_tokens]      the LayerNorm row statistics (as hs_layernorm_fwd)
 *   qkv_out  [dev] bf16 [batch, n_tokens, 3 * channels]  the qkv Linear's output (as hs_gemm_nt would have rounded it)
 *   attn_out [dev] bf16 [batch, n_tokens, channels]      hs_window_attn_fwd's `out`: the proj Linear's input
 *   lse_out  [dev] f32 [batch, num_heads, n_tokens]      hs_window_attn_fwd's `lse` (by shifted position)
 * HBM traffic per token: x in, out written, 5 C saved = 7 C * 2 B (+ 8 + 4 nH bytes of statistics) against 13 C * 2 B for
 * hs_layernorm_fwd -> hs_gemm_nt -> hs_window_attn_fwd -> hs_gemm_nt(HS_EPI_RESID), none of the saved tensors re-read in the forward.
 * norm2_gamma / norm2_beta / n2_out / mean2_out / rstd2_out (all five or none): the block's SECOND LayerNorm (:337) applied to `out`
 * in the same launch: n2_out [dev] bf16 [batch, n_tokens, channels] = LayerNorm(out) (what the MLP reads), its row statistics in
 * mean2_out / rstd2_out -- as hs_layernorm_fwd on `out` would have written them.
 * Same support set and argument meaning as hs_window_attn_module_fwd. */
int hs_window_attn_module_fwd_train(const void* x, void* out, void* xn_out, float* mean_out, float* rstd_out, void* qkv_out,
                                    void* attn_out, float* lse_out, const void* qkv_w, const float* qkv_b, const void* proj_w,
                                    const float* proj_b, const float* ln_gamma, const float* ln_beta, const float* bias,
                                    const float* head_scale, const int32_t* idx, int64_t roll, const uint8_t* labels,
                                    const float* norm2_gamma, const float* norm2_beta, void* n2_out, float* mean2_out, float* rstd2_out,
                                    int batch, int64_t n_tokens, int channels, int num_heads, int window_size, unsigned flags, int dtype,
                                    void* stream);

/* Backward of the same unit in one call: the gradients of out = [x +] proj(attention(qkv([LayerNorm](x)))) with respect to x and every
 * parameter, from dout and the tensors hs_window_attn_module_fwd_train saved.  Host-side chain of the kernels above and below on
 * `stream` (hs_linear_wgrad + hs_gemm_nt for proj, hs_window_attn_bwd, hs_linear_wgrad + hs_gemm_nt for qkv,
 * hs_add_layernorm_bwd for norm1 and the residual), not a fused kernel (DESIGN 4.6 says why).
 *   dout, x, xn, qkv, attn_out [dev] bf16 as written by / passed to the forward; mean, rstd, lse [dev] f32 likewise
 *   qkv_w_t [dev] bf16 [C, 3C], proj_w_t [dev] bf16 [C, C]: the TRANSPOSED weight copies (input-gradient products)
 *   ln_gamma NULL (v2 placement): x is the qkv Linear's input, xn / mean / rstd / dln_* are NULL, dx = dqkv W_q only
 *   dx [dev] bf16 [batch, n_tokens, C]; dqkv_w f32 [3C, C], dqkv_b f32 [3C] or NULL, dproj_w f32 [C, C], dproj_b f32 [C] or NULL,
 *   dln_gamma / dln_beta f32 [C], dbias f32 [nH, Ws, Ws] (NULL iff bias is), dhead_scale f32 [nH]: overwritten, or added to when
 *   accumulate != 0;  workspace [dev] f32 [hs_window_attn_module_bwd_chain_workspace(...)]
 *   flags: HS_ATTN_COSINE, HS_ATTN_RESIDUAL as in the forward call. */
int64_t hs_window_attn_module_bwd_chain_workspace(int batch, int64_t n_tokens, int channels, int num_heads, int window_size);
int hs_window_attn_module_bwd_chain(const void* dout, const void* x, const void* xn, const float* mean, const float* rstd, const void* qkv,
                              const void* attn_out, const float* lse, const void* qkv_w_t, const void* proj_w_t, const float* ln_gamma,
                              const float* bias, const float* head_scale, const int32_t* idx, int64_t roll, const uint8_t* labels,
                              void* dx, float* dqkv_w, float* dqkv_b, float* dproj_w, float* dproj_b, float* dln_gamma, float* dln_beta,
                              float* dbias, float* dhead_scale, float* workspace, int accumulate, int batch, int64_t n_tokens, int channels,
                              int num_heads, int window_size, unsigned flags, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Forward and input-gradient product of the path's Linear layers with the elementwise step behind it fused into the
 * epilogue (bf16 activations; replaces `F.linear` + `nn.GELU` + `nn.Dropout` at models_torch/swin_hp_transformer.py:38-44
 * (Mlp.forward), :136/:172 (qkv, proj), :392 (PatchMerging.reduction), :425/:447 (expand), :772-775 (skip concat + Linear),
 * :785-788 (1x1 head) and their autograd input gradients):
 *     acc[m, n] = sum_k a[m, k] * b[n, k]   (+ sum_k a2[m, k] * b2[n, k]   when k2 > 0)
 *   a  [dev] bf16, row m at a + m*lda (k valid elements); b [dev] bf16 [n rows], row n at b + n*ldb: a weight exactly as
 *   nn.Linear stores it ([out, in]); for an input gradient pass the transposed bf16 copy of the weight.  The second segment
 *   contracts a second operand pair into the same accumulator: cat([x, skip], -1) @ W^T without the concatenation
 *   (b = W, b2 = W + k, ldb = ldb2 = k + k2).  bias [dev] f32[n] or NULL.  c, aux [dev] bf16 [m, n] contiguous.
 *   epilogue HS_EPI_BIAS : c = acc + bias
 *            HS_EPI_GELU : c = h = acc + bias (skipped when c == NULL), aux = dropout(gelu_erf(h), drop_p, seed) [written]
 *            HS_EPI_DGELU: c = acc * dropout_mask * gelu_erf'(aux)          (aux = the saved h [read]; bias ignored)
 *            HS_EPI_RESID: c = acc + bias + aux                              (aux = a residual term [read])
 *   The dropout mask is the one hs_gelu_fwd/bwd draw for the same (seed, element index m*n_cols + n).
 *   k, k2, lda, ldb multiples of 8; n a multiple of 4; dtype must be HS_BF16 (fp32 runs keep the library GEMM).
 * hs_gemm_nt_set_tile: measurement hook (0 = built-in choice, 1 = 128x128 tiles, 2 = 256x128 x 3 stages, 3 = 256x256 with
 * 8 waves, 4 = 256x256 with 4 waves of 128x128 and the accumulators in AGPRs: never chosen by 0, kept for A/B runs).
 * ---------------------------------------------------------------------------------------------- */
#define HS_EPI_BIAS 0
#define HS_EPI_GELU 1
#define HS_EPI_DGELU 2
#define HS_EPI_RESID 3
int hs_gemm_nt(const void* a, int64_t lda, const void* b, int64_t ldb, int k, const void* a2, int64_t lda2, const void* b2,
               int64_t ldb2, int k2, const float* bias, void* c, void* aux, int64_t m, int n, int epilogue, float drop_p,
               uint64_t seed, int dtype, void* stream);
int hs_gemm_nt_set_tile(int variant);

/* ------------------------------------------------------------------------------------------------
 * Fisheye image -> HEALPix projection, the input pipeline in front of the model.  Replaces, per image, the
 * sampling of data/segmentation/project_on_s2.py:344-372; the coordinate table (u, v) of a calibration is built once on
 * the host (heal_swin_amd/projection.py, the reference's project_s2_points_to_img :141-183) and stays resident.
 *   hs_pix2ang_nest     [host] healpy.pixelfunc.pix2ang(nside, ipix, nest=True) (:350) for ipix = first .. first+count-1:
 *                       theta, phi [host] f64[count].
 *   hs_sample_bilinear_u8   sample_bilinear(img, rx, ry).astype(np.uint8) (:38-73, :361): img [dev] u8[batch, channels,
 *                       height, width]; rx (along height), ry (along width) [dev] f64[n]; out [dev] u8[batch, channels, n].
 *                       float64, the reference's operation order, no FMA contraction: bit-exact.  Integer coordinates give 0
 *                       (both of the reference's weights vanish), neighbours outside the image contribute 0.
 *   hs_sample_mask_u8   sample_mask(mask, rx, ry, background) (:76-80, :362): nearest pixel by round-half-to-even,
 *                       outside the image -> background.  mask [dev] u8[batch, height, width]; out [dev] u8[batch, n].
 * ---------------------------------------------------------------------------------------------- */
int hs_pix2ang_nest(int nside, int64_t first, int64_t count, double* theta, double* phi);
int hs_sample_bilinear_u8(const void* img, int batch, int channels, int height, int width, const double* rx, const double* ry,
                          int64_t n, void* out, void* stream);
int hs_sample_mask_u8(const void* mask, int batch, int height, int width, const double* rx, const double* ry, int64_t n,
                      int background, void* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Decoder tail: LayerNorm(C) of FinalPatchExpand_X4 + the 1x1 class head in one pass, so that the normalised
 * [B, 4 N0, C] tensor is never written.  Replaces `self.norm(x)` (models_torch/swin_hp_transformer.py:448-452) followed by
 * `self.output(x)` (:756-761, :785-788) and their backward.  bf16 rows, C in {64, 96, ..., 256}, <= 16 classes.
 *   y [dev] bf16[rows, C] (the expanded rows); logits [dev] bf16[rows, 16] (columns >= n_classes are 0).
 *   forward : wfold [dev] bf16[32, C] = gamma * W[k, :] (rows >= n_classes zero), bvec [dev] f32[32] = sum_c beta_c W[k, c];
 *             mean, rstd [dev] f32[rows] are saved for the backward.
 *   backward: afold [dev] bf16[C, 16] = (gamma * W)^T (columns >= n_classes zero); dlogits [dev] bf16[rows, 16];
 *             dy [dev] bf16[rows, C] = LayerNorm input gradient; dprime [dev] bf16[rows, 16] = dlogits * rstd;
 *             partials [dev] f32[hs_ln_head_partials(rows), 32]: per-wave sums u[k] = sum dlogits (0..15) and
 *             t[k] = sum dprime * mean (16..31).  With X = hs_linear_wgrad(dprime, y) - t:  dW = gamma X + beta u,
 *             dgamma_c = sum_k W X, dbeta_c = sum_k W u   (heal_swin_amd/ops/tail.py:LnHeadFn).
 *   logits_dtype: HS_BF16 (32-byte rows) or HS_F32 (64-byte rows: the logits keep their fp32 accumulator value and xhat enters the
 *             head product as hi + lo; dlogits is then read as fp32 too) -- the decoder tail's roundings are not averaged by
 *             anything downstream and dominate the bf16 logit error, see csrc/ln_head.hip.
 * ---------------------------------------------------------------------------------------------- */
int hs_ln_head_supported(int width, int n_classes, int dtype);
int64_t hs_ln_head_partials(int64_t rows);
int hs_ln_head_fwd(const void* y, const void* wfold, const float* bvec, void* logits, float* mean, float* rstd, int64_t rows,
                   int width, int dtype, int logits_dtype, void* stream);
int hs_ln_head_bwd(const void* y, const float* mean, const float* rstd, const void* dlogits, const void* afold, void* dy,
                   void* dprime, float* partials, int64_t rows, int width, int dtype, int logits_dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * bf16 x 3 products for fp32 activations (the reference trains in fp32, training/train_config.py:95).  hs_split_bf16x3 writes, for
 * fp32 x [rows, k], the bf16 matrix [rows, 3 k] = [hi | hi | lo] (mode 0: activation side) or [hi | lo | hi] (mode 1: weight side),
 * hi = bf16(x), lo = bf16(x - hi): a bf16 GEMM of depth 3 k over a mode-0 and a mode-1 operand with fp32 accumulation equals
 * a_hi b_hi + a_hi b_lo + a_lo b_hi, i.e. the fp32 product to ~1e-5 relative, at 3/16 of the fp32-MFMA time (csrc/split3.hip).
 * hs_linear_wgrad_ld is hs_linear_wgrad (bf16) on the column blocks [ycol0, ycol0 + n_out) / [xcol0, xcol0 + k_in) of matrices with row
 * strides ldy / ldx (elements, multiples of 8; dy / x point to the matrices' first elements): the hi / lo blocks of two mode-0 matrices feed the three weight-gradient products (accumulate = 1 on the second and third).
 * ---------------------------------------------------------------------------------------------- */
int hs_split_bf16x3(const float* x, void* out, int64_t rows, int k, int mode, void* stream);
/* GELU fused with the split: out3 [rows, 3 k] bf16 = [hi | hi | lo] of dropout(gelu_erf(x)) (dy == NULL: the forward, x the
 * pre-activation) or of dy * dropout_mask * gelu_erf'(x) (the backward), x / dy [rows, k] fp32 -- for tensors that only
 * bf16 x 3 products read (the MLP hidden activation and its gradient): no fp32 copy is written.  Same dropout mask as
 * hs_gelu_fwd / hs_gelu_bwd for (seed, element index row * k + column). */
int hs_gelu_split3(const float* dy, const float* x, void* out3, int64_t rows, int k, float drop_p, uint64_t seed, void* stream);
int hs_linear_wgrad_ld(const void* dy, int64_t ldy, int64_t ycol0, const void* x, int64_t ldx, int64_t xcol0, float* dw, float* dbias,
                       float* workspace, int64_t rows, int n_out, int k_in, int accumulate, void* stream);

/* The whole decoder tail forward in one launch: FinalPatchExpand_X4's Linear(C -> 4 C) (models_torch/swin_hp_transformer.py:442-447),
 * the 'b n (p c) -> b (n p) c' view (:449), its LayerNorm(C) (:450-452) and the 1x1 class head (:785-788).  LayerNorm reads the fp32
 * accumulators of the expand product, xhat enters the head as hi + lo, the logits leave in fp32: of the tail's four bf16 roundings
 * only the input's remains (csrc/expand_ln_head.hip).  bf16, 4 children, C in {64, 96, 128} (the expand weight lives in LDS).
 *   xn [dev] bf16[tokens, C] (the norm_up output), xn_lo [dev] bf16[tokens, C] or NULL: its rounding remainder (hs_layernorm_fwd_ex lo_out) --
 *   with it the expand product takes its input as hi + lo and the tail has NO bf16 rounding between norm_up and the logits;
 *   wexp [dev] bf16[4 C, C] as nn.Linear stores it; wfold [dev] bf16[64, C]: rows 0..31 = gamma * W[k, :] rounded to bf16 (rows >=
 *   n_classes zero), rows 32..63 its rounding remainder; bvec as for hs_ln_head_fwd;
 *   logits [dev] f32[4 tokens, 16]; y [dev] bf16[4 tokens, C] + mean, rstd [dev] f32[4 tokens]: what the backward (hs_ln_head_bwd on y,
 *   then the Linear's gradients) needs -- all three NULL for a forward without gradient, in which case the expanded tensor never exists. */
int hs_expand_ln_head_supported(int width, int children, int n_classes, int dtype);
int hs_expand_ln_head_fwd(const void* xn, const void* xn_lo, const void* wexp, const void* wfold, const float* bvec, void* y, float* logits,
                          float* mean, float* rstd, int64_t tokens, int width, int children, int dtype, void* stream);

/* The tail WITH the segmentation caller's loss (nn.CrossEntropyLoss(weight=...)(logits, masks.long()),
 * models_lightning/segmentation/model_lightning_swin_hp.py:39-45, :104-111): the forward kernel above additionally takes the pixel
 * labels, forms the weighted cross-entropy of every row from the fp32 logits it holds in registers and writes per-wavefront partial
 * sums; with logits == NULL the [4 tokens, 16] fp32 logits tensor is never written (training: 403 MB at nside 256, batch 8, plus the
 * loss kernels' 3 passes over it).  The backward recomputes the row's logits from the saved expanded rows y, forms
 * dlogits = scale w[label] (softmax - onehot) in registers and continues as hs_ln_head_bwd (same outputs).
 *   labels [dev] u8[4 tokens] in pixel order (row = 4 token + child); ids >= n_classes carry weight 0;
 *   class_weights [dev] f32[n_classes] or NULL (all ones);
 *   loss_partials [dev] f32[4 * hs_expand_ln_head_blocks(tokens), 2]: sums of w (lse - logit_label) and of w per wavefront:
 *                 loss = sum(col 0) / sum(col 1)  (the reference's weighted mean);
 *   backward: scale [dev] f32[1] = dloss / sum(col 1); wfold [dev] bf16[64, C] and bvec [dev] f32[32] as for the forward but with row
 *             blocks 4..7 and 8..11 EXCHANGED (heal_swin_amd/ops/tail.py:_fold_head_ce; csrc/ln_head.hip says why); afold, dy, dprime,
 *             partials as for hs_ln_head_bwd.  bf16, C in {64, 96, 128}. */
int64_t hs_expand_ln_head_blocks(int64_t tokens);
int hs_expand_ln_head_ce_fwd(const void* xn, const void* xn_lo, const void* wexp, const void* wfold, const float* bvec, const uint8_t* labels,
                             const float* class_weights, int n_classes, void* y, float* logits, float* mean, float* rstd,
                             float* loss_partials, int64_t tokens, int width, int children, int dtype, void* stream);
int hs_ln_head_ce_bwd(const void* y, const float* mean, const float* rstd, const uint8_t* labels, const float* class_weights,
                      const float* scale, int n_classes, const void* wfold, const float* bvec, const void* afold, void* dy, void* dprime,
                      float* partials, int64_t rows, int width, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * PatchMerging / PatchExpand / FinalPatchExpand_X4 as one operator call per module and direction (SURVEY 8b's proposed
 * hs_patch_merge_* / hs_patch_expand_*).  In nested HEALPix order the reference's data movement is a free view -- the four
 * strided slices + cat of PatchMerging (models_torch/swin_hp_transformer.py:385-390) are [B, N, C] -> [B, N/4, 4C], PatchExpand's
 * 'b n (p c) -> b (n p) c' (:427, :449) is [B, N, p c] -> [B, N p, c] -- so each module is a row LayerNorm and a bias-free
 * Linear.  These entry points chain the library's own kernels (hs_layernorm_*, hs_gemm_nt, hs_linear_wgrad) on the caller's
 * stream; bf16 only (HS_ERR_UNSUPPORTED otherwise: fp32 runs compose hs_layernorm_* with a library GEMM).
 *
 * hs_patch_merge_fwd   replaces PatchMerging.forward (:378-395):  out = LN_{4 dim}(x) W^T
 *   x [dev] bf16[rows, 4 dim] = the stage output [B, N, dim] viewed as [B N/4, 4 dim]; gamma, beta [dev] f32[4 dim];
 *   w [dev] bf16[dim_out, 4 dim] (nn.Linear layout; the reference has dim_out = 2 dim); normed [dev] bf16[rows, 4 dim],
 *   mean, rstd [dev] f32[rows]: saved for the backward; out [dev] bf16[rows, dim_out].
 * hs_patch_merge_bwd   its autograd: dx [dev] bf16[rows, 4 dim]; dw f32[dim_out, 4 dim], dgamma, dbeta f32[4 dim] overwritten
 *   (accumulate == 0) or added to (the callers' .grad buffers); w_t [dev] bf16[4 dim, dim_out] = the transposed weight;
 *   dnormed [dev] bf16[rows, 4 dim] scratch; workspace [dev] f32[hs_patch_merge_bwd_workspace(...)].
 * hs_patch_expand_fwd  replaces PatchExpand.forward (:418-430; children = 4, dim_exp = 2 dim) and FinalPatchExpand_X4.forward
 *   (:441-452; children = patch_size, dim_exp = patch_size dim):  out = LN_{dim_exp / children}( view(x W^T) )
 *   x [dev] bf16[rows, dim]; w [dev] bf16[dim_exp, dim]; gamma, beta f32[dim_exp / children]; expanded [dev] bf16[rows, dim_exp],
 *   mean, rstd [dev] f32[rows children]: saved; out [dev] bf16[rows children, dim_exp / children].
 * hs_patch_expand_bwd  its autograd: w_t [dev] bf16[dim, dim_exp]; dexpanded [dev] bf16[rows, dim_exp] scratch; dx [dev]
 *   bf16[rows, dim]; dw f32[dim_exp, dim]; dgamma, dbeta f32[dim_exp / children]; workspace as sized by the _workspace call.
 * ---------------------------------------------------------------------------------------------- */
int hs_patch_merge_fwd(const void* x, const float* gamma, const float* beta, const void* w, void* normed, float* mean, float* rstd,
                       void* out, int64_t rows, int dim, int dim_out, int dtype, void* stream);
int64_t hs_patch_merge_bwd_workspace(int64_t rows, int dim, int dim_out);
int hs_patch_merge_bwd(const void* dout, const void* x, const void* normed, const float* gamma, const float* mean, const float* rstd,
                       const void* w_t, void* dnormed, void* dx, float* dw, float* dgamma, float* dbeta, float* workspace,
                       int accumulate, int64_t rows, int dim, int dim_out, int dtype, void* stream);
int hs_patch_expand_fwd(const void* x, const void* w, const float* gamma, const float* beta, void* expanded, float* mean, float* rstd,
                        void* out, int64_t rows, int dim, int dim_exp, int children, int dtype, void* stream);
int64_t hs_patch_expand_bwd_workspace(int64_t rows, int dim, int dim_exp, int children);
int hs_patch_expand_bwd(const void* dout, const void* x, const void* expanded, const float* gamma, const float* mean, const float* rstd,
                        const void* w_t, void* dexpanded, void* dx, float* dw, float* dgamma, float* dbeta, float* workspace,
                        int accumulate, int64_t rows, int dim, int dim_exp, int children, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HEALSWIN_H */
