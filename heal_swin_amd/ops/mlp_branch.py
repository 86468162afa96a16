"""The block's Mlp branch: fc1 -> GELU -> dropout -> fc2 as GEMM epilogues (`MlpFn`), and as ONE kernel per direction with its
LayerNorm and residual for C = 96 / 128 (`FusedMlpBlockFn`, csrc/mlp_fused.hip).  Reference: swin_hp_transformer.py:21-44, :334-338."""
import os

import torch

from .. import _lib
from .._lib import check, lib, ptr, stream_ptr
from .runtime import RT, _cast_param, _defer_flag, _defer_keep, _draw_seed, _extras, _f32, _require_gpu, _timed  # noqa: F401
from .norm import _norm_param_grads, _norm_param_result  # noqa: F401
from .gemm import (_Split, _bf16x3_ok, _cast_param_t, _input_grad, _lib_linear, _lib_matmul, _param_grads, _split_of, gemm_nt, own_gemm_legal, own_gemm_ok)  # noqa: F401


class MlpFn(torch.autograd.Function):
    """fc1 -> GELU(erf) -> dropout -> fc2 (reference Mlp.forward, swin_hp_transformer.py:38-44, without the output dropout,
    which the caller fuses into the next norm kernel) as ONE autograd node, so that the elementwise steps ride on the GEMMs
    around them: forward `hs_gemm_nt(EPI_GELU)` writes the pre-activation h and dropout(gelu(h)) from one accumulator pass;
    backward `hs_gemm_nt(EPI_DGELU)` turns dy W2 into dh = dy W2 * mask * gelu'(h) in its epilogue.  Where the library GEMM is
    faster (own_gemm_ok) the standalone `hs_gelu_*` kernels are used instead; both forms draw the same dropout mask."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, drop_p, seed, passthrough, residual=None):
        _require_gpu(x, w1, b1, w2, b2, residual)
        c_in, hid = w1.shape[1], w1.shape[0]
        x2 = x.reshape(-1, c_in)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        dt = x.dtype
        x3 = a3 = None  # (fp32 runs) the bf16x3 splits the two forward products made, kept for the weight gradients
        w1c, w2c = _cast_param(w1, dt), _cast_param(w2, dt)
        need_grad = any(ctx.needs_input_grad[:5])
        if own_gemm_ok(_lib.HS_EPI_GELU, hid, c_in, dt):
            h, a = gemm_nt(x2, w1c, b1, _lib.HS_EPI_GELU, want_c=need_grad, drop_p=drop_p, seed=seed)
        else:
            h = _lib_linear(x2, w1c, None if b1 is None else _cast_param(b1, dt))
            x3 = _split_of(x2)
            if _bf16x3_ok(h, w2.shape[0], hid) and w2c.dtype == torch.float32 and w2.shape[0] % 8 == 0 and hid % 8 == 0:
                # fp32 run, fc2 a bf16x3 product: gelu(h) is written as that product's [hi | hi | lo] operand and never as fp32
                a3 = torch.empty((h.shape[0], 3 * hid), dtype=torch.bfloat16, device=h.device)
                check(lib.hs_gelu_split3(None, ptr(h), ptr(a3), h.shape[0], hid, float(drop_p), int(seed), stream_ptr(h.device)),
                      "hs_gelu_split3")
                a = _Split(a3, hid)
            else:
                a = torch.empty_like(h)
                check(lib.hs_gelu_fwd(ptr(h), ptr(a), h.numel(), float(drop_p), int(seed), _lib.dtype_code(dt), stream_ptr(h.device)),
                      "hs_gelu_fwd")
        ctx.has_residual = residual is not None
        if residual is not None and own_gemm_legal(w2.shape[0], hid, dt):
            res2 = residual.reshape(-1, w2.shape[0])
            y = gemm_nt(a, w2c, b2, _lib.HS_EPI_RESID, aux=res2 if res2.is_contiguous() else res2.contiguous())[0]
        elif own_gemm_ok(_lib.HS_EPI_BIAS, w2.shape[0], hid, dt):
            y = gemm_nt(a, w2c, b2)[0]
        else:
            y = _lib_linear(a, w2c, None if b2 is None else _cast_param(b2, dt))
            a3 = _split_of(a)
            if isinstance(a, _Split):
                a = None
        if residual is not None and not own_gemm_legal(w2.shape[0], hid, dt):
            y = y + residual.reshape(-1, w2.shape[0])
        assert not isinstance(a, _Split)
        # (fp32 runs: the weight gradients read the bf16x3 splits the forward products made, not x2 / a themselves)
        ctx.save_for_backward(None if x3 is not None else x2, h, None if a3 is not None else a, w1, w2)
        ctx.biases = (b1, b2)
        ctx.casts = (w1c if w1c.dtype != w1.dtype else None, w2c if w2c.dtype != w2.dtype else None)
        ctx.cast_cache = RT.cast_cache
        ctx.meta = (float(drop_p), int(seed), x.shape)
        ctx.splits = (x3, a3)
        y = y.view(x.shape[:-1] + (w2.shape[0],))
        return (y, x.view_as(x)) if passthrough else y

    @staticmethod
    def backward(ctx, dy, dx_res=None):
        x2, h, a, w1, w2 = ctx.saved_tensors
        b1, b2 = ctx.biases
        w1c, w2c = ctx.casts
        p, seed, xshape = ctx.meta
        if dy is None:  # only the passthrough alias was used downstream
            return dx_res, None, None, None, None, None, None, None, None
        c_out, hid, c_in = w2.shape[0], w1.shape[0], w1.shape[1]
        dy2 = dy.reshape(-1, c_out)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dt = dy2.dtype
        # dh = (dy W2) * mask * gelu'(h)
        if own_gemm_ok(_lib.HS_EPI_DGELU, hid, c_out, dt):
            dh = gemm_nt(dy2, _cast_param_t(w2, dt, ctx.cast_cache), None, _lib.HS_EPI_DGELU, aux=h, drop_p=p, seed=seed)[0]
        else:
            da = _lib_matmul(dy2, w2c if (w2c is not None and w2c.dtype == dt) else w2.to(dt))
            if _bf16x3_ok(da, c_in, hid) and w1.dtype == torch.float32 and c_in % 8 == 0 and hid % 8 == 0:
                # fp32 run: dh is read by fc1's input- and weight-gradient products only, both bf16x3 -- written as their operand
                dh3 = torch.empty((h.shape[0], 3 * hid), dtype=torch.bfloat16, device=h.device)
                check(lib.hs_gelu_split3(ptr(da), ptr(h), ptr(dh3), h.shape[0], hid, p, seed, stream_ptr(h.device)), "hs_gelu_split3")
                dh = _Split(dh3, hid)
            else:
                dh = torch.empty_like(h)
                check(lib.hs_gelu_bwd(ptr(da), ptr(h), ptr(dh), h.numel(), p, seed, _lib.dtype_code(dt), stream_ptr(h.device)), "hs_gelu_bwd")
            del da
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _input_grad(dh, w1, w1c, None if dx_res is None else dx_res.reshape(-1, c_in), ctx.cast_cache).reshape(xshape)
        ctx.casts = ctx.cast_cache = None
        (x3, a3), ctx.splits = ctx.splits, (None, None)
        dw2, db2 = _param_grads(dy2, a, w2, b2, ctx.needs_input_grad[3], b2 is not None and ctx.needs_input_grad[4], a3)
        dw1, db1 = _param_grads(dh, x2, w1, b1, ctx.needs_input_grad[1], b1 is not None and ctx.needs_input_grad[2], x3)
        return dx, dw1, db1, dw2, db2, None, None, None, (dy if ctx.has_residual else None)


def mlp(x, w1, b1, w2, b2, drop_p=0.0, seed=None, passthrough=False, residual=None):
    """fc2(dropout(gelu(fc1(x)))) (+ an alias of x when passthrough, see LinearFn; + residual in fc2's epilogue)."""
    if drop_p > 0.0 and seed is None:
        seed = _draw_seed()
    return MlpFn.apply(x, w1, b1, w2, b2, float(drop_p), int(seed or 0), bool(passthrough), residual)


# ----------------------------------------------------------------------------- fused Mlp block (HBM-bound stages)
FUSED_MLP = os.environ.get("HS_FUSED_MLP", "1") != "0"  # A/B switch: off = LayerNorm -> hs_gemm_nt(GELU) -> hs_gemm_nt(residual)
# gelu(h) kept for fc2's weight gradient (True), or re-applied to the saved h inside that weight-gradient kernel (False:
# hs_linear_wgrad_gelu -- 4 of the forward's 11 row-units and 1.6 GB per stage-0 block of HEAL-SWIN-B less).  Measured on MI355X
# (profiles/r05_mlp_fused_keep_act_ab.txt): the forward kernel gains 864 -> 775 us (it is then bound by its own issue rate, not by
# HBM), the weight gradient loses ~300 us (two waves evaluate every fragment's GELU), the step is unchanged (142.3 vs 142.6 ms) at
# 103.5 instead of 109.9 GB peak: the memory-saving form is an option, the default keeps the activation.
MLP_KEEP_ACT = True


def fused_mlp_ok(x, hidden):
    """Whether `fused_mlp_block` (hs_mlp_fused_fwd / _bwd, csrc/mlp_fused.hip) covers this block: bf16 rows on the GPU, C = 96 / 128,
    hidden = 4 C, a row count that is a multiple of 32."""
    c = x.shape[-1]
    return bool(FUSED_MLP and x.is_cuda and x.dtype == torch.bfloat16 and (x.numel() // c) % 32 == 0 and
                lib.hs_mlp_fused_supported(int(c), int(hidden), _lib.HS_BF16))


class FusedMlpBlockFn(torch.autograd.Function):
    """The block's second residual branch as ONE forward kernel that also writes what the backward reads, and ONE backward kernel for
    the two input-gradient products around gelu' (csrc/mlp_fused.hip):
        v1 placement (reference swin_hp_transformer.py:337-338):  x + fc2(gelu(fc1(LayerNorm(x))))
        v2 placement (`post_norm`, :334-335):                     x + LayerNorm(fc2(gelu(fc1(x))))
    with Mlp.forward :38-44.  Weight / bias gradients come from `hs_linear_wgrad`, the LayerNorm backward from the LayerNorm
    kernels (v1: with the residual gradient folded in; v2: in front of the Mlp backward, whose epilogue adds the residual path)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w1, b1, w2, b2, post_norm=False, stoch=None):
        """stoch (train mode, post_norm only): (row_scale fp32 [B] or None, rows_per_sample, drop_p, seed_hidden, seed_out) -- Mlp.drop
        behind the activation and behind fc2, DropPath as a per-sample factor, all inside the launch (hs_mlp_fused_drop_fwd / _bwd)."""
        _require_gpu(x, ln_w, ln_b, w1, b1, w2, b2)
        assert stoch is None or post_norm
        C, hid = x.shape[-1], w1.shape[0]
        x2 = x.reshape(-1, C)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        rows = x2.shape[0]
        dev = x.device
        need = any(ctx.needs_input_grad)
        w1c, w2c = _cast_param(w1, torch.bfloat16).contiguous(), _cast_param(w2, torch.bfloat16).contiguous()
        g, b = _f32(ln_w), _f32(ln_b)
        out = torch.empty_like(x2)
        n = torch.empty_like(x2) if need else None  # v1: LayerNorm(x) (fc1's input); v2: mlp(x) (the LayerNorm's input)
        mean = torch.empty(rows, dtype=torch.float32, device=dev) if need else None
        rstd = torch.empty(rows, dtype=torch.float32, device=dev) if need else None
        h = torch.empty((rows, hid), dtype=x.dtype, device=dev) if need else None
        # (ops.MLP_KEEP_ACT = False: kept only where fc2's weight gradient cannot take it from h, hs_linear_wgrad_gelu)
        # (with hidden dropout the kept activation is the DROPPED one: fc2's weight gradient cannot take it from h)
        keep_act = need and (MLP_KEEP_ACT or (stoch is not None and stoch[2] > 0) or
                             not lib.hs_linear_wgrad_gelu_supported(rows, C, hid, _lib.HS_BF16))
        act = torch.empty((rows, hid), dtype=x.dtype, device=dev) if keep_act else None
        flags = _lib.HS_ATTN_RESIDUAL | (_lib.HS_MLP_NORM_AFTER if post_norm else 0)
        # algorithmic traffic: x in, out (+ n, h, gelu(h) kept for the backward); flops: the two products
        with _timed("mlp_fused_fwd", dev, 2 * rows * ((3 if need else 2) * C + ((2 if keep_act else 1) * hid if need else 0)), 4 * rows * C * hid):
            if stoch is None:
                check(lib.hs_mlp_fused_fwd(ptr(x2), ptr(g), ptr(b), ptr(w1c), ptr(_f32(b1)), ptr(w2c), ptr(_f32(b2)), ptr(n), ptr(mean), ptr(rstd),
                                           ptr(h), ptr(act), ptr(out), rows, C, hid, flags, _lib.HS_BF16, stream_ptr(dev)),
                      "hs_mlp_fused_fwd")
            else:
                rs, rps, dp, seed_h, seed_o = stoch
                check(lib.hs_mlp_fused_drop_fwd(ptr(x2), ptr(g), ptr(b), ptr(w1c), ptr(_f32(b1)), ptr(w2c), ptr(_f32(b2)), ptr(n), ptr(mean),
                                                ptr(rstd), ptr(h), ptr(act), ptr(out), ptr(rs), rps, dp, seed_h, seed_o, rows, C, hid, flags,
                                                _lib.HS_BF16, stream_ptr(dev)), "hs_mlp_fused_drop_fwd")
        ctx.stoch = stoch
        ctx.save_for_backward(x2, n, mean, rstd, h, act, g, w1, w2)
        ctx.params = (ln_w, ln_b, b1, b2)
        ctx.cast_cache = RT.cast_cache
        ctx.x_shape = x.shape
        ctx.post_norm = bool(post_norm)
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, dout):
        x2, n, mean, rstd, h, act, g, w1, w2 = ctx.saved_tensors
        ln_w, ln_b, b1, b2 = ctx.params
        rows, C = x2.shape
        hid = w1.shape[0]
        dev = x2.device
        dy2 = dout.reshape(rows, C)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        w2t = _cast_param_t(w2, torch.bfloat16, ctx.cast_cache)  # [4C, C]
        w1t = _cast_param_t(w1, torch.bfloat16, ctx.cast_cache)  # [C, 4C]
        ctx.cast_cache = None
        dgamma, dbeta, direct = _norm_param_grads(ln_w, ln_b, C, dev, ctx.needs_input_grad[1] and ctx.needs_input_grad[2])
        ws = torch.empty(int(lib.hs_layernorm_bwd_workspace(rows, C)), dtype=torch.float32, device=dev)
        acc = (1 | _defer_flag(dev)) if direct else 0
        dh = torch.empty_like(h)
        if ctx.post_norm:
            # out = x + LN(m): LayerNorm backward first (dm from dout and the saved m), then the Mlp backward on dm with the
            # residual path's gradient (dout itself) added in its epilogue: dx = dout + dh W1
            dm = torch.empty_like(x2)
            stoch = ctx.stoch
            if stoch is None:
                check(lib.hs_layernorm_bwd(ptr(dy2), ptr(n), ptr(g), ptr(mean), ptr(rstd), ptr(dm), ptr(dgamma), ptr(dbeta), ptr(ws), acc, rows, C,
                                           _lib.HS_BF16, stream_ptr(dev)), "hs_layernorm_bwd")
            else:  # out = x + rs * LN(drop_o(m)): dm = mask_o * LN_bwd(rs * dout)
                rs, rps, dp, seed_h, seed_o = stoch
                check(lib.hs_layernorm_drop_bwd(ptr(dy2), ptr(n), ptr(g), ptr(mean), ptr(rstd), ptr(dm), ptr(dgamma), ptr(dbeta), ptr(ws), acc,
                                                ptr(rs), rps, dp, seed_o, rows, C, _lib.HS_BF16, stream_ptr(dev)), "hs_layernorm_drop_bwd")
            dx = torch.empty_like(x2)
            with _timed("mlp_fused_bwd", dev, 2 * rows * (3 * C + 2 * hid), 4 * rows * C * hid):
                if stoch is None:
                    check(lib.hs_mlp_fused_bwd(ptr(dm), ptr(h), ptr(w2t), ptr(w1t), ptr(dy2), ptr(dh), ptr(dx), rows, C, hid, _lib.HS_BF16,
                                               stream_ptr(dev)), "hs_mlp_fused_bwd")
                else:
                    check(lib.hs_mlp_fused_drop_bwd(ptr(dm), ptr(h), ptr(w2t), ptr(w1t), ptr(dy2), ptr(dh), ptr(dx), dp, seed_h, rows, C, hid,
                                                    _lib.HS_BF16, stream_ptr(dev)), "hs_mlp_fused_drop_bwd")
            dw2, db2 = _param_grads(dm, h if act is None else act, w2, b2, ctx.needs_input_grad[5], b2 is not None and ctx.needs_input_grad[6],
                                    gelu_x=act is None)
            dw1, db1 = _param_grads(dh, x2, w1, b1, ctx.needs_input_grad[3], b1 is not None and ctx.needs_input_grad[4])
        else:
            dn = torch.empty_like(x2)
            with _timed("mlp_fused_bwd", dev, 2 * rows * (2 * C + 2 * hid), 4 * rows * C * hid):
                check(lib.hs_mlp_fused_bwd(ptr(dy2), ptr(h), ptr(w2t), ptr(w1t), None, ptr(dh), ptr(dn), rows, C, hid, _lib.HS_BF16,
                                           stream_ptr(dev)), "hs_mlp_fused_bwd")
            dw2, db2 = _param_grads(dy2, h if act is None else act, w2, b2, ctx.needs_input_grad[5], b2 is not None and ctx.needs_input_grad[6],
                                    gelu_x=act is None)
            dw1, db1 = _param_grads(dh, n, w1, b1, ctx.needs_input_grad[3], b1 is not None and ctx.needs_input_grad[4])
            # norm2 backward with the residual gradient (dy itself) added inside the kernel
            dx = torch.empty_like(x2)
            check(lib.hs_add_layernorm_bwd(ptr(dn), ptr(dy2), ptr(x2), ptr(g), ptr(mean), ptr(rstd), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(ws),
                                           acc, rows, C, _lib.HS_BF16, stream_ptr(dev)), "hs_add_layernorm_bwd")
        if acc & _lib.HS_ACC_DEFER:
            _defer_keep(dev, ws)
        dlw, dlb = _norm_param_result(ln_w, ln_b, dgamma, dbeta, direct)
        return dx.view(ctx.x_shape), dlw, dlb, dw1, db1, dw2, db2, None, None


def fused_mlp_block(x, ln_w, ln_b, w1, b1, w2, b2, post_norm=False, row_scale=None, drop_p=0.0, seeds=None):
    """x + fc2(gelu(fc1(LayerNorm(x)))) -- or, post_norm, x + LayerNorm(fc2(gelu(fc1(x)))) -- in one launch (FusedMlpBlockFn; use
    fused_mlp_ok first).  Train mode, post_norm only: Mlp.drop (drop_p) behind the activation and behind fc2 and the per-sample DropPath
    factor `row_scale` ([B] or None) ride in the same launch -- x + rs * LayerNorm(drop(fc2(drop(gelu(fc1(x))))))."""
    stoch = None
    if row_scale is not None or drop_p:
        ex = _extras(x, row_scale, drop_p, 0)  # (0: no seed drawn here -- the two below are this block's whole share of the host seed stream)
        seed_h, seed_o = seeds if seeds is not None else ((_draw_seed(), _draw_seed()) if drop_p else (0, 0))
        stoch = (ex[0], ex[1], ex[2], int(seed_h), int(seed_o))
    return FusedMlpBlockFn.apply(x, ln_w, ln_b, w1, b1, w2, b2, bool(post_norm), stoch)


def fused_mlp_stochastic_ok(x, post_norm):
    """Whether the stochastic form of the fused Mlp block applies: v2 placement, whole 32-row tiles per sample."""
    return bool(post_norm) and (x.numel() // x.shape[-1] // x.shape[0]) % 32 == 0
