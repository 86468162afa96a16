"""Row LayerNorm (+ fused residual add, dropout, DropPath scaling) and the elementwise GELU / residual-drop ops
(swin_hp_transformer.py:21-44, :315-338)."""

import torch

from .. import _lib
from .._lib import check, lib, ptr, stream_ptr
from .runtime import RT, _defer_flag, _defer_keep, _draw_seed, _extras, _f32, _require_gpu, _sink_buffer  # noqa: F401


def _norm_param_grads(weight, bias, width, device, want):
    """Buffers the LayerNorm backward writes dgamma / dbeta to: under a RT.grad_sink that knows both parameters their fp32
    gradient buffers (the kernel ADDS, autograd sees no gradient and launches no AccumulateGrad kernels), otherwise fresh
    tensors."""
    wbuf = _sink_buffer(weight) if want else None
    bbuf = _sink_buffer(bias) if wbuf is not None else None
    if wbuf is not None and bbuf is not None:
        return wbuf.view(-1), bbuf.view(-1), True
    return (torch.empty(width, dtype=torch.float32, device=device), torch.empty(width, dtype=torch.float32, device=device), False)


def _norm_param_result(weight, bias, dgamma, dbeta, direct):
    if not direct:
        return dgamma.to(weight.dtype), dbeta.to(bias.dtype)
    RT.grad_sink.deposited(weight)
    RT.grad_sink.deposited(bias)
    return None, None


class LayerNormFn(torch.autograd.Function):
    """y = [residual +] rs * LayerNorm(drop(x)) over the last dimension (eps 1e-5), statistics in fp32; rs / drop are the
    optional per-sample DropPath factor and dropout mask (train mode), absent in eval."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, extras, passthrough=False, res_lo=None, want_lo=False, pre=None):
        """pre = (y, mean, rstd): results the fused module kernel already wrote (window_attn_module_train); nothing is launched.
        res_lo / want_lo: compensated residual stream of the v2 placement (y = residual + LN(x) IS the stream): the stream
        operand is residual + res_lo, and with want_lo the call returns (y, y_lo) with y_lo the rounding remainder of y."""
        _require_gpu(x, weight, bias, residual)
        # an output nobody differentiates (the alias, or the non-differentiable y_lo) reaches backward as None instead of a
        # zero-filled activation-sized tensor -- which would also select the residual-gradient form of the kernel
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        width = x.shape[-1]
        rows = x.numel() // width
        dt = _lib.dtype_code(x.dtype)
        g, b = _f32(weight), _f32(bias)
        res = None if residual is None else residual.contiguous()
        if res is not None:
            assert res.shape == x.shape and res.dtype == x.dtype
        y_lo = None
        if pre is None:
            y = torch.empty_like(x)
            need_grad = any(ctx.needs_input_grad[:3])
            mean = torch.empty(rows, dtype=torch.float32, device=x.device) if need_grad else None
            rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if need_grad else None
        if pre is not None:
            assert res is None and extras is None and not want_lo and res_lo is None
            y, mean, rstd = pre
        elif want_lo or res_lo is not None:
            assert not passthrough and (res is not None or res_lo is None)
            y_lo = torch.empty_like(x) if want_lo else None
            rs, rps, p, seed = extras if extras is not None else (None, 1, 0.0, 0)
            check(lib.hs_layernorm_fwd_ex(ptr(x), ptr(res), None, ptr(None if res_lo is None else res_lo.contiguous()), ptr(g), ptr(b),
                                          ptr(y), None, ptr(y_lo), ptr(mean), ptr(rstd), ptr(rs), rps, p, seed, rows, width, dt,
                                          stream_ptr(x.device)), "hs_layernorm_fwd_ex")
        elif extras is None:
            check(lib.hs_layernorm_fwd(ptr(x), ptr(res), ptr(g), ptr(b), ptr(y), ptr(mean), ptr(rstd), rows, width, dt,
                                       stream_ptr(x.device)), "hs_layernorm_fwd")
        else:
            rs, rps, p, seed = extras
            check(lib.hs_layernorm_drop_fwd(ptr(x), ptr(res), ptr(g), ptr(b), ptr(y), ptr(mean), ptr(rstd), ptr(rs), rps, p, seed,
                                            rows, width, dt, stream_ptr(x.device)), "hs_layernorm_drop_fwd")
        ctx.save_for_backward(x, g, mean, rstd, None if extras is None else extras[0])
        ctx.meta = (rows, width, dt, residual is not None, extras)
        ctx.params = (weight, bias)
        ctx.second_is_alias = bool(passthrough)  # (with want_lo the second output is the non-differentiable remainder)
        # passthrough (plain norm only): also hand x back as an alias for a second use (the block's residual connection); its
        # gradient then arrives here with dy and is added inside the backward kernel instead of by a separate elementwise add
        assert not passthrough or (extras is None and residual is None)
        if want_lo:
            ctx.mark_non_differentiable(y_lo)
            return y, y_lo
        return (y, x.view_as(x)) if passthrough else y

    @staticmethod
    def backward(ctx, dy, dx_alias=None):
        x, g, mean, rstd, rs = ctx.saved_tensors
        rows, width, dt, has_res, extras = ctx.meta
        weight, bias = ctx.params
        if not ctx.second_is_alias:
            dx_alias = None
        if dy is None:  # only the alias was used downstream
            return dx_alias, None, None, None, None, None, None, None, None
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dgamma, dbeta, direct = _norm_param_grads(weight, bias, width, x.device, ctx.needs_input_grad[1] and ctx.needs_input_grad[2])
        ws = torch.empty(int(lib.hs_layernorm_bwd_workspace(rows, width)), dtype=torch.float32, device=x.device)
        acc = (1 | _defer_flag(x.device)) if direct else 0
        if dx_alias is not None:
            check(lib.hs_add_layernorm_bwd(ptr(dy), ptr(dx_alias.contiguous()), ptr(x), ptr(g), ptr(mean), ptr(rstd), ptr(dx),
                                           ptr(dgamma), ptr(dbeta), ptr(ws), acc, rows, width, dt, stream_ptr(x.device)),
                  "hs_add_layernorm_bwd")
        elif extras is None:
            check(lib.hs_layernorm_bwd(ptr(dy), ptr(x), ptr(g), ptr(mean), ptr(rstd), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(ws),
                                       acc, rows, width, dt, stream_ptr(x.device)), "hs_layernorm_bwd")
        else:
            _, rps, p, seed = extras
            check(lib.hs_layernorm_drop_bwd(ptr(dy), ptr(x), ptr(g), ptr(mean), ptr(rstd), ptr(dx), ptr(dgamma), ptr(dbeta),
                                            ptr(ws), acc, ptr(rs), rps, p, seed, rows, width, dt, stream_ptr(x.device)),
                  "hs_layernorm_drop_bwd")
        if acc & _lib.HS_ACC_DEFER:
            _defer_keep(x.device, ws)
        dw, db = _norm_param_result(weight, bias, dgamma, dbeta, direct)
        return dx, dw, db, (dy if has_res else None), None, None, None, None, None


def layer_norm(x, weight, bias, residual=None, row_scale=None, drop_p=0.0, seed=None):
    return LayerNormFn.apply(x, weight, bias, residual, _extras(x, row_scale, drop_p, seed))


def layer_norm_hilo(x, weight, bias):
    """(y, y_lo): LayerNorm(x) as the plain activation tensor y plus its rounding remainder y_lo (not differentiable), for a
    consumer that takes its operand as hi + lo (the fused decoder tail, `expand_ln_head`)."""
    return LayerNormFn.apply(x, weight, bias, None, None, False, None, True)


def layer_norm_stream(x, weight, bias, residual, res_lo=None, row_scale=None, drop_p=0.0, seed=None):
    """(y, y_lo) with y + y_lo = residual + res_lo + rs * LN(drop(x)) to 16 mantissa bits: the v2-placement residual stream kept
    compensated (see csrc/layernorm.hip).  y is the plain activation tensor; y_lo is not differentiable."""
    return LayerNormFn.apply(x, weight, bias, residual, _extras(x, row_scale, drop_p, seed), False, res_lo, True)


def layer_norm_passthrough(x, weight, bias):
    """(LayerNorm(x), alias of x): use the alias for the second consumer of x (see LayerNormFn)."""
    return LayerNormFn.apply(x, weight, bias, None, None, True)


class AddLayerNormFn(torch.autograd.Function):
    """(s, y) = (a + rs * drop(b), LayerNorm(s)) in one pass; backward folds the residual-path gradient into the LN backward
    and routes the gradient of b through the same DropPath factor / dropout mask."""

    @staticmethod
    def forward(ctx, a, b, weight, bias, extras, a_lo=None, want_lo=False):
        """a_lo / want_lo: compensated residual stream (csrc/layernorm.hip): the stream operand is a + a_lo, and with want_lo the
        call returns (s, y, s_lo) with s_lo the rounding remainder of the new stream s (not differentiable)."""
        _require_gpu(a, b, weight, bias)
        a, b = a.contiguous(), b.contiguous()
        assert a.shape == b.shape and a.dtype == b.dtype
        width = a.shape[-1]
        rows = a.numel() // width
        dt = _lib.dtype_code(a.dtype)
        g, be = _f32(weight), _f32(bias)
        s = torch.empty_like(a)
        y = torch.empty_like(a)
        mean = torch.empty(rows, dtype=torch.float32, device=a.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=a.device)
        s_lo = None
        if want_lo or a_lo is not None:
            s_lo = torch.empty_like(a) if want_lo else None
            rs, rps, p, seed = extras if extras is not None else (None, 1, 0.0, 0)
            check(lib.hs_layernorm_fwd_ex(ptr(a), None, ptr(b), ptr(None if a_lo is None else a_lo.contiguous()), ptr(g), ptr(be), ptr(y),
                                          ptr(s), ptr(s_lo), ptr(mean), ptr(rstd), ptr(rs), rps, p, seed, rows, width, dt,
                                          stream_ptr(a.device)), "hs_layernorm_fwd_ex")
        elif extras is None:
            check(lib.hs_add_layernorm_fwd(ptr(a), ptr(b), ptr(g), ptr(be), ptr(s), ptr(y), ptr(mean), ptr(rstd), rows, width,
                                           dt, stream_ptr(a.device)), "hs_add_layernorm_fwd")
        else:
            rs, rps, p, seed = extras
            check(lib.hs_add_layernorm_drop_fwd(ptr(a), ptr(b), ptr(g), ptr(be), ptr(s), ptr(y), ptr(mean), ptr(rstd), ptr(rs),
                                                rps, p, seed, rows, width, dt, stream_ptr(a.device)), "hs_add_layernorm_drop_fwd")
        ctx.save_for_backward(s, g, mean, rstd, None if extras is None else extras[0])
        ctx.meta = (rows, width, dt, extras)
        ctx.params = (weight, bias)
        if want_lo:
            ctx.mark_non_differentiable(s_lo)
            return s, y, s_lo
        return s, y

    @staticmethod
    def backward(ctx, ds, dy, ds_lo=None):
        s, g, mean, rstd, rs = ctx.saved_tensors
        rows, width, dt, extras = ctx.meta
        weight, bias = ctx.params
        if dy is None:  # only the sum was used downstream
            if extras is None:
                return ds, ds, None, None, None, None, None
            dy = torch.zeros_like(s)
        dy = dy.contiguous()
        ds_c = None if ds is None else ds.contiguous()
        da = torch.empty_like(s)
        dgamma, dbeta, direct = _norm_param_grads(weight, bias, width, s.device, ctx.needs_input_grad[2] and ctx.needs_input_grad[3])
        ws = torch.empty(int(lib.hs_layernorm_bwd_workspace(rows, width)), dtype=torch.float32, device=s.device)
        acc = (1 | _defer_flag(s.device)) if direct else 0
        if extras is None:
            check(lib.hs_add_layernorm_bwd(ptr(dy), ptr(ds_c), ptr(s), ptr(g), ptr(mean), ptr(rstd), ptr(da), ptr(dgamma),
                                           ptr(dbeta), ptr(ws), acc, rows, width, dt, stream_ptr(s.device)),
                  "hs_add_layernorm_bwd")
            db = da
        else:
            _, rps, p, seed = extras
            db = torch.empty_like(s)
            check(lib.hs_add_layernorm_drop_bwd(ptr(dy), ptr(ds_c), ptr(s), ptr(g), ptr(mean), ptr(rstd), ptr(da), ptr(db),
                                                ptr(dgamma), ptr(dbeta), ptr(ws), acc, ptr(rs), rps, p, seed, rows, width,
                                                dt, stream_ptr(s.device)), "hs_add_layernorm_drop_bwd")
        if acc & _lib.HS_ACC_DEFER:
            _defer_keep(s.device, ws)
        dw, dbias = _norm_param_result(weight, bias, dgamma, dbeta, direct)
        return da, db, dw, dbias, None, None, None


def add_layer_norm(a, b, weight, bias, row_scale=None, drop_p=0.0, seed=None):
    """returns (a + rs*drop(b), LayerNorm(a + rs*drop(b)))"""
    return AddLayerNormFn.apply(a, b, weight, bias, _extras(a, row_scale, drop_p, seed))


def add_layer_norm_stream(a, a_lo, b, weight, bias, row_scale=None, drop_p=0.0, seed=None):
    """(s, y, s_lo): s + s_lo = a + a_lo + rs*drop(b) to 16 mantissa bits (a_lo may be None), y = LayerNorm of that sum: the
    v1-placement residual stream kept compensated.  s is the plain activation tensor; s_lo is not differentiable."""
    return AddLayerNormFn.apply(a, b, weight, bias, _extras(a, row_scale, drop_p, seed), a_lo, True)


# ----------------------------------------------------------------------------- GELU (+ dropout)
class GeluDropoutFn(torch.autograd.Function):
    """y = dropout(gelu(x), p) in one pass; the backward regenerates the mask from the seed (reference Mlp :39-41)."""

    @staticmethod
    def forward(ctx, x, p, seed):
        _require_gpu(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        dt = _lib.dtype_code(x.dtype)
        check(lib.hs_gelu_fwd(ptr(x), ptr(y), x.numel(), float(p), int(seed), dt, stream_ptr(x.device)), "hs_gelu_fwd")
        ctx.save_for_backward(x)
        ctx.meta = (float(p), int(seed), dt)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        p, seed, dt = ctx.meta
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        check(lib.hs_gelu_bwd(ptr(dy), ptr(x), ptr(dx), x.numel(), p, seed, dt, stream_ptr(x.device)), "hs_gelu_bwd")
        return dx, None, None


class ResidualDropFn(torch.autograd.Function):
    """x + rs * drop(t) in one pass (`hs_residual_drop`); the mask is regenerated in the backward from the seed."""

    @staticmethod
    def forward(ctx, x, t, row_scale, p, seed):
        _require_gpu(x, t, row_scale)
        x, t = x.contiguous(), t.contiguous()
        assert x.shape == t.shape and x.dtype == t.dtype
        rs = None if row_scale is None else row_scale.detach().to(torch.float32).contiguous()
        eps = t.numel() // t.shape[0]
        out = torch.empty_like(t)
        dt = _lib.dtype_code(t.dtype)
        check(lib.hs_residual_drop(ptr(x), ptr(t), ptr(out), ptr(rs), eps, t.numel(), float(p), int(seed), dt, stream_ptr(t.device)),
              "hs_residual_drop")
        ctx.save_for_backward(rs)
        ctx.meta = (eps, float(p), int(seed), dt)
        return out

    @staticmethod
    def backward(ctx, dy):
        (rs,) = ctx.saved_tensors
        eps, p, seed, dt = ctx.meta
        dy = dy.contiguous()
        dtv = torch.empty_like(dy)
        check(lib.hs_residual_drop(None, ptr(dy), ptr(dtv), ptr(rs), eps, dy.numel(), p, seed, dt, stream_ptr(dy.device)),
              "hs_residual_drop (backward)")
        return dy, dtv, None, None, None


def residual_drop(x, t, row_scale=None, p=0.0, seed=None):
    """x + rs * dropout(t): DropPath factor per sample (row_scale [B] or None) and dropout with probability p."""
    if p > 0.0 and seed is None:
        seed = _draw_seed()
    return ResidualDropFn.apply(x, t, row_scale, float(p), int(seed or 0))


def gelu_dropout(x, p=0.0, seed=None):
    if p > 0.0 and seed is None:
        seed = _draw_seed()
    return GeluDropoutFn.apply(x, float(p), int(seed or 0))


# ----------------------------------------------------------------------------- Linear with HIP weight gradient
