"""Skip-connection Linear without the concatenation, PatchMerging / PatchExpand as operators, standalone row gather (shift)
(swin_hp_transformer.py:364-452, :772-775; hp_shifting.py)."""

import torch

from .. import _lib
from .._lib import check, lib, ptr, stream_ptr
from .runtime import RT, _cast_param, _f32, _require_gpu, _sink_buffer  # noqa: F401
from .gemm import LinearFn, _cast_param_t, gemm_nt, own_gemm_ok  # noqa: F401


class ConcatLinearFn(torch.autograd.Function):
    """y = cat([x, skip], -1) W^T + b without materialising the concatenation (the decoder's skip connection,
    swin_hp_transformer.py:772-775): W = [Wa | Wb] by columns, y = x Wa^T + skip Wb^T + b.  Saves the concat copy in the
    forward and the strided slices of the concatenated gradient (re-packed by their consumers) in the backward."""

    @staticmethod
    def forward(ctx, x, skip, weight, bias):
        _require_gpu(x, skip, weight, bias)
        c = x.shape[-1]
        w = _cast_param(weight, x.dtype)
        b = None if bias is None else _cast_param(bias, x.dtype)
        x2, s2 = x.reshape(-1, c), skip.reshape(-1, skip.shape[-1])
        if own_gemm_ok(_lib.HS_EPI_BIAS, weight.shape[0], c, x.dtype, k2=s2.shape[1]) and x2.is_contiguous() and s2.is_contiguous():
            y = gemm_nt(x2, w[:, :c], bias, a2=s2, w2=w[:, c:])[0]  # both K segments into one accumulator
        else:
            y = torch.addmm(b, x2, w[:, :c].t()) if b is not None else x2 @ w[:, :c].t()
            y.addmm_(s2, w[:, c:].t())
        ctx.save_for_backward(x, skip, weight)
        ctx.bias_param = bias
        ctx.w_cast = w if w is not weight else None
        ctx.cast_cache = RT.cast_cache
        return y.reshape(x.shape[:-1] + (weight.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        x, skip, weight = ctx.saved_tensors
        bias = ctx.bias_param
        n_out, c = weight.shape[0], x.shape[-1]
        cs = weight.shape[1] - c
        dy2 = dy.reshape(-1, n_out)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        x2, s2 = x.reshape(-1, c), skip.reshape(-1, cs)
        w = ctx.w_cast if (ctx.w_cast is not None and ctx.w_cast.dtype == dy.dtype) else _cast_param(weight, dy.dtype)
        ctx.w_cast = None
        if own_gemm_ok(_lib.HS_EPI_BIAS, c, n_out, dy2.dtype) and own_gemm_ok(_lib.HS_EPI_BIAS, cs, n_out, dy2.dtype):
            wt = _cast_param_t(weight, dy2.dtype, ctx.cast_cache)  # [c + cs, n_out]: the two row blocks are the B operands
            dx = gemm_nt(dy2, wt[:c])[0].reshape(x.shape) if ctx.needs_input_grad[0] else None
            dskip = gemm_nt(dy2, wt[c:])[0].reshape(skip.shape) if ctx.needs_input_grad[1] else None
        else:
            dx = (dy2 @ w[:, :c]).reshape(x.shape) if ctx.needs_input_grad[0] else None
            dskip = (dy2 @ w[:, c:]).reshape(skip.shape) if ctx.needs_input_grad[1] else None
        want_w = ctx.needs_input_grad[2]
        want_b = bias is not None and ctx.needs_input_grad[3]
        if not (want_w or want_b):
            return dx, dskip, None, None
        align = 8 if x.dtype == torch.bfloat16 else 4
        hip_ok = (x.dtype in (torch.bfloat16, torch.float32) and n_out % 4 == 0 and c % align == 0 and cs % align == 0
                  and x2.is_contiguous() and s2.is_contiguous())
        if hip_ok:
            dwa, db32 = LinearFn._wgrad_hip(dy2, x2, n_out, c, want_b)
            dwb, _ = LinearFn._wgrad_hip(dy2, s2, n_out, cs, False)
        else:
            dwa, dwb = dy2.t() @ x2, dy2.t() @ s2
            db32 = dy2.sum(0) if want_b else None
        wbuf = _sink_buffer(weight) if want_w else None
        bbuf = _sink_buffer(bias) if (wbuf is not None and want_b) else None
        if wbuf is not None and (not want_b or bbuf is not None):
            wbuf[:, :c].add_(dwa)
            wbuf[:, c:].add_(dwb)
            if want_b:
                bbuf.add_(db32)
            RT.grad_sink.deposited(weight)
            if want_b:
                RT.grad_sink.deposited(bias)
            return dx, dskip, None, None
        dw = torch.cat([dwa, dwb], 1).to(weight.dtype) if want_w else None
        return dx, dskip, dw, (db32.to(bias.dtype) if want_b else None)


def concat_linear(x, skip, weight, bias=None):
    return ConcatLinearFn.apply(x, skip, weight, bias)


# ----------------------------------------------------------------------------- standalone shift (gather rows)
class PatchMergeFn(torch.autograd.Function):
    """PatchMerging.forward (ref :378-395) through the one-call C-ABI operators `hs_patch_merge_fwd/bwd`: x [B, N, C] bf16 ->
    [B, N/4, dim_out].  The operator-level binding of INTEGRATION.md; the nn.Module mirror composes the same kernels itself
    (per-shape choice between `hs_gemm_nt` and the library GEMM)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, weight):
        _require_gpu(x, gamma, beta, weight)
        B, N, C = x.shape
        assert N % 4 == 0, f"x size {N} is not divisible by 4 as necessary for patching."
        x = x.contiguous()
        rows, dim_out = B * N // 4, weight.shape[0]
        dt = _lib.dtype_code(x.dtype)
        w = weight.detach().to(x.dtype).contiguous()
        g, b = _f32(gamma), _f32(beta)
        normed = torch.empty((rows, 4 * C), dtype=x.dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        out = torch.empty((B, N // 4, dim_out), dtype=x.dtype, device=x.device)
        check(lib.hs_patch_merge_fwd(ptr(x), ptr(g), ptr(b), ptr(w), ptr(normed), ptr(mean), ptr(rstd), ptr(out), rows, C, dim_out, dt,
                                     stream_ptr(x.device)), "hs_patch_merge_fwd")
        ctx.save_for_backward(x, normed, g, mean, rstd, w)
        ctx.meta = (rows, C, dim_out, dt, gamma.dtype, beta.dtype, weight.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, normed, g, mean, rstd, w = ctx.saved_tensors
        rows, C, dim_out, dt, gdt, bdt, wdt = ctx.meta
        dev = x.device
        dout = dout.contiguous()
        w_t = w.t().contiguous()
        dnormed = torch.empty_like(normed)
        dx = torch.empty_like(x)
        dw = torch.empty((dim_out, 4 * C), dtype=torch.float32, device=dev)
        dgamma = torch.empty(4 * C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(4 * C, dtype=torch.float32, device=dev)
        ws = torch.empty(int(lib.hs_patch_merge_bwd_workspace(rows, C, dim_out)), dtype=torch.float32, device=dev)
        check(lib.hs_patch_merge_bwd(ptr(dout), ptr(x), ptr(normed), ptr(g), ptr(mean), ptr(rstd), ptr(w_t), ptr(dnormed), ptr(dx),
                                     ptr(dw), ptr(dgamma), ptr(dbeta), ptr(ws), 0, rows, C, dim_out, dt, stream_ptr(dev)),
              "hs_patch_merge_bwd")
        return dx, dgamma.to(gdt), dbeta.to(bdt), dw.to(wdt)


def patch_merge(x, gamma, beta, weight):
    return PatchMergeFn.apply(x, gamma, beta, weight)


class PatchExpandFn(torch.autograd.Function):
    """PatchExpand.forward (ref :418-430, children = 4) / FinalPatchExpand_X4.forward (:441-452, children = patch_size) through
    `hs_patch_expand_fwd/bwd`: x [B, N, C] bf16 -> [B, N children, dim_exp / children]."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, children):
        _require_gpu(x, gamma, beta, weight)
        B, N, C = x.shape
        x = x.contiguous()
        rows, dim_exp = B * N, weight.shape[0]
        dt = _lib.dtype_code(x.dtype)
        w = weight.detach().to(x.dtype).contiguous()
        g, b = _f32(gamma), _f32(beta)
        expanded = torch.empty((rows, dim_exp), dtype=x.dtype, device=x.device)
        mean = torch.empty(rows * children, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows * children, dtype=torch.float32, device=x.device)
        out = torch.empty((B, N * children, dim_exp // children), dtype=x.dtype, device=x.device)
        check(lib.hs_patch_expand_fwd(ptr(x), ptr(w), ptr(g), ptr(b), ptr(expanded), ptr(mean), ptr(rstd), ptr(out), rows, C, dim_exp,
                                      children, dt, stream_ptr(x.device)), "hs_patch_expand_fwd")
        ctx.save_for_backward(x, expanded, g, mean, rstd, w)
        ctx.meta = (rows, C, dim_exp, children, dt, gamma.dtype, beta.dtype, weight.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, expanded, g, mean, rstd, w = ctx.saved_tensors
        rows, C, dim_exp, children, dt, gdt, bdt, wdt = ctx.meta
        dev = x.device
        dout = dout.contiguous()
        w_t = w.t().contiguous()
        dexp = torch.empty_like(expanded)
        dx = torch.empty_like(x)
        dw = torch.empty((dim_exp, C), dtype=torch.float32, device=dev)
        dgamma = torch.empty(dim_exp // children, dtype=torch.float32, device=dev)
        dbeta = torch.empty(dim_exp // children, dtype=torch.float32, device=dev)
        ws = torch.empty(int(lib.hs_patch_expand_bwd_workspace(rows, C, dim_exp, children)), dtype=torch.float32, device=dev)
        check(lib.hs_patch_expand_bwd(ptr(dout), ptr(x), ptr(expanded), ptr(g), ptr(mean), ptr(rstd), ptr(w_t), ptr(dexp), ptr(dx),
                                      ptr(dw), ptr(dgamma), ptr(dbeta), ptr(ws), 0, rows, C, dim_exp, children, dt, stream_ptr(dev)),
              "hs_patch_expand_bwd")
        return dx, dw.to(wdt), dgamma.to(gdt), dbeta.to(bdt), None


def patch_expand(x, weight, gamma, beta, children=4):
    return PatchExpandFn.apply(x, weight, gamma, beta, children)


class GatherRowsFn(torch.autograd.Function):
    """out[:, j] = x[:, idx[j]]  (or roll); backward gathers with the inverse table."""

    @staticmethod
    def forward(ctx, x, idx, inv, roll):
        _require_gpu(x, idx, inv)
        x = x.contiguous()
        B, N = x.shape[0], x.shape[1]
        row_bytes = (x.numel() // (B * N)) * x.element_size()
        out = torch.empty_like(x)
        check(lib.hs_gather_rows(ptr(x), ptr(out), ptr(idx), int(roll), B, N, row_bytes, stream_ptr(x.device)), "hs_gather_rows")
        ctx.save_for_backward(idx, inv)
        ctx.meta = (B, N, row_bytes, int(roll))
        return out

    @staticmethod
    def backward(ctx, dout):
        idx, inv = ctx.saved_tensors
        B, N, row_bytes, roll = ctx.meta
        dout = dout.contiguous()
        dx = torch.empty_like(dout)
        back_roll = (N - roll) % N
        check(lib.hs_gather_rows(ptr(dout), ptr(dx), ptr(inv), back_roll, B, N, row_bytes, stream_ptr(dout.device)), "hs_gather_rows")
        return dx, None, None, None


def gather_rows(x, idx=None, inv=None, roll=0):
    return GatherRowsFn.apply(x, idx, inv, roll)
