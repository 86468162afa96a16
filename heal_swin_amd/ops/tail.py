"""Decoder tail: final LayerNorm + class head (+ expand, + cross-entropy) in one launch per direction
(swin_hp_transformer.py:433-452, :756-761, :781-788)."""

import torch

from .. import _lib
from .._lib import check, lib, ptr, stream_ptr
from .runtime import RT, _cast_param, _require_gpu, _timed  # noqa: F401
from .gemm import LinearFn, _input_grad, _param_grads  # noqa: F401


FUSED_LN_HEAD = True  # (tests flip the attribute to compare with the unfused tail)


def ln_head_ok(x, width, n_classes):
    """Whether `ln_head` (hs_ln_head_*) runs this decoder tail: bf16 rows on the GPU, C in 64..256 (multiple of 32), <= 16 classes."""
    return bool(FUSED_LN_HEAD and x.is_cuda and x.dtype == torch.bfloat16 and
                lib.hs_ln_head_supported(int(width), int(n_classes), _lib.HS_BF16))


class LnHeadFn(torch.autograd.Function):
    """LayerNorm(C) + bias-free 1x1 head as one pass over the rows, forward and backward (reference: the `norm` of
    FinalPatchExpand_X4, swin_hp_transformer.py:448-452, followed by `self.output`, :785-788): the normalised [rows, C] tensor is
    neither written nor saved.  Returns the padded logits [rows, 16] in FP32 (columns >= f_out are zero); backward takes their gradient.
    Parameter gradients come from ONE weight-gradient product over the raw rows (see csrc/ln_head.hip):
        X[k, c] = sum_rows dlogits[row, k] xhat[row, c] = hs_linear_wgrad(dlogits * rstd, y)[k, c] - sum_rows dlogits rstd mean
        dW = gamma X + beta u,   dgamma_c = sum_k W X,   dbeta_c = sum_k W u,   u[k] = sum_rows dlogits[row, k]."""

    KP = 16

    @staticmethod
    def forward(ctx, y2, gamma, beta, weight):
        _require_gpu(y2, gamma, beta, weight)
        rows, C = y2.shape
        wfold, bvec = _fold_head(gamma, beta, weight, C, y2.device)
        # fp32 logits: the tail's roundings (norm_up -> expand -> xhat -> logits) dominate the bf16 logit error of the whole
        # model (csrc/ln_head.hip); the logits therefore keep their accumulator value and xhat enters the head as hi + lo
        logits = torch.empty((rows, LnHeadFn.KP), dtype=torch.float32, device=y2.device)
        mean = torch.empty(rows, dtype=torch.float32, device=y2.device)
        rstd = torch.empty_like(mean)
        with _timed("ln_head_fwd", y2.device, rows * (2 * C + 4 * LnHeadFn.KP) + 8 * rows, 2 * rows * C * 32):
            check(lib.hs_ln_head_fwd(ptr(y2), ptr(wfold), ptr(bvec), ptr(logits), ptr(mean), ptr(rstd), rows, C, _lib.HS_BF16,
                                     _lib.HS_F32, stream_ptr(y2.device)), "hs_ln_head_fwd")
        ctx.save_for_backward(y2, mean, rstd, gamma, beta, weight)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        y2, mean, rstd, gamma, beta, weight = ctx.saved_tensors
        dy, dgamma, dbeta, dw = _ln_head_backward(y2, mean, rstd, gamma, beta, weight, dlogits, any(ctx.needs_input_grad[1:]))
        return (dy if ctx.needs_input_grad[0] else None), dgamma, dbeta, dw


def _ln_head_backward(y2, mean, rstd, gamma, beta, weight, dlogits, want_params, ce=None):
    """(dy, dgamma, dbeta, dWhead) of logits = head(LayerNorm(y2)) by `hs_ln_head_bwd` + one weight-gradient product (LnHeadFn).
    ce = (labels u8 [rows], class weights or None, scale f32[1]): the logits' gradient is that of the weighted cross-entropy and is
    formed inside the kernel (`hs_ln_head_ce_bwd`) instead of being read."""
    rows, C = y2.shape
    f_out, KP = weight.shape[0], LnHeadFn.KP
    dev = y2.device
    w = weight.detach().reshape(f_out, C).float()
    g32, b32 = gamma.detach().float(), beta.detach().float()
    afold = torch.zeros((C, KP), dtype=torch.bfloat16, device=dev)
    afold[:, :f_out] = (w * g32).t().to(torch.bfloat16)
    dy = torch.empty_like(y2)
    dprime = torch.empty((rows, KP), dtype=torch.bfloat16, device=dev)
    part = torch.empty((int(lib.hs_ln_head_partials(rows)), 32), dtype=torch.float32, device=dev)
    if ce is not None:
        labels, class_w, scale = ce
        wfold, bvec = _fold_head_ce(gamma, beta, weight, C, dev)
        with _timed("ln_head_ce_bwd", dev, rows * (4 * C + 2 * KP + 1) + 8 * rows, 2 * rows * C * (KP + 96)):
            check(lib.hs_ln_head_ce_bwd(ptr(y2), ptr(mean), ptr(rstd), ptr(labels), ptr(class_w), ptr(scale), f_out, ptr(wfold), ptr(bvec),
                                        ptr(afold), ptr(dy), ptr(dprime), ptr(part), rows, C, _lib.HS_BF16, stream_ptr(dev)),
                  "hs_ln_head_ce_bwd")
    else:
        dlogits = dlogits.to(torch.float32).contiguous()
        with _timed("ln_head_bwd", dev, rows * (4 * C + 6 * KP) + 8 * rows, 2 * rows * C * KP):
            check(lib.hs_ln_head_bwd(ptr(y2), ptr(mean), ptr(rstd), ptr(dlogits), ptr(afold), ptr(dy), ptr(dprime), ptr(part), rows, C,
                                     _lib.HS_BF16, _lib.HS_F32, stream_ptr(dev)), "hs_ln_head_bwd")
    dgamma = dbeta = dw = None
    if want_params:
        ut = part.sum(0)
        u, t = ut[:f_out], ut[KP:KP + f_out]
        G = LinearFn._wgrad_hip(dprime, y2, KP, C, False)[0][:f_out]
        X = G - t[:, None]
        dw = (g32 * X + b32 * u[:, None]).to(weight.dtype).view(weight.shape)
        dgamma = (w * X).sum(0).to(gamma.dtype)
        dbeta = (w * u[:, None]).sum(0).to(beta.dtype)
    return dy, dgamma, dbeta, dw


def _fold_head(gamma, beta, weight, C, device):
    """(wfold [64, C] bf16: rows 0..31 = gamma * W rounded to bf16, rows 32..63 the rounding remainder (read by
    hs_expand_ln_head_fwd only); bvec [32] f32 = W beta) of the fused LayerNorm + head kernels."""
    f_out = weight.shape[0]
    w = weight.detach().reshape(f_out, C).float()
    wfold = torch.zeros((64, C), dtype=torch.bfloat16, device=device)
    prod = w * gamma.detach().float()
    wfold[:f_out] = prod.to(torch.bfloat16)
    wfold[32:32 + f_out] = (prod - wfold[:f_out].float()).to(torch.bfloat16)
    bvec = torch.zeros(32, dtype=torch.float32, device=device)
    bvec[:f_out] = w @ beta.detach().float()
    return wfold, bvec


FUSED_EXPAND_HEAD = True


def expand_ln_head_ok(x, width, children, n_classes):
    """Whether `expand_ln_head` (hs_expand_ln_head_fwd) runs the decoder tail: bf16 rows on the GPU, 4 children, C in {64, 96, 128}."""
    return bool(FUSED_EXPAND_HEAD and FUSED_LN_HEAD and x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] == width and
                lib.hs_expand_ln_head_supported(int(width), int(children), int(n_classes), _lib.HS_BF16))


class ExpandLnHeadFn(torch.autograd.Function):
    """FinalPatchExpand_X4 (Linear C -> 4 C, view, LayerNorm(C)) + the 1x1 head as ONE forward kernel (reference
    swin_hp_transformer.py:442-452, :785-788; csrc/expand_ln_head.hip).  xn2 [tokens, C] bf16 -> padded fp32 logits [4 tokens, 16].
    With a gradient wanted the kernel also writes the expanded rows once (the backward's LayerNorm input); the backward is
    `hs_ln_head_bwd` on them followed by the Linear's input / weight gradients.  Without, the [4 tokens, C] tensor never exists."""

    @staticmethod
    def forward(ctx, xn2, wexp, gamma, beta, weight, xn_lo=None):
        _require_gpu(xn2, wexp, gamma, beta, weight, xn_lo)
        tokens, C = xn2.shape
        xn2 = xn2.contiguous()
        xn_lo = None if xn_lo is None else xn_lo.reshape(tokens, C).contiguous()
        P = wexp.shape[0] // C
        wq = _cast_param(wexp, torch.bfloat16).contiguous()
        wfold, bvec = _fold_head(gamma, beta, weight, C, xn2.device)
        need = any(ctx.needs_input_grad)
        rows = tokens * P
        logits = torch.empty((rows, LnHeadFn.KP), dtype=torch.float32, device=xn2.device)
        y = torch.empty((rows, C), dtype=torch.bfloat16, device=xn2.device) if need else None
        mean = torch.empty(rows, dtype=torch.float32, device=xn2.device) if need else None
        rstd = torch.empty_like(mean) if need else None
        # algorithmic traffic: xn in, logits out (+ the expanded rows once in training); flops: expand + head (hi + lo)
        with _timed("expand_ln_head_fwd", xn2.device, 2 * tokens * C + rows * (4 * LnHeadFn.KP + (2 * C + 8 if need else 0)),
                    2 * rows * C * C + 4 * rows * C * 32):
            check(lib.hs_expand_ln_head_fwd(ptr(xn2), ptr(xn_lo), ptr(wq), ptr(wfold), ptr(bvec), ptr(y), ptr(logits), ptr(mean), ptr(rstd),
                                            tokens, C, P, _lib.HS_BF16, stream_ptr(xn2.device)), "hs_expand_ln_head_fwd")
        ctx.save_for_backward(xn2, y, mean, rstd, gamma, beta, weight, wexp)
        ctx.w_cast = wq if wq.dtype != wexp.dtype else None
        ctx.cast_cache = RT.cast_cache
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        xn2, y, mean, rstd, gamma, beta, weight, wexp = ctx.saved_tensors
        tokens, C = xn2.shape
        dy, dgamma, dbeta, dw = _ln_head_backward(y, mean, rstd, gamma, beta, weight, dlogits, any(ctx.needs_input_grad[2:]))
        dy2 = dy.view(tokens, wexp.shape[0])  # 'b (n p) c -> b n (p c)': the children of a token are consecutive rows
        dxn = _input_grad(dy2, wexp, ctx.w_cast, None, ctx.cast_cache) if ctx.needs_input_grad[0] else None
        ctx.w_cast = ctx.cast_cache = None
        dwexp, _ = _param_grads(dy2, xn2, wexp, None, ctx.needs_input_grad[1], False)
        return dxn, dwexp, dgamma, dbeta, dw, None


_CE_PERM = {}


def _fold_head_ce(gamma, beta, weight, C, device):
    """The folded head weight for `hs_ln_head_ce_bwd`: as _fold_head, but with row blocks 4..7 and 8..11 exchanged, so that the
    kernel's accumulator register r < 8 of lane half h is class 8 h + r (csrc/ln_head.hip:ln_head_ce_bwd_kernel)."""
    wfold, bvec = _fold_head(gamma, beta, weight, C, device)
    key = str(device)
    if key not in _CE_PERM:  # (built once per device: six tiny launches per step otherwise)
        perm = list(range(4)) + list(range(8, 12)) + list(range(4, 8)) + list(range(12, 32))
        _CE_PERM[key] = (torch.tensor(perm + [i + 32 for i in perm], device=device), torch.tensor(perm, device=device))
    perm64, perm = _CE_PERM[key]
    return wfold[perm64], bvec[perm]  # (advanced indexing: fresh contiguous tensors)


class ExpandLnHeadCeFn(torch.autograd.Function):
    """The decoder tail AND the segmentation caller's weighted cross-entropy (reference swin_hp_transformer.py:442-452, :785-788 and
    models_lightning/segmentation/model_lightning_swin_hp.py:39-45, :104-111) as one forward and one backward kernel
    (`hs_expand_ln_head_ce_fwd`, `hs_ln_head_ce_bwd`): the [B, Npix, 16] fp32 logits and their gradient never exist in
    HBM.  xn2 [tokens, C] bf16, labels u8 [4 tokens] in pixel order -> scalar loss (fp32)."""

    @staticmethod
    def forward(ctx, xn2, wexp, gamma, beta, weight, labels, class_w, xn_lo):
        _require_gpu(xn2, wexp, gamma, beta, weight, labels, class_w, xn_lo)
        tokens, C = xn2.shape
        xn2 = xn2.contiguous()
        xn_lo = None if xn_lo is None else xn_lo.reshape(tokens, C).contiguous()
        P = wexp.shape[0] // C
        f_out = weight.shape[0]
        wq = _cast_param(wexp, torch.bfloat16).contiguous()
        wfold, bvec = _fold_head(gamma, beta, weight, C, xn2.device)
        need = any(ctx.needs_input_grad[:5])
        rows = tokens * P
        labels = labels.reshape(-1)
        assert labels.dtype == torch.uint8 and labels.numel() == rows and labels.is_contiguous(), "labels: contiguous uint8, one per pixel row"
        y = torch.empty((rows, C), dtype=torch.bfloat16, device=xn2.device) if need else None
        mean = torch.empty(rows, dtype=torch.float32, device=xn2.device) if need else None
        rstd = torch.empty_like(mean) if need else None
        parts = torch.empty((4 * int(lib.hs_expand_ln_head_blocks(tokens)), 2), dtype=torch.float32, device=xn2.device)
        # algorithmic traffic: xn in, labels in (+ the expanded rows once in training); no logits
        with _timed("expand_ln_head_ce_fwd", xn2.device, 2 * tokens * C + rows * (1 + (2 * C + 8 if need else 0)),
                    2 * rows * C * C + 4 * rows * C * 32):
            check(lib.hs_expand_ln_head_ce_fwd(ptr(xn2), ptr(xn_lo), ptr(wq), ptr(wfold), ptr(bvec), ptr(labels), ptr(class_w), f_out,
                                               ptr(y), None, ptr(mean), ptr(rstd), ptr(parts), tokens, C, P, _lib.HS_BF16,
                                               stream_ptr(xn2.device)), "hs_expand_ln_head_ce_fwd")
        tot = parts.sum(0)
        ctx.save_for_backward(xn2, y, mean, rstd, gamma, beta, weight, wexp, labels, class_w, tot)
        ctx.w_cast = wq if wq.dtype != wexp.dtype else None
        ctx.cast_cache = RT.cast_cache
        return tot[0] / tot[1]

    @staticmethod
    def backward(ctx, dloss):
        xn2, y, mean, rstd, gamma, beta, weight, wexp, labels, class_w, tot = ctx.saved_tensors
        tokens, C = xn2.shape
        scale = (dloss.to(torch.float32) / tot[1]).reshape(1)
        dy, dgamma, dbeta, dw = _ln_head_backward(y, mean, rstd, gamma, beta, weight, None, any(ctx.needs_input_grad[2:5]),
                                                  ce=(labels, class_w, scale))
        dy2 = dy.view(tokens, wexp.shape[0])
        dxn = _input_grad(dy2, wexp, ctx.w_cast, None, ctx.cast_cache) if ctx.needs_input_grad[0] else None
        ctx.w_cast = ctx.cast_cache = None
        dwexp, _ = _param_grads(dy2, xn2, wexp, None, ctx.needs_input_grad[1], False)
        return dxn, dwexp, dgamma, dbeta, dw, None, None, None


def expand_ln_head_ce(xn2, wexp, gamma, beta, weight, labels, class_weights=None, xn_lo=None):
    """Weighted cross-entropy of head(LayerNorm(expand(xn2 [+ xn_lo]) viewed per child)) against uint8 pixel labels, without the
    logits (ExpandLnHeadCeFn)."""
    return ExpandLnHeadCeFn.apply(xn2, wexp, gamma, beta, weight, labels, class_weights, xn_lo)


def expand_ln_head(xn2, wexp, gamma, beta, weight, xn_lo=None):
    """Padded fp32 logits [4 tokens, 16] of head(LayerNorm(expand(xn2 [+ xn_lo]) viewed per child)); the caller slices [..., :f_out].
    xn_lo: the rounding remainder of xn2 (`layer_norm_hilo`), used by the forward product only (the gradients take xn2)."""
    return ExpandLnHeadFn.apply(xn2, wexp, gamma, beta, weight, xn_lo)


def ln_head(y2, gamma, beta, weight):
    """Padded logits [rows, 16] of head(LayerNorm(y2)); the caller slices [..., :f_out]."""
    return LnHeadFn.apply(y2, gamma, beta, weight)
