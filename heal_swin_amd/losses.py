"""Caller-side losses of the two reference Lightning modules, so that a train step
is self-contained.  fp32 arithmetic regardless of the logits dtype; the segmentation cross-entropy runs in fused HIP
kernels on device tensors, the depth losses are thin torch compositions."""
from functools import partial

import torch

# MaskedDepthDataStatistics, heal_swin/data/depth_estimation/normalize_depth_data.py:31-40
DEPTH_MEAN = 13.654291032986958
DEPTH_STD = 29.58008801108711


class _SegCrossEntropyFn(torch.autograd.Function):
    """Weighted CE by `hs_seg_ce_fwd/bwd`: reads the logits in place through their strides (the model's [B, Npix, K] output
    viewed as [B, K, Npix] is never transposed or widened to fp32) and recomputes the softmax in the backward."""

    @staticmethod
    def forward(ctx, logits, labels, weights):
        from . import _lib
        from ._lib import check, lib, ptr, stream_ptr
        assert logits.dim() == 3 and labels.shape == (logits.shape[0], logits.shape[2]), "logits [B, K, Npix], labels [B, Npix]"
        if logits.dtype not in (torch.float32, torch.bfloat16):
            logits = logits.float()
        if labels.dtype not in (torch.uint8, torch.int32, torch.int64):
            labels = labels.long()
        labels = labels.contiguous()
        B, K, P = logits.shape
        sb, sk, sp = logits.stride()
        parts = torch.empty((int(lib.hs_seg_ce_partials(B, P)), 2), dtype=torch.float32, device=logits.device)
        dt = _lib.dtype_code(logits.dtype)
        check(lib.hs_seg_ce_fwd(ptr(logits), ptr(labels), ptr(weights), ptr(parts), B, P, K, sb, sk, sp, labels.element_size(), -100,
                                dt, stream_ptr(logits.device)), "hs_seg_ce_fwd")
        tot = parts.sum(0)
        ctx.save_for_backward(logits, labels, weights, tot)
        return tot[0] / tot[1]

    @staticmethod
    def backward(ctx, grad):
        from . import _lib
        from ._lib import check, lib, ptr, stream_ptr
        logits, labels, weights, tot = ctx.saved_tensors
        B, K, P = logits.shape
        sb, sk, sp = logits.stride()
        # (permutation of a dense tensor?  Compared against the contiguous strides of the SHAPE -- `logits.contiguous()` here would
        # copy the whole non-contiguous tensor, 0.5 ms on the headline model, just to read its strides)
        dense = sorted(logits.stride(), reverse=True) == sorted(torch.empty(logits.shape, device="meta").stride(), reverse=True)
        if dense:
            dl = torch.empty_strided(logits.shape, logits.stride(), dtype=logits.dtype, device=logits.device)
        elif sk == 1 and sp > K and sb == P * sp:
            # the model's padded head output ([B, Npix, sp] rows, K of sp columns used, seen as [B, K, Npix]): the gradient is
            # written in the same layout into a zeroed padded buffer, which ops.PadSliceFn hands on whole to the head's
            # backward (no transposing copy, no second zero fill)
            from . import ops
            full = torch.zeros((B, P, sp), dtype=logits.dtype, device=logits.device)
            dl = full[:, :, :K].transpose(1, 2)
            ops.RT.zero_padded_grads[full.data_ptr()] = full  # weak: the entry lives exactly as long as the gradient does
        else:
            dl = torch.empty_like(logits)
        scale = (grad.to(torch.float32) / tot[1]).reshape(1)
        db, dk, dp = dl.stride()
        check(lib.hs_seg_ce_bwd(ptr(logits), ptr(labels), ptr(weights), ptr(scale), ptr(dl), B, P, K, sb, sk, sp, db, dk, dp,
                                labels.element_size(), -100, _lib.dtype_code(logits.dtype), stream_ptr(logits.device)), "hs_seg_ce_bwd")
        return dl, None, None


def seg_loss(logits, labels, class_weights=None):
    """nn.CrossEntropyLoss(weight)(logits[B,K,Npix], labels.long()[B,Npix])
    (heal_swin/models_lightning/segmentation/model_lightning_swin_hp.py:39-45, :104-111) by the fused HIP kernels
    `hs_seg_ce_*` (fp32 arithmetic on bf16 or fp32 logits).  Device tensors only: there is no CPU path."""
    if not logits.is_cuda:
        raise RuntimeError("heal_swin_amd.losses.seg_loss runs only on an MI355X (HIP) device: got CPU logits; "
                           "there is no CPU fallback")
    w = None if class_weights is None else class_weights.to(device=logits.device, dtype=torch.float32).contiguous()
    return _SegCrossEntropyFn.apply(logits, labels, w)


def seg_predictions(logits):
    """`_, preds = torch.max(outputs, 1)` (model_lightning_swin_hp.py:107)"""
    return torch.max(logits, 1)[1]


def depth_standardize(d):
    """normalize_data(..., 'standardize') (normalize_depth_data.py:133-143)"""
    return (d - DEPTH_MEAN) / DEPTH_STD


def depth_unstandardize(d):
    """unnormalize_data(..., 'standardize') (normalize_depth_data.py:146-158)"""
    return d * DEPTH_STD + DEPTH_MEAN


def _finite(target):
    """(keep mask, target with the infinite entries replaced by 0, number of finite entries as an fp32 device scalar).
    The reference selects the finite pixels by boolean-mask indexing (`preds[~isinf(target)]`), which on a device costs a
    `nonzero` and a device-to-host synchronisation per call and cannot be captured in a HIP graph; the same mean is formed
    here as masked sum / count, all on the device."""
    keep = ~torch.isinf(target).detach()
    return keep, torch.where(keep, target, torch.zeros((), dtype=target.dtype, device=target.device)), keep.sum().to(torch.float32)


def _masked_mean(values, keep, count):
    return torch.where(keep, values, torch.zeros((), dtype=values.dtype, device=values.device)).sum() / count


def depth_l1_loss(pred, target, mask_background=False):
    """mean |pred[:,0] - target| over non-inf targets (heal_swin/training/loss_depth_regression.py:41-53)."""
    keep, tgt, n = _finite(target)
    return _masked_mean((pred[:, 0].float() - tgt).abs(), keep, n)


def depth_l2_loss(pred, target, mask_background=False):
    """`mse`: mean (pred[:,0] - target)^2 / 2 over non-inf targets (loss_depth_regression.py:9-21)."""
    keep, tgt, n = _finite(target)
    return _masked_mean((pred[:, 0].float() - tgt) ** 2 / 2, keep, n)


def depth_huber_loss(pred, target, mask_background=False, delta=1.0):
    """SmoothL1Loss(beta=delta, reduction='mean') over non-inf targets (loss_depth_regression.py:56-68); like the reference
    (which indexes all channels of `preds` with the [B,1,Npix] mask) it is defined for one-channel predictions."""
    assert pred.shape[1] == 1, "huber_loss needs a one-channel prediction (reference loss_depth_regression.py:66)"
    keep, tgt, n = _finite(target)
    d = (pred[:, 0].float() - tgt).abs()
    return _masked_mean(torch.where(d < delta, 0.5 * d * d / delta, d - 0.5 * delta), keep, n)


def depth_mean_log_var_loss(pred, target, mask_background=False):
    """mean of log_var/2 + (mean - target)^2 exp(-log_var)/2 over non-inf targets, channel 0 = mean, channel 1 = log variance
    (loss_depth_regression.py:23-38)."""
    keep, tgt, n = _finite(target)
    # (the INPUTS are sanitised at the background pixels, not only the result: the reference's boolean indexing never touches
    # those pixels, while where()'s backward would multiply a 0 gradient with an overflowed exp(-log_var) there: 0 * inf = NaN)
    zero = torch.zeros((), dtype=torch.float32, device=pred.device)
    means, log_var = torch.where(keep, pred[:, 0].float(), zero), torch.where(keep, pred[:, 1].float(), zero)
    return _masked_mean(0.5 * log_var + (means - tgt) ** 2 * (0.5 * torch.exp(-log_var)), keep, n)


def get_depth_loss(common_depth_config):
    """Same selection as the reference's get_depth_loss (loss_depth_regression.py:70-83); the argument needs the fields
    `use_logvar`, `loss` ('l2' | 'l1' | 'huber') and `huber_delta` of CommonDepthConfig (depth_common_config.py:7-10)."""
    if common_depth_config.use_logvar:
        return depth_mean_log_var_loss
    return {"l2": depth_l2_loss, "l1": depth_l1_loss,
            "huber": partial(depth_huber_loss, delta=common_depth_config.huber_delta)}[common_depth_config.loss]
