"""Caller-side losses of the two reference Lightning modules, so that a train step
is self-contained.  Thin torch compositions; fp32 arithmetic regardless of the logits dtype."""
import torch
import torch.nn.functional as F

# MaskedDepthDataStatistics, heal_swin/data/depth_estimation/normalize_depth_data.py:31-40
DEPTH_MEAN = 13.654291032986958
DEPTH_STD = 29.58008801108711


def seg_loss(logits, labels, class_weights=None):
    """nn.CrossEntropyLoss(weight)(logits[B,K,Npix], labels.long()[B,Npix])
    (heal_swin/models_lightning/segmentation/model_lightning_swin_hp.py:39-45, :104-111)."""
    w = None if class_weights is None else class_weights.to(device=logits.device, dtype=torch.float32)
    return F.cross_entropy(logits.float(), labels.long(), weight=w)


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
    return ~torch.isinf(target).detach()


def depth_l1_loss(pred, target, mask_background=False):
    """mean |pred[:,0] - target| over non-inf targets (heal_swin/training/loss_depth_regression.py:41-53)."""
    keep = _finite(target)
    return (pred[:, 0].float()[keep] - target[keep]).abs().mean()


def depth_l2_loss(pred, target, mask_background=False):
    """`mse`: mean (pred[:,0] - target)^2 / 2 over non-inf targets (loss_depth_regression.py:9-21)."""
    keep = _finite(target)
    return ((pred[:, 0].float()[keep] - target[keep]) ** 2 / 2).mean()
