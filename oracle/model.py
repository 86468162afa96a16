"""Floating-point oracle of the HEAL-SWIN forward pass: a functional torch-CPU fp32 restatement that
runs directly off a reference-layout state dict.  Oracle / test infra -- see oracle/__init__.py.

It is written from formulas (explicit mean/var LayerNorm, erf GELU, exp-normalise softmax, explicit
window loops-as-reshapes) rather than nn.Modules so that it shares no code path with either the
reference or the product.  Gradients come from torch autograd over these formulas.

Pinned against the reference by tests/test_oracle_model.py using tests/golden/{modules,models}.npz.
All `file:line` citations are into /root/reference/heal_swin/models_torch/swin_hp_transformer.py unless
another file is named.
"""
import math

import numpy as np
import torch

from . import tables

LN_EPS = 1e-5  # torch.nn.LayerNorm default, used everywhere in the reference (norm_layer=nn.LayerNorm)


# ----------------------------------------------------------------------------- primitives
def layer_norm(x, weight, bias, eps=LN_EPS):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)  # biased variance, as nn.LayerNorm
    return (x - mu) / torch.sqrt(var + eps) * weight + bias


def linear(x, weight, bias=None):
    y = x @ weight.t()
    return y if bias is None else y + bias


def gelu(x):
    """exact (erf) GELU -- nn.GELU default, reference Mlp :27,:40"""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def softmax_lastdim(s):
    s = s - s.max(dim=-1, keepdim=True).values
    e = torch.exp(s)
    return e / e.sum(dim=-1, keepdim=True)


def l2_normalize(x, eps=1e-12):
    """F.normalize(x, dim=-1): x / max(||x||_2, eps)   (:143)"""
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


# ----------------------------------------------------------------------------- shifts
class Shifter:
    """idx / inv / labels triple of oracle.tables as torch tensors (reference hp_shifting classes)."""

    def __init__(self, strategy, n_tokens, base_pix, window_size, shift_size):
        self.window_size = window_size
        if shift_size <= 0:  # NoShift, hp_shifting.py:31-39
            self.idx = self.inv = self.labels = None
            self.mask_is_int = False
            return
        nside = math.isqrt(n_tokens // base_pix)
        assert nside * nside * base_pix == n_tokens, "nside has to be an integer in every layer"  # :272-274
        if strategy == "nest_roll":
            idx, inv, lab = tables.nest_roll_shift(n_tokens, window_size, shift_size)
        elif strategy == "nest_grid_shift":
            idx, inv, lab = tables.nest_grid_shift(nside, base_pix, window_size)
        elif strategy == "ring_shift":
            idx, inv, lab = tables.ring_shift(nside, base_pix, window_size, shift_size)
        else:
            raise KeyError(strategy)
        self.idx = torch.from_numpy(idx)
        self.inv = torch.from_numpy(inv)
        self.labels = torch.from_numpy(lab)
        self.mask_is_int = strategy == "ring_shift"  # int64 mask, hp_shifting.py:380

    def shift(self, x):
        return x if self.idx is None else x[:, self.idx]

    def shift_back(self, x):
        return x if self.inv is None else x[:, self.inv]

    def attn_mask(self):
        """[nW, Ws, Ws] float32 additive mask {0, -100} or None (hp_shifting.py:10-28)."""
        if self.labels is None:
            return None
        m = tables.attn_mask_from_labels(self.labels.numpy(), self.window_size)
        return torch.from_numpy(m.astype(np.float32))


# ----------------------------------------------------------------------------- modules
def window_attention(xw, sd, pre, num_heads, rel_index, mask, use_cos, qk_scale=None):
    """reference WindowAttention.forward :124-174.  xw: [B_, Ws, C] windows, batch-major (b*nW + w)."""
    B_, Ws, C = xw.shape
    hd = C // num_heads
    qkv = linear(xw, sd[pre + "qkv.weight"], sd.get(pre + "qkv.bias"))  # [B_, Ws, 3C], rows [q|k|v][head][hd]
    qkv = qkv.reshape(B_, Ws, 3, num_heads, hd)
    q = qkv[:, :, 0].transpose(1, 2)  # [B_, nH, Ws, hd]
    k = qkv[:, :, 1].transpose(1, 2)
    v = qkv[:, :, 2].transpose(1, 2)
    if use_cos:  # :142-147
        s = l2_normalize(q) @ l2_normalize(k).transpose(-1, -2)
        ls = torch.exp(torch.clamp(sd[pre + "logit_scale"], max=math.log(1.0 / 0.01)))  # [nH,1,1]
        s = s * ls
    else:  # :149-150
        scale = qk_scale or hd ** -0.5
        s = (q * scale) @ k.transpose(-1, -2)
    if rel_index is not None:  # :152-159
        table = sd[pre + "relative_position_bias_table"]  # [(2 side - 1)^2, nH]
        bias = table[rel_index.reshape(-1)].reshape(Ws, Ws, num_heads).permute(2, 0, 1)
        s = s + bias[None]
    if mask is not None:  # :161-164: window w of every image gets mask[w]
        nW = mask.shape[0]
        s = (s.reshape(B_ // nW, nW, num_heads, Ws, Ws) + mask[None, :, None]).reshape(B_, num_heads, Ws, Ws)
    p = softmax_lastdim(s)
    o = (p @ v).transpose(1, 2).reshape(B_, Ws, C)  # heads merged [head][hd], :171
    return linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def mlp(x, sd, pre):
    """reference Mlp.forward :38-44 (dropout p = 0)"""
    return linear(gelu(linear(x, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"])), sd[pre + "fc2.weight"], sd[pre + "fc2.bias"])


def swin_block(x, sd, pre, num_heads, window_size, shifter, rel_index, use_cos, v2_norm, qk_scale=None):
    """reference SwinTransformerBlock.forward :310-340 (drop_path = identity)"""
    B, N, C = x.shape
    ws = min(window_size, N)  # :243-246
    shortcut = x
    if not v2_norm:
        x = layer_norm(x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
    xs = shifter.shift(x)
    xw = xs.reshape(B * (N // ws), ws, C)  # window_partition, hp_windowing.py:18-21
    aw = window_attention(xw, sd, pre + "attn.", num_heads, rel_index, shifter.attn_mask(), use_cos, qk_scale)
    x = shifter.shift_back(aw.reshape(B, N, C))  # window_reverse + shift_back
    if v2_norm:  # :334-335
        x = shortcut + layer_norm(x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
        x = x + layer_norm(mlp(x, sd, pre + "mlp."), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"])
    else:  # :337-338
        x = shortcut + x
        x = x + mlp(layer_norm(x, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"]), sd, pre + "mlp.")
    return x


def patch_merging(x, sd, pre):
    """reference PatchMerging.forward :378-395: 4 sibling pixels -> one row of 4C, LN(4C), Linear(4C->2C)"""
    B, N, C = x.shape
    assert N % 4 == 0
    x = x.reshape(B, N // 4, 4 * C)  # == cat(x[0::4], x[1::4], x[2::4], x[3::4], -1)
    x = layer_norm(x, sd[pre + "norm.weight"], sd[pre + "norm.bias"])
    return linear(x, sd[pre + "reduction.weight"])


def patch_expand(x, sd, pre, p=4):
    """reference PatchExpand.forward :420-430 and FinalPatchExpand_X4.forward :442-452
    (Linear without bias, each token split into p children of C/p channels, LN over the child)."""
    x = linear(x, sd[pre + "expand.weight"])
    B, N, C = x.shape
    x = x.reshape(B, N * p, C // p)  # 'b n (p c) -> b (n p) c'
    return layer_norm(x, sd[pre + "norm.weight"], sd[pre + "norm.bias"])


# ----------------------------------------------------------------------------- whole model
def _stage(x, sd, pre, depth, num_heads, cfg, base_pix, rel_index):
    """reference BasicLayer / BasicLayer_up block loop :539-544, :645-650; odd blocks shifted :516,:622"""
    N = x.shape[1]
    for i in range(depth):
        ws = cfg.window_size
        shift = 0 if i % 2 == 0 else cfg.shift_size
        if N <= ws:  # :243-246
            shift, ws_eff = 0, N
        else:
            ws_eff = ws
        sh = Shifter(cfg.shift_strategy, N, base_pix, ws_eff, shift)
        x = swin_block(x, sd, f"{pre}blocks.{i}.", num_heads, ws_eff, sh, rel_index, cfg.use_cos_attn,
                       cfg.use_v2_norm_placement, cfg.qk_scale)
    return x


def forward(sd, cfg, spec, x, taps=None):
    """reference SwinHPTransformerSys.forward :948-955 (+ forward_features :930-946, UnetDecoder.forward
    :765-791).  `sd`: reference-layout state dict of fp32 tensors; `cfg`: anything with the
    SwinHPTransformerConfig fields (:794-818); `spec`: anything with dim_in, f_in, f_out, base_pix.
    x: [B, f_in, dim_in] -> [B, f_out, dim_in].  All drop rates are treated as 0.
    `taps` (a dict, optional) receives the intermediate activations a parity test wants to localise an error with:
    "patch_embed", "layers.{i}" (output of encoder stage i incl. its PatchMerging), "norm", "decoder.layers_up.{k}".
    """
    def tap(name, t):
        if taps is not None:
            taps[name] = t.detach()

    sd = {k: v for k, v in sd.items()}
    L = len(cfg.depths)
    B, f_in, npix = x.shape
    assert npix == spec.dim_in
    rel_index = None
    if cfg.rel_pos_bias == "flat":
        rel_index = torch.from_numpy(tables.rel_pos_index(cfg.window_size))

    # PatchEmbed :686-694: Conv1d(k = s = patch) == per-patch linear over (channel, offset)
    P = cfg.patch_size
    w = sd["patch_embed.proj.weight"]  # [C, f_in, P]
    xp = x.reshape(B, f_in, npix // P, P).permute(0, 2, 1, 3).reshape(B, npix // P, f_in * P)
    x = xp @ w.reshape(w.shape[0], -1).t() + sd["patch_embed.proj.bias"]
    if cfg.ape:
        x = x + sd["absolute_pos_embed"]
    tap("patch_embed", x)

    skips = []
    for i in range(L):  # :939-943
        skips.append(x)
        x = _stage(x, sd, f"layers.{i}.", cfg.depths[i], cfg.num_heads[i], cfg, spec.base_pix, rel_index)
        if i < L - 1:
            x = patch_merging(x, sd, f"layers.{i}.downsample.")
        tap(f"layers.{i}", x)
    x = layer_norm(x, sd["norm.weight"], sd["norm.bias"])  # :945
    tap("norm", x)

    for inx in range(L):  # :766-778
        pre = f"decoder.layers_up.{inx}."
        if inx == 0:
            x = patch_expand(x, sd, pre)
            tap(f"decoder.layers_up.{inx}", x)
            continue
        down = L - 1 - inx
        x = torch.cat([x, skips[down]], dim=-1)
        x = linear(x, sd[f"decoder.concat_back_dim.{inx}.weight"], sd[f"decoder.concat_back_dim.{inx}.bias"])
        x = _stage(x, sd, pre, cfg.depths[down], cfg.num_heads[down], cfg, spec.base_pix, rel_index)
        if down > 0:
            x = patch_expand(x, sd, pre + "upsample.")
        tap(f"decoder.layers_up.{inx}", x)
    x = layer_norm(x, sd["decoder.norm_up.weight"], sd["decoder.norm_up.bias"])  # :781
    x = patch_expand(x, sd, "decoder.up.", p=cfg.patch_size)  # :782
    wo = sd["decoder.output.weight"]  # [f_out, C, 1], 1x1 conv without bias :756-761
    return (x @ wo[:, :, 0].t()).transpose(1, 2)  # [B, f_out, Npix]


# ----------------------------------------------------------------------------- caller-side losses
def seg_loss(logits, labels, class_weights=None):
    """reference WoodscapeSegmenterSwinHP loss, models_lightning/segmentation/model_lightning_swin_hp.py
    :39-45,:104-111: nn.CrossEntropyLoss(weight)(logits[B,K,Npix], labels.long()[B,Npix]), i.e.
    sum_i w[y_i] * (-log softmax(z_i)[y_i]) / sum_i w[y_i]."""
    B, K, Np = logits.shape
    z = logits.permute(0, 2, 1).reshape(-1, K)
    y = labels.reshape(-1).long()
    zmax = z.max(dim=1, keepdim=True).values
    lse = torch.log(torch.exp(z - zmax).sum(dim=1)) + zmax[:, 0]
    nll = lse - z[torch.arange(z.shape[0]), y]
    w = torch.ones(K, dtype=z.dtype) if class_weights is None else class_weights.to(z.dtype)
    wy = w[y]
    return (wy * nll).sum() / wy.sum()


DEPTH_MEAN = 13.654291032986958  # MaskedDepthDataStatistics, data/depth_estimation/normalize_depth_data.py:31-40
DEPTH_STD = 29.58008801108711


def depth_standardize(d):
    """normalize_data(..., 'standardize'), normalize_depth_data.py:133-143"""
    return (d - DEPTH_MEAN) / DEPTH_STD


def depth_unstandardize(d):
    """unnormalize_data(..., 'standardize'), normalize_depth_data.py:146-158"""
    return d * DEPTH_STD + DEPTH_MEAN


def depth_l1_loss(pred, target):
    """reference training/loss_depth_regression.py:41-53: mean |pred[:,0] - target| over non-inf targets."""
    means = pred[:, 0]
    keep = ~torch.isinf(target)
    return (means[keep] - target[keep]).abs().mean()


def depth_l2_loss(pred, target):
    """reference training/loss_depth_regression.py:9-21 (`mse`): mean (pred-target)^2 / 2 over non-inf."""
    means = pred[:, 0]
    keep = ~torch.isinf(target)
    return ((means[keep] - target[keep]) ** 2 / 2).mean()


def depth_huber_loss(pred, target, delta=1.0):
    """reference training/loss_depth_regression.py:56-68: SmoothL1Loss(beta=delta, mean) over non-inf targets,
    i.e. 0.5 d^2 / delta for |d| < delta, else |d| - 0.5 delta.  The reference indexes `preds` (all channels) with the
    [B,1,Npix] mask, which only works for a one-channel prediction."""
    assert pred.shape[1] == 1, "the reference's huber_loss indexes preds[B,C,Npix] with a [B,1,Npix] mask: C must be 1"
    keep = ~torch.isinf(target)
    d = (pred[:, 0][keep] - target[keep]).abs()
    return torch.where(d < delta, 0.5 * d * d / delta, d - 0.5 * delta).mean()


def depth_mean_log_var_loss(pred, target):
    """reference training/loss_depth_regression.py:23-38: mean of log_var/2 + (mean-target)^2 * exp(-log_var)/2 over
    non-inf targets; channel 0 = mean, channel 1 = log variance."""
    means, log_var = pred[:, 0], pred[:, 1]
    keep = ~torch.isinf(target)
    return (0.5 * log_var[keep] + (means[keep] - target[keep]) ** 2 * (0.5 * torch.exp(-log_var[keep]))).mean()


def get_depth_loss(loss="l2", use_logvar=False, huber_delta=1.0):
    """reference training/loss_depth_regression.py:70-83 (fields of CommonDepthConfig, depth_common_config.py:7-10)"""
    if use_logvar:
        return depth_mean_log_var_loss
    return {"l2": depth_l2_loss, "l1": depth_l1_loss, "huber": lambda p, t: depth_huber_loss(p, t, huber_delta)}[loss]
