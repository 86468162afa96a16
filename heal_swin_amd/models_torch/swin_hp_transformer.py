"""HEAL-SWIN-UNet on the MI355X-native hot path.

Keeps the public surface of the reference module `heal_swin/models_torch/swin_hp_transformer.py`
(class names, constructor signatures, attribute paths and therefore state-dict keys, `forward`
contracts) so that it can stand in for it under the reference's Lightning modules, while the
computation is organised differently:

  * per block there is ONE attention kernel call: shift -> window partition -> attention -> window reverse
    -> shift back are fused into `hs_window_attn_fwd/bwd`, which gathers rows of the un-shifted qkv tensor
    through the shifter's int32 table and scatters its output rows back through the same table
    (row permutations commute with the row-wise LayerNorm / Linear layers around it);
  * masks are per-pixel uint8 region labels, not [nW, Ws, Ws] tensors; the reference's `attn_mask` buffer
    only exists in `state_dict()` / is accepted by `load_state_dict()`;
  * all LayerNorms (block norms, PatchMerging's LN(4C) over 4 sibling pixels, PatchExpand's LN over each
    child row) run in the HIP row-LayerNorm kernel, with the v2-placement residual add fused in;
  * activations run in `compute_dtype` (fp32 or bf16; fp32 statistics/softmax/accumulation either way),
    parameters stay fp32.

Reference line numbers in comments refer to heal_swin/models_torch/swin_hp_transformer.py.
"""
import math
from dataclasses import dataclass, field
from typing import List, Literal, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as checkpoint

from .. import _lib, ops
from ..data_spec import DataSpec
from . import hp_shifting

EMIT_REFERENCE_BUFFERS = True  # state_dict() carries the reference's `attn_mask` buffers (ref :306-308)


# ----------------------------------------------------------------------------- leaf layers
class HSLayerNorm(nn.LayerNorm):
    """nn.LayerNorm parameters, HIP kernel arithmetic (`hs_layernorm_fwd/bwd`)."""

    def forward(self, x, residual=None, row_scale=None, drop_p=0.0):
        assert self.elementwise_affine and len(self.normalized_shape) == 1 and abs(self.eps - 1e-5) < 1e-12
        return ops.layer_norm(x, self.weight, self.bias, residual, row_scale=row_scale, drop_p=drop_p)


def _make_norm(norm_layer, dim):
    return HSLayerNorm(dim) if norm_layer is nn.LayerNorm else norm_layer(dim)


class HSLinear(nn.Linear):
    """fp32 master weights; forward and input-gradient GEMMs in the activation dtype (library GEMM), weight and bias
    gradients by the HIP split-token kernel `hs_linear_wgrad` (ops.LinearFn)."""

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)

    def forward_passthrough(self, x):
        """(self(x), alias of x for a residual connection around the branch): see ops.LinearFn."""
        return ops.linear_passthrough(x, self.weight, self.bias)

    def forward_residual(self, x, residual):
        """self(x) + residual with the add in the product's epilogue."""
        return ops.linear_residual(x, self.weight, self.bias, residual)


class DropPath(nn.Module):
    """Stochastic depth per sample (the reference imports timm's; identity when p == 0 or in eval)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def sample_scale(self, batch, device):
        """Per-sample factor (0 or 1/keep) of one application, or None when inactive; the fused kernels take it as a [B] vector."""
        if self.drop_prob == 0.0 or not self.training:
            return None
        keep = 1.0 - self.drop_prob
        return torch.empty(batch, dtype=torch.float32, device=device).bernoulli_(keep).div_(keep)

    def forward(self, x):
        rs = self.sample_scale(x.shape[0], x.device)
        if rs is None:
            return x
        return x * rs.to(x.dtype).view((x.shape[0],) + (1,) * (x.dim() - 1))


class Mlp(nn.Module):
    """fc1 -> GELU(erf) -> drop -> fc2 -> drop (ref :21-44)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = HSLinear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = HSLinear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x, apply_out_drop=True, residual_alias=False, residual=None):
        """residual_alias: also return an alias of x whose gradient is folded into fc1's input-gradient GEMM.
        residual: added to the output inside fc2's epilogue (ops.RESID_EPILOGUE path; no output dropout then)."""
        exact_gelu = isinstance(self.act, nn.GELU) and getattr(self.act, "approximate", "none") == "none"
        if exact_gelu and x.dtype in (torch.bfloat16, torch.float32) and x.is_cuda:
            # one autograd node: GELU (+ hidden dropout) ride on the GEMM epilogues, masks regenerated in backward (ops.MlpFn)
            out = ops.mlp(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias,
                          drop_p=self.drop.p if self.training else 0.0, passthrough=residual_alias, residual=residual)
            y, x_res = out if residual_alias else (out, None)
        elif residual is not None:
            return self.fc2(self.drop(self.act(self.fc1(x)))) + residual
        else:
            x_res = None
            if residual_alias:
                h, x_res = self.fc1.forward_passthrough(x)
            else:
                h = self.fc1(x)
            y = self.fc2(self.drop(self.act(h)))
        y = self.drop(y) if apply_out_drop else y  # the caller fuses the output dropout into the next norm kernel
        return (y, x_res) if residual_alias else y


# ----------------------------------------------------------------------------- attention
def _labels_from_dense_mask(mask):
    """Recover per-position region labels from a {0, x} [nW, Ws, Ws] mask built by get_attn_mask_from_mask."""
    m = mask.detach().cpu()
    nW, Ws, _ = m.shape
    same = m == 0
    labels = same.to(torch.uint8).argmax(dim=2).to(torch.uint8)  # first position in the same region
    rebuilt = (labels[:, :, None] != labels[:, None, :]).to(m.dtype) * (-100)
    if not torch.equal(rebuilt.to(torch.float32), m.to(torch.float32)):
        raise NotImplementedError("WindowAttention mask is not a {0,-100} region mask; dense masks are not supported")
    return labels.reshape(-1).contiguous()


class WindowAttention(nn.Module):
    """Window multi-head self-attention with relative position bias (ref :47-174)."""

    def __init__(self, dim, window_size, num_heads, rel_pos_bias=None, qkv_bias=True, qk_scale=None,
                 attn_drop=0.0, proj_drop=0.0, use_cos_attn=False):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.use_cos_attn = use_cos_attn
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.rel_pos_bias = rel_pos_bias
        if use_cos_attn:  # ref :84-87
            self.logit_scale = nn.Parameter(torch.log(10 * torch.ones((num_heads, 1, 1))), requires_grad=True)
        if rel_pos_bias == "flat":  # ref :89-114; the table starts at zero (:92-96, :121)
            side = int(round(window_size ** 0.5))
            self.relative_position_bias_table = nn.Parameter(torch.zeros(((2 * side - 1) ** 2, num_heads)))
            rel = _lib.rel_pos_index(window_size)
            self.register_buffer("relative_position_index", torch.from_numpy(rel))
            self.register_buffer("_rel_idx32", torch.from_numpy(rel.astype(np.int32).reshape(-1)), persistent=False)
        self._scale_cache = None  # constant per-head scale tensor of the non-cosine variant, built once per device
        self.qkv = HSLinear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = HSLinear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def extra_repr(self):
        return f"dim={self.dim}, window_size={self.window_size}, num_heads={self.num_heads}"

    def head_scale(self):
        """Per-head multiplier of the raw scores: exp(min(logit_scale, ln 100)) (ref :144-147) or the qk scale."""
        if self.use_cos_attn:
            now = self.__dict__.get("_scale_now")  # made for all blocks at once by the model's forward (SwinHPTransformerSys._prefetch_attn_params)
            if now is not None:
                return now
            if self.logit_scale.is_cuda and self.logit_scale.dtype == torch.float32:
                return ops.cos_head_scale(self.logit_scale)  # one launch forward, one backward
            return torch.exp(torch.clamp(self.logit_scale, max=math.log(1.0 / 0.01))).reshape(-1)
        dev = self.qkv.weight.device
        if self._scale_cache is None or self._scale_cache.device != dev:
            self._scale_cache = torch.full((self.num_heads,), float(self.scale), dtype=torch.float32, device=dev)
        return self._scale_cache

    def bias(self):
        if self.rel_pos_bias is None:
            return None
        now = self.__dict__.get("_bias_now")  # made for all blocks at once by the model's forward
        if now is not None:
            return now
        return ops.RelPosBiasFn.apply(self.relative_position_bias_table, self._rel_idx32, self.window_size)

    def attend(self, x, window_size, idx, roll, labels, apply_proj_drop=True, residual_alias=False, residual=None):
        """x: [B, N, C] in natural order -> attention branch output [B, N, C] in natural order
        (residual_alias: also an alias of x whose gradient is folded into the qkv input-gradient GEMM)."""
        drop = self.attn_drop.p if self.training else 0.0  # dropout on the attention probabilities (ref :169), in-kernel
        if self.rel_pos_bias is not None and window_size != self.window_size:
            raise AssertionError("relative position bias needs input_resolution >= window_size")  # ref quirk :243-251
        x_res = None
        if self.fusable(x, window_size):
            # no-grad forward of a stage whose qkv weights fit the LDS: qkv -> attention -> proj in ONE kernel
            y = self.fused_module(x, window_size, idx, roll, labels)
            return (y, x) if residual_alias else y
        if residual is None and self.trainable_fused(x, window_size):
            # training forward of a stage whose qkv weights fit the LDS, branch without a norm in front (v2 placement): qkv ->
            # attention -> proj in ONE launch that also writes what the backward reads (ops.window_attn_module_train)
            return ops.window_attn_module_train(x, None, None, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias,
                                                self.bias(), self.head_scale(), idx, roll, labels, self.num_heads, window_size,
                                                self.use_cos_attn, residual_alias=residual_alias)
        if residual_alias:
            qkv, x_res = self.qkv.forward_passthrough(x)
        else:
            qkv = self.qkv(x)
        o = ops.window_attn_core(qkv, self.bias(), self.head_scale(), idx, roll, labels, self.num_heads, window_size,
                                 self.use_cos_attn, attn_drop=drop)
        if residual is not None:  # x + proj(o) from the proj product's epilogue (ops.RESID_EPILOGUE; no proj dropout on this path)
            return self.proj.forward_residual(o, residual)
        y = self.proj(o)
        y = self.proj_drop(y) if apply_proj_drop else y  # the caller fuses proj_drop into the next norm kernel
        return (y, x_res) if residual_alias else y

    def fusable(self, x, window_size):
        """The one-launch inference kernel applies: no gradient wanted, supported shape, nothing stochastic switched on."""
        return (ops.window_attn_module_ok(x, self.num_heads, window_size) and
                not (self.training and (self.attn_drop.p > 0 or self.proj_drop.p > 0)) and
                (self.rel_pos_bias is None or window_size == self.window_size))

    def fused_module(self, x, window_size, idx, roll, labels, norm=None, residual=False):
        """[x +] proj(attention(qkv([norm](x)))) by `hs_window_attn_module_fwd` (inference only; ops.window_attn_module_ok)."""
        return ops.window_attn_module(x, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias, self.bias(),
                                      self.head_scale(), idx, roll, labels, self.num_heads, window_size, self.use_cos_attn,
                                      ln_weight=None if norm is None else norm.weight, ln_bias=None if norm is None else norm.bias,
                                      residual=residual)

    def trainable_fused(self, x, window_size):
        """The one-launch TRAINING form applies (ops.window_attn_module_train_ok): gradients wanted, supported shape, nothing
        stochastic inside the branch."""
        return (ops.window_attn_module_train_ok(x, self.num_heads, window_size) and
                not (self.training and (self.attn_drop.p > 0 or self.proj_drop.p > 0)) and
                (self.rel_pos_bias is None or window_size == self.window_size))

    def fused_module_train(self, x, window_size, idx, roll, labels, norm, norm2=None):
        """x + proj(attention(qkv(norm(x)))) by `hs_window_attn_module_fwd_train`, differentiable (ops.window_attn_module_train);
        with norm2 the same launch also applies the block's second LayerNorm: returns (norm2(x1), x1)."""
        return ops.window_attn_module_train(x, norm.weight, norm.bias, self.qkv.weight, self.qkv.bias, self.proj.weight,
                                            self.proj.bias, self.bias(), self.head_scale(), idx, roll, labels, self.num_heads,
                                            window_size, self.use_cos_attn,
                                            norm2=None if norm2 is None else (norm2.weight, norm2.bias))

    def forward(self, x, mask=None):
        """Reference-compatible entry: x [num_windows*B, Ws, C], mask [nW, Ws, Ws] in {0,-100} or None."""
        B_, Ws, C = x.shape
        if mask is None:
            return self.attend(x.reshape(1, B_ * Ws, C), Ws, None, 0, None).reshape(B_, Ws, C)
        nW = mask.shape[0]
        labels = _labels_from_dense_mask(mask).to(x.device)
        return self.attend(x.reshape(B_ // nW, nW * Ws, C), Ws, None, 0, labels).reshape(B_, Ws, C)


class SwinTransformerBlock(nn.Module):
    """One (shifted-)window block (ref :193-340)."""

    def __init__(self, dim, input_resolution, base_pix, num_heads, window_size=4, shift_size=0, shift_strategy="nest_roll",
                 rel_pos_bias=None, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, use_v2_norm_placement=False, use_cos_attn=False):
        super().__init__()
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.window_size, self.shift_size, self.mlp_ratio = window_size, shift_size, mlp_ratio
        self.use_v2_norm_placement = use_v2_norm_placement
        if input_resolution <= window_size:  # a single window: no partition, no shift (ref :243-246)
            self.shift_size, self.window_size = 0, input_resolution

        self.norm1 = _make_norm(norm_layer, dim)
        # as in the reference the attention module is built with the UNclamped window_size (ref :249-251)
        self.attn = WindowAttention(dim, window_size=window_size, num_heads=num_heads, rel_pos_bias=rel_pos_bias,
                                    qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop,
                                    use_cos_attn=use_cos_attn)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = _make_norm(norm_layer, dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

        nside = math.sqrt(input_resolution // base_pix)
        assert nside % 1 == 0, "nside has to be an integer in every layer"
        nside = int(nside)
        if self.shift_size > 0:
            if shift_strategy == "nest_roll":
                self.shifter = hp_shifting.NestRollShift(self.shift_size, self.input_resolution, self.window_size)
            elif shift_strategy == "nest_grid_shift":
                self.shifter = hp_shifting.NestGridShift(nside, base_pix, self.window_size)
            elif shift_strategy == "ring_shift":
                self.shifter = hp_shifting.RingShift(nside, base_pix, self.window_size, self.shift_size)
            else:
                raise KeyError(shift_strategy)
        else:
            self.shifter = hp_shifting.NoShift()
        self._is_roll = isinstance(self.shifter, hp_shifting.NestRollShift)
        self._shifted = self.shift_size > 0

        # reference keeps a dense [nW, Ws, Ws] `attn_mask` buffer in the state dict (:306-308); here it is
        # virtual: emitted by state_dict(), accepted (and dropped) by load_state_dict().
        self.register_buffer("attn_mask", None)
        self._register_state_dict_hook(SwinTransformerBlock._emit_attn_mask)
        self._register_load_state_dict_pre_hook(self._accept_attn_mask, with_module=False)

    @staticmethod
    def _emit_attn_mask(module, state_dict, prefix, local_metadata):
        if EMIT_REFERENCE_BUFFERS and module._shifted:
            state_dict[prefix + "attn_mask"] = module.shifter.get_mask()

    def _accept_attn_mask(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        state_dict.pop(prefix + "attn_mask", None)

    def extra_repr(self):
        return (f"dim={self.dim}, input_resolution={self.input_resolution}, num_heads={self.num_heads}, "
                f"window_size={self.window_size}, shift_size={self.shift_size}, mlp_ratio={self.mlp_ratio}")

    def _attention_branch(self, x, apply_proj_drop=True, residual_alias=False, residual=None):
        if not self._shifted:
            return self.attn.attend(x, self.window_size, None, 0, None, apply_proj_drop, residual_alias, residual)
        idx, _, labels = self.shifter.tables(x.device)
        if self._is_roll:  # modular offset instead of a table
            return self.attn.attend(x, self.window_size, None, self.shift_size % x.shape[1], labels, apply_proj_drop, residual_alias,
                                    residual)
        return self.attn.attend(x, self.window_size, idx, 0, labels, apply_proj_drop, residual_alias, residual)

    def _stochastic(self):
        """Any dropout / DropPath on the two residual branches active right now?"""
        return self.training and ((isinstance(self.drop_path, DropPath) and self.drop_path.drop_prob > 0) or
                                  self.attn.proj_drop.p > 0 or self.mlp.drop.p > 0)

    def _hs_norms(self):
        return isinstance(self.norm1, HSLayerNorm) and isinstance(self.norm2, HSLayerNorm)

    def _path_scale(self, x):
        """per-sample DropPath factor of one branch ([B] tensor) or None"""
        return self.drop_path.sample_scale(x.shape[0], x.device) if isinstance(self.drop_path, DropPath) else None

    def _fused_mlp(self, x1, post_norm=False):
        """x1 + mlp(norm2(x1)) -- post_norm (v2 placement): x1 + norm2(mlp(x1)) -- by the one-launch Mlp block kernel
        (ops.fused_mlp_block) where it applies: HIP norms, exact GELU, hidden = 4 C at C = 96 / 128, nothing stochastic on the
        branch; else None."""
        m = self.mlp
        if not (isinstance(self.norm2, HSLayerNorm) and isinstance(m.fc1, HSLinear) and isinstance(m.fc2, HSLinear) and
                isinstance(m.act, nn.GELU) and getattr(m.act, "approximate", "none") == "none" and
                ops.fused_mlp_ok(x1, m.fc1.weight.shape[0]) and m.fc2.weight.shape[0] == self.dim):
            return None
        rs = self._path_scale(x1) if self.training else None
        dp = m.drop.p if self.training else 0.0
        if rs is None and not dp:
            return ops.fused_mlp_block(x1, self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias,
                                       post_norm=post_norm)
        if ops.fused_mlp_stochastic_ok(x1, post_norm):  # Mlp.drop (both sites) and DropPath inside the launch (v2 placement)
            return ops.fused_mlp_block(x1, self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias,
                                       post_norm=True, row_scale=rs, drop_p=dp)
        return None

    def can_defer(self):
        """v1 placement with the HIP norms: both residual adds -- together with the dropout / DropPath on the added branch --
        ride on the LayerNorm kernel that consumes the sum."""
        return not self.use_v2_norm_placement and self._hs_norms()

    def _shift_args(self, x):
        if not self._shifted:
            return None, 0, None
        idx, _, labels = self.shifter.tables(x.device)
        if self._is_roll:
            return None, self.shift_size % x.shape[1], labels
        return idx, 0, labels

    def forward_deferred(self, x, pending, x_lo=None, comp=None):
        """v1 block on the input `x (+ x_lo) + rs*drop(p)` for pending = (p, rs, drop_p) (or None); returns (x1, pending', x1_lo)
        with the block output = x1 (+ x1_lo) + rs'*drop(m) (ref :337-338 and :316 of the next block).  x_lo / x1_lo are the
        rounding remainders of the compensated residual stream (bf16 runs; None when off or not yet started)."""
        train = self.training
        comp = (ops.COMP_RESIDUAL if comp is None else comp) and x.dtype == torch.bfloat16
        if self.attn.fusable(x, self.window_size) and not (train and isinstance(self.drop_path, DropPath) and self.drop_path.drop_prob > 0):
            # no-grad forward: x1 = xs + proj(attn(qkv(norm1(xs)))) is ONE launch (norm1 as the kernel's prologue, the
            # residual add as its epilogue), xs = x + previous block's MLP branch
            xs = self.resolve_pending(x, pending)
            idx, roll, labels = self._shift_args(xs)
            x1 = self.attn.fused_module(xs, self.window_size, idx, roll, labels, norm=self.norm1, residual=True)
            x2 = self._fused_mlp(x1)  # ... and x2 = x1 + mlp(norm2(x1)) as a second one
            if x2 is not None:
                return x2, None, None
            m = self.mlp(self.norm2(x1), apply_out_drop=False)
            return x1, (m, None, 0.0), None
        if ops.RESID_EPILOGUE and x.dtype == torch.bfloat16 and not comp and x_lo is None and not self._stochastic():
            # a residual add leaves its branch's last product's epilogue WHERE that product runs on hs_gemm_nt anyway (the
            # HBM-bound shapes of stages 0-1); the norm behind it is then a plain LayerNorm whose second output is an alias of its
            # input (the alias' gradient is added inside the LayerNorm backward kernel).  Elsewhere the add stays in the norm kernel.
            C = self.dim
            proj_own = ops.own_gemm_ok(_lib.HS_EPI_BIAS, C, C, x.dtype, m=x.numel() // C)
            fc2_own = ops.own_gemm_ok(_lib.HS_EPI_BIAS, C, self.mlp.fc1.weight.shape[0], x.dtype, m=x.numel() // C)
            if pending is None and self.attn.trainable_fused(x, self.window_size):
                # norm1 -> qkv -> attention -> proj -> residual add in ONE launch that also writes what the backward reads
                idx, roll, labels = self._shift_args(x)
                # ... and the block's norm2 on the sum it has just formed (ops.FUSED_NORM2)
                if ops.FUSED_NORM2:
                    n2, x1 = self.attn.fused_module_train(x, self.window_size, idx, roll, labels, self.norm1, norm2=self.norm2)
                else:
                    x1 = self.attn.fused_module_train(x, self.window_size, idx, roll, labels, self.norm1)
                    x2 = self._fused_mlp(x1)  # norm2 -> fc1 -> GELU -> fc2 -> residual add in one launch as well
                    if x2 is not None:
                        return x2, None, None
                    n2, x1 = ops.layer_norm_passthrough(x1, self.norm2.weight, self.norm2.bias)
                if fc2_own:
                    return self.mlp(n2, apply_out_drop=False, residual=x1), None, None
                return x1, (self.mlp(n2, apply_out_drop=False), None, 0.0), None
            if proj_own or fc2_own:
                if pending is None:
                    n1, xs = ops.layer_norm_passthrough(x, self.norm1.weight, self.norm1.bias)
                else:
                    t, rs, dp = pending
                    xs, n1 = ops.add_layer_norm(x, t, self.norm1.weight, self.norm1.bias, row_scale=rs, drop_p=dp)
                if proj_own:
                    x1 = self._attention_branch(n1, apply_proj_drop=False, residual=xs)
                    x2 = self._fused_mlp(x1)
                    if x2 is not None:
                        return x2, None, None
                    n2, x1 = ops.layer_norm_passthrough(x1, self.norm2.weight, self.norm2.bias)
                else:
                    x1, n2 = ops.add_layer_norm(xs, self._attention_branch(n1, apply_proj_drop=False), self.norm2.weight, self.norm2.bias)
                if fc2_own:
                    return self.mlp(n2, apply_out_drop=False, residual=x1), None, None
                return x1, (self.mlp(n2, apply_out_drop=False), None, 0.0), None
        if pending is None:  # x feeds norm1 AND the residual add below: the alias keeps the two gradients in one kernel
            n1, x = ops.layer_norm_passthrough(x, self.norm1.weight, self.norm1.bias)
        elif comp:
            t, rs, dp = pending
            x, n1, x_lo = ops.add_layer_norm_stream(x, x_lo, t, self.norm1.weight, self.norm1.bias, row_scale=rs, drop_p=dp)
        else:
            t, rs, dp = pending
            x, n1 = ops.add_layer_norm(x, t, self.norm1.weight, self.norm1.bias, row_scale=rs, drop_p=dp)
        a = self._attention_branch(n1, apply_proj_drop=False)
        if comp:
            x1, n2, x1_lo = ops.add_layer_norm_stream(x, x_lo, a, self.norm2.weight, self.norm2.bias, row_scale=self._path_scale(x),
                                                      drop_p=self.attn.proj_drop.p if train else 0.0)
        else:
            x1_lo = None
            x1, n2 = ops.add_layer_norm(x, a, self.norm2.weight, self.norm2.bias, row_scale=self._path_scale(x),
                                        drop_p=self.attn.proj_drop.p if train else 0.0)
        m = self.mlp(n2, apply_out_drop=False)
        return x1, (m, self._path_scale(x), self.mlp.drop.p if train else 0.0), x1_lo

    def can_stream_v2(self):
        """v2 placement with the HIP norms: the block's two `x + norm(branch)` results are the residual stream itself."""
        return self.use_v2_norm_placement and self._hs_norms()

    def forward_stream_v2(self, x, x_lo=None):
        """v2 block (ref :334-335) on the compensated stream x (+ x_lo); returns (x', x'_lo)."""
        train = self.training
        a, xr = self._attention_branch(x, apply_proj_drop=False, residual_alias=True)
        x, x_lo = ops.layer_norm_stream(a, self.norm1.weight, self.norm1.bias, xr, res_lo=x_lo, row_scale=self._path_scale(x),
                                        drop_p=self.attn.proj_drop.p if train else 0.0)
        m, xr = self.mlp(x, apply_out_drop=False, residual_alias=True)
        return ops.layer_norm_stream(m, self.norm2.weight, self.norm2.bias, xr, res_lo=x_lo, row_scale=self._path_scale(x),
                                     drop_p=self.mlp.drop.p if train else 0.0)

    @staticmethod
    def resolve_pending(x, pending):
        """x + rs*drop(p): the standalone form of a deferred residual (end of a stage)."""
        if pending is None:
            return x
        t, rs, dp = pending
        vec = 8 if t.dtype == torch.bfloat16 else 4
        if (dp or rs is not None) and t.is_cuda and t.dtype in (torch.bfloat16, torch.float32) and (t.numel() // t.shape[0]) % vec == 0:
            return ops.residual_drop(x, t, rs, dp)  # dropout, DropPath and the add in one HIP pass
        if dp:
            t = F.dropout(t, dp, True)
        if rs is not None:
            t = t * rs.to(t.dtype).view(-1, 1, 1)
        return x + t

    def forward(self, x):
        B, N, C = x.shape
        assert N == self.input_resolution, f"expected {self.input_resolution} tokens, got {N}"
        if self.can_defer():
            return self.resolve_pending(*self.forward_deferred(x, None)[:2])
        train = self.training
        if self.use_v2_norm_placement and self._hs_norms():  # ref :334-335: x + drop_path(norm(branch)), fused per branch
            # the residual operand is the alias handed back by the branch's first Linear: its gradient is then added inside
            # that Linear's input-gradient GEMM, not by a separate elementwise kernel
            a, xr = self._attention_branch(x, apply_proj_drop=False, residual_alias=True)
            x = self.norm1(a, residual=xr, row_scale=self._path_scale(x), drop_p=self.attn.proj_drop.p if train else 0.0)
            x2 = self._fused_mlp(x, post_norm=True)  # x + norm2(mlp(x)) in one launch where the Mlp is HBM-bound (stage 0)
            if x2 is not None:
                return x2
            m, xr = self.mlp(x, apply_out_drop=False, residual_alias=True)
            return self.norm2(m, residual=xr, row_scale=self._path_scale(x), drop_p=self.mlp.drop.p if train else 0.0)
        if self.use_v2_norm_placement:  # foreign norm layers
            x = x + self.drop_path(self.norm1(self._attention_branch(x)))
            return x + self.drop_path(self.norm2(self.mlp(x)))
        # ref :315-316, :337-338
        x = x + self.drop_path(self._attention_branch(self.norm1(x)))
        return x + self.drop_path(self.mlp(self.norm2(x)))


# ----------------------------------------------------------------------------- resolution changes
class PatchMerging(nn.Module):
    """4 sibling pixels (consecutive in nested order) -> one token: view [B, N/4, 4C] -> LN(4C) -> Linear(4C -> 2C)
    (ref :364-395; the strided slices + cat there are exactly this view)."""

    def __init__(self, dim, dim_scale=2, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.reduction = HSLinear(4 * dim, dim_scale * dim, bias=False)
        self.norm = _make_norm(norm_layer, 4 * dim)

    def forward(self, x):
        B, N, C = x.shape
        assert N % 4 == 0, f"x size {N} is not divisible by 4 as necessary for patching."
        return self.reduction(self.norm(x.reshape(B, N // 4, 4 * C)))


class PatchExpand(nn.Module):
    """Linear(C -> 2C) then every token becomes 4 children of C/2 channels, LN over each child (ref :407-430)."""

    def __init__(self, dim, dim_scale=2, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.expand = HSLinear(dim, dim_scale * dim, bias=False) if dim_scale != 1 else nn.Identity()
        self.norm = _make_norm(norm_layer, dim * dim_scale // 4)

    def forward(self, x):
        x = self.expand(x)
        B, N, C = x.shape
        return self.norm(x.reshape(B, N * 4, C // 4))  # 'b n (p c) -> b (n p) c' is a view in nested order


class FinalPatchExpand_X4(nn.Module):
    """Linear(C -> p*C), p children per token, LN(C) (ref :433-452)."""

    def __init__(self, patch_size, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim, self.patch_size, self.output_dim = dim, patch_size, dim
        self.expand = HSLinear(dim, patch_size * dim, bias=False)
        self.norm = _make_norm(norm_layer, dim)

    def forward(self, x):
        x = self.expand(x)
        B, N, C = x.shape
        return self.norm(x.reshape(B, N * self.patch_size, C // self.patch_size))


# ----------------------------------------------------------------------------- stages
def _build_blocks(dim, input_resolution, depth, num_heads, window_size, base_pix, shift_size, shift_strategy, rel_pos_bias,
                  mlp_ratio, qkv_bias, qk_scale, drop, attn_drop, drop_path, norm_layer, use_v2_norm_placement, use_cos_attn):
    return nn.ModuleList([
        SwinTransformerBlock(dim=dim, input_resolution=input_resolution, base_pix=base_pix, num_heads=num_heads,
                             window_size=window_size, shift_size=shift_size if i % 2 else 0,  # odd blocks shifted (ref :516)
                             shift_strategy=shift_strategy, rel_pos_bias=rel_pos_bias, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                             qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                             drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path, norm_layer=norm_layer,
                             use_v2_norm_placement=use_v2_norm_placement, use_cos_attn=use_cos_attn)
        for i in range(depth)
    ])


class _Stage(nn.Module):
    def _run_blocks(self, x):
        pending = None  # second residual branch of the previous block, added inside the next block's first LayerNorm
        x_lo = None     # rounding remainder of the residual stream (compensated bf16 stream, ops.COMP_RESIDUAL); dropped at the
        #                 end of the stage (one ordinary rounding): every tensor that leaves a stage is a plain activation
        for blk in self.blocks:
            if self.use_checkpoint:
                x, pending, x_lo = SwinTransformerBlock.resolve_pending(x, pending), None, None
                x = checkpoint.checkpoint(blk, x, use_reentrant=False)
            elif blk.can_defer():
                x, pending, x_lo = blk.forward_deferred(x, pending, x_lo, comp=self._comp())
            elif blk.can_stream_v2() and self._comp() and x.dtype == torch.bfloat16 and x.is_cuda:
                x, x_lo = blk.forward_stream_v2(x, x_lo)
            else:
                x, pending, x_lo = SwinTransformerBlock.resolve_pending(x, pending), None, None
                x = blk(x)
        return SwinTransformerBlock.resolve_pending(x, pending)

    comp_residual = None  # None: follow ops.COMP_RESIDUAL; True / False: this stage's own setting (see UnetDecoder)

    def _comp(self):
        return ops.COMP_RESIDUAL if self.comp_residual is None else bool(self.comp_residual)

    def extra_repr(self):
        return f"dim={self.dim}, input_resolution={self.input_resolution}, depth={self.depth}"


class BasicLayer(_Stage):
    """Encoder stage: blocks + optional PatchMerging (ref :455-547)."""

    def __init__(self, dim, input_resolution, depth, num_heads, window_size, base_pix, shift_size, shift_strategy, rel_pos_bias,
                 mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm,
                 downsample=None, use_checkpoint=False, use_v2_norm_placement=False, use_cos_attn=False):
        super().__init__()
        self.dim, self.input_resolution, self.depth, self.use_checkpoint = dim, input_resolution, depth, use_checkpoint
        self.blocks = _build_blocks(dim, input_resolution, depth, num_heads, window_size, base_pix, shift_size, shift_strategy,
                                    rel_pos_bias, mlp_ratio, qkv_bias, qk_scale, drop, attn_drop, drop_path, norm_layer,
                                    use_v2_norm_placement, use_cos_attn)
        self.downsample = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def forward(self, x):
        x = self._run_blocks(x)
        return x if self.downsample is None else self.downsample(x)


class BasicLayer_up(_Stage):
    """Decoder stage: blocks + optional PatchExpand (ref :561-653)."""

    def __init__(self, dim, input_resolution, depth, num_heads, window_size, base_pix, shift_size, shift_strategy, rel_pos_bias,
                 mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm,
                 upsample=None, use_checkpoint=False, use_v2_norm_placement=False, use_cos_attn=False):
        super().__init__()
        self.dim, self.input_resolution, self.depth, self.use_checkpoint = dim, input_resolution, depth, use_checkpoint
        self.blocks = _build_blocks(dim, input_resolution, depth, num_heads, window_size, base_pix, shift_size, shift_strategy,
                                    rel_pos_bias, mlp_ratio, qkv_bias, qk_scale, drop, attn_drop, drop_path, norm_layer,
                                    use_v2_norm_placement, use_cos_attn)
        self.upsample = PatchExpand(dim=dim, dim_scale=2, norm_layer=norm_layer) if upsample is not None else None

    def forward(self, x):
        x = self._run_blocks(x)
        return x if self.upsample is None else self.upsample(x)


class PatchEmbed(nn.Module):
    """`patch_size` consecutive nested pixels -> one token (ref :656-694).  The Conv1d(k = s = patch) parameters are kept
    (state-dict layout [C, f_in, patch]); the arithmetic is a per-patch linear map."""

    def __init__(self, config, data_spec):
        super().__init__()
        assert config.patch_size % 4 == 0, "required for valid nside in deeper layers"
        self.config, self.data_spec = config, data_spec
        self.num_patches = data_spec.dim_in // config.patch_size
        self.proj = nn.Conv1d(data_spec.f_in, config.embed_dim, kernel_size=config.patch_size, stride=config.patch_size)
        # reference quirk (:681-684): the config VALUE is stored, not an instance; only None is usable
        self.norm = config.patch_embed_norm_layer if config.patch_embed_norm_layer is not None else None

    def forward(self, x):
        B, C, N = x.shape
        assert N == self.data_spec.dim_in, f"Input image size ({N}) doesn't match model ({self.data_spec.dim_in})."
        P = self.config.patch_size
        patches = x.reshape(B, C, N // P, P).permute(0, 2, 1, 3).reshape(B, N // P, C * P)
        w = self.proj.weight.reshape(self.proj.weight.shape[0], C * P)
        if x.is_cuda and x.dtype == torch.bfloat16:
            # RGB x 4 pixels = 12 input features = 24-byte rows: zero-padded to 16 so that the product and, above all, its weight
            # gradient (a 128 x 12 output reduced over every token: 1.27 ms in the library, 0.09 ms in hs_linear_wgrad) run in
            # the HIP kernels (16-byte operand rows)
            pad = (-C * P) % 8
            if pad:
                patches, w = F.pad(patches, (0, pad)), F.pad(w, (0, pad))
            x = ops.linear(patches, w, self.proj.bias)
        else:
            x = F.linear(patches, w.to(x.dtype), self.proj.bias.to(x.dtype))
        return x if self.norm is None else self.norm(x)


class UnetDecoder(nn.Module):
    """Expanding path with skip connections (ref :704-791)."""

    def __init__(self, config, data_spec, dpr):
        super().__init__()
        self.config = config
        L = self.num_layers = len(config.depths)
        self.num_features = int(config.embed_dim * 2 ** (L - 1))
        num_patches = data_spec.dim_in // config.patch_size
        self.layers_up = nn.ModuleList()
        self.concat_back_dim = nn.ModuleList()
        for i_layer in range(L):
            down = L - 1 - i_layer
            width = int(config.embed_dim * 2 ** down)
            if i_layer == 0:
                self.concat_back_dim.append(nn.Identity())
                self.layers_up.append(PatchExpand(dim=width, dim_scale=2, norm_layer=config.norm_layer))
                continue
            self.concat_back_dim.append(HSLinear(2 * width, width))
            lo = sum(config.depths[:down])
            self.layers_up.append(BasicLayer_up(
                dim=width, input_resolution=num_patches // (4 ** down), depth=config.depths[down],
                num_heads=config.num_heads[down], window_size=config.window_size, base_pix=data_spec.base_pix,
                shift_size=config.shift_size, shift_strategy=config.shift_strategy, rel_pos_bias=config.rel_pos_bias,
                mlp_ratio=config.mlp_ratio, qkv_bias=config.qkv_bias, qk_scale=config.qk_scale,
                use_cos_attn=config.use_cos_attn, drop=config.drop_rate, attn_drop=config.attn_drop_rate,
                drop_path=dpr[lo:lo + config.depths[down]], norm_layer=config.norm_layer,
                use_v2_norm_placement=config.use_v2_norm_placement, upsample=PatchExpand if down > 0 else None,
                use_checkpoint=config.use_checkpoint))
        if ops.COMP_RESIDUAL_LAST_STAGE and isinstance(self.layers_up[-1], BasicLayer_up):
            # the last decoder stage feeds the tail directly: its residual stream is carried as hi + lo (ops.COMP_RESIDUAL_LAST_STAGE)
            self.layers_up[-1].comp_residual = True
        self.up = FinalPatchExpand_X4(patch_size=config.patch_size, dim=config.embed_dim)
        self.output = nn.Conv1d(in_channels=config.embed_dim, out_channels=data_spec.f_out, kernel_size=1, bias=False)
        self.norm_up = _make_norm(config.norm_layer, config.embed_dim)

    def forward(self, x, x_downsample, ce=None):
        """ce = (labels u8 [B, Npix], class weights or None): return the weighted cross-entropy of the logits instead of the logits
        (SwinHPTransformerSys.forward_seg_loss); fused into the tail kernels where they apply."""
        dbg = self.config.dev_mode
        for inx, layer_up in enumerate(self.layers_up):
            if inx > 0:
                lin = self.concat_back_dim[inx]  # Linear(2c -> c) on cat([x, skip]) (ref :772-775), without the concat copy
                x = ops.concat_linear(x, x_downsample[self.num_layers - 1 - inx], lin.weight, lin.bias)
            x = layer_up(x)
            if dbg:
                print(f"feature shape after decoder layer {inx}: {x.size()}")
        w = self.output.weight  # 1x1 conv without bias (ref :756-761) as the [f_out, C] matrix it is (ops.LinearFn)
        f_out = w.shape[0]
        up = self.up
        if (isinstance(up.norm, HSLayerNorm) and isinstance(up.expand, HSLinear) and up.expand.bias is None and
                ops.expand_ln_head_ok(x, up.dim, up.patch_size, f_out)):
            # the whole tail in one forward kernel (hs_expand_ln_head_fwd): expand -> view -> LayerNorm -> head with fp32 statistics
            # on the expand product's accumulators; the [B, Npix, C] tensor is written once for the backward, or not at all
            xn_lo = None
            if isinstance(self.norm_up, HSLayerNorm):  # norm_up output as hi + lo: no rounding between norm_up and the logits
                xn, xn_lo = ops.layer_norm_hilo(x, self.norm_up.weight, self.norm_up.bias)
            else:
                xn = self.norm_up(x)
            B, N0, _ = xn.shape
            if ce is not None and ce[0].dtype == torch.uint8 and torch.is_grad_enabled():
                # training: expand -> LayerNorm -> head -> weighted CE in one forward kernel; the logits are never written
                return ops.expand_ln_head_ce(xn.reshape(B * N0, up.dim), up.expand.weight, up.norm.weight, up.norm.bias, w,
                                             ce[0].contiguous(), ce[1], xn_lo)
            lg = ops.expand_ln_head(xn.reshape(B * N0, up.dim), up.expand.weight, up.norm.weight, up.norm.bias, w, xn_lo)
            return self._maybe_loss(ops.pad_slice(lg.view(B, N0 * up.patch_size, -1), f_out).transpose(1, 2), ce)  # B, f_out, Npix (fp32)
        if isinstance(up.norm, HSLayerNorm) and ops.ln_head_ok(x, up.dim, f_out):
            # the tail's LayerNorm and the class head in one pass over the expanded rows (hs_ln_head_*): the normalised
            # [B, Npix, C] tensor is neither written nor kept for the backward
            x = up.expand(self.norm_up(x))  # B, N0, p * C: row (b, n) holds the p children of token n back to back
            B, N0, _ = x.shape
            x = ops.ln_head(x.reshape(B * N0 * up.patch_size, up.dim), up.norm.weight, up.norm.bias, w)
            return self._maybe_loss(ops.pad_slice(x.view(B, N0 * up.patch_size, -1), f_out).transpose(1, 2), ce)  # B, f_out, Npix (fp32)
        x = up(self.norm_up(x))  # B, Npix, C
        if x.dtype == torch.bfloat16 and f_out % 8 and f_out > 8:
            # 12 classes: rows padded to 16 so that the input gradient (K = 12 -> 16) runs in hs_gemm_nt: 0.33 ms instead of the
            # library's 0.85 ms; the caller sees the [.., :f_out] view (the loss kernels read logits through their strides)
            x = ops.pad_slice(ops.linear(x, F.pad(w.reshape(f_out, -1), (0, 0, 0, (-f_out) % 8))), f_out)
        else:
            x = ops.linear(x, w)
        return self._maybe_loss(x.float().transpose(1, 2), ce)  # B, f_out, Npix; logits leave the model in fp32 whatever the compute dtype (see ops.LnHeadFn)

    @staticmethod
    def _maybe_loss(logits, ce):
        if ce is None:
            return logits
        from ..losses import seg_loss
        return seg_loss(logits, ce[0], ce[1])


@dataclass
class SwinHPTransformerConfig:
    """Same 23 fields and defaults as the reference config (ref :794-818)."""

    patch_size: int = 4
    window_size: int = 4
    shift_size: int = 2
    shift_strategy: Literal["nest_roll", "nest_grid_shift", "ring_shift"] = "nest_roll"
    rel_pos_bias: Optional[Literal["flat"]] = None
    embed_dim: int = 96
    patch_embed_norm_layer: Optional[Literal[nn.LayerNorm]] = None
    depths: List[int] = field(default_factory=lambda: [2, 2, 2, 2])
    num_heads: List[int] = field(default_factory=lambda: [3, 6, 12, 24])
    mlp_ratio: float = 4.0
    qkv_bias: bool = True
    qk_scale: Optional[float] = None
    use_cos_attn: bool = False
    drop_rate: float = 0.0
    attn_drop_rate: float = 0.0
    drop_path_rate: float = 0.1
    norm_layer: Literal[nn.LayerNorm] = nn.LayerNorm
    use_v2_norm_placement: bool = False
    ape: bool = False
    patch_norm: bool = True
    use_checkpoint: bool = False
    dev_mode: bool = False
    decoder_class: Literal[UnetDecoder] = UnetDecoder


class SwinHPTransformerSys(nn.Module):
    """HEAL-SWIN-UNet: forward(x[B, f_in, Npix]) -> [B, f_out, Npix] (ref :821-955)."""

    def __init__(self, config: SwinHPTransformerConfig, data_spec: DataSpec, **kwargs):
        super().__init__()
        self.config, self.data_spec = config, data_spec
        L = self.num_layers = len(config.depths)
        self.num_features = int(config.embed_dim * 2 ** (L - 1))
        self.num_features_up = int(config.embed_dim * 2)
        self.compute_dtype = kwargs.pop("compute_dtype", None)  # None: follow autocast, else the input dtype

        self.patch_embed = PatchEmbed(config, data_spec=data_spec)
        num_patches = self.patch_embed.num_patches
        if config.ape:
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, num_patches, config.embed_dim))
            nn.init.trunc_normal_(self.absolute_pos_embed, std=0.02)
        self.pos_drop = nn.Dropout(p=config.drop_rate)
        dpr = [v.item() for v in torch.linspace(0, config.drop_path_rate, sum(config.depths))]  # ref :871-873

        self.layers = nn.ModuleList()
        for i in range(L):
            lo = sum(config.depths[:i])
            self.layers.append(BasicLayer(
                dim=int(config.embed_dim * 2 ** i), input_resolution=num_patches // (4 ** i), depth=config.depths[i],
                num_heads=config.num_heads[i], window_size=config.window_size, base_pix=data_spec.base_pix,
                shift_size=config.shift_size, shift_strategy=config.shift_strategy, rel_pos_bias=config.rel_pos_bias,
                mlp_ratio=config.mlp_ratio, qkv_bias=config.qkv_bias, qk_scale=config.qk_scale,
                use_cos_attn=config.use_cos_attn, drop=config.drop_rate, attn_drop=config.attn_drop_rate,
                drop_path=dpr[lo:lo + config.depths[i]], norm_layer=config.norm_layer,
                use_v2_norm_placement=config.use_v2_norm_placement,
                downsample=PatchMerging if i < L - 1 else None, use_checkpoint=config.use_checkpoint))
        # a reference config object names the reference's own UnetDecoder class: use this package's counterpart
        decoder_cls = config.decoder_class
        if getattr(decoder_cls, "__name__", "") == "UnetDecoder":
            decoder_cls = UnetDecoder
        self.decoder = decoder_cls(config, data_spec, dpr)
        self.norm = _make_norm(config.norm_layer, self.num_features)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):  # ref :912-919 (Conv1d layers keep the torch default init)
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"absolute_pos_embed"}

    @torch.jit.ignore
    def no_weight_decay_keywords(self):
        return {"relative_position_bias_table"}

    def _activation_dtype(self, x):
        if self.compute_dtype is not None:
            return self.compute_dtype
        if torch.is_autocast_enabled():
            return torch.get_autocast_gpu_dtype()
        return x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32

    def _prefetch_attn_params(self):
        """The relative-position bias tiles and the cosine score scales of ALL blocks in one launch each (per window size), handed to
        the blocks for this forward; the backward scatters / differentiates them in one launch each as well."""
        sink = ops.RT.grad_sink
        if not ops.BATCH_ATTN_PARAMS or self.config.use_checkpoint or (sink is not None and getattr(sink, "world", 1) > 1):
            # (a checkpointed block recomputes its forward later, outside this call: it must see the same per-block tensors both times;
            # under data parallelism the batched backward would report every block's table gradient at the END of the pass, and no
            # gradient bucket could start its exchange before that: the per-block nodes keep the exchange overlapped with the backward)
            return
        mods = self.__dict__.get("_attn_mods")
        if mods is None:
            mods = self.__dict__["_attn_mods"] = [m for m in self.modules() if isinstance(m, WindowAttention)]
        by_ws = {}
        for m in mods:
            t = getattr(m, "relative_position_bias_table", None)
            if m.rel_pos_bias == "flat" and t is not None and t.is_cuda and t.dtype == torch.float32:
                by_ws.setdefault((m.window_size, t.shape[0]), []).append(m)
        for (ws, _), group in by_ws.items():
            if len(group) > 1:
                for m, b in zip(group, ops.rel_pos_bias_many(group[0]._rel_idx32, ws, [m.relative_position_bias_table for m in group])):
                    m.__dict__["_bias_now"] = b
        cos = [m for m in mods if m.use_cos_attn and m.logit_scale.is_cuda and m.logit_scale.dtype == torch.float32 and m.num_heads <= 64]
        if len(cos) > 1:
            for m, sc in zip(cos, ops.cos_head_scale_many([m.logit_scale for m in cos])):
                m.__dict__["_scale_now"] = sc

    def _clear_attn_params(self):
        for m in self.__dict__.get("_attn_mods") or ():
            m.__dict__.pop("_bias_now", None)
            m.__dict__.pop("_scale_now", None)

    def forward_features(self, x):
        x = self.patch_embed(x)
        if self.config.ape:
            x = x + self.absolute_pos_embed.to(x.dtype)
        x = self.pos_drop(x)
        x_downsample = []
        for k, layer in enumerate(self.layers):
            x_downsample.append(x)  # the INPUT of encoder stage k is the skip tensor (ref :939-941)
            x = layer(x)
            if self.config.dev_mode:
                print(f"feature shape after basic layer {k}: {x.size()}")
        return self.norm(x), x_downsample

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("SwinHPTransformerSys (heal_swin_amd) runs only on an MI355X (HIP) device; there is no CPU path")
        dt = self._activation_dtype(x)
        prev, ops.RT.cast_cache = ops.RT.cast_cache, self._param_casts(dt)
        ops.RT.last_cast_cache = ops.RT.cast_cache
        try:
            with torch.autocast(device_type="cuda", enabled=False):
                self._prefetch_attn_params()
                x, x_downsample = self.forward_features(x.to(dt))
                return self.decoder(x, x_downsample)
        finally:
            self._clear_attn_params()
            ops.RT.cast_cache = prev

    def forward_seg_loss(self, x, labels, class_weights=None):
        """nn.CrossEntropyLoss(weight=class_weights)(self(x), labels.long()) -- the segmentation caller's training step
        (models_lightning/segmentation/model_lightning_swin_hp.py:39-45, :104-111) -- as ONE call, so that the loss rides on the
        decoder tail's kernels: in bf16 training the [B, f_out, Npix] logits and their gradient are never written (SURVEY 8f N2;
        csrc/expand_ln_head.hip, csrc/ln_head.hip).  Where the fused tail does not apply (fp32, other widths, no gradient) this is
        exactly losses.seg_loss(self(x), labels, class_weights).  labels: [B, Npix] integer class ids."""
        if not x.is_cuda:
            raise RuntimeError("SwinHPTransformerSys (heal_swin_amd) runs only on an MI355X (HIP) device; there is no CPU path")
        if labels.dtype != torch.uint8 and self.data_spec.f_out <= 255:
            # the kernels read one byte per pixel and ignore ids >= f_out.  A plain cast would WRAP (256 -> class 0, and
            # CrossEntropyLoss' ignore_index -100 -> 156): out-of-range ids are mapped to 255 (ignored) before narrowing
            labels = torch.where((labels < 0) | (labels > 254), 255, labels).to(torch.uint8)
        w = None if class_weights is None else class_weights.to(device=x.device, dtype=torch.float32).contiguous()
        dt = self._activation_dtype(x)
        prev, ops.RT.cast_cache = ops.RT.cast_cache, self._param_casts(dt)
        ops.RT.last_cast_cache = ops.RT.cast_cache
        try:
            with torch.autocast(device_type="cuda", enabled=False):
                self._prefetch_attn_params()
                x, x_downsample = self.forward_features(x.to(dt))
                return self.decoder(x, x_downsample, ce=(labels, w))
        finally:
            self._clear_attn_params()
            ops.RT.cast_cache = prev

    def _param_casts(self, dt):
        """bf16 copies of the Linear parameters, re-made in one multi-tensor kernel after each optimizer step (ops.ParamCastCache)."""
        ops.note_forward(torch.is_grad_enabled())
        if dt == torch.float32:
            return None
        cache = self.__dict__.get("_cast_cache")
        params = [p for m in self.modules() if isinstance(m, HSLinear) for p in (m.weight, m.bias) if p is not None]
        if (cache is None or cache.dtype != dt or len(cache.params) != len(params) or any(a is not b for a, b in zip(cache.params, params))
                or any(sh.device != p.device for sh, p in zip(cache.shadows[:1], params[:1]))):
            cache = ops.ParamCastCache(params, dt, shadow_of=self.__dict__.get("_shadow_provider"))  # (optim.FlatAdam)
            self.__dict__["_cast_cache"] = cache  # not a module attribute: stays out of state_dict / .to()
        cache.refresh(force=torch.is_grad_enabled())  # (fused optimizers do not bump parameter versions: see ops.ParamCastCache)
        return cache

    def invalidate_param_casts(self):
        """Force the bf16 parameter copies to be re-made at the next forward (needed only after in-place writes through
        `param.data`, which PyTorch's version counters do not see; see ops.ParamCastCache)."""
        cache = self.__dict__.get("_cast_cache")
        if cache is not None:
            cache.invalidate()
