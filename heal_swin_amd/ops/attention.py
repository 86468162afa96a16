"""Window attention: relative-position bias gather, cosine head scales, the attention core (hs_window_attn_fwd / _bwd) and the
one-launch WindowAttention module kernels (swin_hp_transformer.py:47-174)."""
import ctypes
import os

import torch

from .. import _lib
from .._lib import check, lib, ptr, stream_ptr
from .runtime import RT, _cast_param, _f32, _require_gpu, _sink_buffer, _timed  # noqa: F401
from .norm import LayerNormFn  # noqa: F401
from .gemm import LinearFn  # noqa: F401


class RelPosBiasFn(torch.autograd.Function):
    """bias[h,i,j] = table[rel_idx[i,j], h]   (reference swin_hp_transformer.py:152-159)"""

    @staticmethod
    def forward(ctx, table, rel_idx, window_size):
        _require_gpu(table, rel_idx)
        assert rel_idx.dtype == torch.int32 and rel_idx.is_contiguous()
        t = _f32(table)
        rows, nh = t.shape
        bias = torch.empty((nh, window_size, window_size), dtype=torch.float32, device=t.device)
        check(lib.hs_rel_bias_gather(ptr(t), ptr(rel_idx), ptr(bias), rows, nh, window_size, stream_ptr(t.device)),
              "hs_rel_bias_gather")
        ctx.save_for_backward(rel_idx)
        ctx.shape = (rows, nh, window_size)
        ctx.table_dtype = table.dtype
        ctx.table = table if table.dtype == torch.float32 else None
        return bias

    @staticmethod
    def backward(ctx, dbias):
        (rel_idx,) = ctx.saved_tensors
        rows, nh, ws = ctx.shape
        dbias = dbias.to(torch.float32).contiguous()
        order, offsets = _rel_idx_groups(rel_idx, rows)
        buf = _sink_buffer(ctx.table)
        if buf is not None:  # straight into the gradient sink's buffer (no AccumulateGrad add kernel)
            check(lib.hs_rel_bias_scatter_grad_sorted_add(ptr(dbias), ptr(order), ptr(offsets), ptr(buf), rows, nh, ws,
                                                          stream_ptr(dbias.device)), "hs_rel_bias_scatter_grad_sorted_add")
            RT.grad_sink.deposited(ctx.table)
            return None, None, None
        dtable = torch.empty((rows, nh), dtype=torch.float32, device=dbias.device)
        check(lib.hs_rel_bias_scatter_grad_sorted(ptr(dbias), ptr(order), ptr(offsets), ptr(dtable), rows, nh, ws,
                                                  stream_ptr(dbias.device)), "hs_rel_bias_scatter_grad_sorted")
        return dtable.to(ctx.table_dtype), None, None


class CosHeadScaleFn(torch.autograd.Function):
    """exp(min(logit_scale, ln 100)) per head (reference swin_hp_transformer.py:144-147) in one launch, backward in one launch that
    deposits straight into the gradient sink where one is installed (torch: clamp, exp + mul, compare, where, add_)."""

    @staticmethod
    def forward(ctx, logit_scale):
        _require_gpu(logit_scale)
        ls = logit_scale.detach().reshape(-1)
        out = torch.empty_like(ls)
        check(lib.hs_cos_head_scale_fwd(ptr(ls), ptr(out), ls.numel(), stream_ptr(ls.device)), "hs_cos_head_scale_fwd")
        ctx.param = logit_scale
        return out

    @staticmethod
    def backward(ctx, dscale):
        p = ctx.param
        ls = p.detach().reshape(-1)
        dscale = dscale.to(torch.float32).contiguous()
        buf = _sink_buffer(p)
        if buf is not None:
            check(lib.hs_cos_head_scale_bwd(ptr(ls), ptr(dscale), ptr(buf.view(-1)), ls.numel(), 1, stream_ptr(ls.device)), "hs_cos_head_scale_bwd")
            RT.grad_sink.deposited(p)
            return None
        d = torch.empty_like(ls)
        check(lib.hs_cos_head_scale_bwd(ptr(ls), ptr(dscale), ptr(d), ls.numel(), 0, stream_ptr(ls.device)), "hs_cos_head_scale_bwd")
        return d.view(p.shape)


def cos_head_scale(logit_scale):
    return CosHeadScaleFn.apply(logit_scale)


# ---- every attention block of a model in one launch each (the model calls these once per forward and hands the results to its blocks:
# HEAL-SWIN-T at nside 128 is bound by its ~600 launches per step, 2 x 22 (+ 2 x 22 with cosine attention) of which were these)
BATCH_ATTN_PARAMS = True  # (A/B: tools/policy_ab.py BATCH_ATTN_PARAMS=False)


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _int_array(values):
    return (ctypes.c_int * len(values))(*[int(v) for v in values])


class RelPosBiasManyFn(torch.autograd.Function):
    """RelPosBiasFn for a list of tables that share one index (one window size): biases as views of ONE buffer, one gather launch;
    the backward scatters every block's d bias in one launch, straight into the gradient sink where one is installed."""

    @staticmethod
    def forward(ctx, rel_idx, window_size, *tables):
        _require_gpu(rel_idx, *tables)
        assert rel_idx.dtype == torch.int32 and rel_idx.is_contiguous()
        assert all(t.dtype == torch.float32 and t.is_contiguous() and t.shape[0] == tables[0].shape[0] for t in tables)
        rows, heads = tables[0].shape[0], [t.shape[1] for t in tables]
        flat = torch.empty((sum(heads), window_size, window_size), dtype=torch.float32, device=rel_idx.device)
        check(lib.hs_rel_bias_gather_many(_ptr_array([t.detach() for t in tables]), _int_array(heads), len(tables), ptr(rel_idx), ptr(flat), rows,
                                          window_size, stream_ptr(rel_idx.device)), "hs_rel_bias_gather_many")
        ctx.save_for_backward(rel_idx)
        ctx.tables, ctx.meta = tables, (rows, heads, window_size)
        ctx.set_materialize_grads(False)  # a block whose bias takes no gradient costs no job (and gets no zero gradient)
        return tuple(flat.split(heads, 0))

    @staticmethod
    def backward(ctx, *dbiases):
        (rel_idx,) = ctx.saved_tensors
        rows, heads, ws = ctx.meta
        order, offsets = _rel_idx_groups(rel_idx, rows)
        grads = [None] * len(heads)
        src, dst, nh, acc, sunk = [], [], [], [], []
        for j, db in enumerate(dbiases):
            if db is None:
                continue
            db = db.to(torch.float32).contiguous()
            buf = _sink_buffer(ctx.tables[j])
            if buf is not None:
                sunk.append(ctx.tables[j])
                out, a = buf, 1
            else:
                out, a = torch.empty((rows, heads[j]), dtype=torch.float32, device=db.device), 0
                grads[j] = out
            src.append(db), dst.append(out), nh.append(heads[j]), acc.append(a)
        if src:
            check(lib.hs_rel_bias_scatter_grad_sorted_many(_ptr_array(src), _ptr_array(dst), _int_array(nh), _int_array(acc), len(src), ptr(order),
                                                           ptr(offsets), rows, ws, stream_ptr(src[0].device)), "hs_rel_bias_scatter_grad_sorted_many")
        for t in sunk:  # (after the launch: a bucket's exchange may start the moment its last gradient is reported)
            RT.grad_sink.deposited(t)
        return (None, None, *grads)


def rel_pos_bias_many(rel_idx, window_size, tables):
    """[bias_j] with bias_j[h, i, j] = tables[j][rel_idx[i, j], h] -- all blocks in one launch (fp32 tables on the GPU)."""
    return RelPosBiasManyFn.apply(rel_idx, int(window_size), *tables)


class CosHeadScaleManyFn(torch.autograd.Function):
    """CosHeadScaleFn for every cosine-attention block of a model: one launch forward, one backward."""

    @staticmethod
    def forward(ctx, *logit_scales):
        _require_gpu(*logit_scales)
        ls = [p.detach().reshape(-1) for p in logit_scales]
        heads = [t.numel() for t in ls]
        flat = torch.empty(sum(heads), dtype=torch.float32, device=ls[0].device)
        outs = list(flat.split(heads))
        check(lib.hs_cos_head_scale_many(_ptr_array(ls), None, _ptr_array(outs), _int_array(heads), None, len(ls), stream_ptr(flat.device)),
              "hs_cos_head_scale_many")
        ctx.params, ctx.heads = logit_scales, heads
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dscales):
        grads = [None] * len(ctx.heads)
        ls, ds, dst, nh, acc, sunk = [], [], [], [], [], []
        for j, d in enumerate(dscales):
            if d is None:
                continue
            p = ctx.params[j]
            buf = _sink_buffer(p)
            if buf is not None:
                sunk.append(p)
                out, a = buf.view(-1), 1
            else:
                out, a = torch.empty(ctx.heads[j], dtype=torch.float32, device=d.device), 0
                grads[j] = out.view(p.shape)
            ls.append(p.detach().reshape(-1)), ds.append(d.to(torch.float32).contiguous()), dst.append(out), nh.append(ctx.heads[j]), acc.append(a)
        if ls:
            check(lib.hs_cos_head_scale_many(_ptr_array(ls), _ptr_array(ds), _ptr_array(dst), _int_array(nh), _int_array(acc), len(ls),
                                             stream_ptr(ls[0].device)), "hs_cos_head_scale_many (backward)")
        for p in sunk:
            RT.grad_sink.deposited(p)
        return tuple(grads)


def cos_head_scale_many(logit_scales):
    return CosHeadScaleManyFn.apply(*logit_scales)


_REL_IDX_GROUPS = {}


def _rel_idx_groups(rel_idx, rows):
    """(order, offsets) of `hs_rel_bias_scatter_grad_sorted` for an index buffer, built once per buffer (the index is a
    registered buffer of the module: constant)."""
    key = (rel_idx.data_ptr(), rel_idx.numel(), rows, rel_idx.device)
    hit = _REL_IDX_GROUPS.get(key)
    if hit is None:
        flat = rel_idx.flatten().long()
        order = torch.argsort(flat, stable=True).to(torch.int32)
        counts = torch.bincount(flat, minlength=rows)[:rows]
        offsets = torch.zeros(rows + 1, dtype=torch.int32, device=rel_idx.device)
        offsets[1:] = torch.cumsum(counts, 0).to(torch.int32)
        hit = _REL_IDX_GROUPS[key] = (order.contiguous(), offsets, rel_idx)  # (keeps the keyed buffer alive)
    return hit[0], hit[1]


# ----------------------------------------------------------------------------- fused shift + window attention
FORCE_VALU_ATTENTION = False  # tests / A-B runs: route the attention op to the generic fp32-VALU kernels (HS_ATTN_FORCE_VALU)


class WindowAttnCoreFn(torch.autograd.Function):
    """shift -> window_partition -> (cos|scaled) QK^T + bias + mask -> softmax -> @V -> window_reverse -> shift_back
    on the un-shifted qkv tensor (reference swin_hp_transformer.py:319-330 around :136-171)."""

    @staticmethod
    def forward(ctx, qkv, bias, head_scale, idx, roll, labels, num_heads, window_size, cosine, attn_drop=0.0, seed=0, pre=None):
        """pre = (out, lse): results the fused module kernel already wrote (window_attn_module_train); nothing is launched."""
        _require_gpu(qkv, bias, head_scale, idx, labels)
        B, N, C3 = qkv.shape
        C = C3 // 3
        qkv = qkv.contiguous()
        dt = _lib.dtype_code(qkv.dtype)
        hs = _f32(head_scale).reshape(-1)
        assert hs.numel() == num_heads
        bias_c = _f32(bias)
        flags = (_lib.HS_ATTN_COSINE if cosine else 0) | (_lib.HS_ATTN_FORCE_VALU if FORCE_VALU_ATTENTION else 0)
        if pre is not None:
            out, lse = pre
        else:
            out = torch.empty((B, N, C), dtype=qkv.dtype, device=qkv.device)
            need_grad = any(ctx.needs_input_grad[:3])
            lse = torch.empty((B, num_heads, N), dtype=torch.float32, device=qkv.device) if need_grad else None
            # algorithmic traffic: q,k,v read + o written once; flops: QK^T and PV, 2*Ws*hd each per (row, head)
            with _timed("window_attn_fwd", qkv.device, 4 * B * N * C * qkv.element_size(), 4 * B * N * C * window_size):
                check(lib.hs_window_attn_fwd(ptr(qkv), ptr(out), ptr(lse), ptr(bias_c), ptr(hs), ptr(idx), int(roll), ptr(labels),
                                             B, N, C, num_heads, window_size, flags, float(attn_drop), int(seed), dt,
                                             stream_ptr(qkv.device)),
                      "hs_window_attn_fwd")
        ctx.save_for_backward(qkv, out, lse, bias_c, hs, idx, labels)
        ctx.args = (B, N, C, num_heads, window_size, flags, dt, int(roll))
        ctx.drop = (float(attn_drop), int(seed))
        ctx.bias_dtype = None if bias is None else bias.dtype
        ctx.scale_meta = (head_scale.dtype, head_scale.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, bias_c, hs, idx, labels = ctx.saved_tensors
        B, N, C, nh, ws, flags, dt, roll = ctx.args
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        # dbias and dscale are accumulated into by the kernels (C ABI): one zero fill for both
        nb = bias_c.numel() if bias_c is not None else 0
        mfma_bf16 = qkv.dtype == torch.bfloat16 and ws == 64 and C == 32 * nh and not (flags & _lib.HS_ATTN_FORCE_VALU)
        if mfma_bf16:  # that path writes its parameter gradients (HS_ATTN_OVERWRITE_GRADS): no zero fill
            flags |= _lib.HS_ATTN_OVERWRITE_GRADS
            zeros = torch.empty(nb + hs.numel(), dtype=torch.float32, device=qkv.device)
        else:
            zeros = torch.zeros(nb + hs.numel(), dtype=torch.float32, device=qkv.device)
        dbias = zeros[:nb].view(bias_c.shape) if bias_c is not None else None
        dscale = zeros[nb:].view(hs.shape)
        nws = int(lib.hs_window_attn_bwd_workspace(B, N, C, nh, ws, dt))
        wsp = torch.empty(nws, dtype=torch.float32, device=qkv.device) if nws else None
        # algorithmic traffic: qkv (3C) + dout (C) read, dqkv (3C) written -- plus out (C) in the fp32 / VALU kernels; the bf16
        # MFMA kernel forms D = rowsum(P o dP) from its own registers and never reads `out`; flops: 5 contractions of 2*Ws*hd
        streams = 7 if mfma_bf16 else 8
        with _timed("window_attn_bwd", qkv.device, streams * B * N * C * qkv.element_size(), 10 * B * N * C * ws):
            check(lib.hs_window_attn_bwd(ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), ptr(dbias), ptr(dscale), ptr(wsp),
                                         ptr(bias_c), ptr(hs), ptr(idx), roll, ptr(labels),
                                         B, N, C, nh, ws, flags, ctx.drop[0], ctx.drop[1], dt, stream_ptr(qkv.device)),
                  "hs_window_attn_bwd")
        dbias_out = None if dbias is None else dbias.to(ctx.bias_dtype)
        sdt, sshape = ctx.scale_meta
        dscale_out = dscale.to(sdt).reshape(sshape) if (flags & _lib.HS_ATTN_COSINE) else None
        return dqkv, dbias_out, dscale_out, None, None, None, None, None, None, None, None, None


def window_attn_core(qkv, bias, head_scale, idx, roll, labels, num_heads, window_size, cosine, attn_drop=0.0, seed=None):
    """attn_drop > 0 applies the reference's dropout on the attention probabilities; `seed` (64-bit) fixes the mask,
    by default it is drawn from torch's CPU generator (so torch.manual_seed makes runs repeatable)."""
    if attn_drop > 0.0 and seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    return WindowAttnCoreFn.apply(qkv, bias, head_scale, idx, roll, labels, num_heads, window_size, cosine,
                                  float(attn_drop), int(seed or 0))


# Compensated residual stream (bf16 runs): the two residual adds of every block keep their rounding remainder in a second bf16
# tensor that only the next add reads (csrc/layernorm.hip).  Opt-in (HS_COMP_RESIDUAL=1): measured on HEAL-SWIN-B / nside 256 it
# halves the error of the stage outputs (enc.2: 2.6e-2 -> 1.3e-2 of scale) but moves the LOGIT error by only 0-12 % (the decoder
# tail's roundings dominate it, tests/experiments/bf16_error_budget.py) and costs 2.4 % of the step (158.7 -> 162.6 ms).
COMP_RESIDUAL = os.environ.get("HS_COMP_RESIDUAL", "0") == "1"
# the same for the LAST decoder stage only (the two blocks in front of the tail; 2 of 46 blocks of HEAL-SWIN-B): experiment switch
COMP_RESIDUAL_LAST_STAGE = False  # (set by tests / experiments; no environment switch)


FUSED_ATTN_MODULE = True  # the no-grad fused module path (tests flip the attribute to compare with the composition)


def window_attn_module_ok(x, num_heads, window_size):
    """Whether `hs_window_attn_module_fwd` covers this call: no gradient needed, bf16, window 64, head_dim 32, C in {96, 128}."""
    return (FUSED_ATTN_MODULE and x.is_cuda and x.dtype == torch.bfloat16 and not torch.is_grad_enabled() and
            bool(lib.hs_window_attn_module_supported(x.shape[-1], num_heads, window_size, _lib.HS_BF16)))


def window_attn_module(x, qkv_w, qkv_b, proj_w, proj_b, bias, head_scale, idx, roll, labels, num_heads, window_size, cosine,
                       ln_weight=None, ln_bias=None, residual=False):
    """[x +] proj(window_attention(qkv([LayerNorm](x)))) in one launch (inference; see include/healswin.h).  x [B, N, C] bf16 in
    natural order; qkv_w / proj_w in any float dtype (bf16 copies come from the weight cache)."""
    _require_gpu(x, qkv_w, proj_w, bias, head_scale, idx, labels)
    B, N, C = x.shape
    x = x.contiguous()
    out = torch.empty_like(x)
    wq, wp = _cast_param(qkv_w, torch.bfloat16).contiguous(), _cast_param(proj_w, torch.bfloat16).contiguous()
    hs = _f32(head_scale).reshape(-1)
    flags = (_lib.HS_ATTN_COSINE if cosine else 0) | (_lib.HS_ATTN_RESIDUAL if residual else 0)
    # algorithmic traffic: x in, out written (+ x again for the residual); flops: qkv + scores + P V + proj
    nbytes = (3 if residual else 2) * B * N * C * 2
    flops = B * N * (8 * C * C + 4 * window_size * C)
    with _timed("window_attn_module_fwd", x.device, nbytes, flops):
        check(lib.hs_window_attn_module_fwd(ptr(x), ptr(out), ptr(wq), ptr(_f32(qkv_b)), ptr(wp), ptr(_f32(proj_b)), ptr(_f32(ln_weight)),
                                            ptr(_f32(ln_bias)), ptr(_f32(bias)), ptr(hs), ptr(idx), int(roll), ptr(labels), B, N, C,
                                            num_heads, window_size, flags, _lib.HS_BF16, stream_ptr(x.device)),
              "hs_window_attn_module_fwd")
    return out


# The TRAINING form of the module kernel (`hs_window_attn_module_fwd_train`): x + proj(attention(qkv(LayerNorm(x)))) in one launch that
# also writes what the backward reads.  HS_FUSED_ATTN_TRAIN=0 keeps the four-kernel composition (A/B runs).
FUSED_ATTN_MODULE_TRAIN = os.environ.get("HS_FUSED_ATTN_TRAIN", "1") != "0"
# the block's norm2 as that kernel's epilogue: built, parity-tested, time-NEUTRAL on the step (144.3-144.5 ms either way: the standalone
# LayerNorm streams at 4.7 TB/s, the one-wave-per-SIMD module kernel pays about as much for the extra phase) -- off by default
FUSED_NORM2 = False  # superseded: norm2 is now the PROLOGUE of the fused Mlp block (csrc/mlp_fused.hip); kept as a tested kernel option


def window_attn_module_train_ok(x, num_heads, window_size):
    return (FUSED_ATTN_MODULE_TRAIN and x.is_cuda and x.dtype == torch.bfloat16 and torch.is_grad_enabled() and
            not FORCE_VALU_ATTENTION and bool(lib.hs_window_attn_module_supported(x.shape[-1], num_heads, window_size, _lib.HS_BF16)))


def window_attn_module_train(x, ln_weight, ln_bias, qkv_w, qkv_b, proj_w, proj_b, bias, head_scale, idx, roll, labels, num_heads,
                             window_size, cosine, residual_alias=False, norm2=None):
    """x + proj(window_attention(qkv(LayerNorm(x)))) for a block on the training path (reference :315-316 around :124-174).  ONE
    kernel computes it and writes LayerNorm(x) with its statistics, qkv, the attention output and the log-sum-exp rows; the four
    autograd nodes of the composed path (LayerNormFn, LinearFn, WindowAttnCoreFn, LinearFn with the residual) are then recorded
    around those tensors WITHOUT launching anything (`pre=`), so the backward is exactly the composed path's.
    ln_weight None (v2 norm placement, ref :334-335): proj(window_attention(qkv(x))) without norm and residual; with residual_alias
    the call returns (y, alias of x) as `LinearFn`'s passthrough form does (the alias' gradient rides on the qkv input-gradient GEMM).
    norm2 = (weight, bias) of the block's second LayerNorm (v1 placement only): the same launch also writes LayerNorm(out); the
    call then returns (n2, out) as `layer_norm_passthrough(out, ...)` would."""
    _require_gpu(x, qkv_w, proj_w, bias, head_scale, idx, labels)
    B, N, C = x.shape
    x = x.contiguous()
    dev = x.device
    has_ln = ln_weight is not None
    out, o = torch.empty_like(x), torch.empty_like(x)
    xn = torch.empty_like(x) if has_ln else None
    qkv = torch.empty((B, N, 3 * C), dtype=x.dtype, device=dev)
    mean = torch.empty(B * N, dtype=torch.float32, device=dev) if has_ln else None
    rstd = torch.empty(B * N, dtype=torch.float32, device=dev) if has_ln else None
    lse = torch.empty((B, num_heads, N), dtype=torch.float32, device=dev)
    n2 = mean2 = rstd2 = None
    if norm2 is not None:
        assert has_ln and not residual_alias
        n2 = torch.empty_like(x)
        mean2 = torch.empty(B * N, dtype=torch.float32, device=dev)
        rstd2 = torch.empty(B * N, dtype=torch.float32, device=dev)
    wq, wp = _cast_param(qkv_w, torch.bfloat16).contiguous(), _cast_param(proj_w, torch.bfloat16).contiguous()
    hs = _f32(head_scale).reshape(-1)
    flags = (_lib.HS_ATTN_COSINE if cosine else 0) | (_lib.HS_ATTN_RESIDUAL if has_ln else 0)
    # algorithmic traffic: x in (+ again for the residual), out + LayerNorm(x) + qkv + attention output written; flops as the module
    with _timed("window_attn_module_fwd_train", dev, ((9 if has_ln else 6) + (1 if n2 is not None else 0)) * B * N * C * 2,
                B * N * (8 * C * C + 4 * window_size * C)):
        check(lib.hs_window_attn_module_fwd_train(ptr(x), ptr(out), ptr(xn), ptr(mean), ptr(rstd), ptr(qkv), ptr(o), ptr(lse), ptr(wq),
                                                  ptr(_f32(qkv_b)), ptr(wp), ptr(_f32(proj_b)), ptr(_f32(ln_weight)), ptr(_f32(ln_bias)),
                                                  ptr(_f32(bias)), ptr(hs), ptr(idx), int(roll), ptr(labels),
                                                  ptr(None if n2 is None else _f32(norm2[0])), ptr(None if n2 is None else _f32(norm2[1])),
                                                  ptr(n2), ptr(mean2), ptr(rstd2), B, N, C, num_heads,
                                                  window_size, flags, _lib.HS_BF16, stream_ptr(dev)),
              "hs_window_attn_module_fwd_train")
    if not has_ln:
        x_res = None
        if residual_alias:
            qkv_t, x_res = LinearFn.apply(x, qkv_w, qkv_b, True, None, (qkv,))
        else:
            qkv_t = LinearFn.apply(x, qkv_w, qkv_b, False, None, (qkv,))
        o_t = WindowAttnCoreFn.apply(qkv_t, bias, head_scale, idx, roll, labels, num_heads, window_size, cosine, 0.0, 0, (o, lse))
        y = LinearFn.apply(o_t, proj_w, proj_b, False, None, (out,))
        return (y, x_res) if residual_alias else y
    assert not residual_alias
    n1, xs = LayerNormFn.apply(x, ln_weight, ln_bias, None, None, True, None, False, (xn, mean, rstd))
    qkv_t = LinearFn.apply(n1, qkv_w, qkv_b, False, None, (qkv,))
    o_t = WindowAttnCoreFn.apply(qkv_t, bias, head_scale, idx, roll, labels, num_heads, window_size, cosine, 0.0, 0, (o, lse))
    x1 = LinearFn.apply(o_t, proj_w, proj_b, False, xs, (out,))
    if n2 is None:
        return x1
    return LayerNormFn.apply(x1, norm2[0], norm2[1], None, None, True, None, False, (n2, mean2, rstd2))


# ----------------------------------------------------------------------------- row LayerNorm (+ residual, + train-mode extras)
