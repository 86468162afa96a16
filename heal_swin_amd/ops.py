"""torch.autograd.Function wrappers around the C ABI of libhealswin.so.

PyTorch owns the memory (caching allocator), the stream and the autograd graph; every arithmetic step
below runs in the HIP library.  All ops require CUDA(HIP) tensors and raise otherwise -- there is no CPU path.
"""
import contextlib
import ctypes
import math
import os
import weakref

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr


# Optional live kernel timing (bench.py): when KERNEL_TIMINGS is a list, the attention launches are bracketed
# with events recorded on the stream the kernel runs on, and (tag, start, end, algorithmic_bytes, flops) is appended.
KERNEL_TIMINGS = None
TIMED_PREFIXES = None  # None: every tagged launch; else only tags starting with one of these (each bracket costs ~3 us of stream time)


class RuntimeState:
    """The process-wide switches the ops consult, in ONE object (`ops.RT`) instead of six module globals:
      grad_sink          gradient sink with flat fp32 buckets (a parallel.GradBucketAllReduce) that kernels ADD parameter gradients into, or None
      async_wgrad        an AsyncWgrad (side stream for the weight-gradient kernels), or None
      cast_cache         the ParamCastCache of the model forward that is running (set by SwinHPTransformerSys.forward), or None
      last_cast_cache    that of the most recent forward: what a backward falls back to when its node kept none
      prefer_own_gemm    every legal bf16 Linear product on hs_gemm_nt (set while CUs are reserved for a communication library)
      zero_padded_grads  data_ptr -> zero-padded gradient buffer written by losses.seg_loss' backward (weak values, see PadSliceFn)
      weight_epoch       generation counter of "the parameters may have changed" for caches that cannot rely on `_version`
    One training setup per process is the supported configuration (as with DistributedDataParallel); `scoped` swaps fields for the
    duration of a block and restores them, which is how nested / temporary configurations should be expressed."""

    def __init__(self):
        self.grad_sink = None
        self.async_wgrad = None
        self.cast_cache = None
        self.last_cast_cache = None
        self.prefer_own_gemm = False
        self.zero_padded_grads = weakref.WeakValueDictionary()
        self.weight_epoch = 0         # moves with every grad-enabled model forward (parameters may have been stepped): _weight_split
        self._epoch_dirty = False

    @contextlib.contextmanager
    def scoped(self, **fields):
        prev = {k: getattr(self, k) for k in fields}
        for k, v in fields.items():
            setattr(self, k, v)
        try:
            yield self
        finally:
            for k, v in prev.items():
                setattr(self, k, v)


RT = RuntimeState()


def note_forward(grad_enabled):
    """Called by the model at the start of every forward: a grad-enabled forward (a training step: an optimizer step follows, and
    fused optimizers do not bump `_version`) and the first no-grad forward after one open a new weight epoch."""
    if grad_enabled or RT._epoch_dirty:
        RT.weight_epoch += 1
    RT._epoch_dirty = bool(grad_enabled)


class _timed:
    def __init__(self, tag, device, nbytes, flops):
        self.on = KERNEL_TIMINGS is not None and (TIMED_PREFIXES is None or tag.startswith(TIMED_PREFIXES))
        if self.on:
            self.tag, self.nbytes, self.flops = tag, nbytes, flops
            self.start = torch.cuda.Event(enable_timing=True)
            self.end = torch.cuda.Event(enable_timing=True)
            self.stream = torch.cuda.current_stream(device)

    def __enter__(self):
        if self.on:
            self.start.record(self.stream)
        return self

    def __exit__(self, *exc):
        if self.on:
            self.end.record(self.stream)
            KERNEL_TIMINGS.append((self.tag, self.start, self.end, self.nbytes, self.flops))
        return False


def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "heal_swin_amd ops run only on an MI355X (HIP) device: got a CPU tensor. "
                "There is no CPU fallback; move the model and inputs to 'cuda'."
            )


def _f32(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


def _draw_seed():
    """64-bit seed from torch's CPU generator (repeatable under torch.manual_seed)."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


# ----------------------------------------------------------------------------- rel-pos bias
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
def _extras(x, row_scale, drop_p, seed):
    """(row_scale fp32 or None, rows_per_sample, drop_p, seed) for the *_drop_* kernels; None if nothing stochastic is on."""
    if row_scale is None and not drop_p:
        return None
    rows = x.numel() // x.shape[-1]
    rs, rps = None, 1
    if row_scale is not None:
        rs = row_scale.detach().to(torch.float32).contiguous()
        assert rows % rs.numel() == 0
        rps = rows // rs.numel()
    if drop_p and seed is None:
        seed = _draw_seed()
    return rs, rps, float(drop_p or 0.0), int(seed or 0)


def _sink_buffer(p):
    """fp32 gradient buffer of parameter p that a kernel may ADD into (a view into RT.grad_sink's flat buckets), or None when no
    sink is installed or p is not registered with it."""
    sink = RT.grad_sink
    return None if (sink is None or p is None) else sink.grad_buffer(p)


# Deferred parameter-gradient reductions (csrc/reduce_many.hip, include/healswin.h: HS_ACC_DEFER).  A kernel that deposits into the
# gradient sink's buffers queues its final "sum the partial records" step instead of launching it; the sink flushes the queue --
# ONE launch for up to 44 sums -- before it exchanges a bucket and at the end of the pass (GradBucketAllReduce._launch / finish).
# The partial records live in the call's workspace, which therefore stays referenced here until the flush.
DEFER_REDUCTIONS = os.environ.get("HS_DEFER_REDUCE", "1") != "0"
_DEFER_KEEP = {}   # stream handle -> workspaces of the queued sums
_DEFER_FLUSH_AT = 32


def _defer_flag(device):
    """HS_ACC_DEFER if a direct-deposit call on `device`'s current stream may queue its reduction, else 0: a sink that flushes is
    installed, and the weight-gradient kernels are not on a side stream."""
    sink = RT.grad_sink
    ok = DEFER_REDUCTIONS and sink is not None and RT.async_wgrad is None and getattr(sink, "flushes_reductions", False)
    return _lib.HS_ACC_DEFER if ok else 0


def _defer_keep(device, *workspaces):
    s = torch.cuda.current_stream(device).cuda_stream
    keep = _DEFER_KEEP.setdefault(s, [])
    keep.extend(workspaces)
    if int(lib.hs_reduce_pending(ctypes.c_void_p(s))) >= _DEFER_FLUSH_AT:
        flush_reductions(device)


def flush_reductions(device=None):
    """Launch every queued parameter-gradient sum of the current stream (of `device`, default: the current device) and release
    the workspaces they read.  Cheap when nothing is queued (no launch)."""
    if not torch.cuda.is_available():
        return
    s = torch.cuda.current_stream(device).cuda_stream
    check(lib.hs_reduce_flush(ctypes.c_void_p(s)), "hs_reduce_flush")
    keep = _DEFER_KEEP.get(s)
    if keep:
        keep.clear()


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
class AsyncWgrad:
    """Opt-in: run the weight/bias-gradient kernels of every Linear on a SIDE stream (their results go straight into the
    RT.grad_sink's buffers).  Nothing on the backward critical path consumes dW, and the wgrad kernels are MFMA work while much
    of the rest of backward (LayerNorm, attention) is HBM-bound, so the two can co-schedule on the chip.
    `sync()` makes the current stream wait for everything enqueued so far (the sink calls it before it exchanges a bucket
    and at the end of the pass)."""

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)

    def sync(self):
        cur = torch.cuda.current_stream(self.stream.device)
        cur.wait_stream(self.stream)


# Direct gradient deposit: an object with `grad_buffer(param) -> fp32 tensor | None` and `deposited(param)` (installed by
# parallel.GradBucketAllReduce).  Linear / LayerNorm parameter gradients of the parameters it knows are accumulated by the
# kernels straight into those buffers and autograd sees no gradient for them (no AccumulateGrad kernels, no dtype round trip).


class ParamCastCache:
    """Activation-dtype copies of the fp32 master parameters of the Linear layers, refreshed for ALL registered parameters by
    one multi-tensor copy instead of one cast kernel per parameter and forward.  When:
      * every GRAD-ENABLED forward (`force`): a training forward is followed by an optimizer step, and the FUSED optimizers
        (`torch.optim.Adam(fused=True)`, what bench.py and the Lightning trainer use) update the parameters WITHOUT bumping their
        `_version` counters (verified: 0 -> 0 across `step()`), so a version check alone left the forward on the bf16 weights of
        step 0 for a whole run (found in round 3 by tests/test_gpu_graphs.py::test_eager_forward_between_replays_...);
      * a no-grad forward after a grad-enabled one (`dirty`), or whenever (data_ptr, _version) of a parameter changed
        (load_state_dict, `param.data = ...`, non-fused optimizers);
      * after `invalidate()` (`model.invalidate_param_casts()`): HIP-graph replays and in-place writes through `param.data`
        (EMA / SWA utilities, manual weight surgery) are invisible to both rules.
    Cost: one read of the fp32 masters and one bf16 write per training step (0.9 GB for HEAL-SWIN-B: ~0.2 ms of a 155 ms step)."""

    always_refresh = False

    def __init__(self, params, dtype, shadow_of=None):
        """shadow_of(p, dtype) -> tensor | None: storage for p's copy owned by someone who keeps it current (optim.FlatAdam writes
        the bf16 parameters from its step kernel and calls mark_refreshed_externally(): no copy pass at the next forward)."""
        self.params = [p for p in params if p.dtype != dtype]
        self.dtype = dtype
        ext = [None if shadow_of is None else shadow_of(p, dtype) for p in self.params]
        self.all_external = bool(ext) and all(e is not None and e.shape == p.shape and e.device == p.device for e, p in zip(ext, self.params))
        self.shadows = ext if self.all_external else [torch.empty_like(p, dtype=dtype) for p in self.params]
        self.external_fresh = False
        self.index = {id(p): i for i, p in enumerate(self.params)}
        self.versions = None
        self.dirty = False    # a grad-enabled forward has run since the last refresh: the parameters are about to change
        self.transposed = {}  # index -> [transposed shadow, the `versions` list object it was made from]
        self._jobs = None     # device table of hs_transpose_many_16 over the entries of `transposed`

    def invalidate(self):
        self.versions = None

    def mark_refreshed_externally(self):
        """The owner of the shadows (optim.FlatAdam.step) has just written every one of them from the updated parameters."""
        if self.all_external:
            self.versions = [(p.data_ptr(), p._version) for p in self.params]  # (a new list: the transposed copies are re-made)
            self.external_fresh = True

    def refresh(self, force=False):
        versions = [(p.data_ptr(), p._version) for p in self.params]
        if self.external_fresh and versions == self.versions and not self.always_refresh:
            self.dirty = False  # the optimizer that steps these parameters keeps the copies current: nothing to do
            return
        if force or self.dirty or self.always_refresh or versions != self.versions:
            with torch.no_grad():
                torch._foreach_copy_(self.shadows, self.params)
            self.versions = versions
        self.dirty = bool(force)

    def get(self, p, dtype):
        i = self.index.get(id(p)) if dtype == self.dtype else None
        return None if i is None else self.shadows[i]

    def current(self, p):
        """The shadow of p still reflects p (p unchanged since the last refresh)."""
        i = self.index.get(id(p))
        return i is not None and self.versions is not None and self.versions[i] == (p.data_ptr(), p._version)

    def get_t(self, p, dtype):
        """[in, out] (transposed) activation-dtype copy of a 2-D weight: the B operand of its input-gradient product in
        `hs_gemm_nt`.  Made on first use; after a refresh ALL copies made so far are re-made together by one launch
        (`hs_transpose_many_16`) the first time any of them is asked for."""
        i = self.index.get(id(p)) if dtype == self.dtype else None
        if i is None:
            return None
        ent = self.transposed.get(i)
        if ent is None:
            t = self.shadows[i].t().contiguous()
            self.transposed[i] = ent = [t, self.versions]
            if self._jobs is not None:  # a captured hs_transpose_many_16 node may still read the old table: never free it
                self.__dict__.setdefault("_retired_jobs", []).append(self._jobs)
            self._jobs = None  # the job table is rebuilt with this entry
        elif ent[1] is not self.versions:
            self._retranspose_all()
        return ent[0]

    def _retranspose_all(self):
        ents = sorted(self.transposed.items())
        if self.dtype.itemsize != 2 or not self.shadows[0].is_cuda:
            for i, ent in ents:
                ent[0].copy_(self.shadows[i].t())
                ent[1] = self.versions
            return
        if getattr(self, "_jobs", None) is None:
            rec = []
            for i, ent in ents:
                rows, cols = self.shadows[i].shape
                rec.append([self.shadows[i].data_ptr(), ent[0].data_ptr(), rows, cols])
            self._jobs = torch.tensor(rec, dtype=torch.int64).to(self.shadows[0].device)
            self._job_blocks = int(min(64, max(1, max((r[2] + 31) // 32 * ((r[3] + 31) // 32) for r in rec))))
        check(lib.hs_transpose_many_16(ptr(self._jobs), len(ents), self._job_blocks, stream_ptr(self.shadows[0].device)),
              "hs_transpose_many_16")
        for _, ent in ents:
            ent[1] = self.versions


# the cache of the most recent forward: the BACKWARD of that forward takes the transposed weight copies from it (the
# parameters have not changed in between: an optimizer step bumps the versions and the next forward refreshes)


def _cast_param(p, dtype):
    if p.dtype == dtype:
        return p
    c = RT.cast_cache.get(p, dtype) if RT.cast_cache is not None else None
    return p.to(dtype) if c is None else c


# ----------------------------------------------------------------------------- hs_gemm_nt (own bf16 GEMM with fused epilogues)
OWN_GEMM = os.environ.get("HS_OWN_GEMM", "auto")  # "auto": per-shape choice below; "0": library GEMMs only; "1": own kernel wherever legal


# Set by parallel.GradBucketAllReduce while compute units are reserved for a co-resident gradient exchange (world > 1): every
# bf16 Linear product then runs on hs_gemm_nt, whose persistent grids honour hs_set_reserved_cus.  The library GEMMs fill all
# 256 CUs and cannot be masked: with 8 foreign workgroups resident they lose 64 % (256 -> 420 us, profiles/r03_cu_contention.json).
# Costs ~3 ms per step on an idle chip (HS_OWN_GEMM=1 measurement of round 3), saves ~27 ms under contention (r04_cu_contention.json).
OWN_GELU_MAX_K = 4096
OWN_DGELU_MAX_K = 1024
# (n, k) -> bool: measured exceptions to the class rule of own_gemm_ok for the bias / residual products, in situ against the TunableOp
# picks (profiles/r05_gemm_shape_table_ab.txt, us per launch lib -> own): T stage-1 qkv 146 -> 122, T stage-2 qkv 103 -> 76, T stage-2
# proj 39 / 43 -> 33, B stage-1 qkv 259 -> 217.  The long reductions of the same stages stay with the library (T fc2 141 vs 160, 101 vs 113).
OWN_SHAPE_TABLE = {(576, 192): True, (1152, 384): True, (384, 384): True, (768, 256): True}
OWN_SHAPE_TABLE_MIN_M = 49152  # measured at m = 65536 ... 393216 rows only
OWN_BIAS_MAX_K = 0  # (> 0 would send every bias / residual product with k <= this to hs_gemm_nt: measured, slower -- profiles/r03_gemm_policy_ab.txt)


def own_gemm_ok(epi, n, k, dtype, k2=0, m=None):
    """Whether `hs_gemm_nt` should run this product (else the library GEMM + the standalone elementwise kernel).
    Measured on MI355X against hipBLASLt on the B / nside 256 / batch 8 shapes (tools/bench_gemm_nt.py,
    profiles/r02_gemm_nt_vs_library.*): the own kernel wins where the product is HBM-bound (short reductions, narrow outputs:
    stages 0-1), ties the untuned hipBLASLt on the K = 512 shapes (and loses to the TunableOp-selected solutions bench.py
    loads) and loses the long reductions (K >= 1024: 0.96-1.06 vs 1.26 PFLOP/s).  A GELU forward epilogue pays while the
    product is HBM-bound (it has to write h AND gelu(h): at K = 512 the 256x256 tile needs 355-368 us against 197 us tuned
    library GEMM + 141 us standalone GELU pass); the GELU-gradient epilogue (reads h, writes once) wins at every stage
    (K = 1024: 265 us against 175-188 us library GEMM + 105 us GELU' pass)."""
    if dtype != torch.bfloat16 or OWN_GEMM == "0" or k % 8 or k2 % 8 or n % 8 or n < 16:
        return False  # (n % 8: whole-row-segment stores; the model pads the 12-class head to 16 rows)
    if OWN_GEMM == "1" or RT.prefer_own_gemm:
        return True
    kk = k + k2
    if epi == _lib.HS_EPI_DGELU:
        return kk <= OWN_DGELU_MAX_K
    if epi == _lib.HS_EPI_GELU:
        return kk <= OWN_GELU_MAX_K
    if OWN_BIAS_MAX_K > 0:
        return kk <= OWN_BIAS_MAX_K
    if m is None or m >= OWN_SHAPE_TABLE_MIN_M:
        pick = OWN_SHAPE_TABLE.get((n, kk))
        if pick is not None:
            return pick
    return kk <= 128 or n <= 128 or (n <= 256 and kk <= 256)


RESID_DGRAD_OWN = True  # (T@256 paper config, same box: 46.2 vs 46.9 ms per step)


def own_gemm_legal(n, k, dtype):
    """Whether `hs_gemm_nt` CAN run an [*, k] x [n, k]^T product (the policy question is own_gemm_ok)."""
    return dtype == torch.bfloat16 and OWN_GEMM != "0" and k % 8 == 0 and n % 8 == 0 and n >= 16


# Residual adds in the GEMM epilogue (v1 blocks without stochastic regularisers): x1 = x + proj(o) and x2 = x1 + fc2(act) leave the
# proj / fc2 product's epilogue (EPI_RESID: acc + bias + residual, ONE rounding), so the LayerNorm that follows is a plain
# LayerNorm (reads 1, writes 1) instead of the fused add + LayerNorm (reads 2, writes 2): 2 of 8 tensor-units per block.
RESID_EPILOGUE = True


def gemm_nt(a2d, w, bias=None, epi=0, aux=None, a2=None, w2=None, want_c=True, drop_p=0.0, seed=0):
    """c = epilogue(a2d @ w^T (+ a2 @ w2^T) + bias) through `hs_gemm_nt`; returns (c, aux).  a2d [m, k] bf16 (row stride free),
    w [n, k] bf16 (row stride free), bias fp32 [n] or None."""
    m, k = a2d.shape
    n = w.shape[0]
    assert a2d.stride(1) == 1 and w.stride(1) == 1 and a2d.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
    c = torch.empty((m, n), dtype=torch.bfloat16, device=a2d.device) if want_c else None
    if epi == _lib.HS_EPI_GELU:
        aux = torch.empty((m, n), dtype=torch.bfloat16, device=a2d.device)
    k2 = 0 if a2 is None else a2.shape[1]
    if bias is not None and bias.dtype != torch.float32:
        bias = bias.float()
    with _timed(f"hs_gemm_nt epi={epi} m={m} n={n} k={k + k2}", a2d.device, 2 * (m * (k + k2) + n * (k + k2) + m * n * (2 if (epi and want_c) else 1)), 2 * m * n * (k + k2)):
        check(lib.hs_gemm_nt(ptr(a2d), a2d.stride(0), ptr(w), w.stride(0), k, ptr(a2), 0 if a2 is None else a2.stride(0), ptr(w2),
                             0 if w2 is None else w2.stride(0), k2, ptr(bias), ptr(c), ptr(aux), m, n, epi, float(drop_p), int(seed),
                             _lib.HS_BF16, stream_ptr(a2d.device)), "hs_gemm_nt")
    return c, aux


def _lib_tag(kind, m, n, k):
    """Tag of a library-GEMM call in KERNEL_TIMINGS (bench.py --kernel-table): which shapes hipBLASLt still runs, and how fast."""
    return f"lib {kind} m={m} n={n} k={k}"


# fp32 activations (the reference's precision): "bf16x3" forms every Linear product as ONE bf16 GEMM of three-fold depth over
# hi / lo splits of both operands with fp32 accumulation (csrc/split3.hip: a_hi b_hi + a_hi b_lo + a_lo b_hi, ~1e-5 relative to the
# fp32 product, 3/16 of the fp32-MFMA time); "strict" keeps exact-fp32 GEMMs (library fp32 GEMM, v_mfma_f32_32x32x2_f32 weight
# gradients) -- the reference form, used by the finite-difference tests.
FP32_GEMM = os.environ.get("HS_FP32_GEMM", "bf16x3")
_BF16X3_MIN = 160  # n k / (n + k) from which a product takes the bf16x3 form (sweep: see _bf16x3_ok)
_SPLIT_MEMO = []  # the last few splits (key, tensor): dy is split once for the input- and the weight-gradient product
_MM_OUT_DTYPE = [None]  # whether torch.mm(..., out_dtype=) is available in this build (probed on first use)


def _bf16x3_ok(x, n=None, k=None):
    """bf16x3 for this fp32 product?  Only where the exact-fp32 product is MFMA-bound: n k / (n + k) >= 160 (e.g. 512 x 256; sweep 64 / 100 /
    128 / 192 / 300 on B@256 fp32: 635 / 625 / 607 / 606 / 632 ms).  Narrow products (C = 96 / 128: stage 0, the whole first stages of HEAL-SWIN-T) move 4 (n + k) bytes per row against
    2 n k flops at 110 TFLOP/s -- they are HBM-bound in fp32 already and the split passes would only add traffic (measured: the T
    depth-head companion 35.9 -> 33.2 images/s with bf16x3 everywhere)."""
    ok = FP32_GEMM == "bf16x3" and x.dtype == torch.float32 and x.is_cuda and x.shape[-1] % 8 == 0
    ok = ok and (n is None or n * k >= _BF16X3_MIN * (n + k))
    if ok and _MM_OUT_DTYPE[0] is None:
        _probe_mm_out_dtype(x.device)
    return ok and _MM_OUT_DTYPE[0] is not False


def _probe_mm_out_dtype(device):
    """bf16x3 needs `torch.mm(bf16, bf16, out_dtype=float32)`; a build without it runs the exact-fp32 products instead (one warning)."""
    try:
        a = torch.zeros((8, 8), dtype=torch.bfloat16, device=device)
        torch.mm(a, a, out_dtype=torch.float32)
        _MM_OUT_DTYPE[0] = True
    except Exception:  # noqa: BLE001  (a build without mm.dtype)
        _MM_OUT_DTYPE[0] = False
        import warnings
        warnings.warn("heal_swin_amd: torch.mm(..., out_dtype=) is unavailable in this PyTorch build; fp32 Linear products run as "
                      "exact fp32 GEMMs (HS_FP32_GEMM=strict behaviour) instead of bf16x3")


def split3(x2d, mode):
    """bf16 [rows, 3 k] = [hi | hi | lo] (mode 0) or [hi | lo | hi] (mode 1) of fp32 x2d [rows, k] (`hs_split_bf16x3`)."""
    x2d = x2d.contiguous()
    key = (x2d.data_ptr(), tuple(x2d.shape), x2d._version, mode)
    for kk, _, t in _SPLIT_MEMO:
        if kk == key:
            return t
    rows, k = x2d.shape
    out = torch.empty((rows, 3 * k), dtype=torch.bfloat16, device=x2d.device)
    check(lib.hs_split_bf16x3(ptr(x2d), ptr(out), rows, k, mode, stream_ptr(x2d.device)), "hs_split_bf16x3")
    if mode == 0:
        # (the entry keeps the SOURCE alive: its address cannot be recycled for another tensor while the key is in the memo)
        _SPLIT_MEMO.append((key, x2d, out))
        del _SPLIT_MEMO[:-2]
    return out


class _Split:
    """An fp32 [rows, k] operand that exists ONLY as its bf16x3 split t3 [rows, 3 k] (`hs_gelu_split3`): stands in for the tensor
    in _lib_linear / _lib_matmul / _param_grads (duck-typed: shape, dtype, device, is_contiguous, record_stream)."""

    def __init__(self, t3, k):
        self.t3, self.shape, self.dtype, self.device = t3, (t3.shape[0], k), torch.float32, t3.device

    def is_contiguous(self):
        return True

    def record_stream(self, stream):
        self.t3.record_stream(stream)

    def dim(self):
        return 2


def _split_of(x2d):
    """The [hi | hi | lo] split of x2d a forward product just made (still in the memo), or None: the Linear keeps it for its weight
    gradient instead of splitting the same activations again in the backward (a third of the fp32 step's split passes)."""
    if isinstance(x2d, _Split):
        return x2d.t3
    if x2d is None or x2d.dtype != torch.float32:
        return None
    key = (x2d.data_ptr(), tuple(x2d.shape), x2d._version, 0)
    for kk, _, t in _SPLIT_MEMO:
        if kk == key:
            return t
    return None


_WSPLIT = {}  # (storage address, storage offset, shape, transposed) -> (version, weight epoch, view of the source, [hi | lo | hi] split)
_WSPLIT_CAPACITY = 1024


def _weight_split(w2d, transposed):
    """The [hi | lo | hi] operand (mode 1) of the fp32 weight w2d [n, k] -- or of its transpose [k, n] -- for the bf16x3 products.
    Cached per weight: callers hand over fresh VIEWS of the parameter (`w.view(n, k)`), so an entry is identified by the storage
    it views (address + offset + shape; the entry keeps a view alive, so the address cannot be recycled while it is cached) and is
    valid while the parameter's `_version` (shared by all its views) AND `RT.weight_epoch` are unchanged.  The epoch moves with
    every grad-enabled model forward, because fused optimizers update parameters WITHOUT bumping `_version` (see ParamCastCache):
    in a training loop every weight is therefore split once per step and direction (forward, transposed for the input gradient);
    evaluation loops, gradient accumulation under no_grad re-forwards and activation checkpointing re-use the cached operand."""
    key = (w2d.untyped_storage().data_ptr(), w2d.storage_offset(), tuple(w2d.shape), tuple(w2d.stride()), bool(transposed))
    hit = _WSPLIT.get(key)
    if hit is not None and hit[0] == w2d._version and hit[1] == RT.weight_epoch:
        return hit[3]
    src = w2d.t().contiguous() if transposed else w2d.contiguous()
    rows, k = src.shape
    out = torch.empty((rows, 3 * k), dtype=torch.bfloat16, device=src.device)
    check(lib.hs_split_bf16x3(ptr(src), ptr(out), rows, k, 1, stream_ptr(src.device)), "hs_split_bf16x3")
    if len(_WSPLIT) >= _WSPLIT_CAPACITY:  # (models come and go in a test session: bounded, oldest entries first)
        for old in list(_WSPLIT)[:_WSPLIT_CAPACITY // 2]:
            del _WSPLIT[old]
    _WSPLIT[key] = (w2d._version, RT.weight_epoch, w2d.detach(), out)
    return out


def _mm_f32(a3, b3t, bias=None):
    """fp32 result of the bf16 product a3 @ b3t (+ bias): hipBLASLt with an fp32 output (`out_dtype`)."""
    if _MM_OUT_DTYPE[0] is None:
        _probe_mm_out_dtype(a3.device)
    if not _MM_OUT_DTYPE[0]:
        raise RuntimeError("HS_FP32_GEMM=bf16x3 needs torch.mm(..., out_dtype=torch.float32); set HS_FP32_GEMM=strict")
    if bias is None:
        return torch.mm(a3, b3t, out_dtype=torch.float32)
    if _MM_OUT_DTYPE[0] is True:  # addend (bias vector or residual matrix, fp32) in the GEMM's epilogue where the build has addmm.dtype
        try:
            y = torch.addmm(bias, a3, b3t, out_dtype=torch.float32)
            _MM_OUT_DTYPE[0] = "addmm"
            return y
        except Exception:  # noqa: BLE001
            _MM_OUT_DTYPE[0] = "mm"
    if _MM_OUT_DTYPE[0] == "addmm":
        return torch.addmm(bias, a3, b3t, out_dtype=torch.float32)
    return torch.mm(a3, b3t, out_dtype=torch.float32).add_(bias)


def _lib_linear(x2, w, b):
    if isinstance(x2, _Split):
        m, k = x2.shape
        with _timed(_lib_tag("fwd bf16x3", m, w.shape[0], 3 * k), x2.device, 4 * (m * k + m * w.shape[0]), 6 * m * k * w.shape[0]):
            return _mm_f32(x2.t3, _weight_split(w.reshape(w.shape[0], k), False).t(), b)
    m, k = x2.shape[0] if x2.dim() == 2 else x2.numel() // x2.shape[-1], x2.shape[-1]
    if _bf16x3_ok(x2, w.shape[0], k) and w.dtype == torch.float32 and w.shape[0] % 8 == 0:
        with _timed(_lib_tag("fwd bf16x3", m, w.shape[0], 3 * k), x2.device, 4 * (m * k + m * w.shape[0]), 6 * m * k * w.shape[0]):
            y = _mm_f32(split3(x2.reshape(m, k), 0), _weight_split(w.reshape(w.shape[0], k), False).t(), b)
        return y.view(x2.shape[:-1] + (w.shape[0],))
    with _timed(_lib_tag("fwd", m, w.shape[0], k), x2.device, 2 * (m * k + m * w.shape[0]), 2 * m * k * w.shape[0]):
        return torch.nn.functional.linear(x2, w, b)


def _lib_matmul(dy2, w, res=None):
    m, n = dy2.shape
    if isinstance(dy2, _Split):
        with _timed(_lib_tag("dgrad bf16x3", m, w.shape[1], 3 * n), dy2.device, 4 * (m * n + m * w.shape[1]), 6 * m * n * w.shape[1]):
            return _mm_f32(dy2.t3, _weight_split(w, True).t(), res)
    if _bf16x3_ok(dy2, w.shape[1], n) and w.dtype == torch.float32 and w.shape[1] % 8 == 0:
        with _timed(_lib_tag("dgrad bf16x3", m, w.shape[1], 3 * n), dy2.device, 4 * (m * n + m * w.shape[1]), 6 * m * n * w.shape[1]):
            dx = _mm_f32(split3(dy2, 0), _weight_split(w, True).t(), res)
        return dx
    with _timed(_lib_tag("dgrad", m, w.shape[1], n), dy2.device, 2 * (m * n + m * w.shape[1]), 2 * m * n * w.shape[1]):
        return dy2 @ w if res is None else torch.addmm(res, dy2, w)


def _cast_param_t(p, dtype, cache=None):
    """[in, out] copy of weight p ([out, in, ...]) in `dtype` for the input-gradient product.  `cache`: the ParamCastCache the
    FORWARD of this autograd node ran under (kept on its ctx, so that several models in one process each take their own
    copies); falls back to the cache of the most recent forward."""
    if cache is None:
        cache = RT.cast_cache if RT.cast_cache is not None else RT.last_cast_cache
    c = cache.get_t(p, dtype) if (cache is not None and p.dim() == 2 and cache.current(p)) else None
    if c is None:
        n_out = p.shape[0]
        c = p.detach().to(dtype).view(n_out, -1).t().contiguous()
    return c


def _param_grads(dy2, x2, weight, bias, want_w, want_b, x3=None, gelu_x=False):
    """Weight / bias gradient of y = x W^T + b from dy2 [rows, n_out], x2 [rows, k_in]: deposited straight into the gradient
    sink's buffers when one knows the parameters (returns (None, None)), else returned in the parameters' dtype.
    gelu_x: the Linear's input was gelu(x2) and only the pre-activation x2 was kept (`hs_linear_wgrad_gelu`)."""
    n_out = weight.shape[0]
    k_in = weight.numel() // n_out
    if not (want_w or want_b):
        return None, None
    if x2 is None:  # (fp32 runs) only the bf16x3 split of the input was kept: the three-product weight gradient reads nothing else
        assert x3 is not None and dy2.dtype == torch.float32
        hip_ok = dy2.is_contiguous() and n_out % 8 == 0
        assert hip_ok
    else:
        hip_ok = (x2.is_contiguous() and dy2.is_contiguous() and n_out % 4 == 0 and
                  ((x2.dtype == torch.bfloat16 and k_in % 8 == 0) or (x2.dtype == torch.float32 and k_in % 4 == 0)))
    wbuf = _sink_buffer(weight) if (hip_ok and want_w) else None
    bbuf = _sink_buffer(bias) if (wbuf is not None and want_b) else None
    if wbuf is not None and (not want_b or bbuf is not None):
        # accumulate dW (and db) straight into the sink's gradient buffers (no autograd AccumulateGrad kernels, no dtype
        # round trip); optionally on the side stream
        aw = RT.async_wgrad
        wbuf = wbuf.view(n_out, k_in)
        if aw is not None:
            cur = torch.cuda.current_stream(dy2.device)
            aw.stream.wait_stream(cur)
            dy2.record_stream(aw.stream)
            (x2 if x2 is not None else x3).record_stream(aw.stream)
            with torch.cuda.stream(aw.stream):
                LinearFn._wgrad_hip(dy2, x2, n_out, k_in, want_b, wbuf, bbuf, x3, gelu_x)
        else:
            LinearFn._wgrad_hip(dy2, x2, n_out, k_in, want_b, wbuf, bbuf, x3, gelu_x)
        RT.grad_sink.deposited(weight)
        if want_b:
            RT.grad_sink.deposited(bias)
        return None, None
    dw = db = None
    if hip_ok:
        dw32, db32 = LinearFn._wgrad_hip(dy2, x2, n_out, k_in, want_b, x3=x3, gelu_x=gelu_x)
        dw = dw32.to(weight.dtype).view(weight.shape) if want_w else None
        db = db32.to(bias.dtype) if want_b else None
    else:  # odd widths: library GEMM
        assert not gelu_x
        if want_w:
            dw = (dy2.t() @ x2).to(weight.dtype).view(weight.shape)
        if want_b:
            db = dy2.sum(0).to(bias.dtype)
    return dw, db


class LinearFn(torch.autograd.Function):
    """y = x W^T + b with fp32 master parameters and activations in x.dtype.
    forward / input gradient: library GEMM; weight + bias gradient: `hs_linear_wgrad` (split over the token axis, fp32
    results straight into the master dtype).  `weight` may carry trailing singleton dimensions (the decoder's 1x1 Conv1d head,
    [f_out, C, 1]): it is used as the [n_out, k_in] matrix it is, so the PARAMETER itself (a leaf) receives the gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, passthrough=False, residual=None, pre=None):
        """pre = (y,): the product the fused module kernel already wrote (window_attn_module_train); nothing is launched."""
        _require_gpu(x, weight, bias, residual)
        n_out = weight.shape[0]
        k_in = weight.numel() // n_out
        w = _cast_param(weight, x.dtype).view(n_out, k_in)
        ctx.x_shape = x.shape
        ctx.bias_param = bias
        ctx.w_cast = w if w.dtype != weight.dtype else None  # activation-dtype copy, reused by the input-gradient GEMM
        ctx.cast_cache = RT.cast_cache
        ctx.passthrough = passthrough
        ctx.has_residual = residual is not None
        ctx.x3 = None
        if pre is not None:
            y = pre[0]
        elif residual is not None:
            # y = x W^T + b + residual: the add rides on the product's epilogue (one rounding); its gradient is dy itself
            assert not passthrough
            if own_gemm_legal(n_out, k_in, x.dtype) and x.is_contiguous():
                res2 = residual.reshape(-1, n_out)
                res2 = res2 if res2.is_contiguous() else res2.contiguous()
                y = gemm_nt(x.reshape(-1, k_in), w, bias, _lib.HS_EPI_RESID, aux=res2)[0].view(x.shape[:-1] + (n_out,))
            else:
                y = _lib_linear(x, w, None if bias is None else _cast_param(bias, x.dtype)) + residual
                ctx.x3 = _split_of(x.reshape(-1, k_in)) if x.is_contiguous() else None
        elif own_gemm_ok(_lib.HS_EPI_BIAS, n_out, k_in, x.dtype, m=x.numel() // k_in) and x.is_contiguous():
            y = gemm_nt(x.reshape(-1, k_in), w, bias)[0].view(x.shape[:-1] + (n_out,))  # fp32 master bias added in the epilogue
        else:
            y = _lib_linear(x, w, None if bias is None else _cast_param(bias, x.dtype))
            ctx.x3 = _split_of(x.reshape(-1, k_in)) if x.is_contiguous() else None
        # (fp32 runs) where the forward product made a bf16x3 split of x, the weight gradient reads THAT (6 bytes per element) and x
        # itself (4) is not kept for it
        ctx.save_for_backward(None if ctx.x3 is not None else x, weight)
        # passthrough: also hand x back (an alias) for the block's residual connection.  The gradient of that second use then
        # arrives HERE together with dy, and the input-gradient GEMM adds it as its beta * C term instead of autograd
        # launching a separate add over the whole activation (v2 norm placement: x + LN(branch(x)), ref :334-335)
        return (y, x.view_as(x)) if passthrough else y

    @staticmethod
    def _wgrad_hip(dy2, x2, n_out, k_in, want_b, dw_out=None, db_out=None, x3=None, gelu_x=False):
        """dW (and db) of one Linear.  With dw_out/db_out (existing fp32 gradient buffers) the result is ADDED there."""
        rows = dy2.shape[0]
        dev = dy2.device
        accumulate = 1 if dw_out is not None else 0
        # deposits into the gradient sink's buffers queue their slice sums (one launch per ~32 layers, ops.flush_reductions)
        defer = _defer_flag(dev) if (dw_out is not None and (db_out is not None or not want_b)) else 0
        dw32 = dw_out if dw_out is not None else torch.empty((n_out, k_in), dtype=torch.float32, device=dev)
        db32 = None
        if want_b:
            db32 = db_out if db_out is not None else torch.empty(n_out, dtype=torch.float32, device=dev)
        nws = int(lib.hs_linear_wgrad_workspace(rows, n_out, k_in))
        ws = torch.empty(nws, dtype=torch.float32, device=dev)
        if x3 is not None or (_bf16x3_ok(x2, n_out, k_in) and n_out % 8 == 0 and x2.dtype == dy2.dtype):
            # dW = dY^T X as three bf16 weight-gradient products over the hi / lo column blocks of the [hi | hi | lo] splits
            # (the split of dY is shared with the input-gradient product): hi^T hi + hi^T lo + lo^T hi; the bias gradient takes
            # the column sums of dY_hi and dY_lo
            dy3 = dy2.t3 if isinstance(dy2, _Split) else split3(dy2, 0)
            x3 = x3 if x3 is not None else split3(x2, 0)
            aw = RT.async_wgrad
            if aw is not None and torch.cuda.current_stream(dev) == aw.stream:
                # the splits were allocated on the main stream (memo / forward) and are read here on the side stream: tell the
                # caching allocator, or a block evicted from the memo could be recycled under the lagging weight-gradient kernels
                dy3.record_stream(aw.stream)
                x3.record_stream(aw.stream)
            with _timed("linear_wgrad bf16x3", dev, 3 * 2 * rows * (n_out + k_in), 6 * rows * n_out * k_in):
                for i, (yo, xo, dbp) in enumerate(((0, 0, db32), (0, 2 * k_in, None), (2 * n_out, 0, db32))):
                    # the three sums share dw32 (two of them db32): jobs of one hs_reduce_flush launch run side by side and would
                    # race on it, so only the LAST product's sum is queued -- the first two land at once, in stream order, and
                    # the workspace is free again when the next product writes it
                    d = defer if i == 2 else 0
                    check(lib.hs_linear_wgrad_ld(ptr(dy3), 3 * n_out, yo, ptr(x3), 3 * k_in, xo, ptr(dw32), ptr(dbp), ptr(ws), rows,
                                                 n_out, k_in, (1 if (accumulate or i) else 0) | d, stream_ptr(dev)), "hs_linear_wgrad_ld")
                if defer:
                    _defer_keep(dev, ws)
            return dw32, db32
        with _timed("linear_wgrad", dev, x2.element_size() * rows * (n_out + k_in), 2 * rows * n_out * k_in):
            if gelu_x:  # dW = dY^T gelu(x2): the activation is applied to the operand fragments inside the kernel
                check(lib.hs_linear_wgrad_gelu(ptr(dy2), ptr(x2), ptr(dw32), ptr(db32), ptr(ws), rows, n_out, k_in, accumulate | defer,
                                               _lib.dtype_code(x2.dtype), stream_ptr(dev)), "hs_linear_wgrad_gelu")
            else:
                check(lib.hs_linear_wgrad(ptr(dy2), ptr(x2), ptr(dw32), ptr(db32), ptr(ws), rows, n_out, k_in, accumulate | defer,
                                          _lib.dtype_code(x2.dtype), stream_ptr(dev)), "hs_linear_wgrad")
        if defer:
            _defer_keep(dev, ws)
        return dw32, db32

    @staticmethod
    def backward(ctx, dy, dx_res=None):
        x, weight = ctx.saved_tensors
        bias = ctx.bias_param
        n_out = weight.shape[0]
        k_in = weight.numel() // n_out
        if dy is None:  # only the passthrough alias was used downstream
            return dx_res, None, None, None, None, None
        dy2 = dy.reshape(-1, n_out)
        x2 = None if x is None else x.reshape(-1, k_in)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _input_grad(dy2, weight, ctx.w_cast, None if dx_res is None else dx_res.reshape(-1, k_in), ctx.cast_cache).reshape(ctx.x_shape)
        ctx.w_cast = ctx.cast_cache = None
        x3, ctx.x3 = ctx.x3, None
        dw, db = _param_grads(dy2, x2, weight, bias, ctx.needs_input_grad[1], bias is not None and ctx.needs_input_grad[2], x3)
        return dx, dw, db, None, (dy if ctx.has_residual else None), None


def _input_grad(dy2, weight, w_cast, dx_res2=None, cache=None):
    """dx = dy2 @ W (+ dx_res2): `hs_gemm_nt` on the transposed weight copy where that wins, else the library GEMM."""
    n_out = weight.shape[0]
    k_in = weight.numel() // n_out
    epi = _lib.HS_EPI_BIAS if dx_res2 is None else _lib.HS_EPI_RESID
    # with a residual-path gradient to add (v2 placement), the library form is torch.addmm(res, dy, W): a device-to-device copy of
    # `res` into the result and THEN the product with beta = 1 -- a whole extra pass (36 copies, ~1 ms per HEAL-SWIN-T @ 256 step);
    # hs_gemm_nt reads the addend in its epilogue instead
    if own_gemm_ok(epi, k_in, n_out, dy2.dtype, m=dy2.shape[0]) or (dx_res2 is not None and RESID_DGRAD_OWN and own_gemm_legal(k_in, n_out, dy2.dtype)):
        res = None if dx_res2 is None else dx_res2.to(dy2.dtype).contiguous()
        return gemm_nt(dy2, _cast_param_t(weight, dy2.dtype, cache), None, epi, aux=res)[0]
    w = w_cast if (w_cast is not None and w_cast.dtype == dy2.dtype) else (
        weight if weight.dtype == dy2.dtype else weight.to(dy2.dtype)).view(n_out, k_in)
    if dx_res2 is not None:
        return _lib_matmul(dy2, w, dx_res2.to(dy2.dtype))
    return _lib_matmul(dy2, w)


def linear(x, weight, bias=None):
    return LinearFn.apply(x, weight, bias)


def linear_residual(x, weight, bias, residual):
    """x W^T + b + residual with the add in the product's epilogue (LinearFn)."""
    return LinearFn.apply(x, weight, bias, False, residual)


def linear_passthrough(x, weight, bias=None):
    """(x W^T + b, alias of x): use the alias for a residual connection around the branch this Linear opens."""
    return LinearFn.apply(x, weight, bias, True)


# data_ptr -> zero-padded gradient buffer written by losses.seg_loss' backward (see PadSliceFn).  WEAK values: an entry exists only
# while the buffer itself is alive (i.e. while autograd still holds the gradient view into it), so nothing is retained when no
# PadSliceFn consumes it, and a recycled address cannot resurrect a dead buffer.


class PadSliceFn(torch.autograd.Function):
    """x[..., :n] of the padded head output.  Backward: when the incoming gradient is the [..., :n] view of a zero-padded
    buffer of x's shape (losses.seg_loss writes its gradient that way), that buffer IS the gradient of x; otherwise the
    gradient is copied into a zeroed buffer, as autograd's slice backward does."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.shape, ctx.n = x.shape, n
        return x[..., :n]

    @staticmethod
    def backward(ctx, g):
        full = RT.zero_padded_grads.pop(g.data_ptr(), None)
        if (full is not None and g._base is full and full.numel() == math.prod(ctx.shape) and full.dtype == g.dtype and
                g.shape == ctx.shape[:-1] + (ctx.n,) and g.stride() == full.view(ctx.shape)[..., :ctx.n].stride()):
            return full.view(ctx.shape), None
        out = g.new_zeros(ctx.shape)
        out[..., :ctx.n] = g
        return out, None


def pad_slice(x, n):
    return PadSliceFn.apply(x, n)


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


def _fold_head_ce(gamma, beta, weight, C, device):
    """The folded head weight for `hs_ln_head_ce_bwd`: as _fold_head, but with row blocks 4..7 and 8..11 exchanged, so that the
    kernel's accumulator register r < 8 of lane half h is class 8 h + r (csrc/ln_head.hip:ln_head_ce_bwd_kernel)."""
    wfold, bvec = _fold_head(gamma, beta, weight, C, device)
    perm = torch.arange(32, device=device)
    perm[4:8], perm[8:12] = torch.arange(8, 12, device=device), torch.arange(4, 8, device=device)
    perm64 = torch.cat([perm, perm + 32])
    return wfold[perm64].contiguous(), bvec[perm].contiguous()


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
