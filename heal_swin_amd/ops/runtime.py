"""Process-wide state of the op layer: runtime switches shared by the ops (`RT`), live kernel timing for bench.py, the deferred
parameter-gradient sums (hs_reduce_flush), the side stream for weight gradients and the bf16 / transposed parameter copies."""
import contextlib
import ctypes
import os
import weakref

import torch

from .. import _lib
from .._lib import check, lib, ptr, stream_ptr


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
