"""Data-parallel gradient exchange for the HEAL-SWIN train step: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).

The reference's only collective is Lightning-DDP's gradient all-reduce (heal_swin/train.py:182-189).  Here:
  * every parameter's .grad is a VIEW into a few large flat fp32 buckets (no flatten/unflatten copies);
  * buckets are filled in reverse parameter order (the order backward produces gradients) and an async
    all-reduce is launched from a post-accumulate-grad hook the moment a bucket's last gradient lands, so
    the exchange overlaps the rest of backward;
  * bucket size defaults to 64 MiB: xGMI is point-to-point and ring steps are per-link bound, so few large
    messages beat many small ones (298-596 MB per step for the B model).

Step protocol (one optimizer step):

    dp.zero_grad()                      # or optimizer.zero_grad(set_to_none=True): detected, see _begin_pass
    with dp.no_sync():                  # optional gradient accumulation: all micro-batches but the last
        loss_1.backward(); dp.finish()
    loss_k.backward(); dp.finish()      # last micro-batch: buckets are exchanged (sum of the micro-batches, averaged)
    optimizer.step()

A backward pass that would add local gradients on top of already exchanged ones (a second backward() after an
exchanging pass without zeroing in between) raises instead of silently letting the replicas diverge.
"""
import contextlib

import torch
import torch.distributed as dist


def _graph_task_id():
    """Id of the autograd graph task (one per backward() call) this thread is executing, -1 outside one, None if unavailable."""
    fn = getattr(torch._C, "_current_graph_task_id", None)
    return fn() if fn is not None else None


class GradBucketAllReduce:
    def __init__(self, params, bucket_bytes=64 << 20, process_group=None, async_wgrad=False, direct_wgrad=True,
                 exchange_single_rank=False, comm_dtype=None, reserved_cus="auto"):
        """comm_dtype: None / torch.float32 exchanges the fp32 buckets themselves; torch.bfloat16 exchanges a bf16 copy of each
        bucket (half the bytes on the xGMI links: 298 instead of 596 MB per step for HEAL-SWIN-B) -- the gradients are still
        ACCUMULATED in the fp32 buckets (kernels' direct deposit, micro-batches under no_sync()); only the wire format and the
        cross-rank sum are bf16, as with DDP's bf16 compression hook.
        reserved_cus: compute units the library's chip-filling launches leave free for RCCL's kernels while this exchange is
        active ("auto": 16 when more than one rank exchanges, else 0; see include/healswin.h:hs_set_reserved_cus and
        profiles/archive_r01_r04/r03_cu_contention.json)."""
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.async_wgrad = None
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # a one-rank group normally skips the collectives; exchange_single_rank keeps them (exercises the RCCL path on one GPU)
        self._exchange = self.world > 1 or (exchange_single_rank and dist.is_initialized())
        self._avg_in_collective = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.buckets = []       # flat fp32 tensors
        self._counts = []       # parameters per bucket
        self._where = {}        # param -> bucket id
        self._views = {}        # param -> its .grad view into the bucket
        self._build(bucket_bytes)
        self.comm_dtype = None if comm_dtype in (None, torch.float32) else comm_dtype
        self._comm = [torch.empty_like(f, dtype=self.comm_dtype) for f in self.buckets] if self.comm_dtype is not None else None
        self._sync = True       # False inside no_sync()
        self._reduced = False   # a bucket has been exchanged since the gradients were last zeroed
        self._stepped = False   # an attached optimizer has stepped since the last exchange
        self.timeline, self._in_finish = None, False  # record_timeline(): (bucket, event, "hook" | "finish") per exchange of a pass
        self._reset_pass()
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self._direct = False
        on_gpu = bool(self.params) and self.params[0].is_cuda
        self._reserved_prev = None
        self._prefer_prev = None
        if on_gpu:
            from . import _lib
            want = (16 if (self._exchange and self.world > 1) else 0) if reserved_cus == "auto" else int(reserved_cus)
            self._reserved_prev = int(_lib.lib.hs_get_reserved_cus())
            if want != self._reserved_prev:
                _lib.check(_lib.lib.hs_set_reserved_cus(want), "hs_set_reserved_cus")
            if want > 0:  # CUs are reserved for the exchange: keep every GEMM on the kernels that honour the reservation
                from . import ops as _ops
                self._prefer_prev = _ops.RT.prefer_own_gemm
                _ops.RT.prefer_own_gemm = True
        if (direct_wgrad or async_wgrad) and on_gpu:
            # kernels accumulate Linear / LayerNorm parameter gradients straight into the bucket views of the parameters
            # REGISTERED HERE (ops asks grad_buffer(p) per parameter; other models in the process are unaffected)
            from . import ops

            ops.RT.grad_sink = self
            self._direct = True
        if async_wgrad and on_gpu:
            # the Linear weight-gradient kernels additionally run on a side stream (ops.AsyncWgrad)
            from . import ops

            self.async_wgrad = ops.AsyncWgrad(self.params[0].device)
            ops.RT.async_wgrad = self.async_wgrad

    # ------------------------------------------------------------------ construction
    def _build(self, bucket_bytes):
        order = list(reversed(self.params))
        groups, cur, cur_bytes = [], [], 0
        for p in order:
            nbytes = p.numel() * 4
            if cur and (cur_bytes + nbytes > bucket_bytes or cur[0].device != p.device):
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        for b, ps in enumerate(groups):
            flat = torch.zeros(sum(p.numel() for p in ps), dtype=torch.float32, device=ps[0].device)
            off = 0
            for p in ps:
                assert p.dtype == torch.float32, "master parameters are fp32"
                view = flat[off:off + p.numel()].view_as(p)
                p.grad = view
                off += p.numel()
                self._where[p] = b
                self._views[p] = view
            self.buckets.append(flat)
            self._counts.append(len(ps))

    def _reset_pass(self):
        self._pending = list(self._counts)   # per bucket: gradients still missing in this backward pass
        self._launched = [False] * len(self.buckets)
        self._seen = set()                   # parameters already counted in this pass
        self._works = []
        self._pass_open = False
        self._pass_task = None               # autograd graph-task id of the backward() that opened this pass

    # ------------------------------------------------------------------ step protocol
    def reset(self):
        """Forget an unfinished backward pass (e.g. after an exception inside backward()): in-flight exchanges are waited for and
        dropped, the next gradient event opens a fresh pass.  zero_grad() does the same."""
        for _, w in self._works:
            w.wait()
        self._reset_pass()

    def zero_grad(self):
        """Zero the buckets in place and (re-)attach every .grad view.  Also closes a pass left open by an aborted backward()
        (see reset()); restrictions by design: ONE backward() per pass -- nested / re-entrant backwards (reentrant activation
        checkpointing) and `loss1.backward(retain_graph=True); loss2.backward()` need finish() in between or no_sync()."""
        for _, w in getattr(self, "_works", []):
            w.wait()
        if getattr(self, "_direct", False):
            self._flush_reductions()  # (sums an aborted pass left queued must not land on the zeroed buckets)
        for flat in self.buckets:
            flat.zero_()
        for p, view in self._views.items():
            if p.grad is not view:
                p.grad = view
        self._reduced = False
        self._stepped = False
        self._reset_pass()

    @contextlib.contextmanager
    def no_sync(self):
        """Backward passes inside accumulate locally (no exchange), as DistributedDataParallel.no_sync()."""
        prev, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = prev

    @contextlib.contextmanager
    def suspended(self):
        """No direct deposit inside (e.g. around torch.autograd.grad calls that need the parameter gradients returned)."""
        from . import ops
        prev_sink, prev_aw = ops.RT.grad_sink, ops.RT.async_wgrad
        if prev_sink is self:
            ops.RT.grad_sink = None
        if prev_aw is self.async_wgrad:
            ops.RT.async_wgrad = None
        try:
            yield
        finally:
            ops.RT.grad_sink, ops.RT.async_wgrad = prev_sink, prev_aw

    def attach_optimizer(self, optimizer):
        """Lets a step of `optimizer` mark the start of a new iteration, for callers that zero gradients in place with
        optimizer.zero_grad(set_to_none=False) instead of dp.zero_grad()."""
        optimizer.register_step_post_hook(lambda *a, **k: setattr(self, "_stepped", True))
        return optimizer

    def _detached(self, p):
        g = p.grad
        return g is None or g.data_ptr() != self._views[p].data_ptr()

    def _begin_pass(self):
        """First gradient event of a backward pass.  `.grad` of a registered parameter that is None (the PyTorch / Lightning
        default optimizer.zero_grad(set_to_none=True)) or a fresh tensor installed by autograd means "zero so far": the
        view is zeroed, takes over what autograd may already have stored, and is attached again, so the direct-deposit
        path and the flat buckets stay in use whatever way the caller zeroes."""
        det = [p for p in self.params if self._detached(p)]
        if det:
            with torch.no_grad():
                if len(det) == len(self.params):
                    for flat in self.buckets:
                        flat.zero_()
                    self._reduced = False
                else:
                    for p in det:
                        self._views[p].zero_()
                for p in det:
                    if p.grad is not None:
                        self._views[p].copy_(p.grad)
                    p.grad = self._views[p]
        if self._reduced:
            if not self._stepped:
                raise RuntimeError(
                    "GradBucketAllReduce: a backward pass started on gradients that were already exchanged and not zeroed "
                    "since.  For gradient accumulation run all micro-batches but the last under dp.no_sync(); between "
                    "optimizer steps call dp.zero_grad() or optimizer.zero_grad(set_to_none=True) (or attach_optimizer()).")
            self._reduced = False  # the attached optimizer stepped: the caller zeroed in place (as with DDP, not verified)
        self._stepped = False
        self._pass_open = True
        self._pass_task = _graph_task_id()

    def _check_same_backward(self):
        """A gradient event of ANOTHER backward() while the previous pass is still open means finish() was skipped: its
        in-flight buckets were never waited for, and the direct-deposit kernels would add this pass's gradients on top of
        the last one's while autograd-managed parameters start afresh."""
        t = _graph_task_id()
        if self._pass_open and t is not None and self._pass_task is not None and t >= 0 and self._pass_task >= 0 and t != self._pass_task:
            raise RuntimeError("GradBucketAllReduce: a new backward() started before finish() closed the previous pass; call "
                               "dp.finish() after every backward() (inside dp.no_sync() for all micro-batches but the last)")

    # ------------------------------------------------------------------ gradient events
    def grad_buffer(self, p):
        """fp32 buffer a kernel may ADD p's gradient into (the bucket view), or None if p is not registered here."""
        if p not in self._where:
            return None
        if not self._pass_open:
            self._begin_pass()
        else:
            self._check_same_backward()
        view = self._views[p]
        if p.grad is not view and self._detached(p):
            p.grad = view  # only reachable if the caller dropped .grad in the middle of a pass
        return view

    def deposited(self, p):
        """A kernel has enqueued p's gradient into grad_buffer(p)."""
        self._on_grad(p)

    def _on_grad(self, p):
        # idempotent per pass: a parameter whose gradient is deposited directly by a kernel is announced by the op itself,
        # and PyTorch may ALSO run its post-accumulate hook (it does, with an undefined gradient)
        if not self._pass_open:
            self._begin_pass()
        else:
            self._check_same_backward()
        if p in self._seen:
            return
        self._seen.add(p)
        b = self._where[p]
        self._pending[b] -= 1
        if self._pending[b] == 0 and self._sync and self._exchange:
            self._launch(b)

    flushes_reductions = True  # (ops._defer_flag) this sink launches the queued parameter-gradient sums before it reads a bucket

    @staticmethod
    def _flush_reductions():
        if torch.cuda.is_available():
            from . import ops
            ops.flush_reductions()

    def _launch(self, b):
        flat = self.buckets[b]
        if self._direct:
            self._flush_reductions()  # kernels may have QUEUED their final sums into this bucket (ops.flush_reductions)
        if self.async_wgrad is not None:
            self.async_wgrad.sync()  # gradients deposited from the side stream must have landed before the exchange
        buf = flat
        if self._comm is not None:  # wire format: a bf16 copy of the bucket (copied back into the fp32 bucket in finish())
            buf = self._comm[b]
            if self._avg_in_collective:
                buf.copy_(flat)
            else:
                torch.mul(flat, 1.0 / self.world, out=flat)
                buf.copy_(flat)
        elif not self._avg_in_collective:
            flat.mul_(1.0 / self.world)  # average, as DDP does (gloo has no AVG op)
        op = dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM  # RCCL averages inside the all-reduce
        if self.timeline is not None and flat.is_cuda:  # (diagnostic: when did this bucket become ready, from a hook or only in finish()?)
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.timeline.append((b, ev, "finish" if self._in_finish else "hook"))
        self._works.append((b, dist.all_reduce(buf, op=op, group=self.group, async_op=True)))
        self._launched[b] = True
        self._reduced = True

    def finish(self):
        """End of a backward pass: wait for every in-flight bucket (and, outside no_sync(), exchange the buckets no hook
        has launched); call after backward(), before optimizer.step()."""
        if self.async_wgrad is not None:
            self.async_wgrad.sync()
        if self._direct:
            self._flush_reductions()
        if self._exchange and self._sync:
            self._in_finish = True
            try:
                for b, left in enumerate(self._pending):
                    if self._launched[b]:
                        continue
                    if 0 < left < self._counts[b] and self._pass_open:
                        raise RuntimeError("a gradient bucket was only partially produced; unused parameters are not supported")
                    # untouched in this pass (parameters without gradient, or gradients accumulated under no_sync() earlier)
                    self._launch(b)
            finally:
                self._in_finish = False
        for b, w in self._works:
            w.wait()
            if self._comm is not None:
                self.buckets[b].copy_(self._comm[b])
        if self.timeline is not None and self.buckets and self.buckets[0].is_cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()  # (the current stream has waited for every exchange: the optimizer may start here)
            self.timeline.append((-1, ev, "exchanged"))
        self._reset_pass()

    def record_timeline(self, on=True):
        """Diagnostic for multi-GPU runs: with `on`, every bucket exchange of the following passes records an event on the compute
        stream at the moment it is launched (from a gradient hook during the backward, or only in finish()), and finish() one when
        every exchange has been waited for; `timeline_ms(start_event)` turns them into milliseconds since a caller's event."""
        self.timeline = [] if on else None

    def timeline_ms(self, start):
        torch.cuda.synchronize()
        return [dict(bucket=b, ms=round(start.elapsed_time(ev), 3), launched_from=where) for b, ev, where in (self.timeline or [])]

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        # queued parameter-gradient sums hold raw pointers into the buckets: land them before anything here can be freed
        from . import ops as _ops_flush
        _ops_flush.flush_reductions()
        if self._reserved_prev is not None:
            from . import _lib
            _lib.lib.hs_set_reserved_cus(self._reserved_prev)
            self._reserved_prev = None
        if self._prefer_prev is not None:
            from . import ops as _ops
            _ops.RT.prefer_own_gemm = self._prefer_prev
            self._prefer_prev = None
        if not self._direct and self.async_wgrad is None:
            return
        from . import ops

        if self.async_wgrad is not None:
            if ops.RT.async_wgrad is self.async_wgrad:
                ops.RT.async_wgrad = None
            self.async_wgrad = None
        if self._direct:
            if ops.RT.grad_sink is self:
                ops.RT.grad_sink = None
            self._direct = False
