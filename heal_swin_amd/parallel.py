"""Data-parallel gradient exchange for the HEAL-SWIN train step: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).

The reference's only collective is Lightning-DDP's gradient all-reduce (heal_swin/train.py:182-189).  Here:
  * every parameter's .grad is a VIEW into a few large flat fp32 buckets (no flatten/unflatten copies);
  * buckets are filled in reverse parameter order (the order backward produces gradients) and an async
    all-reduce is launched from a post-accumulate-grad hook the moment a bucket's last gradient lands, so
    the exchange overlaps the rest of backward;
  * bucket size defaults to 64 MiB: xGMI is point-to-point and ring steps are per-link bound, so few large
    messages beat many small ones (298-596 MB per step for the B model).
"""
import torch
import torch.distributed as dist


class GradBucketAllReduce:
    def __init__(self, params, bucket_bytes=64 << 20, process_group=None, async_wgrad=False, direct_wgrad=True,
                 exchange_single_rank=False):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.async_wgrad = None
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # a one-rank group normally skips the collectives; exchange_single_rank keeps them (exercises the RCCL path on one GPU)
        self._exchange = self.world > 1 or (exchange_single_rank and dist.is_initialized())
        self.buckets = []       # flat fp32 tensors
        self._pending = []      # per bucket: number of grads still missing this step
        self._counts = []
        self._where = {}        # param -> bucket id
        self._works = []
        self._seen = set()      # parameters already counted in this step
        self._build(bucket_bytes)
        self._hooks = []
        if self._exchange:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._direct = False
        if direct_wgrad and not async_wgrad and self.params and self.params[0].is_cuda:
            # Linear weight/bias gradients are accumulated by the wgrad kernel straight into the bucket views
            from . import ops

            ops.GRAD_SINK = self._on_grad if self._exchange else True
            self._direct = True
        if async_wgrad and self.params and self.params[0].is_cuda:
            # Linear weight/bias gradients are produced on a side stream straight into the bucket views (ops.AsyncWgrad)
            from . import ops

            self.async_wgrad = ops.AsyncWgrad(self.params[0].device, sink=self._on_grad if self._exchange else None)
            ops.ASYNC_WGRAD = self.async_wgrad

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
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                self._where[p] = b
            self.buckets.append(flat)
            self._counts.append(len(ps))
        self._pending = list(self._counts)

    def zero_grad(self):
        """Zero the buckets in place (keeps the .grad views alive; use instead of optimizer.zero_grad(set_to_none=True))."""
        for flat in self.buckets:
            flat.zero_()
        self._pending = list(self._counts)
        self._works = []
        self._seen = set()

    def _on_grad(self, p):
        # idempotent per step: a parameter whose gradient is deposited directly by a kernel is announced by the op itself,
        # and PyTorch may ALSO run its post-accumulate hook (it does, with an undefined gradient)
        if p in self._seen:
            return
        self._seen.add(p)
        b = self._where[p]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            flat = self.buckets[b]
            if self.async_wgrad is not None:
                self.async_wgrad.sync()  # gradients deposited from the side stream must have landed before the exchange
            flat.mul_(1.0 / self.world)  # average, as DDP does (gloo has no AVG op)
            self._works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Wait for every in-flight bucket; call after backward(), before optimizer.step()."""
        if self.async_wgrad is not None:
            self.async_wgrad.sync()
        if self._exchange:
            # parameters that received no gradient this step (unused) still need their bucket exchanged
            for b, left in enumerate(self._pending):
                if left not in (0, ) and left != self._counts[b]:
                    raise RuntimeError("a gradient bucket was only partially produced; unused parameters are not supported")
                if left == self._counts[b]:
                    self.buckets[b].mul_(1.0 / self.world)
                    self._works.append(dist.all_reduce(self.buckets[b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in self._works:
            w.wait()
        self._works = []

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self.async_wgrad is not None:
            from . import ops

            if ops.ASYNC_WGRAD is self.async_wgrad:
                ops.ASYNC_WGRAD = None
            self.async_wgrad = None
        if self._direct:
            from . import ops

            ops.GRAD_SINK = None
            self._direct = False
