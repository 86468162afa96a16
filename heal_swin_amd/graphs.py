"""Whole-step HIP graph: zero_grad -> forward -> loss -> backward -> optimizer step captured once and replayed.

The hot path issues ~1500 kernel launches per training step (every `hs_*` kernel is a ctypes call behind an autograd node).
On the large workloads the GPU is the bottleneck and the host keeps ahead of it; on the small ones (BASELINE configs[0], [1]:
HEAL-SWIN-T at nside <= 128) the step is bound by the host's launch rate (22.5 ms for a step whose kernels take < 10 ms).
Every launch of the library goes to the caller's CURRENT stream, allocates nothing and never synchronises, so a whole step can
be recorded into one `hipGraph` through PyTorch's capture and replayed with a single launch -- the MI355X-native answer to
what the reference would have needed a tracing compiler for.

    step = GraphedTrainStep(model, lambda logits, y: seg_loss(logits, y), optimizer, images, labels)
    for images, labels in loader:
        loss = step(images, labels)          # copies the batch into the graph's static buffers and replays

Constraints (checked): a `capturable=True` optimizer, fixed shapes.  Data parallelism (world > 1, round 6): collectives are not captured, so
the step becomes TWO graphs around an eager exchange -- graph 1: zero_grad -> forward -> loss -> backward with the gradients accumulated
locally in the sink's flat buckets (`GradBucketAllReduce.no_sync()`: the hooks launch nothing); eager: `sink.finish()` all-reduces every
bucket; graph 2: the optimizer step.  That gives up the overlap of the exchange with the backward (all buckets leave after the backward) for
the host launch rate: it pays on the launch-bound workloads (HEAL-SWIN-T at nside <= 128: 18-19 ms eager, 15 ms replayed, ~1-2 ms of
exposed exchange for 165 MB of gradients), not on the GPU-bound ones.  Needs the package's gradient sink (`grad_sink=`); torch's
DistributedDataParallel launches its collectives from autograd hooks inside the backward and cannot be split this way.  Dropout / DropPath: the kernels' seeds are drawn on the host per call and are frozen into the graph; a model with
drop rates > 0 is replayed with a device-side step counter registered with the library (`hs_set_seed_epoch`): every mask generator
adds counter x odd constant to its frozen seed, the captured step ends with counter += 1, so every replay draws fresh masks and the
forward and backward of a step agree.  DropPath's per-sample factors come from torch's CUDA generator, which is graph-safe by itself.
"""
import torch

from . import _lib


class GraphedTrainStep:
    def __init__(self, model, loss_fn, optimizer, example_inputs, example_targets, warmup=2, pre_forward=None, grad_sink=None):
        """model: a module of this package in train mode; loss_fn(model_output, targets) -> scalar tensor;
        optimizer: constructed with capturable=True (e.g. torch.optim.Adam(..., fused=True, capturable=True));
        example_inputs / example_targets: CUDA tensors of the step's fixed shapes (copied, not kept);
        pre_forward: optional callable applied to the static input inside the graph (e.g. `lambda x: x.float()`);
        grad_sink: optional `parallel.GradBucketAllReduce` of a single-process run (its flat buckets then receive the kernels'
        direct gradient deposits; `zero_grad()` / `finish()` are part of the captured step)."""
        cfg = getattr(model, "config", None)
        stochastic = model.training and cfg is not None and any(getattr(cfg, n, 0.0) for n in ("drop_rate", "attn_drop_rate", "drop_path_rate"))
        self._epoch = None
        self._dp = bool(torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1)
        if self._dp and (grad_sink is None or getattr(grad_sink, "world", 1) <= 1):
            raise ValueError("GraphedTrainStep under data parallelism needs grad_sink=GradBucketAllReduce(...): the step is replayed as two "
                             "graphs around the sink's eager exchange (torch DDP's in-backward collectives cannot be captured)")
        for group in optimizer.param_groups:
            if not group.get("capturable", False):
                raise ValueError("GraphedTrainStep: the optimizer must be constructed with capturable=True")
        if not (example_inputs.is_cuda and example_targets.is_cuda):
            raise ValueError("GraphedTrainStep needs an MI355X: example tensors must live on the GPU (no CPU path)")
        self.model, self.loss_fn, self.optimizer = model, loss_fn, optimizer
        self.pre_forward, self.grad_sink = pre_forward, grad_sink
        self.inputs = example_inputs.detach().clone()
        self.targets = example_targets.detach().clone()
        if stochastic:  # replay counter of the mask generators (process-wide in the library: one stochastic graphed step at a time)
            if _lib.lib.hs_get_seed_epoch():
                raise ValueError("GraphedTrainStep: another graphed step with dropout is alive in this process (one seed counter per process)")
            self._epoch = torch.zeros(1, dtype=torch.int64, device=self.inputs.device)
            _lib.check(_lib.lib.hs_set_seed_epoch(_lib.ptr(self._epoch)), "hs_set_seed_epoch")

        try:
            self._warm_up_and_capture(warmup)
        except BaseException:
            self.close()  # a failed warm-up / capture must not leave the library pointing at this object's counter
            raise

    def _warm_up_and_capture(self, warmup):
        # warm-up on a side stream (PyTorch's capture protocol): lazy initialisations -- kernel attributes, the bf16 weight
        # shadows, optimizer state, the allocator's blocks -- happen here, not inside the capture
        side = torch.cuda.Stream(device=self.inputs.device)
        side.wait_stream(torch.cuda.current_stream(self.inputs.device))
        with torch.cuda.stream(side):
            # at least TWO eager steps: the second one builds the device-side job table of the batched weight transposes
            # (ops.ParamCastCache), a pageable host-to-device copy that must not happen inside the capture
            for _ in range(max(2, warmup)):
                self._eager()
        torch.cuda.current_stream(self.inputs.device).wait_stream(side)
        torch.cuda.synchronize(self.inputs.device)

        self.graph = torch.cuda.CUDAGraph()
        if not self._dp:
            with torch.cuda.graph(self.graph):
                self.loss = self._eager()
            return
        # data parallel: [graph 1: zero_grad, forward, loss, backward into the local buckets] -> eager exchange -> [graph 2: optimizer]
        self.graph_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = self._fwd_bwd_local()
        self.grad_sink.finish()  # (an exchange between the two captures keeps every rank's collective sequence aligned with the replays')
        with torch.cuda.graph(self.graph_opt):
            self._opt_step()

    def _eager(self):
        if self.grad_sink is not None:
            self.grad_sink.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=False)  # gradients keep their addresses: the graph writes into them
        x = self.inputs if self.pre_forward is None else self.pre_forward(self.inputs)
        loss = self.loss_fn(self.model(x), self.targets)
        loss.backward()
        if self.grad_sink is not None:
            self.grad_sink.finish()
        self.optimizer.step()
        if self._epoch is not None:
            self._epoch.add_(1)  # the next step (replay) draws new masks
        return loss.detach()

    def _fwd_bwd_local(self):
        with self.grad_sink.no_sync():  # gradients accumulate in the flat buckets; no hook launches a collective
            self.grad_sink.zero_grad()
            x = self.inputs if self.pre_forward is None else self.pre_forward(self.inputs)
            loss = self.loss_fn(self.model(x), self.targets)
            loss.backward()
            self.grad_sink.finish()  # (local: queued parameter-gradient sums land, nothing is exchanged)
        return loss.detach()

    def _opt_step(self):
        self.optimizer.step()
        if self._epoch is not None:
            self._epoch.add_(1)

    def close(self):
        """Unregister the replay counter (stochastic models); the graph must not be replayed afterwards."""
        if self._epoch is not None:
            torch.cuda.synchronize(self._epoch.device)
            _lib.lib.hs_set_seed_epoch(None)
            self._epoch = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __call__(self, inputs, targets):
        """One training step on (inputs, targets); returns the loss (a static device tensor, overwritten by the next call)."""
        self.inputs.copy_(inputs, non_blocking=True)
        self.targets.copy_(targets, non_blocking=True)
        self.graph.replay()
        if self._dp:
            self.grad_sink.finish()   # eager: every bucket all-reduced (averaged) and waited for on the current stream
            self.graph_opt.replay()
        # The replay updated the parameters on the device without touching their Python-side version counters, which is what
        # ops.ParamCastCache keys its bf16 copies on: an EAGER forward after this replay (validation, the no-grad fused path)
        # must re-make them.  (The replayed step itself refreshes its copies inside the graph.)  Host-only, no launch.
        invalidate = getattr(self.model, "invalidate_param_casts", None)
        if invalidate is not None:
            invalidate()
        return self.loss
