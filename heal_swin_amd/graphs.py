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

Constraints (checked): single process (collectives are not captured -- under DP use the eager step), no dropout / DropPath
(their seeds are drawn on the host per call and would be frozen into the graph), a `capturable=True` optimizer, fixed shapes.
"""
import torch


class GraphedTrainStep:
    def __init__(self, model, loss_fn, optimizer, example_inputs, example_targets, warmup=2, pre_forward=None, grad_sink=None):
        """model: a module of this package in train mode; loss_fn(model_output, targets) -> scalar tensor;
        optimizer: constructed with capturable=True (e.g. torch.optim.Adam(..., fused=True, capturable=True));
        example_inputs / example_targets: CUDA tensors of the step's fixed shapes (copied, not kept);
        pre_forward: optional callable applied to the static input inside the graph (e.g. `lambda x: x.float()`);
        grad_sink: optional `parallel.GradBucketAllReduce` of a single-process run (its flat buckets then receive the kernels'
        direct gradient deposits; `zero_grad()` / `finish()` are part of the captured step)."""
        cfg = getattr(model, "config", None)
        for name in ("drop_rate", "attn_drop_rate", "drop_path_rate"):
            if cfg is not None and getattr(cfg, name, 0.0) and model.training:
                raise ValueError(f"GraphedTrainStep: {name} > 0 draws a host-side seed per call, which a graph would freeze")
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            raise ValueError("GraphedTrainStep: gradient all-reduce is not captured; use the eager step under data parallelism")
        for group in optimizer.param_groups:
            if not group.get("capturable", False):
                raise ValueError("GraphedTrainStep: the optimizer must be constructed with capturable=True")
        if not (example_inputs.is_cuda and example_targets.is_cuda):
            raise ValueError("GraphedTrainStep needs an MI355X: example tensors must live on the GPU (no CPU path)")
        self.model, self.loss_fn, self.optimizer = model, loss_fn, optimizer
        self.pre_forward, self.grad_sink = pre_forward, grad_sink
        self.inputs = example_inputs.detach().clone()
        self.targets = example_targets.detach().clone()

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
        with torch.cuda.graph(self.graph):
            self.loss = self._eager()

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
        return loss.detach()

    def __call__(self, inputs, targets):
        """One training step on (inputs, targets); returns the loss (a static device tensor, overwritten by the next call)."""
        self.inputs.copy_(inputs, non_blocking=True)
        self.targets.copy_(targets, non_blocking=True)
        self.graph.replay()
        # The replay updated the parameters on the device without touching their Python-side version counters, which is what
        # ops.ParamCastCache keys its bf16 copies on: an EAGER forward after this replay (validation, the no-grad fused path)
        # must re-make them.  (The replayed step itself refreshes its copies inside the graph.)  Host-only, no launch.
        invalidate = getattr(self.model, "invalidate_param_casts", None)
        if invalidate is not None:
            invalidate()
        return self.loss
