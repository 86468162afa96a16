"""FlatAdam: torch.optim.Adam / AdamW (what the reference's trainer builds, heal_swin/training/optimizer.py:57-66) on flat buffers.

    dp = GradBucketAllReduce(model.parameters())            # gradients: a few flat fp32 buckets (parallel.py)
    opt = FlatAdam(model.parameters(), dp, lr=1e-3, model=model)
    ...
    dp.zero_grad(); loss.backward(); dp.finish(); opt.step()

Parameters and both moments are laid out exactly like the gradient buckets (every `p.data` becomes a view into a flat fp32
buffer; values are preserved), so a step is ONE `hs_adam_step` launch per bucket (csrc/adam.hip) -- and that launch also writes
the bf16 copy of the updated parameters that the model's next forward reads (`ops.ParamCastCache`), which otherwise costs a
second pass over all parameters per training step.  Same arithmetic as torch.optim.Adam (amsgrad = False, maximize = False;
`decoupled_weight_decay=True` = AdamW); the step counter lives on the device, so the whole training step stays capturable in
one HIP graph (`graphs.GraphedTrainStep`, `bench.py --graph`).

It is a torch.optim.Optimizer: learning-rate schedulers act on `param_groups[0]["lr"]` (read on the host at every `step()`; under
graph replay pass `lr` as a 0-dim CUDA tensor and update it in place), `state_dict()` / `load_state_dict()` carry `exp_avg`,
`exp_avg_sq` and `step` per parameter in torch.optim.Adam's layout.  One parameter group, GPU only -- there is no CPU path.
"""
import torch

from . import _lib, ops
from ._lib import check, lib, ptr, stream_ptr


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, grad_sink, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled_weight_decay=False,
                 model=None, lowp_dtype=torch.bfloat16):
        """grad_sink: the parallel.GradBucketAllReduce that owns the gradients of exactly these parameters.
        model: a SwinHPTransformerSys whose bf16 parameter copies this optimizer should keep current (optional)."""
        params = [p for p in params if p.requires_grad]
        if not params or not all(p.is_cuda and p.dtype == torch.float32 for p in params):
            raise RuntimeError("FlatAdam runs on fp32 master parameters on an MI355X (HIP) device; there is no CPU path")
        if set(map(id, params)) != set(map(id, grad_sink.params)):
            raise ValueError("FlatAdam and its gradient sink must be built over the same parameters")
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or eps < 0.0 or weight_decay < 0.0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay,
                                      decoupled_weight_decay=bool(decoupled_weight_decay), capturable=True, fused=True))
        if len(self.param_groups) != 1:
            raise NotImplementedError("FlatAdam supports one parameter group")
        self.sink = grad_sink
        self.lowp_dtype = lowp_dtype if model is not None else None
        dev = params[0].device
        self._step = torch.zeros((), dtype=torch.int64, device=dev)
        self._flat_p, self._flat_m, self._flat_v, self._flat_lowp = [], [], [], []
        self._lowp_view = {}
        with torch.no_grad():
            for b, g in enumerate(grad_sink.buckets):
                P = torch.zeros_like(g)  # (alignment gaps between parameters stay finite under the Adam arithmetic)
                M, V = torch.zeros_like(g), torch.zeros_like(g)
                S = torch.empty_like(g, dtype=self.lowp_dtype) if self.lowp_dtype is not None else None
                for p in grad_sink.params:
                    if grad_sink._where[p] != b:
                        continue
                    view = grad_sink._views[p]
                    off = view.storage_offset() - g.storage_offset()
                    n = p.numel()
                    pv = P[off:off + n].view_as(p)
                    pv.copy_(p)
                    p.data = pv  # the parameter now lives in the flat buffer (same values, same shape / dtype / device)
                    self.state[p] = {"step": self._step, "exp_avg": M[off:off + n].view_as(p), "exp_avg_sq": V[off:off + n].view_as(p)}
                    if S is not None:
                        self._lowp_view[id(p)] = S[off:off + n].view_as(p)
                if S is not None:
                    S.copy_(P)
                self._flat_p.append(P)
                self._flat_m.append(M)
                self._flat_v.append(V)
                self._flat_lowp.append(S)
        self._model = model
        if model is not None:  # the model's cast cache takes its bf16 shadows from here (built at its next forward)
            model.__dict__["_shadow_provider"] = self.lowp_copy
            model.__dict__.pop("_cast_cache", None)

    def lowp_copy(self, p, dtype):
        """The bf16 view of parameter p that `step()` keeps current, or None (other dtype / foreign parameter)."""
        return self._lowp_view.get(id(p)) if dtype == self.lowp_dtype else None

    def zero_grad(self, set_to_none=False):  # gradients are the sink's bucket views: zeroed in place, never detached
        self.sink.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        grp = self.param_groups[0]
        lr = grp["lr"]
        lr_dev = lr if isinstance(lr, torch.Tensor) else None
        if lr_dev is not None and not (lr_dev.is_cuda and lr_dev.dtype == torch.float32 and lr_dev.numel() == 1):
            raise ValueError("a tensor learning rate must be a 0-dim float32 CUDA tensor")
        b1, b2 = grp["betas"]
        dev = self._step.device
        s = stream_ptr(dev)
        for P, G, M, V, S in zip(self._flat_p, self.sink.buckets, self._flat_m, self._flat_v, self._flat_lowp):
            check(lib.hs_adam_step(ptr(P), ptr(G), ptr(M), ptr(V), ptr(S), P.numel(), 0.0 if lr_dev is not None else float(lr), ptr(lr_dev),
                                   float(b1), float(b2), float(grp["eps"]), float(grp["weight_decay"]),
                                   int(grp["decoupled_weight_decay"]), ptr(self._step), s), "hs_adam_step")
        check(lib.hs_adam_advance(ptr(self._step), s), "hs_adam_advance")
        # the update went through raw pointers (no parameter `_version` moved): caches keyed by weight contents -- the bf16x3
        # splits of fp32 weights, ops._weight_split -- are told here, also for layers used outside a model's forward
        ops.RT.weight_epoch += 1
        cache = None if self._model is None else self._model.__dict__.get("_cast_cache")
        if cache is not None:
            cache.mark_refreshed_externally()
        return loss

    def load_state_dict(self, state_dict):
        """Moments and step count are COPIED into the flat buffers (the views must keep pointing there)."""
        own = {id(p): st for p, st in self.state.items()}
        keep = {id(p): dict(st) for p, st in self.state.items()}
        super().load_state_dict(state_dict)
        with torch.no_grad():
            step = None
            for p in self.param_groups[0]["params"]:
                new, old = self.state.get(p, {}), keep[id(p)]
                for k in ("exp_avg", "exp_avg_sq"):
                    if k not in new:
                        old[k].zero_()  # an optimizer that never stepped: torch would start from zero moments
                    elif new[k] is not old[k]:
                        old[k].copy_(new[k])
                if "step" in new and new["step"] is not old["step"]:
                    step = new["step"]
                self.state[p] = own[id(p)]
                self.state[p].update(old)
            # no step count in the loaded state = a state that never stepped: moments (zeroed above) and bias correction agree
            self._step.fill_(int(float(step)) if step is not None else 0)
        ops.RT.weight_epoch += 1
