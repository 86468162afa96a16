"""torch.autograd.Function wrappers around the C ABI of libhealswin.so.

PyTorch owns the memory (caching allocator), the stream and the autograd graph; every arithmetic step
below runs in the HIP library.  All ops require CUDA(HIP) tensors and raise otherwise -- there is no CPU path.

One module per concern; everything is re-exported here, so `from heal_swin_amd import ops; ops.linear(...)` is the interface:

    runtime    `RT` (process-wide switches of the op layer), live kernel timing, deferred parameter-gradient sums, parameter copies
    norm       LayerNorm family, GELU / residual-drop elementwise ops
    gemm       hs_gemm_nt + its policy, bf16 x 3 products, weight gradients, `LinearFn`
    attention  relative-position bias, cosine scales, attention core, the one-launch WindowAttention module
    tail       decoder tail: LayerNorm + head (+ expand, + cross-entropy)
    mlp_branch the Mlp branch as GEMM epilogues and as one fused kernel per direction
    patch      skip-connection Linear, PatchMerging / PatchExpand, standalone row gather

Module-level SETTINGS (`ops.FUSED_MLP`, `ops.FP32_GEMM`, `ops.KERNEL_TIMINGS`, ...) live in the submodule that reads them; assigning
one on this package forwards the value to every submodule that holds the name, so `ops.FUSED_MLP = False` keeps working for tests,
bench.py and A/B tools.
"""
import sys
import types

from . import runtime, norm, gemm, attention, tail, mlp_branch, patch  # noqa: F401  (dependency order)

_SUBMODULES = (runtime, norm, gemm, attention, tail, mlp_branch, patch)

for _m in _SUBMODULES:
    for _k, _v in vars(_m).items():
        if not _k.startswith("__") and not isinstance(_v, types.ModuleType):
            globals()[_k] = _v
del _m, _k, _v


class _OpsPackage(types.ModuleType):
    def __setattr__(self, name, value):
        for m in _SUBMODULES:
            if name in m.__dict__:
                setattr(m, name, value)
        super().__setattr__(name, value)


sys.modules[__name__].__class__ = _OpsPackage
