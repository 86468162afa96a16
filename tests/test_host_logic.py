"""Host-side logic that needs no GPU: the per-step cast cache of the Linear parameters."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_param_cast_cache_refreshes_only_after_parameter_updates():
    from heal_swin_amd import ops
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(8, 4)), torch.nn.Parameter(torch.randn(8)), torch.nn.Parameter(torch.randn(3, 8))]
    cache = ops.ParamCastCache(params, torch.bfloat16)
    cache.refresh()
    for p in params:
        sh = cache.get(p, torch.bfloat16)
        assert sh is not None and sh.dtype == torch.bfloat16 and not sh.requires_grad
        assert torch.equal(sh, p.detach().to(torch.bfloat16))
    assert cache.get(params[0], torch.float16) is None            # other dtype: caller casts itself
    assert cache.get(torch.nn.Parameter(torch.zeros(2)), torch.bfloat16) is None  # unregistered parameter
    before = [cache.get(p, torch.bfloat16).clone() for p in params]
    cache.refresh()                                               # nothing changed: shadows untouched (same storage, same data)
    assert all(torch.equal(a, cache.get(p, torch.bfloat16)) for a, p in zip(before, params))
    with torch.no_grad():
        params[1].add_(1.0)                                       # an optimizer step bumps the version counter
    cache.refresh()
    assert torch.equal(cache.get(params[1], torch.bfloat16), params[1].detach().to(torch.bfloat16))
    assert not torch.equal(cache.get(params[1], torch.bfloat16), before[1])


def test_cast_param_falls_back_to_a_plain_cast_without_a_cache():
    from heal_swin_amd import ops
    p = torch.nn.Parameter(torch.randn(4, 4))
    assert ops.CAST_CACHE is None
    assert ops._cast_param(p, torch.float32) is p
    c = ops._cast_param(p, torch.bfloat16)
    assert c.dtype == torch.bfloat16 and torch.equal(c, p.detach().to(torch.bfloat16))
