"""Deferred, batched parameter-gradient reductions (csrc/reduce_many.hip; include/healswin.h HS_ACC_DEFER): the queued form of
`hs_linear_wgrad` and of the LayerNorm backward must give the gradients of the immediate form -- through the C ABI on single
layers (every tile geometry, tall and short slice stacks, queue overflow) and on a whole training step with the gradient sink."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _lib():
    from heal_swin_amd import _lib
    return _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("rows,n_out,k_in,bias", [
    (8192, 288, 96, True),      # HEAL-SWIN-T stage 0 qkv: three 128-wide tiles, ~170 slices (16 row phases)
    (8192, 96, 384, True),
    (65536, 2048, 512, True),   # HEAL-SWIN-B stage 2 fc1: 16 tiles of 256 x 256, 16 slices (4 phases)
    (4096, 16, 128, False),     # the padded class head
    (300, 128, 128, True),      # fewer rows than one resident round of slices
])
def test_deferred_linear_wgrad_equals_immediate(rows, n_out, k_in, bias):
    L = _lib()
    lib, ptr = L.lib, L.ptr
    g = torch.Generator(device=DEV).manual_seed(rows + n_out)
    dy = torch.randn((rows, n_out), generator=g, device=DEV).to(torch.bfloat16)
    x = torch.randn((rows, k_in), generator=g, device=DEV).to(torch.bfloat16)
    base_w = torch.randn((n_out, k_in), generator=g, device=DEV)
    base_b = torch.randn(n_out, generator=g, device=DEV)
    nws = int(lib.hs_linear_wgrad_workspace(rows, n_out, k_in))
    out = {}
    for mode in ("immediate", "deferred"):
        for acc in (0, 1):
            dw, db = base_w.clone(), (base_b.clone() if bias else None)
            ws = torch.empty(nws, dtype=torch.float32, device=DEV)
            flag = acc | (L.HS_ACC_DEFER if mode == "deferred" else 0)
            L.check(lib.hs_linear_wgrad(ptr(dy), ptr(x), ptr(dw), ptr(db), ptr(ws), rows, n_out, k_in, flag, L.HS_BF16, _stream()),
                    "hs_linear_wgrad")
            if mode == "deferred":
                assert int(lib.hs_reduce_pending(_stream())) == 1
                if not acc:
                    assert torch.equal(dw, base_w), "a deferred call must not touch the gradient buffer before the flush"
                L.check(lib.hs_reduce_flush(_stream()), "hs_reduce_flush")
                assert int(lib.hs_reduce_pending(_stream())) == 0
            out[mode, acc] = (dw, db)
    ref = dy.float().t() @ x.float()
    for acc in (0, 1):
        a, b = out["immediate", acc], out["deferred", acc]
        scale = float(ref.abs().max())
        assert float((a[0] - b[0]).abs().max()) <= 2e-6 * scale + 1e-6, (acc, float((a[0] - b[0]).abs().max()), scale)
        want = ref + (base_w if acc else 0)
        assert float((b[0] - want).abs().max()) <= 2e-3 * scale
        if bias:
            bref = dy.float().sum(0) + (base_b if acc else 0)
            assert float((a[1] - b[1]).abs().max()) <= 2e-6 * float(bref.abs().max()) + 1e-6
            assert float((b[1] - bref).abs().max()) <= 2e-3 * float(bref.abs().max())


def test_deferred_queue_overflow_flushes_itself_and_keeps_every_sum():
    """More queued sums than one launch holds (44): the queue flushes itself on the producing stream; every destination is right."""
    L = _lib()
    lib, ptr = L.lib, L.ptr
    g = torch.Generator(device=DEV).manual_seed(5)
    rows, n_out, k_in, n = 2048, 64, 64, 100
    nws = int(lib.hs_linear_wgrad_workspace(rows, n_out, k_in))
    keep, dws, refs = [], [], []
    for i in range(n):
        dy = torch.randn((rows, n_out), generator=g, device=DEV).to(torch.bfloat16)
        x = torch.randn((rows, k_in), generator=g, device=DEV).to(torch.bfloat16)
        dw = torch.zeros((n_out, k_in), device=DEV)
        ws = torch.empty(nws, dtype=torch.float32, device=DEV)
        L.check(lib.hs_linear_wgrad(ptr(dy), ptr(x), ptr(dw), None, ptr(ws), rows, n_out, k_in, 1 | L.HS_ACC_DEFER, L.HS_BF16, _stream()),
                "hs_linear_wgrad")
        keep.append(ws)
        dws.append(dw)
        refs.append(dy.float().t() @ x.float())
    assert 0 < int(lib.hs_reduce_pending(_stream())) < 44
    L.check(lib.hs_reduce_flush(_stream()), "hs_reduce_flush")
    for dw, ref in zip(dws, refs):
        assert float((dw - ref).abs().max()) <= 2e-3 * float(ref.abs().max())


@pytest.mark.parametrize("rows,width,dtype", [(98304, 512, torch.bfloat16), (4099, 96, torch.bfloat16), (8192, 128, torch.float32)])
def test_deferred_layernorm_param_grads_equal_immediate(rows, width, dtype):
    L = _lib()
    lib, ptr = L.lib, L.ptr
    g = torch.Generator(device=DEV).manual_seed(rows)
    x = torch.randn((rows, width), generator=g, device=DEV).to(dtype)
    dy = torch.randn((rows, width), generator=g, device=DEV).to(dtype)
    gamma = torch.rand(width, generator=g, device=DEV) + 0.5
    mean = x.float().mean(1).contiguous()
    rstd = (x.float().var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    dt = L.dtype_code(dtype)
    nws = int(lib.hs_layernorm_bwd_workspace(rows, width))
    res = {}
    for mode in ("immediate", "deferred"):
        dx = torch.empty_like(x)
        dg, db = torch.ones(width, device=DEV), torch.ones(width, device=DEV)
        ws = torch.empty(nws, dtype=torch.float32, device=DEV)
        L.check(lib.hs_layernorm_bwd(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dg), ptr(db), ptr(ws),
                                     1 | (L.HS_ACC_DEFER if mode == "deferred" else 0), rows, width, dt, _stream()), "hs_layernorm_bwd")
        if mode == "deferred":
            assert int(lib.hs_reduce_pending(_stream())) == 1
            L.check(lib.hs_reduce_flush(_stream()), "hs_reduce_flush")
        res[mode] = (dx, dg, db)
    xhat = (x.float() - mean[:, None]) * rstd[:, None]
    for i, ref in ((1, 1 + (dy.float() * xhat).sum(0)), (2, 1 + dy.float().sum(0))):
        a, b = res["immediate"][i], res["deferred"][i]
        scale = float(ref.abs().max())
        assert float((a - b).abs().max()) <= 1e-5 * scale, (i, float((a - b).abs().max()), scale)
        assert float((b - ref).abs().max()) <= 1e-3 * scale
    assert torch.equal(res["immediate"][0], res["deferred"][0])


def test_training_step_with_deferred_sums_equals_the_immediate_step():
    """One fwd + bwd of a small model under the gradient sink with and without deferral: same loss, gradients equal to
    summation-order noise; nothing is left queued after finish()."""
    from heal_swin_amd import ops
    from heal_swin_amd.data_spec import DataSpec
    from heal_swin_amd.models_torch.swin_hp_transformer import SwinHPTransformerConfig, SwinHPTransformerSys
    from heal_swin_amd.parallel import GradBucketAllReduce
    L = _lib()
    spec = DataSpec(dim_in=12 * 32 * 32, f_in=3, f_out=12, base_pix=12, class_names=[])
    cfg = SwinHPTransformerConfig(patch_size=4, window_size=64, shift_size=32, rel_pos_bias="flat", embed_dim=64, depths=[2, 2],
                                  num_heads=[2, 4], drop_path_rate=0.0)
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randint(0, 256, (2, 3, spec.dim_in), generator=g, device=DEV).float()
    y = torch.randint(0, 12, (2, spec.dim_in), generator=g, device=DEV, dtype=torch.uint8)
    grads = {}
    prev = ops.DEFER_REDUCTIONS
    try:
        for on in (False, True):
            ops.DEFER_REDUCTIONS = on
            torch.manual_seed(0)
            model = SwinHPTransformerSys(cfg, spec).to(DEV).train()
            model.compute_dtype = torch.bfloat16
            dp = GradBucketAllReduce(model.parameters())
            try:
                dp.zero_grad()
                loss = model.forward_seg_loss(x, y)
                loss.backward()
                if on:
                    assert int(L.lib.hs_reduce_pending(_stream())) > 0, "nothing was deferred"
                dp.finish()
                assert int(L.lib.hs_reduce_pending(_stream())) == 0
                grads[on] = (float(loss), {k: p.grad.clone() for k, p in model.named_parameters()})
            finally:
                dp.remove()
    finally:
        ops.DEFER_REDUCTIONS = prev
    assert grads[False][0] == grads[True][0]
    for k, a in grads[False][1].items():
        b = grads[True][1][k]
        scale = float(a.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 1e-5 * scale, (k, float((a - b).abs().max()), scale)


def test_pending_sums_into_one_destination_do_not_race():
    """Several queued (and immediate) sums into the SAME dw / db -- a parameter used twice in one backward, micro-batches without a
    flush in between, the three bf16x3 products of an fp32 weight gradient: jobs of one flush launch run side by side without
    atomics, so the queue flushes itself when a destination repeats; every contribution must arrive (include/healswin.h)."""
    L = _lib()
    lib, ptr = L.lib, L.ptr
    g = torch.Generator(device=DEV).manual_seed(17)
    rows, n_out, k_in = 16384, 256, 256
    nws = int(lib.hs_linear_wgrad_workspace(rows, n_out, k_in))
    dw, db = torch.zeros((n_out, k_in), device=DEV), torch.zeros(n_out, device=DEV)
    other = torch.zeros((n_out, k_in), device=DEV)
    ref_w, ref_b, keep = torch.zeros_like(dw), torch.zeros_like(db), []
    for i, flag in enumerate((1 | L.HS_ACC_DEFER, 1 | L.HS_ACC_DEFER, 1, 1 | L.HS_ACC_DEFER, 1 | L.HS_ACC_DEFER)):
        dy = torch.randn((rows, n_out), generator=g, device=DEV).to(torch.bfloat16)
        x = torch.randn((rows, k_in), generator=g, device=DEV).to(torch.bfloat16)
        ws = torch.empty(nws, dtype=torch.float32, device=DEV)
        keep.append(ws)
        L.check(lib.hs_linear_wgrad(ptr(dy), ptr(x), ptr(dw), ptr(db), ptr(ws), rows, n_out, k_in, flag, L.HS_BF16, _stream()),
                "hs_linear_wgrad")
        assert int(lib.hs_reduce_pending(_stream())) <= 1, "two sums into one destination were left pending together"
        ref_w += dy.float().t() @ x.float()
        ref_b += dy.float().sum(0)
        if i == 0:  # an unrelated destination may share the flush
            ws2 = torch.empty(nws, dtype=torch.float32, device=DEV)
            keep.append(ws2)
            L.check(lib.hs_linear_wgrad(ptr(dy), ptr(x), ptr(other), None, ptr(ws2), rows, n_out, k_in, 1 | L.HS_ACC_DEFER, L.HS_BF16,
                                        _stream()), "hs_linear_wgrad")
            assert int(lib.hs_reduce_pending(_stream())) == 2
    L.check(lib.hs_reduce_flush(_stream()), "hs_reduce_flush")
    assert float((dw - ref_w).abs().max()) <= 2e-3 * float(ref_w.abs().max())
    assert float((db - ref_b).abs().max()) <= 2e-3 * float(ref_b.abs().max())


def test_fp32_training_step_with_deferred_sums_equals_autograd():
    """fp32 activations = three bf16 products per weight gradient, all into one buffer (ops.LinearFn._wgrad_hip): under the gradient
    sink with deferral ON (the bench's fp32 / depth_fp32 companions) the gradients must equal the ones plain autograd collects
    without a sink -- a race between the three queued sums loses whole products (round-5 review)."""
    from heal_swin_amd import ops
    from heal_swin_amd.data_spec import DataSpec
    from heal_swin_amd.models_torch.swin_hp_transformer import SwinHPTransformerConfig, SwinHPTransformerSys
    from heal_swin_amd.parallel import GradBucketAllReduce
    L = _lib()
    spec = DataSpec(dim_in=12 * 32 * 32, f_in=3, f_out=12, base_pix=12, class_names=[])
    cfg = SwinHPTransformerConfig(patch_size=4, window_size=64, shift_size=32, rel_pos_bias="flat", embed_dim=64, depths=[2, 2],
                                  num_heads=[2, 4], drop_path_rate=0.0)
    g = torch.Generator(device=DEV).manual_seed(4)
    x = torch.randint(0, 256, (2, 3, spec.dim_in), generator=g, device=DEV).float()
    y = torch.randint(0, 12, (2, spec.dim_in), generator=g, device=DEV, dtype=torch.uint8)
    assert ops.DEFER_REDUCTIONS, "the default is what the bench runs"
    grads = {}
    for sink in (False, True):
        torch.manual_seed(0)
        model = SwinHPTransformerSys(cfg, spec).to(DEV).train()
        model.compute_dtype = torch.float32
        dp = GradBucketAllReduce(model.parameters()) if sink else None
        try:
            if dp is not None:
                dp.zero_grad()
            loss = model.forward_seg_loss(x, y)
            loss.backward()
            if dp is not None:
                dp.finish()
                assert int(L.lib.hs_reduce_pending(_stream())) == 0
            grads[sink] = (float(loss), {k: p.grad.clone() for k, p in model.named_parameters()})
        finally:
            if dp is not None:
                dp.remove()
    assert abs(grads[False][0] - grads[True][0]) <= 1e-6 * abs(grads[False][0])
    for k, a in grads[False][1].items():
        b = grads[True][1][k]
        scale = float(a.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 2e-5 * scale, (k, float((a - b).abs().max()), scale)
