"""hs_window_attn_module_fwd (qkv -> attention -> proj, optional LayerNorm prologue / residual epilogue, one launch) against
the oracle composition on bf16-rounded inputs, and the whole model in no-grad mode (fused path) against its own autograd-mode
forward (three-kernel path) and the oracle."""
import types

import pytest
import torch

from _util import assert_close
from test_gpu_kernels import _oracle_core

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _reference(x, wqkv, bqkv, wp, bp, bias, hscale, idx, labels, nH, cosine, ln, residual):
    from oracle import model as OM
    xin = OM.layer_norm(x, ln[0], ln[1]) if ln is not None else x
    qkv = xin @ wqkv.t() + (bqkv if bqkv is not None else 0)
    o = _oracle_core(qkv, bias, hscale, idx, labels, nH, 64, cosine)
    y = o @ wp.t() + (bp if bp is not None else 0)
    return x + y if residual else y


CASES = [
    # C, nH, B, nside, strategy, shift, cosine, bias, ln, residual, qkv_bias
    (128, 4, 2, 16, "none", 0, False, True, False, False, True),
    (128, 4, 2, 16, "nest_roll", 32, False, True, True, True, True),
    (128, 4, 1, 16, "ring_shift", 4, True, True, True, True, True),
    (128, 4, 1, 16, "nest_grid_shift", 32, True, False, False, True, False),
    (96, 3, 2, 16, "nest_roll", 32, False, True, True, True, True),
    (96, 3, 1, 16, "ring_shift", 4, True, True, False, False, True),
    (96, 3, 1, 8, "none", 0, False, False, True, False, False),
]


@pytest.mark.parametrize("C,nH,B,nside,strategy,shift,cosine,use_bias,use_ln,residual,qkv_bias", CASES)
def test_module_kernel_vs_oracle(C, nH, B, nside, strategy, shift, cosine, use_bias, use_ln, residual, qkv_bias):
    from heal_swin_amd import ops
    from oracle import tables as T
    N = 8 * nside * nside
    g = torch.Generator().manual_seed(C + nside + shift)
    bf = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
    x = bf(torch.randn(B, N, C, generator=g) * (3.0 if use_ln else 1.0) + (0.5 if use_ln else 0.0))
    wqkv, wp = bf(torch.randn(3 * C, C, generator=g) * C ** -0.5), bf(torch.randn(C, C, generator=g) * C ** -0.5)
    bqkv = torch.randn(3 * C, generator=g) * 0.2 if qkv_bias else None
    bp = torch.randn(C, generator=g) * 0.2
    bias = torch.randn(nH, 64, 64, generator=g) if use_bias else None
    hscale = torch.rand(nH, generator=g) * (8 if cosine else 0.3) + 0.1
    ln = (torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2) if use_ln else None
    if strategy == "none":
        idx = labels = None
    else:
        fn = {"nest_roll": lambda: T.nest_roll_shift(N, 64, shift), "nest_grid_shift": lambda: T.nest_grid_shift(nside, 8, 64),
              "ring_shift": lambda: T.ring_shift(nside, 8, 64, shift)}[strategy]
        idx_np, _, lab_np = fn()
        idx, labels = torch.from_numpy(idx_np), torch.from_numpy(lab_np)
    ref = _reference(x, wqkv, bqkv, wp, bp, bias, hscale, idx, labels, nH, cosine, ln, residual)

    d = lambda t: None if t is None else t.to(DEV)  # noqa: E731
    use_roll = strategy == "nest_roll"
    with torch.no_grad():
        assert ops.window_attn_module_ok(x.to(DEV).to(torch.bfloat16), nH, 64)
        y = ops.window_attn_module(x.to(DEV).to(torch.bfloat16), d(wqkv), d(bqkv), d(wp), d(bp), d(bias), d(hscale),
                                   None if (use_roll or idx is None) else idx.to(torch.int32).to(DEV), shift if use_roll else 0,
                                   None if labels is None else labels.to(torch.uint8).to(DEV), nH, 64, cosine,
                                   ln_weight=None if ln is None else d(ln[0]), ln_bias=None if ln is None else d(ln[1]),
                                   residual=residual)
    # bf16 intermediates (normalised x, q / k / v, P, O) inside the kernel: bf16 tolerance
    assert_close(y, ref, 1.5e-2, "module out")


@pytest.mark.parametrize("cfgkw", [dict(embed_dim=128, num_heads=[4, 8], shift_strategy="nest_roll", shift_size=32, bp=12),
                                   dict(embed_dim=96, num_heads=[3, 6], shift_strategy="ring_shift", shift_size=4, bp=8),
                                   dict(embed_dim=96, num_heads=[3, 6], shift_strategy="ring_shift", shift_size=4, bp=8, use_cos_attn=True,
                                        use_v2_norm_placement=True)])
def test_model_no_grad_uses_the_fused_module_and_matches(cfgkw):
    """Under torch.no_grad() the stage-0 blocks run the one-launch module kernel (v1 placement: with norm1 and the residual
    add inside); logits must agree with the autograd-mode forward (three-kernel path) and with the oracle."""
    from heal_swin_amd import ops
    from heal_swin_amd.data_spec import DataSpec
    from heal_swin_amd.models_torch import swin_hp_transformer as M
    from oracle import model as OM
    kw = dict(cfgkw)
    bp = kw.pop("bp")
    cfg = dict(patch_size=4, window_size=64, shift_size=32, shift_strategy="nest_roll", rel_pos_bias="flat", embed_dim=96, depths=[2, 2],
               num_heads=[3, 6], mlp_ratio=4.0, qkv_bias=True, qk_scale=None, use_cos_attn=False, drop_rate=0.0, attn_drop_rate=0.0,
               drop_path_rate=0.0, use_v2_norm_placement=False, ape=False)
    cfg.update(kw)
    spec = dict(dim_in=bp * 32 * 32, f_in=3, f_out=12, base_pix=bp, class_names=[])
    torch.manual_seed(3)
    model = M.SwinHPTransformerSys(M.SwinHPTransformerConfig(**cfg), DataSpec(**spec))
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("relative_position_bias_table"):
                p.normal_(0, 0.02)
    sd = {k: v.clone() for k, v in model.state_dict().items() if not k.endswith("attn_mask")}
    x = torch.randint(0, 256, (2, 3, spec["dim_in"])).float()
    y_ref = OM.forward(sd, types.SimpleNamespace(**cfg), types.SimpleNamespace(**spec), x)
    model = model.to(DEV).eval()
    model.compute_dtype = torch.bfloat16
    calls = []
    real = ops.window_attn_module
    ops.window_attn_module = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad():
            y_fused = model(x.to(DEV))
        n_fused = len(calls)
        y_plain = model(x.to(DEV))  # autograd mode: qkv GEMM -> hs_window_attn_fwd -> proj GEMM
    finally:
        ops.window_attn_module = real
    assert n_fused == 4 and len(calls) == 4  # the 2 + 2 stage-0 blocks (encoder, decoder), only in no-grad mode
    assert_close(y_fused, y_ref, 1e-2, "fused logits vs oracle")
    assert_close(y_fused, y_plain, 1e-2, "fused vs three-kernel logits")


def test_module_kernel_batches_beyond_the_2gib_descriptor_range():
    """B * N * C * 2 bytes > 2 GiB (batch >= 43 at stage 0 of HEAL-SWIN-B / nside 256) used to raise
    HS_ERR_UNSUPPORTED on the default no-grad path.  The entry point now runs such calls in chunks of whole images; the result
    equals the per-image calls bit for bit."""
    import torch
    from heal_swin_amd import ops
    B, N, C, nH = 44, 196608, 128, 4
    assert B * N * C * 2 > 0x7FFFFE00
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B, N, C, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    qkv_w = (torch.randn(3 * C, C, generator=g, device="cuda") * 0.05)
    proj_w = (torch.randn(C, C, generator=g, device="cuda") * 0.05)
    qkv_b = torch.randn(3 * C, generator=g, device="cuda") * 0.1
    proj_b = torch.randn(C, generator=g, device="cuda") * 0.1
    bias = torch.randn(nH, 64, 64, generator=g, device="cuda") * 0.2
    hs = torch.full((nH,), 32 ** -0.5, device="cuda")
    with torch.no_grad():
        assert ops.window_attn_module_ok(x, nH, 64)
        y = ops.window_attn_module(x, qkv_w, qkv_b, proj_w, proj_b, bias, hs, None, 32, None, nH, 64, False)
        for b in (0, 41, 42, 43):  # images on both sides of the chunk boundary (42 images fit the descriptor)
            yb = ops.window_attn_module(x[b:b + 1], qkv_w, qkv_b, proj_w, proj_b, bias, hs, None, 32, None, nH, 64, False)
            assert torch.equal(y[b:b + 1], yb), b


# ----------------------------------------------------------------------------- training form (hs_window_attn_module_fwd_train)
TRAIN_CASES = [
    # C, nH, B, nside, strategy, shift, cosine, bias, qkv_bias, v1 (LayerNorm in front + residual behind; False: v2 placement, neither)
    (128, 4, 2, 16, "none", 0, False, True, True, True),
    (128, 4, 2, 16, "nest_roll", 32, False, True, True, True),
    (128, 4, 1, 16, "ring_shift", 4, True, True, True, True),
    (128, 4, 1, 16, "nest_grid_shift", 32, True, False, False, True),
    (96, 3, 2, 16, "nest_roll", 32, False, True, True, True),
    (96, 3, 1, 16, "ring_shift", 4, True, True, True, True),
    (96, 3, 1, 8, "nest_grid_shift", 32, False, False, False, True),
    (96, 3, 2, 16, "ring_shift", 4, True, True, True, False),
    (128, 4, 1, 16, "nest_roll", 32, False, True, True, False),
    (128, 4, 1, 16, "nest_grid_shift", 32, True, False, True, False),
]


@pytest.mark.parametrize("C,nH,B,nside,strategy,shift,cosine,use_bias,qkv_bias,v1", TRAIN_CASES)
def test_module_train_form_vs_oracle_and_composition(C, nH, B, nside, strategy, shift, cosine, use_bias, qkv_bias, v1):
    """out = x + proj(attention(qkv(LayerNorm(x)))) by ONE launch that also saves LayerNorm(x), its statistics, qkv, the attention
    output and the log-sum-exp rows; the backward is the composed path's on those tensors.  Output and EVERY gradient (x, norm
    weight / bias, qkv and proj weight / bias, bias table, head scale) against the oracle's autograd on bf16-rounded inputs, and
    the saved tensors + gradients against the four-kernel composition (same rounding points: much tighter)."""
    from heal_swin_amd import ops
    from oracle import tables as T
    N = 8 * nside * nside
    g = torch.Generator().manual_seed(C + nside + shift + 17)
    bf = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
    x = bf(torch.randn(B, N, C, generator=g) * (3.0 if v1 else 1.0) + (0.5 if v1 else 0.0))
    P = dict(wqkv=bf(torch.randn(3 * C, C, generator=g) * C ** -0.5), wp=bf(torch.randn(C, C, generator=g) * C ** -0.5),
             bp=torch.randn(C, generator=g) * 0.2, hscale=torch.rand(nH, generator=g) * (8 if cosine else 0.3) + 0.1)
    if v1:
        P.update(lng=torch.rand(C, generator=g) + 0.5, lnb=torch.randn(C, generator=g) * 0.2)
    if qkv_bias:
        P["bqkv"] = torch.randn(3 * C, generator=g) * 0.2
    if use_bias:
        P["bias"] = torch.randn(nH, 64, 64, generator=g)
    dout = bf(torch.randn(B, N, C, generator=g))
    if strategy == "none":
        idx = labels = None
    else:
        fn = {"nest_roll": lambda: T.nest_roll_shift(N, 64, shift), "nest_grid_shift": lambda: T.nest_grid_shift(nside, 8, 64),
              "ring_shift": lambda: T.ring_shift(nside, 8, 64, shift)}[strategy]
        idx_np, _, lab_np = fn()
        idx, labels = torch.from_numpy(idx_np), torch.from_numpy(lab_np)

    # ---- oracle (CPU fp32 autograd on the oracle's formulas)
    xr = x.clone().requires_grad_(True)
    R = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref = _reference(xr, R["wqkv"], R.get("bqkv"), R["wp"], R["bp"], R.get("bias"), R["hscale"], idx, labels, nH, cosine,
                     (R["lng"], R["lnb"]) if v1 else None, v1)
    if not v1:  # the block's `x + norm(branch)` uses x a second time: the alias handed back by the call carries that gradient
        ref = ref + 0.5 * xr
    ref.backward(dout)

    use_roll = strategy == "nest_roll"
    idx_d = None if (use_roll or idx is None) else idx.to(torch.int32).to(DEV)
    lab_d = None if labels is None else labels.to(torch.uint8).to(DEV)
    roll = shift if use_roll else 0

    def run(fused):
        xd = x.to(DEV).to(torch.bfloat16).requires_grad_(True)
        D = {k: v.to(DEV).clone().requires_grad_(True) for k, v in P.items()}
        saved = {}
        if fused:
            assert ops.window_attn_module_train_ok(xd, nH, 64)
            real = ops.lib.hs_window_attn_module_fwd_train
            out = ops.window_attn_module_train(xd, D.get("lng"), D.get("lnb"), D["wqkv"], D.get("bqkv"), D["wp"], D["bp"], D.get("bias"),
                                               D["hscale"], idx_d, roll, lab_d, nH, 64, cosine, residual_alias=not v1)
            if not v1:
                out = out[0] + 0.5 * out[1]
            assert real is ops.lib.hs_window_attn_module_fwd_train
            # the recorded graph: proj LinearFn <- WindowAttnCoreFn <- qkv LinearFn <- LayerNormFn, saved tensors from the kernel
            node = out.grad_fn if v1 else out.grad_fn.next_functions[0][0]
            saved["o"] = node.saved_tensors[0]
        elif v1:
            n1, xs = ops.layer_norm_passthrough(xd, D["lng"], D["lnb"])
            qkv = ops.linear(n1, D["wqkv"], D.get("bqkv"))
            o = ops.window_attn_core(qkv, D.get("bias"), D["hscale"], idx_d, roll, lab_d, nH, 64, cosine)
            out = ops.linear_residual(o, D["wp"], D["bp"], xs)
            saved.update(xn=n1.detach(), qkv=qkv.detach(), o=o.detach())
        else:
            qkv, xs = ops.linear_passthrough(xd, D["wqkv"], D.get("bqkv"))
            o = ops.window_attn_core(qkv, D.get("bias"), D["hscale"], idx_d, roll, lab_d, nH, 64, cosine)
            out = ops.linear(o, D["wp"], D["bp"]) + 0.5 * xs
            saved.update(qkv=qkv.detach(), o=o.detach())
        out.backward(dout.to(DEV).to(torch.bfloat16))
        return out.detach(), xd.grad, {k: v.grad for k, v in D.items()}, saved

    out_f, dx_f, G_f, S_f = run(True)
    out_c, dx_c, G_c, S_c = run(False)
    # vs the oracle: bf16 intermediates inside the kernel
    assert_close(out_f, ref, 1.5e-2, "train-form out vs oracle")
    assert_close(dx_f, xr.grad, 4e-2, "train-form dx vs oracle")
    for k in P:
        if k == "hscale" and not cosine:
            continue
        tol = 8e-2 if k in ("bias", "hscale") else 4e-2  # (the table / scale gradients: the noisy families of tests/test_gpu_model.py)
        assert_close(G_f[k], R[k].grad, tol, f"train-form d{k} vs oracle")
    # vs the composition: same formulas, same rounding points, different summation order in the two products
    assert_close(out_f, out_c, 1e-2, "train-form out vs composition")
    assert_close(S_f["o"], S_c["o"], 1e-2, "saved attention output vs composition")
    assert_close(dx_f, dx_c, 2e-2, "train-form dx vs composition")
    for k in P:
        if k == "hscale" and not cosine:
            continue
        assert_close(G_f[k], G_c[k], 3e-2, f"train-form d{k} vs composition")


def test_module_train_form_saved_tensors_equal_the_separate_kernels():
    """C ABI: xn / mean / rstd are those of hs_layernorm_fwd bit for bit (same arithmetic per row), qkv and the attention output
    those of hs_gemm_nt / hs_window_attn_fwd on them to bf16 rounding, lse to fp32 rounding of the same scores."""
    from heal_swin_amd import ops, _lib
    from heal_swin_amd._lib import check, lib, ptr
    B, N, C, nH = 2, 8 * 16 * 16, 128, 4
    g = torch.Generator(device=DEV).manual_seed(5)
    x = (torch.randn(B, N, C, generator=g, device=DEV) * 2 + 0.3).to(torch.bfloat16)
    wq = (torch.randn(3 * C, C, generator=g, device=DEV) * C ** -0.5).to(torch.bfloat16)
    wp = (torch.randn(C, C, generator=g, device=DEV) * C ** -0.5).to(torch.bfloat16)
    bq, bp = torch.randn(3 * C, generator=g, device=DEV) * 0.2, torch.randn(C, generator=g, device=DEV) * 0.2
    lg, lb = torch.rand(C, generator=g, device=DEV) + 0.5, torch.randn(C, generator=g, device=DEV) * 0.2
    bias = torch.randn(nH, 64, 64, generator=g, device=DEV)
    hs = torch.full((nH,), 32 ** -0.5, device=DEV)
    out, xn, o = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    qkv = torch.empty(B, N, 3 * C, dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.empty(B * N, device=DEV), torch.empty(B * N, device=DEV)
    lse = torch.empty(B, nH, N, device=DEV)
    l2g, l2b = torch.rand(C, generator=g, device=DEV) + 0.5, torch.randn(C, generator=g, device=DEV) * 0.2
    n2, mean_2, rstd_2 = torch.empty_like(x), torch.empty(B * N, device=DEV), torch.empty(B * N, device=DEV)
    check(lib.hs_window_attn_module_fwd_train(ptr(x), ptr(out), ptr(xn), ptr(mean), ptr(rstd), ptr(qkv), ptr(o), ptr(lse), ptr(wq), ptr(bq),
                                              ptr(wp), ptr(bp), ptr(lg), ptr(lb), ptr(bias), ptr(hs), None, 32, None, ptr(l2g), ptr(l2b), ptr(n2),
                                              ptr(mean_2), ptr(rstd_2), B, N, C, nH, 64, _lib.HS_ATTN_RESIDUAL, _lib.HS_BF16, None), "train")
    # the block's norm2 in the same launch: hs_layernorm_fwd on the `out` rows it has just written
    n2r, m2r, r2r = torch.empty_like(x), torch.empty_like(mean_2), torch.empty_like(rstd_2)
    check(lib.hs_layernorm_fwd(ptr(out), None, ptr(l2g), ptr(l2b), ptr(n2r), ptr(m2r), ptr(r2r), B * N, C, _lib.HS_BF16, None), "ln2")
    assert_close(mean_2, m2r, 1e-5, "norm2 mean")
    assert_close(rstd_2, r2r, 1e-5, "norm2 rstd")
    assert_close(n2, n2r, 4e-3, "norm2(out)")
    xn2, mean2, rstd2 = torch.empty_like(x), torch.empty_like(mean), torch.empty_like(rstd)
    check(lib.hs_layernorm_fwd(ptr(x), None, ptr(lg), ptr(lb), ptr(xn2), ptr(mean2), ptr(rstd2), B * N, C, _lib.HS_BF16, None), "ln")
    assert_close(mean, mean2, 1e-5, "mean")
    assert_close(rstd, rstd2, 1e-5, "rstd")
    assert_close(xn, xn2, 4e-3, "LayerNorm(x)")  # (one bf16 ulp where the two kernels' fp32 statistics differ in the last bit)
    with torch.no_grad():
        qkv2 = ops.gemm_nt(xn.reshape(-1, C), wq, bq)[0].view(B, N, 3 * C)
        assert_close(qkv, qkv2, 4e-3, "qkv")
        o2 = torch.empty_like(o)
        lse2 = torch.empty_like(lse)
        check(lib.hs_window_attn_fwd(ptr(qkv), ptr(o2), ptr(lse2), ptr(bias), ptr(hs), None, 32, None, B, N, C, nH, 64, 0, 0.0, 0,
                                     _lib.HS_BF16, None), "core")
        assert_close(o, o2, 4e-3, "attention output")
        assert_close(lse, lse2, 1e-5, "lse")
        out2 = ops.gemm_nt(o.reshape(-1, C), wp, bp, _lib.HS_EPI_RESID, aux=x.reshape(-1, C))[0].view(B, N, C)
        assert_close(out, out2, 4e-3, "out")


def test_model_training_step_uses_the_train_form_and_matches_the_composition():
    """Stage-0 blocks of a bf16 model in autograd mode run hs_window_attn_module_fwd_train (4 calls: 2 encoder + 2 decoder
    blocks); loss and every parameter gradient agree with the same model on the four-kernel composition."""
    from heal_swin_amd import ops
    from heal_swin_amd.data_spec import DataSpec
    from heal_swin_amd.models_torch import swin_hp_transformer as M
    cfg = dict(patch_size=4, window_size=64, shift_size=32, shift_strategy="nest_roll", rel_pos_bias="flat", embed_dim=128, depths=[2, 2],
               num_heads=[4, 8], mlp_ratio=4.0, qkv_bias=True, qk_scale=None, use_cos_attn=False, drop_rate=0.0, attn_drop_rate=0.0,
               drop_path_rate=0.0, use_v2_norm_placement=False, ape=False)
    spec = dict(dim_in=12 * 32 * 32, f_in=3, f_out=12, base_pix=12, class_names=[])
    torch.manual_seed(11)
    model = M.SwinHPTransformerSys(M.SwinHPTransformerConfig(**cfg), DataSpec(**spec))
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("relative_position_bias_table"):
                p.normal_(0, 0.02)
    model = model.to(DEV).train()
    model.compute_dtype = torch.bfloat16
    x = torch.randint(0, 256, (2, 3, spec["dim_in"])).float().to(DEV)
    w = torch.randn(2, 12, spec["dim_in"], device=DEV)

    def step(fused):
        prev = ops.FUSED_ATTN_MODULE_TRAIN
        ops.FUSED_ATTN_MODULE_TRAIN = fused
        calls = []
        real = ops.window_attn_module_train
        ops.window_attn_module_train = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        try:
            model.zero_grad(set_to_none=True)
            y = model(x)
            (y.float() * w).mean().backward()
        finally:
            ops.window_attn_module_train = real
            ops.FUSED_ATTN_MODULE_TRAIN = prev
        return y.detach(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}, len(calls)

    y_f, g_f, n_f = step(True)
    y_c, g_c, n_c = step(False)
    assert n_f == 4 and n_c == 0
    prev2, ops.FUSED_NORM2 = ops.FUSED_NORM2, not ops.FUSED_NORM2  # ... and with the block's norm2 inside the same launch (or outside)
    try:
        y_n, g_n, n_n = step(True)
    finally:
        ops.FUSED_NORM2 = prev2
    assert n_n == 4
    assert_close(y_n, y_f, 1e-2, "logits, norm2 fused vs separate")
    for n in g_f:
        from _util import errors as _e
        assert _e(g_n[n], g_f[n])["scale_err"] <= 6e-2, n
    assert_close(y_f, y_c, 1e-2, "logits, train form vs composition")
    assert set(g_f) == set(g_c)
    worst = 0.0
    from _util import errors
    for n in g_f:
        e = errors(g_f[n], g_c[n])
        worst = max(worst, e["scale_err"])
        assert e["scale_err"] <= 6e-2, (n, e)
    print(f"train form vs composition: worst parameter-gradient scale error {worst:.2e}")


@pytest.mark.parametrize("v1,cosine,strategy", [(True, False, "nest_roll"), (True, True, "ring_shift"), (False, True, "nest_grid_shift")])
def test_module_bwd_entry_point_equals_the_autograd_path(v1, cosine, strategy):
    """C ABI: hs_window_attn_module_fwd_train followed by hs_window_attn_module_bwd_chain (one call each way) gives the gradients the
    Python mirror's recorded autograd nodes give (ops.window_attn_module_train) -- same kernels, chained by the library."""
    from heal_swin_amd import ops, _lib
    from heal_swin_amd._lib import check, lib, ptr
    from oracle import tables as T
    B, nside, C, nH, shift = 2, 16, 128, 4, 32
    N = 8 * nside * nside
    g = torch.Generator(device=DEV).manual_seed(9)
    rnd = lambda *s, k=1.0: torch.randn(*s, generator=g, device=DEV) * k  # noqa: E731
    x = (rnd(B, N, C) * 2 + 0.3).to(torch.bfloat16)
    P = dict(wq=rnd(3 * C, C, k=C ** -0.5), bq=rnd(3 * C, k=0.2), wp=rnd(C, C, k=C ** -0.5), bp=rnd(C, k=0.2), bias=rnd(nH, 64, 64),
             hs=torch.rand(nH, generator=g, device=DEV) * (8 if cosine else 0.3) + 0.1)
    if v1:
        P.update(lg=torch.rand(C, generator=g, device=DEV) + 0.5, lb=rnd(C, k=0.2))
    dout = rnd(B, N, C).to(torch.bfloat16)
    if strategy == "nest_roll":
        idx, roll, lab_np = None, shift, T.nest_roll_shift(N, 64, shift)[2]
    else:
        idx_np, _, lab_np = T.nest_grid_shift(nside, 8, 64) if strategy == "nest_grid_shift" else T.ring_shift(nside, 8, 64, 4)
        idx, roll = torch.from_numpy(idx_np).to(torch.int32).to(DEV), 0
    labels = torch.from_numpy(lab_np).to(torch.uint8).to(DEV)
    # ---- autograd path
    xd = x.clone().requires_grad_(True)
    D = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    y = ops.window_attn_module_train(xd, D.get("lg"), D.get("lb"), D["wq"], D["bq"], D["wp"], D["bp"], D["bias"], D["hs"], idx, roll, labels,
                                     nH, 64, cosine)
    y.backward(dout)
    # ---- C ABI, one call each way
    wq16, wp16 = P["wq"].to(torch.bfloat16).contiguous(), P["wp"].to(torch.bfloat16).contiguous()
    out, xn, o = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    qkv = torch.empty(B, N, 3 * C, dtype=torch.bfloat16, device=DEV)
    mean, rstd, lse = torch.empty(B * N, device=DEV), torch.empty(B * N, device=DEV), torch.empty(B, nH, N, device=DEV)
    flags = (_lib.HS_ATTN_COSINE if cosine else 0) | (_lib.HS_ATTN_RESIDUAL if v1 else 0)
    check(lib.hs_window_attn_module_fwd_train(ptr(x), ptr(out), ptr(xn) if v1 else None, ptr(mean) if v1 else None, ptr(rstd) if v1 else None,
                                              ptr(qkv), ptr(o), ptr(lse), ptr(wq16), ptr(P["bq"]), ptr(wp16), ptr(P["bp"]),
                                              ptr(P["lg"]) if v1 else None, ptr(P["lb"]) if v1 else None, ptr(P["bias"]), ptr(P["hs"]),
                                              ptr(idx), roll, ptr(labels), None, None, None, None, None, B, N, C, nH, 64, flags, _lib.HS_BF16, None), "fwd_train")
    assert torch.equal(out, y.detach())
    ws = torch.empty(int(lib.hs_window_attn_module_bwd_chain_workspace(B, N, C, nH, 64)), device=DEV)
    dx = torch.empty_like(x)
    G = dict(wq=torch.empty(3 * C, C, device=DEV), bq=torch.empty(3 * C, device=DEV), wp=torch.empty(C, C, device=DEV), bp=torch.empty(C, device=DEV),
             lg=torch.empty(C, device=DEV), lb=torch.empty(C, device=DEV), bias=torch.empty(nH, 64, 64, device=DEV), hs=torch.empty(nH, device=DEV))
    check(lib.hs_window_attn_module_bwd_chain(ptr(dout), ptr(x), ptr(xn) if v1 else None, ptr(mean) if v1 else None, ptr(rstd) if v1 else None, ptr(qkv),
                                        ptr(o), ptr(lse), ptr(wq16.t().contiguous()), ptr(wp16.t().contiguous()), ptr(P["lg"]) if v1 else None,
                                        ptr(P["bias"]), ptr(P["hs"]), ptr(idx), roll, ptr(labels), ptr(dx), ptr(G["wq"]), ptr(G["bq"]),
                                        ptr(G["wp"]), ptr(G["bp"]), ptr(G["lg"]) if v1 else None, ptr(G["lb"]) if v1 else None, ptr(G["bias"]),
                                        ptr(G["hs"]), ptr(ws), 0, B, N, C, nH, 64, flags, _lib.HS_BF16, None), "module_bwd")
    assert_close(dx, xd.grad, 1e-6, "dx")
    for k in D:
        if k == "hs" and not cosine:
            continue
        assert_close(G[k], D[k].grad, 1e-6, f"d{k}")
