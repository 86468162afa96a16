"""hs_mlp_fused_fwd / hs_mlp_fused_bwd (csrc/mlp_fused.hip): the block's second residual branch
    x + fc2(gelu(fc1(LayerNorm(x))))        (reference swin_hp_transformer.py:337-338, Mlp.forward :38-44, norm2 :262)
in one launch per direction.  Checked through the C ABI against the oracle's formulas (oracle.model.layer_norm / linear / gelu,
evaluated in float64 on the same bf16-rounded inputs), output and every tensor saved for the backward; the autograd node against
the oracle's autograd on the same formulas, and against the three-kernel composition it replaces."""
import ctypes

import pytest
import torch

from _util import GRAD_TOL, TOL, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _L():
    from heal_swin_amd import _lib
    return _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _case(C, rows, seed, scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    H = 4 * C
    t = dict(
        x=(torch.randn((rows, C), generator=g, device=DEV) * 2 + torch.randn((rows, 1), generator=g, device=DEV)).to(BF),
        ln_w=torch.rand(C, generator=g, device=DEV) + 0.5, ln_b=torch.randn(C, generator=g, device=DEV) * 0.2,
        w1=(torch.randn((H, C), generator=g, device=DEV) * scale / C ** 0.5).to(BF), b1=torch.randn(H, generator=g, device=DEV) * 0.3,
        w2=(torch.randn((C, H), generator=g, device=DEV) * scale / H ** 0.5).to(BF), b2=torch.randn(C, generator=g, device=DEV) * 0.3)
    return t


def _oracle_fwd(t, ln=True, residual=True):
    """float64 evaluation of the oracle's formulas with the kernel's rounding points: LayerNorm(x) -> bf16, h -> bf16 for storage
    (gelu takes the unrounded h, as hs_gemm_nt's epilogue does), gelu(h) -> bf16 (the operand of fc2)."""
    from oracle import model as OM
    x = t["x"].double()
    n = OM.layer_norm(x, t["ln_w"].double(), t["ln_b"].double()) if ln else x
    n_r = n.to(BF).double()
    h = OM.linear(n_r, t["w1"].double(), t["b1"].double())
    act = OM.gelu(h)
    y = OM.linear(act.to(BF).double(), t["w2"].double(), t["b2"].double())
    mean = x.mean(1)
    rstd = (x.var(1, unbiased=False) + 1e-5).rsqrt()
    return dict(n=n, h=h, act=act, out=(x + y) if residual else y, mean=mean, rstd=rstd)


@pytest.mark.parametrize("C", [96, 128])
@pytest.mark.parametrize("rows", [32, 96, 8192 + 64])
@pytest.mark.parametrize("ln,residual,keep", [(True, True, True), (True, True, False), (False, False, True), (False, True, True)])
def test_mlp_fused_forward_vs_oracle(C, rows, ln, residual, keep):
    L = _L()
    lib, ptr = L.lib, L.ptr
    t = _case(C, rows, 7 * C + rows)
    H = 4 * C
    out = torch.full((rows, C), 7.0, device=DEV, dtype=BF)
    n = torch.full((rows, C), 7.0, device=DEV, dtype=BF) if (keep and ln) else None
    mean = torch.empty(rows, device=DEV) if (keep and ln) else None
    rstd = torch.empty(rows, device=DEV) if (keep and ln) else None
    h = torch.full((rows, H), 7.0, device=DEV, dtype=BF) if keep else None
    act = torch.full((rows, H), 7.0, device=DEV, dtype=BF) if keep else None
    L.check(lib.hs_mlp_fused_fwd(ptr(t["x"]), ptr(t["ln_w"] if ln else None), ptr(t["ln_b"] if ln else None), ptr(t["w1"]), ptr(t["b1"]),
                                 ptr(t["w2"]), ptr(t["b2"]), ptr(n), ptr(mean), ptr(rstd), ptr(h), ptr(act), ptr(out), rows, C, H,
                                 L.HS_ATTN_RESIDUAL if residual else 0, L.HS_BF16, _stream()), "hs_mlp_fused_fwd")
    ref = _oracle_fwd(t, ln, residual)
    tag = f"mlp_fused fwd C={C} rows={rows}"
    checks = [(out, ref["out"], TOL[BF], "out")]
    if keep:
        checks += [(h, ref["h"], TOL[BF], "h"), (act, ref["act"], TOL[BF], "gelu(h)")]
        if ln:
            checks += [(n, ref["n"], TOL[BF], "LayerNorm(x)"), (mean, ref["mean"], 1e-5, "mean"), (rstd, ref["rstd"], 1e-5, "rstd")]
    bad = []
    for a, b, tol, what in checks:  # (every tensor is judged before the test fails: the message names all that are off)
        try:
            assert_close(a, b, tol, f"{tag} {what}")
        except AssertionError as e:
            bad.append(str(e))
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("C", [96, 128])
@pytest.mark.parametrize("rows", [32, 4096 + 32])
@pytest.mark.parametrize("with_res", [False, True])
def test_mlp_fused_backward_vs_oracle(C, rows, with_res):
    """dh = (dy W2) * gelu'(h), dn = dh W1 against the oracle's autograd through linear / gelu on the same saved h."""
    from oracle import model as OM
    L = _L()
    lib, ptr = L.lib, L.ptr
    t = _case(C, rows, 11 * C + rows)
    H = 4 * C
    g = torch.Generator(device=DEV).manual_seed(5)
    dy = torch.randn((rows, C), generator=g, device=DEV).to(BF)
    h = (torch.randn((rows, H), generator=g, device=DEV) * 1.5).to(BF)
    w2t, w1t = t["w2"].t().contiguous(), t["w1"].t().contiguous()
    dh = torch.full((rows, H), 7.0, device=DEV, dtype=BF)
    dn = torch.full((rows, C), 7.0, device=DEV, dtype=BF)
    res = torch.randn((rows, C), generator=g, device=DEV).to(BF) if with_res else None
    L.check(lib.hs_mlp_fused_bwd(ptr(dy), ptr(h), ptr(w2t), ptr(w1t), ptr(res), ptr(dh), ptr(dn), rows, C, H, L.HS_BF16, _stream()), "hs_mlp_fused_bwd")
    hd = h.double().requires_grad_(True)
    act = OM.gelu(hd)
    (dact,) = [dy.double() @ t["w2"].double()]
    (dh_ref,) = torch.autograd.grad(act, hd, dact)
    dn_ref = dh_ref.to(BF).double() @ t["w1"].double()
    if with_res:  # v2 placement: the residual path's gradient rides on the epilogue
        dn_ref = dn_ref + res.double()
    tag = f"mlp_fused bwd C={C} rows={rows}"
    assert_close(dh, dh_ref, TOL[BF], tag + " dh")
    assert_close(dn, dn_ref, TOL[BF], tag + " dn")


@pytest.mark.parametrize("C", [96, 128])
@pytest.mark.parametrize("rows", [64, 8192 + 32])
def test_mlp_fused_forward_post_norm_vs_oracle(C, rows):
    """v2 placement through the C ABI: out = x + LayerNorm(fc2(gelu(fc1(x)))), the kept tensor is mlp(x) with ITS statistics."""
    from oracle import model as OM
    L = _L()
    lib, ptr = L.lib, L.ptr
    t = _case(C, rows, 13 * C + rows)
    H = 4 * C
    out = torch.full((rows, C), 7.0, device=DEV, dtype=BF)
    m = torch.full((rows, C), 7.0, device=DEV, dtype=BF)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    h, act = torch.empty((rows, H), device=DEV, dtype=BF), torch.empty((rows, H), device=DEV, dtype=BF)
    L.check(lib.hs_mlp_fused_fwd(ptr(t["x"]), ptr(t["ln_w"]), ptr(t["ln_b"]), ptr(t["w1"]), ptr(t["b1"]), ptr(t["w2"]), ptr(t["b2"]), ptr(m),
                                 ptr(mean), ptr(rstd), ptr(h), ptr(act), ptr(out), rows, C, H, L.HS_ATTN_RESIDUAL | L.HS_MLP_NORM_AFTER,
                                 L.HS_BF16, _stream()), "hs_mlp_fused_fwd")
    x = t["x"].double()
    hr = OM.linear(x, t["w1"].double(), t["b1"].double())
    mr = OM.linear(OM.gelu(hr).to(BF).double(), t["w2"].double(), t["b2"].double())
    mb = mr.to(BF).double()  # the LayerNorm reads the rounded rows, as the composed path's kernel does
    ref = x + OM.layer_norm(mb, t["ln_w"].double(), t["ln_b"].double())
    tag = f"mlp_fused fwd post-norm C={C} rows={rows}"
    assert_close(h, hr, TOL[BF], tag + " h")
    assert_close(m, mr, TOL[BF], tag + " mlp(x)")
    assert_close(out, ref, TOL[BF], tag + " out")
    assert_close(mean, m.double().mean(1), 1e-5, tag + " mean")
    assert_close(rstd, (m.double().var(1, unbiased=False) + 1e-5).rsqrt(), 1e-5, tag + " rstd")


@pytest.mark.parametrize("C", [96, 128])
def test_fused_mlp_block_post_norm_autograd_vs_oracle(C):
    """ops.fused_mlp_block(post_norm=True): output and every gradient against the oracle's float64 autograd over its own formulas."""
    from heal_swin_amd import ops
    from oracle import model as OM
    rows = 2048
    t = _case(C, rows, 5 * C)
    g = torch.Generator(device=DEV).manual_seed(19)
    dy = torch.randn((rows, C), generator=g, device=DEV).to(BF)
    names = ("ln_w", "ln_b", "w1", "b1", "w2", "b2")
    x = t["x"].clone().requires_grad_(True)
    ps = {k: t[k].float().clone().requires_grad_(True) for k in names}
    out = ops.fused_mlp_block(x, ps["ln_w"], ps["ln_b"], ps["w1"], ps["b1"], ps["w2"], ps["b2"], post_norm=True)
    out.backward(dy)
    xo = t["x"].double().requires_grad_(True)
    po = {k: t[k].double().requires_grad_(True) for k in names}
    yo = xo + OM.layer_norm(OM.linear(OM.gelu(OM.linear(xo, po["w1"], po["b1"])), po["w2"], po["b2"]), po["ln_w"], po["ln_b"])
    yo.backward(dy.double())
    assert_close(out, yo.detach(), TOL[BF], f"fused_mlp_block post-norm C={C} out")
    assert_close(x.grad, xo.grad, GRAD_TOL[BF], f"fused_mlp_block post-norm C={C} dx")
    for k in names:
        assert_close(ps[k].grad, po[k].grad, GRAD_TOL[BF], f"fused_mlp_block post-norm C={C} d{k}")


@pytest.mark.parametrize("C", [96, 128])
def test_fused_mlp_block_autograd_vs_oracle_and_composition(C):
    """ops.fused_mlp_block: output and EVERY gradient (x, norm2, fc1, fc2) against the oracle's autograd over its own formulas in
    float64, and against LayerNorm -> hs_gemm_nt(GELU) -> hs_gemm_nt(residual) -- the composition the block ran before."""
    from heal_swin_amd import ops
    from oracle import model as OM
    rows = 2048
    t = _case(C, rows, 3 * C)
    g = torch.Generator(device=DEV).manual_seed(9)
    dy = torch.randn((rows, C), generator=g, device=DEV).to(BF)
    names = ("ln_w", "ln_b", "w1", "b1", "w2", "b2")

    def leaves():
        x = t["x"].clone().requires_grad_(True)
        ps = {k: t[k].float().clone().requires_grad_(True) for k in names}
        return x, ps

    assert ops.fused_mlp_ok(t["x"], 4 * C)
    x, ps = leaves()
    out = ops.fused_mlp_block(x, ps["ln_w"], ps["ln_b"], ps["w1"], ps["b1"], ps["w2"], ps["b2"])
    out.backward(dy)
    fused = dict(out=out.detach(), x=x.grad, **{k: ps[k].grad for k in names})

    # oracle: float64 autograd on the bf16-rounded inputs (no intermediate rounding)
    xo = t["x"].double().requires_grad_(True)
    po = {k: t[k].double().requires_grad_(True) for k in names}
    yo = xo + OM.linear(OM.gelu(OM.linear(OM.layer_norm(xo, po["ln_w"], po["ln_b"]), po["w1"], po["b1"])), po["w2"], po["b2"])
    yo.backward(dy.double())
    assert_close(fused["out"], yo.detach(), TOL[BF], f"fused_mlp_block C={C} out")
    assert_close(fused["x"], xo.grad, GRAD_TOL[BF], f"fused_mlp_block C={C} dx")
    for k in names:
        assert_close(fused[k], po[k].grad, GRAD_TOL[BF], f"fused_mlp_block C={C} d{k}")

    # the composition it replaces (same kernels for the parameter gradients): agreement well inside the bf16 bound
    prev = ops.FUSED_MLP
    try:
        ops.FUSED_MLP = False
        x2, p2 = leaves()
        n2, xa = ops.layer_norm_passthrough(x2, p2["ln_w"], p2["ln_b"])
        out2 = ops.mlp(n2, p2["w1"], p2["b1"], p2["w2"], p2["b2"], residual=xa)
        out2.backward(dy)
    finally:
        ops.FUSED_MLP = prev
    assert_close(fused["out"], out2.detach(), 1e-2, f"fused vs composed C={C} out")
    assert_close(fused["x"], x2.grad, 2e-2, f"fused vs composed C={C} dx")
    for k in names:
        assert_close(fused[k], p2[k].grad, 2e-2, f"fused vs composed C={C} d{k}")


def test_fused_mlp_rejects_unsupported_shapes():
    L = _L()
    assert L.lib.hs_mlp_fused_supported(128, 512, L.HS_BF16) and L.lib.hs_mlp_fused_supported(96, 384, L.HS_BF16)
    assert not L.lib.hs_mlp_fused_supported(256, 1024, L.HS_BF16) and not L.lib.hs_mlp_fused_supported(128, 256, L.HS_BF16)
    assert not L.lib.hs_mlp_fused_supported(128, 512, L.HS_F32)
    x = torch.zeros((48, 128), device=DEV, dtype=BF)  # rows not a multiple of 32
    w = torch.zeros((512, 128), device=DEV, dtype=BF)
    with pytest.raises(AssertionError):
        L.check(L.lib.hs_mlp_fused_fwd(L.ptr(x), None, None, L.ptr(w), None, L.ptr(w), None, None, None, None, None, None, L.ptr(x), 48, 128, 512,
                                       0, L.HS_BF16, _stream()), "hs_mlp_fused_fwd")


@pytest.mark.parametrize("rows,n_out,k_in", [(8192, 128, 512), (4096 + 32, 96, 384), (300, 128, 128)])
def test_linear_wgrad_gelu_equals_wgrad_of_gelu(rows, n_out, k_in):
    """hs_linear_wgrad_gelu(dy, h) = dY^T gelu(h): against float64 with the oracle's gelu on the same bf16 h, and against
    hs_linear_wgrad on a materialised bf16 gelu(h) (the form it replaces; they differ by that tensor's rounding only)."""
    from oracle import model as OM
    L = _L()
    lib, ptr = L.lib, L.ptr
    assert lib.hs_linear_wgrad_gelu_supported(rows, n_out, k_in, L.HS_BF16)
    g = torch.Generator(device=DEV).manual_seed(rows)
    dy = torch.randn((rows, n_out), generator=g, device=DEV).to(BF)
    h = (torch.randn((rows, k_in), generator=g, device=DEV) * 1.5).to(BF)
    nws = int(lib.hs_linear_wgrad_workspace(rows, n_out, k_in))
    dw, db = torch.zeros((n_out, k_in), device=DEV), torch.zeros(n_out, device=DEV)
    L.check(lib.hs_linear_wgrad_gelu(ptr(dy), ptr(h), ptr(dw), ptr(db), ptr(torch.empty(nws, device=DEV)), rows, n_out, k_in, 0, L.HS_BF16,
                                     _stream()), "hs_linear_wgrad_gelu")
    act64 = OM.gelu(h.double())
    ref = dy.double().t() @ act64.to(BF).double()
    assert_close(dw, ref, 3e-3, f"wgrad_gelu {rows}x{n_out}x{k_in} dW")
    assert_close(db, dy.double().sum(0), 1e-3, f"wgrad_gelu {rows}x{n_out}x{k_in} db")
    dw2 = torch.zeros_like(dw)
    act = act64.to(BF)
    L.check(lib.hs_linear_wgrad(ptr(dy), ptr(act), ptr(dw2), None, ptr(torch.empty(nws, device=DEV)), rows, n_out, k_in, 0, L.HS_BF16,
                                _stream()), "hs_linear_wgrad")
    assert_close(dw, dw2, 3e-3, f"wgrad_gelu vs wgrad(gelu) {rows}x{n_out}x{k_in}")
    assert not lib.hs_linear_wgrad_gelu_supported(98304, 2048, 512, L.HS_BF16)  # the 256-wide tiles keep the plain form


@pytest.mark.parametrize("keep", [False, True])
def test_fused_mlp_block_with_and_without_kept_activation(keep):
    """The two forms of the fused block's saved state -- h only (fc2's weight gradient re-applies GELU) and h + gelu(h) -- give the
    same gradients to bf16 rounding."""
    from heal_swin_amd import ops
    C, rows = 128, 4096
    t = _case(C, rows, 77)
    g = torch.Generator(device=DEV).manual_seed(3)
    dy = torch.randn((rows, C), generator=g, device=DEV).to(BF)
    names = ("ln_w", "ln_b", "w1", "b1", "w2", "b2")
    res = {}
    prev = ops.MLP_KEEP_ACT
    try:
        for mode in (keep, not keep):
            ops.MLP_KEEP_ACT = mode
            x = t["x"].clone().requires_grad_(True)
            ps = {k: t[k].float().clone().requires_grad_(True) for k in names}
            out = ops.fused_mlp_block(x, ps["ln_w"], ps["ln_b"], ps["w1"], ps["b1"], ps["w2"], ps["b2"])
            out.backward(dy)
            res[mode] = dict(out=out.detach(), x=x.grad, **{k: ps[k].grad for k in names})
    finally:
        ops.MLP_KEEP_ACT = prev
    assert torch.equal(res[True]["out"], res[False]["out"])
    for k in ("x",) + names:
        assert_close(res[False][k], res[True][k], 3e-3, f"fused mlp keep_act on/off d{k}")


def _mask_of(seed, p, shape):
    """The generator's keep mask (x 1 / (1 - p)) for a [rows, width] bf16 tensor, read off the elementwise GELU kernel: gelu(16) = 16."""
    L = _L()
    src = torch.full(shape, 16.0, device=DEV, dtype=BF)
    dst = torch.empty_like(src)
    L.check(L.lib.hs_gelu_fwd(L.ptr(src), L.ptr(dst), src.numel(), float(p), int(seed), L.HS_BF16, _stream()), "hs_gelu_fwd")
    return (dst.float() != 0).double() / (1.0 - p)


@pytest.mark.parametrize("C", [96, 128])
@pytest.mark.parametrize("drop_p,with_path", [(0.1, True), (0.25, False), (0.0, True)])
def test_fused_mlp_block_stochastic_vs_oracle_and_composition(C, drop_p, with_path):
    """Train-mode v2 block x + rs * LayerNorm(drop(fc2(drop(gelu(fc1(x)))))) in one launch per direction (hs_mlp_fused_drop_fwd / _bwd):
    against the oracle's float64 autograd with the generator's own masks, and against the composition (ops.mlp with hidden dropout ->
    hs_layernorm_drop) on the same seeds -- the masks are functions of (seed, element index), so both paths drop the same elements."""
    from heal_swin_amd import ops
    from oracle import model as OM
    B, per = 4, 1024
    rows = B * per
    t = _case(C, rows, 7 * C)
    g = torch.Generator(device=DEV).manual_seed(23)
    dy = torch.randn((B, per, C), generator=g, device=DEV).to(BF)
    rs = torch.tensor([0.0, 1.25, 1.25, 1.25], device=DEV) if with_path else None
    seeds = (0x1234567890ABCDEF, 0x0FEDCBA987654321)
    names = ("ln_w", "ln_b", "w1", "b1", "w2", "b2")

    def run(fused):
        x = t["x"].clone().view(B, per, C).requires_grad_(True)
        ps = {k: t[k].float().clone().requires_grad_(True) for k in names}
        if fused:
            out = ops.fused_mlp_block(x, ps["ln_w"], ps["ln_b"], ps["w1"], ps["b1"], ps["w2"], ps["b2"], post_norm=True, row_scale=rs,
                                      drop_p=drop_p, seeds=seeds)
        else:
            m = ops.mlp(x, ps["w1"], ps["b1"], ps["w2"], ps["b2"], drop_p=drop_p, seed=seeds[0])
            out = ops.layer_norm(m, ps["ln_w"], ps["ln_b"], residual=x, row_scale=rs, drop_p=drop_p, seed=seeds[1])
        out.backward(dy)
        return out.detach(), x.grad, {k: ps[k].grad for k in names}

    assert ops.fused_mlp_stochastic_ok(t["x"].view(B, per, C), True)
    out_f, dx_f, gp_f = run(True)
    out_c, dx_c, gp_c = run(False)
    mh = _mask_of(seeds[0], drop_p, (rows, 4 * C)) if drop_p else 1.0
    mo = _mask_of(seeds[1], drop_p, (rows, C)) if drop_p else 1.0
    if drop_p:
        frac = 1.0 - (mh != 0).double().mean().item()
        assert abs(frac - drop_p) < 0.01, f"hidden drop fraction {frac} vs {drop_p}"
    rsv = rs.double().repeat_interleave(per).view(rows, 1) if with_path else 1.0
    xo = t["x"].double().requires_grad_(True)
    po = {k: t[k].double().requires_grad_(True) for k in names}
    mlp_o = OM.linear(OM.gelu(OM.linear(xo, po["w1"], po["b1"])) * mh, po["w2"], po["b2"])
    yo = xo + rsv * OM.layer_norm(mlp_o * mo, po["ln_w"], po["ln_b"])
    yo.backward(dy.double().view(rows, C))
    tag = f"stochastic fused_mlp_block C={C} p={drop_p} path={with_path}"
    assert_close(out_f.view(rows, C), yo.detach(), TOL[BF], tag + " out vs oracle")
    assert_close(dx_f.view(rows, C), xo.grad, GRAD_TOL[BF], tag + " dx vs oracle")
    for k in names:
        assert_close(gp_f[k], po[k].grad, GRAD_TOL[BF], tag + f" d{k} vs oracle")
    assert_close(out_f, out_c, TOL[BF], tag + " out vs composition")
    assert_close(dx_f, dx_c, GRAD_TOL[BF], tag + " dx vs composition")
    for k in names:
        assert_close(gp_f[k], gp_c[k], GRAD_TOL[BF], tag + f" d{k} vs composition")
    if with_path:  # a dropped sample passes x through unchanged
        assert torch.equal(out_f[0], t["x"].view(B, per, C)[0])


def test_model_with_paper_drop_rates_fused_and_composed_mlp_agree():
    """The train-mode fused Mlp block inside the model (v2 placement, drop_rate = attn_drop_rate = drop_path_rate = 0.1, the paper's
    rates): with the host seed stream and torch's CUDA generator reset, the run that takes `hs_mlp_fused_drop_*` at stage 0 and the run
    on the composition (`ops.FUSED_MLP = False`) draw the same seeds in the same order and the same DropPath factors -- logits and every
    parameter gradient agree to bf16 rounding; and the fused path was actually taken."""
    from heal_swin_amd import ops
    from heal_swin_amd.data_spec import DataSpec
    from heal_swin_amd.models_torch.swin_hp_transformer import SwinHPTransformerConfig, SwinHPTransformerSys
    cfg = dict(patch_size=4, window_size=64, shift_size=4, shift_strategy="ring_shift", rel_pos_bias="flat", embed_dim=96,
               depths=[2, 2], num_heads=[3, 6], mlp_ratio=4.0, qkv_bias=True, qk_scale=None, use_cos_attn=True, drop_rate=0.1,
               attn_drop_rate=0.1, drop_path_rate=0.1, use_v2_norm_placement=True, ape=False)
    spec = dict(dim_in=8 * 16 * 16, f_in=3, f_out=12, base_pix=8, class_names=[])
    torch.manual_seed(0)
    model = SwinHPTransformerSys(SwinHPTransformerConfig(**cfg), DataSpec(**spec)).to(DEV).train()
    model.compute_dtype = BF
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randint(0, 256, (4, 3, spec["dim_in"]), generator=g, device=DEV).float()
    dy = torch.randn(4, 12, spec["dim_in"], generator=g, device=DEV)
    calls = []
    orig = ops.lib.hs_mlp_fused_drop_fwd

    def run(fused):
        prev = ops.FUSED_MLP
        ops.FUSED_MLP = fused
        try:
            torch.manual_seed(1234)  # the CPU stream of ops._draw_seed and the CUDA generator of DropPath.sample_scale
            model.zero_grad(set_to_none=True)
            y = model(x)
            y.backward(dy)
            return y.detach().float(), {n: p.grad.detach().float().clone() for n, p in model.named_parameters()}
        finally:
            ops.FUSED_MLP = prev

    class Spy:  # counts the launches of the stochastic fused kernel without changing them
        def __call__(self, *a):
            calls.append(1)
            return orig(*a)
    ops.lib.hs_mlp_fused_drop_fwd = Spy()
    try:
        y_f, g_f = run(True)
    finally:
        ops.lib.hs_mlp_fused_drop_fwd = orig
    assert len(calls) == 4, f"stage 0 (2 encoder + 2 decoder blocks) should take the fused train-mode Mlp block, took it {len(calls)} times"
    y_c, g_c = run(False)
    y_f2, _ = run(True)
    assert torch.equal(y_f, y_f2)  # same seeds, same masks: the stochastic step is reproducible
    assert_close(y_f, y_c, TOL[BF], "paper drop rates: logits, fused vs composed Mlp")
    for n in g_f:
        tol = 8e-2 if n.endswith(("logit_scale", "relative_position_bias_table")) else 4e-2
        assert_close(g_f[n], g_c[n], tol, f"paper drop rates: d{n}, fused vs composed Mlp")
