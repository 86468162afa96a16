"""HIP kernels (through the C ABI / autograd ops) vs the oracle and the golden vectors.  Needs an MI355X."""
import os
import sys

import numpy as np
import pytest
import torch

from _golden import REFINIT_WA_CASES, case, state_dict
from _util import TOL, GRAD_TOL, assert_close

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def _mods():
    from heal_swin_amd import ops
    from heal_swin_amd.models_torch import swin_hp_transformer as M
    from heal_swin_amd.models_torch import hp_shifting as S
    return ops, M, S


# ----------------------------------------------------------------------------- WindowAttention vs golden
WA_CASES = [f"{a}_{m}" for a in ("scaled", "cos") for m in ("nomask", "rollmask", "ringmask")]


# The cosine-attention goldens push one head's logit_scale to the clamp (x100, ref :144-146).  A x100 logit gain turns the
# 2^-9 relative rounding of bf16 q/k inputs into O(0.4) logit noise, so for those cases bf16 is bounded at 6e-2 here; the
# kernel arithmetic itself is held to the tight bound against the oracle on identically rounded inputs
# (test_attn_core_vs_oracle), and default-initialised cosine attention (x10) meets 1e-2 (test_gpu_model).
def _bf16_slack(name, dtype):
    """STRESS case, not the north_star bound: the cosine goldens pin one head's logit_scale at the x100 clamp (ref :144-146),
    which turns bf16's 2^-9 rounding of q, k into O(0.4) logit noise; bf16 is bounded at 6e-2 here, the kernel arithmetic at
    the tight bound in test_attn_core_vs_oracle (identically rounded inputs) and the 1e-2 logit bound on default-initialised
    models in tests/test_gpu_baseline_configs.py."""
    return 6.0 if (dtype == torch.bfloat16 and name.startswith("cos")) else 1.0


def _run_module(mod, c, fwd, dtype, slack=1.0):
    mod = mod.to(DEV)
    x = torch.from_numpy(c["x"]).to(DEV).to(dtype).requires_grad_(True)
    y = fwd(mod, x)
    assert y.dtype == dtype
    assert_close(y, c["y"], TOL[dtype] * slack, "y")
    y.backward(torch.from_numpy(c["dy"]).to(DEV).to(dtype))
    assert_close(x.grad, c["dx"], GRAD_TOL[dtype] * slack, "dx")
    params = dict(mod.named_parameters())
    for k, g in c["grad"].items():
        got = params[k].grad
        got = torch.zeros_like(params[k]) if got is None else got
        assert_close(got, g, GRAD_TOL[dtype] * slack, "grad " + k)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", WA_CASES)
def test_window_attention_golden(name, dtype):
    _, M, _ = _mods()
    c = case("modules", "window_attention/" + name)
    wa = M.WindowAttention(96, 64, 3, rel_pos_bias="flat", use_cos_attn=name.startswith("cos"))
    wa.load_state_dict(state_dict(c))
    mask = torch.from_numpy(c["mask"].astype(np.float32)) if "mask" in c else None
    _run_module(wa, c, lambda m, x: m(x, mask=mask), dtype, _bf16_slack(name, dtype))


def _zero_floor(c, k):
    """Absolute error accepted for a bias gradient that is exactly zero in exact arithmetic (the k bias of attention: softmax is
    shift-invariant along the keys): 1e-4 of the scale of its weight's gradient."""
    return 1e-4 * float(np.abs(c["grad"][k.replace(".bias", ".weight")]).max()) if k.endswith(".bias") else 0.0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", REFINIT_WA_CASES)
def test_window_attention_reference_scale_golden(name, dtype):
    """WindowAttention (C 128, 4 heads of 32, window 64 -- stage 0 of HEAL-SWIN-B) at the reference's OWN initialisation scale
    (trunc-normal 0.02, logit_scale ln 10, table N(0, 0.02)), tensors produced by the reference itself (refinit.npz):
    north_star's 1e-3 (fp32) / 1e-2 (bf16) on the output with NO multiplier; gradients at 1e-3 / 3e-2 of each tensor's own
    scale, d logit_scale included."""
    _, M, _ = _mods()
    c = case("refinit", "window_attention/" + name)
    wa = M.WindowAttention(128, 64, 4, rel_pos_bias="flat", use_cos_attn=name.startswith("cos"))
    wa.load_state_dict(state_dict(c))
    mask = torch.from_numpy(c["mask"].astype(np.float32)) if "mask" in c else None
    wa = wa.to(DEV)
    x = torch.from_numpy(c["x"]).to(DEV).to(dtype).requires_grad_(True)
    y = wa(x, mask=mask)
    assert_close(y, c["y"], TOL[dtype], "refinit y")
    y.backward(torch.from_numpy(c["dy"]).to(DEV).to(dtype))
    assert_close(x.grad, c["dx"], GRAD_TOL[dtype], "refinit dx")
    params = dict(wa.named_parameters())
    for k, g in c["grad"].items():
        got = params[k].grad
        got = torch.zeros_like(params[k]) if got is None else got
        # d logit_scale: one scalar per head summed over every (window, query, key): 5e-2 of its value in bf16
        tol = 5e-2 if (k.endswith("logit_scale") and dtype == torch.bfloat16) else GRAD_TOL[dtype]
        assert_close(got, g, tol, "refinit grad " + k, floor=_zero_floor(c, k))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tag,C,nh,ws", [("w16", 16, 2, 16), ("w4", 2, 1, 4)])
def test_window_attention_plain_golden(tag, C, nh, ws, dtype):
    _, M, _ = _mods()
    c = case("modules", "window_attention/plain_" + tag)
    wa = M.WindowAttention(C, ws, nh, rel_pos_bias=None, qkv_bias=False)
    wa.load_state_dict(state_dict(c))
    _run_module(wa, c, lambda m, x: m(x), dtype)


# ----------------------------------------------------------------------------- attention core vs oracle, larger / odd shapes
def _oracle_core(qkv, bias, hscale, idx, labels, nH, Ws, cosine):
    """oracle formulas for the fused op on CPU fp32 (oracle.model primitives)."""
    from oracle import model as OM
    B, N, C3 = qkv.shape
    C = C3 // 3
    hd = C // nH
    xs = qkv if idx is None else qkv[:, idx]
    t = xs.reshape(B, N // Ws, Ws, 3, nH, hd)
    q, k, v = (t[:, :, :, i].permute(0, 1, 3, 2, 4) for i in range(3))  # [B, nW, nH, Ws, hd]
    if cosine:
        s = OM.l2_normalize(q) @ OM.l2_normalize(k).transpose(-1, -2)
    else:
        s = q @ k.transpose(-1, -2)
    s = s * hscale.reshape(1, 1, nH, 1, 1)
    if bias is not None:
        s = s + bias[None, None]
    if labels is not None:
        lab = labels.reshape(N // Ws, Ws)
        s = s + ((lab[:, :, None] != lab[:, None, :]).float() * -100.0)[None, :, None]
    o = (OM.softmax_lastdim(s) @ v).permute(0, 1, 3, 2, 4).reshape(B, N, C)
    if idx is not None:
        inv = torch.empty_like(idx)
        inv[idx] = torch.arange(N)
        o = o[:, inv]
    return o


CORE_CASES = [
    # B, nside, C, nH, Ws, strategy, shift, cosine, bias
    (2, 16, 96, 3, 64, "ring_shift", 4, True, True),
    (2, 16, 128, 4, 64, "nest_roll", 32, False, True),
    (1, 16, 64, 2, 64, "nest_grid_shift", 32, False, False),
    (3, 8, 48, 3, 16, "nest_roll", 8, True, True),
    (2, 8, 24, 3, 16, "none", 0, False, True),
    (1, 8, 40, 5, 4, "ring_shift", 2, True, False),
    (1, 16, 32, 1, 256, "nest_roll", 128, False, True),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,nside,C,nH,Ws,strategy,shift,cosine,use_bias", CORE_CASES)
def test_attn_core_vs_oracle(B, nside, C, nH, Ws, strategy, shift, cosine, use_bias, dtype):
    ops, _, _ = _mods()
    from oracle import tables as T
    N = 8 * nside * nside
    g = torch.Generator().manual_seed(7)
    qkv = torch.randn(B, N, 3 * C, generator=g)
    bias = torch.randn(nH, Ws, Ws, generator=g) if use_bias else None
    hscale = torch.rand(nH, generator=g) * (8 if cosine else 0.3) + 0.1
    dout = torch.randn(B, N, C, generator=g)
    if strategy == "none":
        idx = labels = None
    else:
        fn = {"nest_roll": lambda: T.nest_roll_shift(N, Ws, shift), "nest_grid_shift": lambda: T.nest_grid_shift(nside, 8, Ws),
              "ring_shift": lambda: T.ring_shift(nside, 8, Ws, shift)}[strategy]
        idx_np, _, lab_np = fn()
        idx, labels = torch.from_numpy(idx_np), torch.from_numpy(lab_np)

    # oracle on the values the kernel actually sees (inputs rounded to the activation dtype)
    qkv_r = qkv.to(dtype).float().clone().requires_grad_(True)
    bias_r = None if bias is None else bias.clone().requires_grad_(True)
    hs_r = hscale.clone().requires_grad_(True)
    o_ref = _oracle_core(qkv_r, bias_r, hs_r, idx, labels, nH, Ws, cosine)
    o_ref.backward(dout.to(dtype).float())

    qkv_d = qkv.to(DEV).to(dtype).requires_grad_(True)
    bias_d = None if bias is None else bias.to(DEV).requires_grad_(True)
    hs_d = hscale.to(DEV).requires_grad_(True)
    idx_d = None if idx is None else idx.to(torch.int32).to(DEV)
    lab_d = None if labels is None else labels.to(torch.uint8).to(DEV)
    use_roll = strategy == "nest_roll"
    o = ops.window_attn_core(qkv_d, bias_d, hs_d, None if use_roll else idx_d, shift if use_roll else 0, lab_d, nH, Ws, cosine)
    assert_close(o, o_ref, TOL[dtype], "out")
    o.backward(dout.to(DEV).to(dtype))
    assert_close(qkv_d.grad, qkv_r.grad, GRAD_TOL[dtype], "dqkv")
    if bias is not None:
        assert_close(bias_d.grad, bias_r.grad, GRAD_TOL[dtype], "dbias")
    if cosine:
        assert_close(hs_d.grad, hs_r.grad, GRAD_TOL[dtype], "dhead_scale")


# ----------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,width", [(7, 2), (130, 16), (1000, 96), (513, 128), (300, 384), (64, 1536), (33, 2048), (50, 100)])
@pytest.mark.parametrize("residual", [False, True])
def test_layernorm_vs_oracle(rows, width, residual, dtype):
    ops, _, _ = _mods()
    from oracle import model as OM
    g = torch.Generator().manual_seed(rows * 31 + width)
    x = (torch.randn(rows, width, generator=g) * 3 + 1.5)
    w = 1 + 0.3 * torch.randn(width, generator=g)
    b = 0.2 * torch.randn(width, generator=g)
    r = torch.randn(rows, width, generator=g) if residual else None
    dy = torch.randn(rows, width, generator=g)

    xr = x.to(dtype).float().clone().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    rr = None if r is None else r.to(dtype).float().clone().requires_grad_(True)
    y_ref = OM.layer_norm(xr, wr, br) + (0 if rr is None else rr)
    y_ref.backward(dy.to(dtype).float())

    xd = x.to(DEV).to(dtype).requires_grad_(True)
    wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    rd = None if r is None else r.to(DEV).to(dtype).requires_grad_(True)
    y = ops.layer_norm(xd, wd, bd, rd)
    assert_close(y, y_ref, TOL[dtype], "y")
    y.backward(dy.to(DEV).to(dtype))
    assert_close(xd.grad, xr.grad, GRAD_TOL[dtype], "dx")
    assert_close(wd.grad, wr.grad, GRAD_TOL[dtype], "dgamma")
    assert_close(bd.grad, br.grad, GRAD_TOL[dtype], "dbeta")
    if residual:
        assert_close(rd.grad, rr.grad, GRAD_TOL[dtype], "dres")


# ----------------------------------------------------------------------------- compensated residual stream
@pytest.mark.parametrize("placement", ["v1", "v2"])
def test_compensated_stream_tracks_the_fp32_sum(placement):
    """36 residual adds in a row (the depth of HEAL-SWIN-B's stage 2) in bf16: the plain stream rounds after every add, the
    compensated one (hi + lo, `hs_layernorm_fwd_ex`) follows the fp32 sum to 2^-16; the normalised output is LN of the
    un-rounded sum; the gradients are those of the plain kernels."""
    ops, _, _ = _mods()
    from oracle import model as OM
    g = torch.Generator().manual_seed(3)
    rows, width, depth = 1000, 512, 36
    x0 = torch.randn(rows, width, generator=g).to(torch.bfloat16)
    w = 1 + 0.3 * torch.randn(width, generator=g)
    b = 0.2 * torch.randn(width, generator=g)
    branches = [(0.3 * torch.randn(rows, width, generator=g)).to(torch.bfloat16) for _ in range(depth)]
    wd, bd = w.to(DEV), b.to(DEV)
    ref = x0.float()
    plain, comp, lo = x0.to(DEV), x0.to(DEV), None
    for t in branches:
        td = t.to(DEV)
        if placement == "v1":   # s = a + t, y = LN(s)
            ref = ref + t.float()
            y_ref = OM.layer_norm(ref, w, b)
            plain, _ = ops.add_layer_norm(plain, td, wd, bd)
            comp, y, lo = ops.add_layer_norm_stream(comp, lo, td, wd, bd)
        else:                   # y = res + LN(t): the output is the stream
            ref = ref + OM.layer_norm(t.float(), w, b)
            plain = ops.layer_norm(td, wd, bd, plain)
            comp, lo = ops.layer_norm_stream(td, wd, bd, comp, res_lo=lo)
    scale = float(ref.abs().max())
    err_plain = float((plain.float().cpu() - ref).abs().max()) / scale
    err_hi = float((comp.float().cpu() - ref).abs().max()) / scale
    err_comp = float((comp.float().cpu() + lo.float().cpu() - ref).abs().max()) / scale
    import conftest
    conftest.NOTES.append(f"residual stream after {depth} bf16 adds ({placement}): plain {err_plain:.2e}, compensated hi {err_hi:.2e}, "
                          f"hi + lo {err_comp:.2e} of the fp32 sum's scale")
    assert err_comp < 3e-5 and err_comp < err_plain / 50 and err_hi <= 2.0 ** -8
    if placement == "v1":
        assert_close(y, y_ref, 6e-3, "LN of the compensated sum")  # (bf16 output rounding only)
    # gradients: identical to the plain kernels' (the remainder is not differentiable)
    a = x0.to(DEV).requires_grad_(True)
    t = branches[0].to(DEV).requires_grad_(True)
    wg, bg = wd.clone().requires_grad_(True), bd.clone().requires_grad_(True)
    dy = torch.randn(rows, width, generator=g).to(torch.bfloat16).to(DEV)
    got = {}
    for mode in ("plain", "comp"):
        for v in (a, t, wg, bg):
            v.grad = None
        if placement == "v1":
            out = ops.add_layer_norm(a, t, wg, bg) if mode == "plain" else ops.add_layer_norm_stream(a, None, t, wg, bg)[:2]
            (out[0].float() * 0.5 + out[1].float() * dy.float()).sum().backward()
        else:
            out = ops.layer_norm(t, wg, bg, a) if mode == "plain" else ops.layer_norm_stream(t, wg, bg, a)[0]
            (out.float() * dy.float()).sum().backward()
        got[mode] = [v.grad.clone() for v in (a, t, wg, bg)]
    for p_, c_ in zip(got["plain"], got["comp"]):
        assert_close(c_, p_, 2e-2 if placement == "v1" else 1e-6, "gradient of the compensated call")


# ----------------------------------------------------------------------------- shifters (standalone gather) bit-exact
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.uint8])
def test_shifters_bit_exact(dtype):
    _, _, S = _mods()
    from oracle import tables as T
    nside, ws = 16, 16
    N = 8 * nside * nside
    x = torch.arange(2 * N * 6).reshape(2, N, 6).to(dtype).to(DEV)
    cases = [
        (S.NestRollShift(8, N, ws), T.nest_roll_shift(N, ws, 8)),
        (S.NestGridShift(nside, 8, ws), T.nest_grid_shift(nside, 8, ws)),
        (S.RingShift(nside, 8, ws, 4), T.ring_shift(nside, 8, ws, 4)),
    ]
    for sh, (idx, inv, lab) in cases:
        xs = sh.shift(x)
        assert torch.equal(xs.cpu(), x.cpu()[:, torch.from_numpy(idx)])
        assert torch.equal(sh.shift_back(xs).cpu(), x.cpu())
        assert np.array_equal(sh.get_mask(False).numpy().astype(np.int64), lab)
        assert np.array_equal(sh.get_mask().numpy().astype(np.int64), T.attn_mask_from_labels(lab, ws))


def test_gather_rows_autograd():
    ops, _, S = _mods()
    sh = S.RingShift(8, 8, 16, 4)
    x = torch.randn(2, 512, 24, device=DEV, requires_grad=True)
    y = sh.shift(x)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    assert torch.equal(x.grad.cpu(), w.cpu()[:, sh.back_shift_idcs])


# ----------------------------------------------------------------------------- loud failure without a GPU tensor
def test_cpu_tensors_are_rejected():
    ops, _, _ = _mods()
    with pytest.raises(RuntimeError):
        ops.layer_norm(torch.randn(4, 8), torch.ones(8), torch.zeros(8))


# ----------------------------------------------------------------------------- Linear weight / bias gradient
@pytest.mark.parametrize("rows,n_out,k_in", [(4096, 128, 128), (5000, 384, 128), (777, 96, 288), (33000, 512, 2048),
                                              (2048, 256, 1024), (100, 8, 16), (65536, 128, 512), (3000, 24, 40),
                                              (70000, 12, 128), (5000, 20, 96), (40000, 512, 512), (1, 8, 8), (31, 16, 24),
                                              (2000, 300, 200), (513, 1024, 264)])
@pytest.mark.parametrize("bias", [True, False])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_linear_wgrad_vs_fp32(rows, n_out, k_in, bias, dtype):
    ops, _, _ = _mods()
    g = torch.Generator().manual_seed(rows + n_out)
    x = torch.randn(rows, k_in, generator=g).to(DEV).to(dtype)
    dy = torch.randn(rows, n_out, generator=g).to(DEV).to(dtype)
    w = (torch.randn(n_out, k_in, generator=g) * 0.05).to(DEV).requires_grad_(True)
    b = torch.zeros(n_out, device=DEV, requires_grad=True) if bias else None
    xin = x.clone().requires_grad_(True)
    y = ops.linear(xin, w, b)
    y.backward(dy)
    # fp32 reference on the same bf16-rounded operands: the kernel accumulates in fp32 and never rounds the result
    ref_w = dy.float().t() @ x.float()
    scale = float(ref_w.abs().max())
    assert float((w.grad - ref_w).abs().max()) <= 2e-4 * max(1.0, scale) * max(1.0, (rows / 4096) ** 0.5)
    if bias:
        ref_b = dy.float().sum(0)
        assert float((b.grad - ref_b).abs().max()) <= 2e-4 * max(1.0, float(ref_b.abs().max()))
    ref_dx = (dy.float() @ w.detach().to(dtype).float())
    assert_close(xin.grad, ref_dx, 1e-2 if dtype == torch.bfloat16 else 1e-4, "dx")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,width", [(9, 16), (1000, 96), (513, 128), (64, 1024), (50, 100)])
def test_add_layernorm_vs_oracle(rows, width, dtype):
    ops, _, _ = _mods()
    from oracle import model as OM
    g = torch.Generator().manual_seed(rows + 7 * width)
    a = torch.randn(rows, width, generator=g) * 2
    b = torch.randn(rows, width, generator=g)
    w = 1 + 0.3 * torch.randn(width, generator=g)
    be = 0.2 * torch.randn(width, generator=g)
    ds, dy = torch.randn(rows, width, generator=g), torch.randn(rows, width, generator=g)

    ar = a.to(dtype).float().clone().requires_grad_(True)
    br = b.to(dtype).float().clone().requires_grad_(True)
    wr, ber = w.clone().requires_grad_(True), be.clone().requires_grad_(True)
    s_ref = (ar + br)
    s_round = s_ref + (s_ref.detach().to(dtype).float() - s_ref.detach())  # the kernel normalises the stored (rounded) sum
    y_ref = OM.layer_norm(s_round, wr, ber)
    (s_ref * ds.to(dtype).float()).sum().backward(retain_graph=True)
    (y_ref * dy.to(dtype).float()).sum().backward()

    ad = a.to(DEV).to(dtype).requires_grad_(True)
    bd = b.to(DEV).to(dtype).requires_grad_(True)
    wd, bed = w.to(DEV).requires_grad_(True), be.to(DEV).requires_grad_(True)
    s, y = ops.add_layer_norm(ad, bd, wd, bed)
    assert_close(s, s_ref, TOL[dtype], "sum")
    assert_close(y, y_ref, TOL[dtype], "y")
    torch.autograd.backward([s, y], [ds.to(DEV).to(dtype), dy.to(DEV).to(dtype)])
    assert_close(ad.grad, ar.grad, GRAD_TOL[dtype], "da")
    assert torch.equal(ad.grad, bd.grad)
    assert_close(wd.grad, wr.grad, GRAD_TOL[dtype], "dgamma")
    assert_close(bed.grad, ber.grad, GRAD_TOL[dtype], "dbeta")


# ----------------------------------------------------------------------------- attention dropout (ref :169)
@pytest.mark.parametrize("dtype,C,nH,Ws", [(torch.bfloat16, 128, 4, 64), (torch.float32, 64, 2, 64), (torch.float32, 48, 3, 16)])
def test_attention_dropout_statistics_and_adjoint(dtype, C, nH, Ws):
    ops, _, _ = _mods()
    B, N, p_drop, seed = 2, 4096, 0.25, 1234567
    g = torch.Generator(device=DEV).manual_seed(5)
    qkv = torch.randn(B, N, 3 * C, generator=g, device=DEV).to(dtype)
    hs = torch.full((nH,), 0.2, device=DEV)
    f = lambda t, s=seed, pd=p_drop: ops.window_attn_core(t, None, hs, None, 0, None, nH, Ws, False, attn_drop=pd, seed=s)
    # 1. uniform probabilities (q = 0) and V = 1: every output is (#kept keys / Ws) / (1 - p)
    u = qkv.clone()
    u[:, :, :C] = 0
    u[:, :, 2 * C:] = 1
    o = f(u).float()
    assert abs(float(o.mean()) - 1.0) < 0.01
    expect_std = (p_drop / (Ws * (1 - p_drop))) ** 0.5
    assert abs(float(o.std()) - expect_std) < 0.15 * expect_std + 5e-3
    # 2. reproducible for a seed, different for another, and p = 0 is the plain op
    assert torch.equal(f(qkv), f(qkv))
    assert not torch.equal(f(qkv), f(qkv, seed + 1))
    assert torch.equal(f(qkv, seed, 0.0), ops.window_attn_core(qkv, None, hs, None, 0, None, nH, Ws, False))
    # 3. adjoint identity: out is linear in V for a fixed mask, so <dO, out(V)> == <dV, V> iff the backward regenerates
    #    exactly the forward's mask
    x = qkv.clone().requires_grad_(True)
    out = f(x)
    dO = torch.randn(out.shape, generator=g, device=DEV).to(dtype)
    out.backward(dO)
    terms = dO.double() * out.double()
    lhs = float(terms.sum())
    rhs = float((x.grad[:, :, 2 * C:].double() * qkv[:, :, 2 * C:].double()).sum())
    # both sides are signed sums of ~1e6 products of values rounded to `dtype`: compare against the noise scale of such
    # a sum (2-norm of the terms), not against the sum itself
    noise = float(terms.pow(2).sum().sqrt())
    assert abs(lhs - rhs) <= (2e-2 if dtype == torch.bfloat16 else 1e-4) * noise, (lhs, rhs, noise)
    assert torch.isfinite(x.grad.float()).all()


def test_attention_dropout_mask_is_path_independent():
    """The MFMA kernels (bf16) and the fp32-VALU kernels draw the same mask for the same seed."""
    ops, _, _ = _mods()
    B, N, C, nH, Ws = 1, 2048, 64, 2, 64
    g = torch.Generator(device=DEV).manual_seed(9)
    qkv = torch.randn(B, N, 3 * C, generator=g, device=DEV).to(torch.bfloat16)
    hs = torch.full((nH,), 0.2, device=DEV)
    dO = torch.randn(B, N, C, device=DEV).to(torch.bfloat16)
    res = []
    for t in (qkv, qkv.float()):
        x = t.clone().requires_grad_(True)
        o = ops.window_attn_core(x, None, hs, None, 0, None, nH, Ws, False, attn_drop=0.3, seed=77)
        o.backward(dO.to(t.dtype))
        res.append((o.float(), x.grad.float()))
    assert_close(res[0][0], res[1][0], 1e-2, "out")
    assert_close(res[0][1], res[1][1], 3e-2, "dqkv")


def test_window_attention_module_dropout_train_vs_eval():
    _, M, _ = _mods()
    torch.manual_seed(0)
    wa = M.WindowAttention(96, 64, 3, rel_pos_bias="flat", attn_drop=0.1, proj_drop=0.0).to(DEV)
    x = torch.randn(4, 64, 96, device=DEV)
    wa.eval()
    y_eval = wa(x)
    wa2 = M.WindowAttention(96, 64, 3, rel_pos_bias="flat", attn_drop=0.0).to(DEV)
    wa2.load_state_dict(wa.state_dict())
    assert torch.equal(y_eval, wa2(x))  # eval mode: dropout is the identity
    wa.train()
    xt = x.clone().requires_grad_(True)
    y = wa(xt)
    assert not torch.equal(y, y_eval)
    y.square().mean().backward()
    assert torch.isfinite(xt.grad).all() and float(xt.grad.abs().sum()) > 0


# ----------------------------------------------------------------------------- GELU (+ dropout)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [1, 37, 4096, 1000003])
def test_gelu_vs_oracle(n, dtype):
    ops, _, _ = _mods()
    from oracle import model as OM
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g) * 2.5
    dy = torch.randn(n, generator=g)
    xr = x.to(dtype).float().clone().requires_grad_(True)
    yr = OM.gelu(xr)
    yr.backward(dy.to(dtype).float())
    xd = x.to(DEV).to(dtype).requires_grad_(True)
    y = ops.gelu_dropout(xd)
    assert_close(y, yr, 1e-5 if dtype == torch.float32 else TOL[dtype], "gelu")
    y.backward(dy.to(DEV).to(dtype))
    assert_close(xd.grad, xr.grad, 1e-5 if dtype == torch.float32 else GRAD_TOL[dtype], "gelu'")


def test_gelu_matches_erf_gelu():
    """The division-free GELU of csrc/hs_gelu.h (Phi(-|x|) = exp2(degree-5 polynomial)) against the exact erf form in float64 on a
    dense grid that covers the tail, the origin and large arguments: ABSOLUTE error of GELU and GELU' <= 2e-6 (analysis: 8.3e-7 /
    9.6e-7 in fp32 arithmetic), far inside the 1e-4 a bf16 result can resolve.  nn.GELU is the reference op (ref :31)."""
    ops, _, _ = _mods()
    x = torch.cat([torch.linspace(-40, 40, 2_000_001), torch.linspace(-1e-3, 1e-3, 20001), torch.tensor([0.0, -0.0, 1e4, -1e4, 3e38, -3e38])])
    xd = x.to(DEV).requires_grad_(True)
    y = ops.gelu_dropout(xd)
    y.backward(torch.ones_like(y))
    x64 = x.double()
    phi = 0.5 * (1 + torch.erf(x64 / 2 ** 0.5))
    ref = torch.where(x64.abs() > 1e30, x64.clamp_min(0), x64 * phi)
    dref = phi + x64 * torch.exp(-x64.clamp(-1e3, 1e3) ** 2 / 2) / (2 * torch.pi) ** 0.5
    e = (y.detach().cpu().double() - ref).abs()
    e = torch.where(ref.abs() > 1e3, e / ref.abs().clamp_min(1), e)  # relative for the huge arguments
    de = (xd.grad.cpu().double() - dref).abs()
    assert torch.isfinite(y).all() and torch.isfinite(xd.grad).all()
    assert float(e.max()) <= 2e-6, float(e.max())
    assert float(de.max()) <= 2e-6, float(de.max())
    from conftest import NOTES
    NOTES.append(f"GELU (exp2-polynomial tail) vs erf form, fp32 kernels, |x| <= 40 dense: max abs err {float(e.max()):.2e}, GELU' {float(de.max()):.2e}")


def test_gelu_dropout_mask_consistency():
    ops, _, _ = _mods()
    n, p = 1 << 20, 0.1
    x = (torch.randn(n, device=DEV) * 2).to(torch.bfloat16).requires_grad_(True)
    y = ops.gelu_dropout(x, p, seed=42)
    y0 = ops.gelu_dropout(x.detach(), 0.0)
    kept = (y != 0) | (y0 == 0)
    frac = float(kept.float().mean())
    assert abs(frac - (1 - p)) < 3e-3
    # survivors are scaled by 1/(1-p); the rest are exactly zero
    assert_close(y[kept].float(), y0[kept].float() / (1 - p), 1e-2, "scaled survivors")
    # backward uses the same mask: gradient is zero exactly where the output was dropped
    y.backward(torch.ones_like(y))
    dropped = ~kept
    assert float(x.grad[dropped].abs().max()) == 0.0 and float(x.grad[kept].abs().sum()) > 0
    assert torch.equal(ops.gelu_dropout(x.detach(), p, seed=42), y.detach())
    assert not torch.equal(ops.gelu_dropout(x.detach(), p, seed=43), y.detach())


# ----------------------------------------------------------------------------- norm kernels with fused dropout / DropPath
@pytest.mark.parametrize("rows_per_sample,width", [(64, 128), (256, 96), (16, 512)])
def test_stochastic_norm_variants_vs_explicit_formula(rows_per_sample, width):
    """Recover the kernel's effective multiplier M = rs * mask/(1-p) from the fused-add output in fp32, then check
    s, y and every gradient of both variants (v1: LN(a + M*b); v2: res + rs*LN(mask*x)) against autograd on the formula."""
    ops, _, _ = _mods()
    from oracle import model as OM
    B, p, seed = 4, 0.2, 987654321
    rows = B * rows_per_sample
    g = torch.Generator().manual_seed(width)
    a, b = torch.randn(rows, width, generator=g), torch.randn(rows, width, generator=g) + 3.0  # b != 0 everywhere
    w, be = 1 + 0.3 * torch.randn(width, generator=g), 0.2 * torch.randn(width, generator=g)
    rs = torch.tensor([0.0, 1.25, 1.25, 1.25])
    ds, dy = torch.randn(rows, width, generator=g), torch.randn(rows, width, generator=g)

    ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    wd, bed = w.to(DEV).requires_grad_(True), be.to(DEV).requires_grad_(True)
    s, y = ops.add_layer_norm(ad, bd, wd, bed, row_scale=rs.to(DEV), drop_p=p, seed=seed)
    M = ((s.detach().cpu() - a) / b)
    mask = M.clone()
    mask[rows_per_sample:] /= 1.25  # rows of sample 0 carry rs = 0: their mask is not observable (and irrelevant)
    vals = mask[rows_per_sample:].round(decimals=4).unique()
    assert set(vals.tolist()) <= {0.0, 1.25} and abs(float((mask[rows_per_sample:] > 0).float().mean()) - (1 - p)) < 0.02
    assert float(M[:rows_per_sample].abs().max()) == 0.0  # dropped sample: the branch vanishes
    Mfix = torch.where(M.abs() > 0, torch.full_like(M, 1.25 * 1.25), torch.zeros_like(M))
    Mfix[:rows_per_sample] = 0

    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    wr, ber = w.clone().requires_grad_(True), be.clone().requires_grad_(True)
    s_ref = ar + Mfix * br
    y_ref = OM.layer_norm(s_ref, wr, ber)
    torch.autograd.backward([s_ref, y_ref], [ds, dy])
    assert_close(s, s_ref, 1e-5, "sum")
    assert_close(y, y_ref, 1e-4, "y")
    torch.autograd.backward([s, y], [ds.to(DEV), dy.to(DEV)])
    assert_close(ad.grad, ar.grad, 1e-4, "da")
    assert_close(bd.grad, br.grad, 1e-4, "db")
    assert_close(wd.grad, wr.grad, 1e-4, "dgamma")
    assert_close(bed.grad, ber.grad, 1e-4, "dbeta")

    # v2 form on the same shape and seed (same element indexing -> same mask): y = res + rs * LN(mask * x)
    keep = torch.where(M.abs() > 0, torch.full_like(M, 1.25), torch.zeros_like(M))
    # sample 0's mask is unobservable above; use samples 1.. only
    sl = slice(rows_per_sample, None)
    xd = b.to(DEV).requires_grad_(True)
    rd = a.to(DEV).requires_grad_(True)
    wd2, bed2 = w.to(DEV).requires_grad_(True), be.to(DEV).requires_grad_(True)
    y2 = ops.layer_norm(xd, wd2, bed2, residual=rd, row_scale=rs.to(DEV), drop_p=p, seed=seed)
    xr, rr = b.clone().requires_grad_(True), a.clone().requires_grad_(True)
    wr2, ber2 = w.clone().requires_grad_(True), be.clone().requires_grad_(True)
    y2_ref = rr[sl] + 1.25 * OM.layer_norm(keep[sl] * xr[sl], wr2, ber2)
    y2_ref.backward(dy[sl])
    assert_close(y2[sl], y2_ref, 1e-4, "v2 y")
    assert_close(y2[:rows_per_sample], a[:rows_per_sample], 1e-6, "v2 dropped sample = residual")
    (y2[sl] * dy[sl].to(DEV)).sum().backward()
    assert_close(xd.grad[sl], xr.grad[sl], 1e-4, "v2 dx")
    assert_close(rd.grad[sl], rr.grad[sl], 1e-6, "v2 dres")
    assert_close(wd2.grad, wr2.grad, 1e-4, "v2 dgamma")
    assert_close(bed2.grad, ber2.grad, 1e-4, "v2 dbeta")


@pytest.mark.parametrize("v2", [False, True])
def test_block_with_paper_drop_rates_runs_and_is_identity_in_eval(v2):
    _, M, _ = _mods()
    torch.manual_seed(1)
    kw = dict(window_size=64, shift_size=32, shift_strategy="nest_roll", rel_pos_bias="flat", use_v2_norm_placement=v2, use_cos_attn=v2)
    blk = M.SwinTransformerBlock(64, 768, 12, 2, drop=0.1, attn_drop=0.1, drop_path=0.1, **kw).to(DEV)
    ref = M.SwinTransformerBlock(64, 768, 12, 2, **kw).to(DEV)
    ref.load_state_dict(blk.state_dict())
    x = torch.randn(8, 768, 64, device=DEV).to(torch.bfloat16)
    blk.eval(), ref.eval()
    assert torch.equal(blk(x), ref(x))  # every stochastic element is the identity in eval mode
    blk.train()
    xt = x.clone().requires_grad_(True)
    y = blk(xt)
    assert not torch.equal(y, ref(x))
    y.float().square().mean().backward()
    assert torch.isfinite(xt.grad.float()).all()
    for n, prm in blk.named_parameters():
        assert prm.grad is not None and torch.isfinite(prm.grad).all(), n


def test_layernorm_param_grads_accumulate_into_existing_buffers():
    """`accumulate` of hs_layernorm_bwd / hs_add_layernorm_bwd: d_gamma / d_beta are ADDED to what the buffers hold (the
    direct deposit into a parameter's fp32 .grad), bit-equal to overwrite-then-add."""
    from heal_swin_amd import _lib
    lib, ptr, check = _lib.lib, _lib.ptr, _lib.check
    g = torch.Generator().manual_seed(5)
    rows, width = 3000, 256
    x = torch.randn(rows, width, generator=g).to(DEV).to(torch.bfloat16)
    dy = torch.randn(rows, width, generator=g).to(DEV).to(torch.bfloat16)
    gamma = (1 + 0.1 * torch.randn(width, generator=g)).to(DEV)
    beta = torch.zeros(width, device=DEV)
    y = torch.empty_like(x)
    mean = torch.empty(rows, device=DEV)
    rstd = torch.empty(rows, device=DEV)
    check(lib.hs_layernorm_fwd(ptr(x), None, ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), rows, width, _lib.HS_BF16, None), "fwd")
    ws = torch.empty(int(lib.hs_layernorm_bwd_workspace(rows, width)), device=DEV)
    dx = torch.empty_like(x)
    dg0, db0 = torch.empty(width, device=DEV), torch.empty(width, device=DEV)
    check(lib.hs_layernorm_bwd(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dg0), ptr(db0), ptr(ws), 0,
                               rows, width, _lib.HS_BF16, None), "bwd")
    start_g, start_b = torch.randn(width, generator=g).to(DEV), torch.randn(width, generator=g).to(DEV)
    dg1, db1 = start_g.clone(), start_b.clone()
    check(lib.hs_layernorm_bwd(ptr(dy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dg1), ptr(db1), ptr(ws), 1,
                               rows, width, _lib.HS_BF16, None), "bwd acc")
    assert torch.equal(dg1, start_g + dg0) and torch.equal(db1, start_b + db0)
    dsum = torch.randn(rows, width, generator=g).to(DEV).to(torch.bfloat16)
    dg2, db2 = start_g.clone(), start_b.clone()
    dg3, db3 = torch.empty(width, device=DEV), torch.empty(width, device=DEV)
    check(lib.hs_add_layernorm_bwd(ptr(dy), ptr(dsum), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dg3), ptr(db3), ptr(ws),
                                   0, rows, width, _lib.HS_BF16, None), "add bwd")
    check(lib.hs_add_layernorm_bwd(ptr(dy), ptr(dsum), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dg2), ptr(db2), ptr(ws),
                                   1, rows, width, _lib.HS_BF16, None), "add bwd acc")
    assert torch.equal(dg2, start_g + dg3) and torch.equal(db2, start_b + db3)


# ----------------------------------------------------------------------------- fused segmentation cross-entropy
@pytest.mark.parametrize("dtype", DTYPES)
def test_seg_cross_entropy_layouts_label_types_and_ignore_index(dtype):
    """hs_seg_ce_fwd/bwd vs torch's fp32 CrossEntropyLoss on the host and vs the oracle: the model's pixel-major logits viewed as [B, K, Npix]
    (no copy), a class-major tensor, uint8 / int32 / int64 labels, class weights, ignore_index = -100."""
    from heal_swin_amd import losses as L
    from oracle import model as OM
    g = torch.Generator().manual_seed(3)
    B, K, P = 3, 12, 5000
    raw = (torch.randn(B, P, K, generator=g) * 3).to(dtype)           # what the decoder head produces
    labels = torch.randint(0, K, (B, P), generator=g)
    labels[0, :100] = -100
    w = torch.rand(K, generator=g) + 0.5
    ref_in = raw.float().transpose(1, 2).clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(ref_in, labels, weight=w)  # torch fp32 on the host: ignore_index semantics
    ref.backward()
    tol = 1e-6 if dtype == torch.float32 else 1e-2  # bf16: the gradient is rounded to bf16 on store
    for layout in ("pixel_major_view", "class_major"):
        z = raw.to(DEV).clone().requires_grad_(True)
        logits = z.transpose(1, 2) if layout == "pixel_major_view" else z.transpose(1, 2).contiguous()
        loss = L.seg_loss(logits, labels.to(DEV), w.to(DEV))
        assert abs(float(loss) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref))), layout
        loss.backward()
        got = z.grad.float().cpu().transpose(1, 2)
        assert float((got - ref_in.grad).abs().max()) <= tol * float(ref_in.grad.abs().max()) + 1e-9, layout
        assert float(got[0, :, :100].abs().max()) == 0.0  # ignored pixels
    keep = labels.clamp(min=0)
    base = float(L.seg_loss(raw.to(DEV).transpose(1, 2), keep.to(DEV)))
    for lt in (torch.uint8, torch.int32, torch.int64):
        assert float(L.seg_loss(raw.to(DEV).transpose(1, 2), keep.to(lt).to(DEV))) == base
    assert abs(base - float(OM.seg_loss(raw.float().transpose(1, 2), keep))) <= 2e-6 * max(1.0, abs(base))


def test_linear_wgrad_register_staged_fallback_kernel():
    """The register-staged predecessor of the LDS-DMA kernel stays in the library as the fallback for token slices beyond
    the 2 GiB buffer-offset range; HS_WGRAD_VARIANT=0 (read once per process) selects it, so it is checked in a subprocess."""
    import subprocess
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from heal_swin_amd._lib import check, lib, ptr
for rows, n, k, bias in [(4096, 128, 128, True), (777, 96, 288, True), (33000, 512, 2048, False), (100, 8, 16, True), (70000, 256, 512, True)]:
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, k, generator=g).cuda().to(torch.bfloat16)
    dy = torch.randn(rows, n, generator=g).cuda().to(torch.bfloat16)
    dw = torch.empty(n, k, device="cuda"); db = torch.empty(n, device="cuda")
    ws = torch.empty(int(lib.hs_linear_wgrad_workspace(rows, n, k)), device="cuda")
    check(lib.hs_linear_wgrad(ptr(dy), ptr(x), ptr(dw), ptr(db) if bias else None, ptr(ws), rows, n, k, 0, 1, None), "wgrad")
    ref = dy.float().t() @ x.float()
    assert float((dw - ref).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max())) * max(1.0, (rows / 4096) ** 0.5), (rows, n, k)
    if bias:
        rb = dy.float().sum(0)
        assert float((db - rb).abs().max()) <= 2e-4 * max(1.0, float(rb.abs().max())), (rows, n, k)
print("fallback ok")
''' % ROOT
    env = dict(os.environ, HS_WGRAD_VARIANT="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fallback ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("dtype", DTYPES)
def test_residual_drop_matches_the_explicit_formula(dtype):
    """hs_residual_drop: out = x + rs * drop(t) and dt = dy * rs * mask, with the mask recovered from the output itself
    (statistics: keep rate 1 - p; survivors scaled by exactly 1 / (1 - p); backward uses the same mask)."""
    ops, _, _ = _mods()
    g = torch.Generator().manual_seed(11)
    B, N, C, p = 4, 1000, 64, 0.2
    x = torch.randn(B, N, C, generator=g).to(DEV).to(dtype)
    t = (torch.rand(B, N, C, generator=g) * 2 + 1).to(DEV).to(dtype)  # in [1, 3]: the mask is readable from the result
    rs = torch.tensor([1.25, 0.0, 1.25, 1.25], device=DEV)           # DropPath with keep 0.8: sample 1 dropped
    tt = t.clone().requires_grad_(True)
    xx = x.clone().requires_grad_(True)
    out = ops.residual_drop(xx, tt, rs, p, seed=77)
    branch = (out.float() - x.float())
    mask = (branch.abs() > 1e-3).float()
    assert float(mask[1].sum()) == 0.0
    keep_rate = float(mask[[0, 2, 3]].mean())
    assert abs(keep_rate - (1 - p)) < 5e-3
    expect = x.float() + mask * t.float() * rs.view(-1, 1, 1) / (1 - p)
    assert_close(out, expect, 1e-5 if dtype == torch.float32 else 1e-2, "out")
    dy = torch.randn(B, N, C, generator=g).to(DEV).to(dtype)
    out.backward(dy)
    assert torch.equal(xx.grad, dy)
    assert_close(tt.grad, dy.float() * mask * rs.view(-1, 1, 1) / (1 - p), 1e-5 if dtype == torch.float32 else 1e-2, "dt")
    # same seed -> same mask; p = 0 and no DropPath -> plain add
    assert torch.equal(ops.residual_drop(x, t, rs, p, seed=77), out.detach())
    assert torch.equal(ops.residual_drop(x, t, None, 0.0), x + t)


@pytest.mark.parametrize("ws,nh", [(64, 16), (16, 3), (256, 4)])
def test_rel_bias_scatter_grad_sorted_matches_the_scan_kernel(ws, nh):
    """The grouped (argsort) form of the bias-table gradient against the scanning kernel and a torch index_add."""
    from heal_swin_amd import _lib, ops
    from heal_swin_amd._lib import check, lib, ptr, stream_ptr

    torch.manual_seed(ws + nh)
    rel = torch.from_numpy(_lib.rel_pos_index(ws).astype("int32")).cuda().contiguous()
    rows = int(rel.max()) + 1
    dbias = torch.randn(nh, ws, ws, device="cuda")
    ref = torch.zeros(rows, nh, device="cuda", dtype=torch.float64).index_add_(0, rel.flatten().long(), dbias.reshape(nh, -1).t().double())
    a = torch.empty(rows, nh, device="cuda")
    b = torch.empty_like(a)
    check(lib.hs_rel_bias_scatter_grad(ptr(dbias), ptr(rel), ptr(a), rows, nh, ws, stream_ptr(a.device)), "scan")
    order, offsets = ops._rel_idx_groups(rel, rows)
    check(lib.hs_rel_bias_scatter_grad_sorted(ptr(dbias), ptr(order), ptr(offsets), ptr(b), rows, nh, ws, stream_ptr(a.device)), "sorted")
    assert_close(b, ref, 1e-6, f"rel_bias scatter sorted ws={ws}")
    assert_close(a, ref, 1e-6, f"rel_bias scatter scan ws={ws}")


@pytest.mark.parametrize("K", [12, 8, 16, 4])
def test_seg_cross_entropy_padded_row_fast_path(K):
    """The model's own logits layout (K classes in 16-wide bf16 rows, seen as [B, K, Npix]): `seg_ce_*_row16` must give the
    loss and the gradient of the generic strided kernels bit for bit, leave the pad columns of the gradient buffer zero, and
    `ops.pad_slice` must hand that buffer on whole."""
    from heal_swin_amd import losses as L, ops
    g = torch.Generator().manual_seed(K)
    B, P = 2, 7001
    z16 = torch.zeros(B, P, 16, dtype=torch.bfloat16)
    z16[..., :K] = (torch.randn(B, P, K, generator=g) * 3).to(torch.bfloat16)
    labels = torch.randint(0, K, (B, P), generator=g)
    labels[1, 5:50] = -100
    w = torch.rand(K, generator=g) + 0.5
    # generic path: the same values in a dense pixel-major [B, P, K] tensor
    zd = z16[..., :K].contiguous().to(DEV).requires_grad_(True)
    loss_d = L.seg_loss(zd.transpose(1, 2), labels.to(DEV), w.to(DEV))
    loss_d.backward()
    # fast path through the padded rows and pad_slice
    zp = z16.to(DEV).requires_grad_(True)
    loss_p = L.seg_loss(ops.pad_slice(zp, K).transpose(1, 2), labels.to(DEV), w.to(DEV))
    loss_p.backward()
    assert float(loss_p) == float(loss_d)
    assert torch.equal(zp.grad[..., :K], zd.grad)
    assert not zp.grad[..., K:].any()
    assert not ops.RT.zero_padded_grads  # the buffer was consumed by pad_slice's backward
    ref_in = z16[..., :K].float().transpose(1, 2).clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(ref_in, labels, weight=w)
    ref.backward()
    assert abs(float(loss_p) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    assert float((zp.grad[..., :K].float().cpu().transpose(1, 2) - ref_in.grad).abs().max()) <= 1e-2 * float(ref_in.grad.abs().max())


def test_batched_weight_transposes_follow_the_parameters():
    """`ops.ParamCastCache.get_t`: the [in, out] bf16 copies of the weights are re-made by ONE launch (`hs_transpose_many_16`) after
    every refresh -- odd shapes included -- and equal `param.to(bf16).t()` exactly."""
    ops, _, _ = _mods()
    g = torch.Generator().manual_seed(5)
    shapes = [(384, 128), (128, 512), (40, 24), (33, 100), (2048, 512), (16, 128)]
    params = [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in shapes]
    cache = ops.ParamCastCache(params, torch.bfloat16)
    for step in range(3):
        cache.refresh(force=True)
        for p in params:
            t = cache.get_t(p, torch.bfloat16)
            assert t.shape == (p.shape[1], p.shape[0]) and t.is_contiguous()
            assert torch.equal(t, p.detach().to(torch.bfloat16).t()), (step, tuple(p.shape))
        with torch.no_grad():
            for p in params:
                p.add_(0.37 * (step + 1))  # an optimizer step (fused ones do not bump versions: hence force=True above)


def test_buffer_scalar_offset_takes_part_in_the_range_check():
    """the role-separated DMA of hs_gemm_nt puts the k offset and the 32-row group of an operand piece into the
    SCALAR offset of buffer_load ... lds and relies on the descriptor's range check covering it (rows past an operand's end must
    read zeros, not the bytes behind the allocation).  Pinned on the hardware: a 256-byte descriptor inside a buffer of ones."""
    from heal_swin_amd._lib import check, lib, ptr
    src = torch.full((1024,), 0x3f800000, dtype=torch.int32, device=DEV)  # ones behind the descriptor's end as well
    out = torch.empty(128, dtype=torch.int32, device=DEV)
    for soff, n_valid in ((0, 64), (128, 32), (252, 1), (256, 0), (4096, 0)):
        check(lib.hs_debug_buffer_soffset_probe(ptr(src), 256, soff, ptr(out), None), "probe")
        got = out.cpu().numpy()
        want = np.where(np.arange(64) < n_valid, 0x3f800000, 0)
        assert (got[:64] == want).all(), ("register load", soff, got[:64])
        assert (got[64:] == want).all(), ("LDS-DMA load", soff, got[64:])


@pytest.mark.parametrize("drop_p", [0.0, 0.3])
def test_gelu_split3_equals_gelu_then_split(drop_p):
    """hs_gelu_split3 (fp32 runs: the GELU output / the hidden gradient written directly as the [hi | hi | lo] operand of the
    bf16x3 products that read them) is hs_gelu_fwd / hs_gelu_bwd followed by hs_split_bf16x3 (forward: bit for bit), same dropout mask."""
    from heal_swin_amd import _lib
    from heal_swin_amd._lib import check, lib, ptr
    rows, k, seed = 777, 264, 12345
    g = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randn(rows, k, generator=g, device=DEV) * 2
    dy = torch.randn(rows, k, generator=g, device=DEV)
    for bwd in (False, True):
        ref = torch.empty_like(x)
        if bwd:
            check(lib.hs_gelu_bwd(ptr(dy), ptr(x), ptr(ref), x.numel(), drop_p, seed, _lib.HS_F32, None), "gelu_bwd")
        else:
            check(lib.hs_gelu_fwd(ptr(x), ptr(ref), x.numel(), drop_p, seed, _lib.HS_F32, None), "gelu_fwd")
        ref3 = torch.empty(rows, 3 * k, dtype=torch.bfloat16, device=DEV)
        check(lib.hs_split_bf16x3(ptr(ref), ptr(ref3), rows, k, 0, None), "split")
        out3 = torch.empty_like(ref3)
        check(lib.hs_gelu_split3(ptr(dy) if bwd else None, ptr(x), ptr(out3), rows, k, drop_p, seed, None), "gelu_split3")
        if not bwd and drop_p == 0.0:  # (the products with dy / the dropout factor are contracted differently by the compiler in the
            assert torch.equal(out3.view(torch.int16), ref3.view(torch.int16))  # two kernels: fp32 ulps, signed zeros)
        assert torch.equal(out3[:, :k], out3[:, k:2 * k])
        assert torch.equal(out3[:, :k] == 0, ref == 0)  # the same dropout mask
        # and hi + lo reproduces the fp32 value to 2^-16
        hi, lo = out3[:, :k].float(), out3[:, 2 * k:].float()
        assert float((hi + lo - ref).abs().max()) <= 2.0 ** -15 * float(ref.abs().max())


def test_cos_head_scale_matches_the_torch_formula():
    """exp(clamp(logit_scale, max=ln 100)) (ref :144-147) and its gradient in one launch each, including a head beyond the clamp and
    one exactly on it; `accumulate` adds into an existing buffer."""
    import math
    ops, _, _ = _mods()
    from heal_swin_amd._lib import check, lib, ptr
    ls = torch.tensor([[[0.3]], [[2.3026]], [[math.log(100.0)]], [[5.0]], [[-1.0]]], device=DEV, requires_grad=True)
    ref_in = ls.detach().clone().requires_grad_(True)
    ref = torch.exp(torch.clamp(ref_in, max=math.log(1.0 / 0.01))).reshape(-1)
    out = ops.cos_head_scale(ls)
    g = torch.tensor([1.0, -2.0, 0.5, 3.0, 0.25], device=DEV)
    ref.backward(g)
    out.backward(g)
    assert_close(out, ref, 1e-6, "scale")
    assert_close(ls.grad, ref_in.grad, 1e-6, "d logit_scale")
    assert float(ls.grad.reshape(-1)[3]) == 0.0
    buf = torch.ones(5, device=DEV)
    check(lib.hs_cos_head_scale_bwd(ptr(ls.detach().reshape(-1)), ptr(g), ptr(buf), 5, 1, None), "bwd")
    assert_close(buf - 1.0, ref_in.grad.reshape(-1), 1e-6, "accumulated d logit_scale")


def test_rel_bias_scatter_grad_sorted_add_accumulates():
    ops, _, _ = _mods()
    from heal_swin_amd import _lib
    from heal_swin_amd._lib import check, lib, ptr
    ws, nh = 64, 4
    rel = torch.from_numpy(_lib.rel_pos_index(ws).astype(np.int32).reshape(-1)).to(DEV)
    rows = int(rel.max()) + 1
    dbias = torch.randn(nh, ws, ws, device=DEV)
    order, offsets = ops._rel_idx_groups(rel, rows)
    ref = torch.empty(rows, nh, device=DEV)
    check(lib.hs_rel_bias_scatter_grad_sorted(ptr(dbias), ptr(order), ptr(offsets), ptr(ref), rows, nh, ws, None), "scatter")
    acc = torch.full((rows, nh), 2.0, device=DEV)
    check(lib.hs_rel_bias_scatter_grad_sorted_add(ptr(dbias), ptr(order), ptr(offsets), ptr(acc), rows, nh, ws, None), "scatter add")
    assert torch.equal(acc, ref + 2.0)


def test_batched_bias_gather_and_head_scale_equal_the_per_block_ops():
    """hs_rel_bias_gather_many / hs_rel_bias_scatter_grad_sorted_many / hs_cos_head_scale_many (all attention blocks of a model in one
    launch each, `SwinHPTransformerSys._prefetch_attn_params`) against the per-block entry points: identical values, identical gradients
    -- also when some blocks' outputs are unused -- with different head counts per block (the stages of a model)."""
    ops, _, _ = _mods()
    from oracle import tables as OT
    ws = 64
    rel = torch.from_numpy(OT.rel_pos_index(ws).astype(np.int32).reshape(-1)).to(DEV)
    rows = int(rel.max()) + 1
    g = torch.Generator(device=DEV).manual_seed(2)
    heads = [3, 3, 6, 12, 24, 4]
    tabs = [torch.randn(rows, h, generator=g, device=DEV).requires_grad_(True) for h in heads]
    tabs2 = [t.detach().clone().requires_grad_(True) for t in tabs]
    many = ops.rel_pos_bias_many(rel, ws, tabs)
    single = [ops.RelPosBiasFn.apply(t, rel, ws) for t in tabs2]
    dys = [torch.randn(h, ws, ws, generator=g, device=DEV) for h in heads]
    for a, b in zip(many, single):
        assert torch.equal(a, b)
    used = [0, 2, 3, 5]  # (blocks 1 and 4 take no gradient)
    torch.autograd.backward([many[j] for j in used], [dys[j] for j in used])
    torch.autograd.backward([single[j] for j in used], [dys[j] for j in used])
    for j, (a, b) in enumerate(zip(tabs, tabs2)):
        if j in used:
            assert torch.equal(a.grad, b.grad), j
        else:
            assert a.grad is None and b.grad is None
    ls = [torch.randn(h, 1, 1, generator=g, device=DEV).mul(2).add(3).requires_grad_(True) for h in heads]  # some beyond ln 100
    ls2 = [t.detach().clone().requires_grad_(True) for t in ls]
    sm = ops.cos_head_scale_many(ls)
    ss = [ops.cos_head_scale(t) for t in ls2]
    ds = [torch.randn(h, generator=g, device=DEV) for h in heads]
    for a, b in zip(sm, ss):
        assert torch.equal(a, b)
    torch.autograd.backward(list(sm), ds)
    torch.autograd.backward(ss, ds)
    for a, b in zip(ls, ls2):
        assert torch.equal(a.grad, b.grad)
