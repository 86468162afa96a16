"""Parity at BASELINE's full sizes (where the CPU oracle would take minutes) through size-independent properties of the
fused shift + window attention op, and by cross-checking the two independent device implementations (MFMA path vs the
fp32-VALU path) on identical inputs.  HEAL-SWIN-B stage 0 at nside 256 / 12 base pixels: N = 196608 tokens, C = 128."""
import pytest
import torch

from _util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"
N, C, NH, WS = 196608, 128, 4, 64


def _ops():
    from heal_swin_amd import ops
    from heal_swin_amd.models_torch import hp_shifting as S
    return ops, S


def _inputs(B=2, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    qkv = torch.randn(B, N, 3 * C, generator=g, device=DEV).to(torch.bfloat16)
    bias = torch.randn(NH, WS, WS, generator=g, device=DEV) * 0.5
    hs = torch.full((NH,), 32 ** -0.5, device=DEV)
    return qkv, bias, hs


@pytest.mark.parametrize("shifted", [False, True])
def test_mfma_path_matches_valu_path_full_size(shifted):
    """bf16 MFMA kernels vs the fp32-VALU kernels run on the SAME bf16 tensors viewed as fp32 inputs: two independent
    implementations of the op, forward and backward, at the full stage-0 size."""
    ops, S = _ops()
    qkv, bias, hs = _inputs()
    roll, labels = 0, None
    if shifted:
        sh = S.NestRollShift(32, N, WS)
        roll, labels = 32, sh.tables(DEV)[2]
    dout = torch.randn(2, N, C, device=DEV).to(torch.bfloat16)

    a = qkv.clone().requires_grad_(True)
    ba = bias.clone().requires_grad_(True)
    oa = ops.window_attn_core(a, ba, hs, None, roll, labels, NH, WS, False)  # MFMA path (bf16)
    oa.backward(dout)

    b = qkv.float().requires_grad_(True)
    bb = bias.clone().requires_grad_(True)
    ops.FORCE_VALU_ATTENTION = True
    try:
        ob = ops.window_attn_core(b, bb, hs, None, roll, labels, NH, WS, False)  # fp32-VALU path on the same values
        ob.backward(dout.float())
    finally:
        ops.FORCE_VALU_ATTENTION = False

    assert_close(oa, ob, 1e-2, "out")
    assert_close(a.grad, b.grad, 3e-2, "dqkv")
    # bias gradient: a sum over 6144 windows; compare relative to its scale
    assert_close(ba.grad / ba.grad.abs().max(), bb.grad / bb.grad.abs().max(), 2e-2, "dbias")

    # third implementation: the fp32 MFMA kernels (v_mfma_f32_32x32x2_f32) on the same fp32 values -- fp32 tolerance
    c = qkv.float().requires_grad_(True)
    bc = bias.clone().requires_grad_(True)
    oc = ops.window_attn_core(c, bc, hs, None, roll, labels, NH, WS, False)
    oc.backward(dout.float())
    assert_close(oc, ob, 1e-5, "out fp32 mfma vs valu")
    assert_close(c.grad, b.grad, 1e-4, "dqkv fp32 mfma vs valu")
    assert_close(bc.grad / bb.grad.abs().max(), bb.grad / bb.grad.abs().max(), 1e-4, "dbias fp32 mfma vs valu")


def test_softmax_rows_sum_to_one_full_size():
    """V = 1 everywhere  =>  every output element is exactly the row sum of the probabilities = 1."""
    ops, S = _ops()
    qkv, bias, hs = _inputs(B=1, seed=1)
    qkv[:, :, 2 * C:] = 1.0
    sh = S.NestRollShift(32, N, WS)
    out = ops.window_attn_core(qkv, bias, hs, None, 32, sh.tables(DEV)[2], NH, WS, False)
    assert float((out.float() - 1.0).abs().max()) <= 8e-3  # bf16 rounding of P (64 terms) and of the output


def test_linear_in_v_full_size():
    """out(q, k, a*v1 + v2) == a*out(q, k, v1) + out(q, k, v2)"""
    ops, _ = _ops()
    qkv, bias, hs = _inputs(B=1, seed=2)
    v2 = torch.randn(1, N, C, device=DEV).to(torch.bfloat16)
    x1 = qkv.clone()
    x2 = qkv.clone()
    x2[:, :, 2 * C:] = v2
    x3 = qkv.clone()
    x3[:, :, 2 * C:] = (2.0 * qkv[:, :, 2 * C:].float() + v2.float()).to(torch.bfloat16)
    f = lambda x: ops.window_attn_core(x, bias, hs, None, 0, None, NH, WS, False).float()
    assert_close(f(x3), 2.0 * f(x1) + f(x2), 2e-2, "linearity")


def test_windows_are_independent_full_size():
    """Changing the tokens of one window changes only that window's outputs (bit-exact elsewhere), including under a shift."""
    ops, S = _ops()
    qkv, bias, hs = _inputs(B=1, seed=3)
    sh = S.NestRollShift(32, N, WS)
    lab = sh.tables(DEV)[2]
    o1 = ops.window_attn_core(qkv, bias, hs, None, 32, lab, NH, WS, False)
    w = 1234  # shifted window w covers natural tokens (w*64 + 32 .. w*64 + 95)
    x = qkv.clone()
    x[:, w * 64 + 32: w * 64 + 96] += 1.0
    o2 = ops.window_attn_core(x, bias, hs, None, 32, lab, NH, WS, False)
    same = torch.ones(N, dtype=torch.bool, device=DEV)
    same[w * 64 + 32: w * 64 + 96] = False
    assert torch.equal(o1[:, same], o2[:, same])
    assert not torch.equal(o1[:, ~same], o2[:, ~same])


def test_ring_shift_gather_roundtrip_full_size():
    """shift followed by shift_back is the identity at nside 128 (the stage-0 token grid of the paper config), bit-exact,
    and the fused kernel with the ring table equals  shift -> unshifted kernel -> shift_back."""
    ops, S = _ops()
    n = 8 * 128 * 128
    sh = S.RingShift(128, 8, WS, 4)
    x = torch.randn(2, n, 96, device=DEV).to(torch.bfloat16)
    assert torch.equal(sh.shift_back(sh.shift(x)), x)
    idx, inv, lab = sh.tables(DEV)
    qkv = torch.randn(1, n, 3 * 96, device=DEV).to(torch.bfloat16)
    hs = torch.full((3,), 0.2, device=DEV)
    fused = ops.window_attn_core(qkv, None, hs, idx, 0, lab, 3, WS, False)
    staged = sh.shift_back(ops.window_attn_core(sh.shift(qkv), None, hs, None, 0, lab, 3, WS, False))
    assert torch.equal(fused, staged)


@pytest.mark.parametrize("dtype,n_out,k_in", [(torch.bfloat16, 384, 128), (torch.bfloat16, 2048, 512), (torch.float32, 288, 96)])
def test_linear_wgrad_checksum_full_size(dtype, n_out, k_in):
    """sum over all entries of dW equals (column sums of dY) . (column sums of X): a checksum of checksums at M = 1.5 M rows
    (bf16: the 128-wide and the 256 x 256 tile kernels; fp32: the v_mfma_f32_32x32x2 kernel)."""
    ops, _ = _ops()
    M = 1572864 if n_out < 1024 else 98304 * 4
    x = (torch.randn(M, k_in, device=DEV) * 0.5).to(dtype)
    dy = (torch.randn(M, n_out, device=DEV) * 0.5).to(dtype)
    w = torch.zeros(n_out, k_in, device=DEV, requires_grad=True)
    b = torch.zeros(n_out, device=DEV, requires_grad=True)
    ops.linear(x.requires_grad_(False), w, b).backward(dy)
    # row sums of dW against an independent fp64 evaluation of  dY^T (X 1)
    ref_rows = (dy.double().t() @ x.double().sum(1))
    got_rows = w.grad.double().sum(1)
    assert float((got_rows - ref_rows).abs().max()) <= 1e-3 * float(ref_rows.abs().max())
    assert float((b.grad.double() - dy.double().sum(0)).abs().max()) <= 1e-3 * float(dy.double().sum(0).abs().max())
