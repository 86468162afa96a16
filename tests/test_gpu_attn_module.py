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
