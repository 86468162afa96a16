"""hs_gemm_nt (bf16 NT GEMM with fused epilogues) through the C ABI vs fp32 torch on the same bf16-rounded operands:
every epilogue, ragged M / N / K (tile tails, K not a multiple of the 64-deep step), the two-segment (skip-concat) form,
both tile shapes, and the dropout mask against the standalone GELU kernel's."""
import pytest
import torch

from _util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _gemm(a, b, bias=None, epi=0, aux=None, a2=None, b2=None, want_c=True, p=0.0, seed=0):
    from heal_swin_amd import _lib
    from heal_swin_amd._lib import check, lib, ptr, stream_ptr
    m, k = a.shape
    n = b.shape[0]
    c = torch.empty((m, n), dtype=torch.bfloat16, device=a.device) if want_c else None
    if epi == _lib.HS_EPI_GELU:
        aux = torch.empty((m, n), dtype=torch.bfloat16, device=a.device)
    k2 = 0 if a2 is None else a2.shape[1]
    check(lib.hs_gemm_nt(ptr(a), a.stride(0), ptr(b), b.stride(0), k, ptr(a2), 0 if a2 is None else a2.stride(0), ptr(b2),
                         0 if b2 is None else b2.stride(0), k2, ptr(bias), ptr(c), ptr(aux), m, n, epi, p, seed, _lib.HS_BF16,
                         stream_ptr(a.device)), "hs_gemm_nt")
    return c, aux


def _gelu(x):
    return torch.nn.functional.gelu(x)


def _dgelu(x):
    return 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5


@pytest.mark.parametrize("tile", [0, 1, 2, 3])
@pytest.mark.parametrize("m,n,k", [(256, 128, 64), (300, 132, 96), (1000, 384, 200), (129, 12, 128), (4096, 512, 2048), (77, 260, 8),
                                   (10277, 2048, 128)])  # the last: more tiles than resident workgroups (epilogue -> next tile)
def test_gemm_nt_epilogues_match_fp32(m, n, k, tile):
    from heal_swin_amd import _lib
    _lib.lib.hs_gemm_nt_set_tile(tile)
    try:
        g = torch.Generator().manual_seed(m + n + k)
        a = torch.randn(m, k, generator=g).to(torch.bfloat16).to(DEV)
        b = (torch.randn(n, k, generator=g) * k ** -0.5).to(torch.bfloat16).to(DEV)
        bias = torch.randn(n, generator=g).to(DEV)
        ref = a.float() @ b.float().t()
        c, _ = _gemm(a, b)
        assert_close(c, ref, 6e-3, "plain")
        c, _ = _gemm(a, b, bias)
        assert_close(c, ref + bias, 6e-3, "bias")
        h, act = _gemm(a, b, bias, epi=_lib.HS_EPI_GELU)
        assert_close(h, ref + bias, 6e-3, "gelu: h")
        assert_close(act, _gelu(ref + bias), 6e-3, "gelu: act")
        _, act2 = _gemm(a, b, bias, epi=_lib.HS_EPI_GELU, want_c=False)
        assert torch.equal(act, act2)
        hsaved = (ref + bias).to(torch.bfloat16)
        d, _ = _gemm(a, b, None, epi=_lib.HS_EPI_DGELU, aux=hsaved)
        assert_close(d, ref * _dgelu(hsaved.float()), 8e-3, "dgelu")
        res = torch.randn(m, n, generator=g).to(torch.bfloat16).to(DEV)
        r, _ = _gemm(a, b, bias, epi=_lib.HS_EPI_RESID, aux=res)
        assert_close(r, ref + bias + res.float(), 6e-3, "resid")
    finally:
        _lib.lib.hs_gemm_nt_set_tile(0)


@pytest.mark.parametrize("m,n,k1,k2", [(500, 96, 96, 96), (1024, 256, 256, 256), (130, 64, 40, 72)])
def test_gemm_nt_two_segments_equal_the_concatenated_product(m, n, k1, k2):
    g = torch.Generator().manual_seed(k1 * 7 + k2)
    x = torch.randn(m, k1, generator=g).to(torch.bfloat16).to(DEV)
    skip = torch.randn(m, k2, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(n, k1 + k2, generator=g) * (k1 + k2) ** -0.5).to(torch.bfloat16).to(DEV)
    bias = torch.randn(n, generator=g).to(DEV)
    ref = torch.cat([x, skip], 1).float() @ w.float().t() + bias
    c, _ = _gemm(x, w[:, :k1], bias, a2=skip, b2=w[:, k1:])
    assert_close(c, ref, 6e-3, "two segments")


def test_gemm_nt_dropout_mask_equals_the_gelu_kernels():
    """Same (seed, element index) -> same keep decision as hs_gelu_fwd, so a model may mix fused and unfused MLPs, and the
    backward epilogue regenerates the forward's mask."""
    from heal_swin_amd import _lib, ops
    m, n, k, p, seed = 512, 384, 128, 0.3, 123456789012345
    g = torch.Generator().manual_seed(1)
    a = torch.randn(m, k, generator=g).to(torch.bfloat16).to(DEV)
    b = (torch.randn(n, k, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    bias = torch.full((n,), 3.0, device=DEV)  # pre-activations well away from 0: every surviving gelu(h) is non-zero
    h, act = _gemm(a, b, bias, epi=_lib.HS_EPI_GELU, p=p, seed=seed)
    plain = ops.GeluDropoutFn.apply(h, p, seed)
    assert torch.equal(act == 0, plain == 0)
    assert abs(float((act != 0).float().mean()) - (1 - p)) < 5e-3
    keep = (act != 0)
    assert_close(act[keep], plain[keep], 1e-2, "survivors")
    da = torch.randn(m, k, generator=g).to(torch.bfloat16).to(DEV)  # reuse shapes: d = (da b^T) * mask/(1-p) * gelu'(h)
    d, _ = _gemm(da, b, None, epi=_lib.HS_EPI_DGELU, aux=h, p=p, seed=seed)
    assert torch.equal(d == 0, act == 0)
    ref = (da.float() @ b.float().t()) * _dgelu(h.float()) / (1 - p)
    assert_close(d[keep], ref[keep], 1e-2, "dgelu survivors")


def test_gemm_nt_rejects_unsupported_arguments():
    from heal_swin_amd import _lib
    a = torch.zeros(64, 12, dtype=torch.bfloat16, device=DEV)
    b = torch.zeros(16, 12, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="multiples of 8"):
        _gemm(a, b)
    with pytest.raises(AssertionError, match="needs aux"):
        _gemm(a[:, :8].contiguous(), b[:, :8].contiguous(), epi=_lib.HS_EPI_RESID)


@pytest.mark.parametrize("p", [0.0, 0.25])
@pytest.mark.parametrize("passthrough", [False, True])
def test_mlp_node_own_and_library_paths_agree_with_fp32(p, passthrough):
    """ops.MlpFn (fc1 -> GELU -> dropout -> fc2 as one autograd node) through `hs_gemm_nt` epilogues (HS_OWN_GEMM=1) and through
    library GEMMs + standalone GELU kernels (=0): same dropout mask, outputs and all gradients within bf16 rounding of an
    fp32 torch MLP that uses the recovered mask."""
    from heal_swin_amd import ops
    g = torch.Generator().manual_seed(3)
    rows, c, hid = 640, 96, 384
    x = torch.randn(2, rows // 2, c, generator=g)
    w1, b1 = torch.randn(hid, c, generator=g) * c ** -0.5, torch.randn(hid, generator=g) * 0.1
    w2, b2 = torch.randn(c, hid, generator=g) * hid ** -0.5, torch.randn(c, generator=g) * 0.1
    dy = torch.randn(2, rows // 2, c, generator=g)
    seed = 4242
    out = {}
    prev = ops.OWN_GEMM
    try:
        for mode in ("1", "0"):
            ops.OWN_GEMM = mode
            xs = x.to(DEV).to(torch.bfloat16).requires_grad_(True)
            ps = [t.to(DEV).requires_grad_(True) for t in (w1, b1, w2, b2)]
            r = ops.mlp(xs, *ps, drop_p=p, seed=seed, passthrough=passthrough)
            y = r[0] + r[1] if passthrough else r  # the alias carries the residual connection's gradient
            y.backward(dy.to(DEV).to(torch.bfloat16))
            out[mode] = (y.detach().float().cpu(), xs.grad.float().cpu(), [t.grad.float().cpu() for t in ps])
    finally:
        ops.OWN_GEMM = prev
    # fp32 reference on the bf16-rounded input, with the mask recovered from a probe call (mask depends only on seed, index)
    xr = x.to(torch.bfloat16).float().requires_grad_(True)
    pr = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
    h = xr.reshape(-1, c) @ pr[0].t() + pr[1]
    if p > 0:
        probe = ops.GeluDropoutFn.apply(torch.full((rows, hid), 3.0, device=DEV), p, seed).cpu()
        mask = (probe != 0).float() / (1 - p)
    else:
        mask = torch.ones(rows, hid)
    yr = ((torch.nn.functional.gelu(h) * mask) @ pr[2].t() + pr[3]).view(2, rows // 2, c)
    if passthrough:
        yr = yr + xr
    yr.backward(dy.to(torch.bfloat16).float())
    for mode in ("1", "0"):
        y, dx, grads = out[mode]
        assert_close(y, yr, 1.5e-2, f"mlp[{mode}] y")
        assert_close(dx, xr.grad, 2e-2, f"mlp[{mode}] dx")
        for got, ref, nm in zip(grads, pr, ("w1", "b1", "w2", "b2")):
            assert_close(got, ref.grad, 2e-2, f"mlp[{mode}] d{nm}")


def test_linear_and_concat_linear_own_kernel_vs_library():
    from heal_swin_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 200, 128, generator=g).to(torch.bfloat16)
    skip = torch.randn(3, 200, 128, generator=g).to(torch.bfloat16)
    w, b = torch.randn(128, 128, generator=g) * 0.1, torch.randn(128, generator=g) * 0.1
    wc, bc = torch.randn(128, 256, generator=g) * 0.08, torch.randn(128, generator=g) * 0.1
    dy = torch.randn(3, 200, 128, generator=g).to(torch.bfloat16)
    res = {}
    prev = ops.OWN_GEMM
    try:
        for mode in ("1", "0"):
            ops.OWN_GEMM = mode
            xs, ss = x.to(DEV).requires_grad_(True), skip.to(DEV).requires_grad_(True)
            ps = [t.to(DEV).requires_grad_(True) for t in (w, b, wc, bc)]
            y, alias = ops.linear_passthrough(xs, ps[0], ps[1])
            z = ops.concat_linear(y + alias, ss, ps[2], ps[3])
            z.backward(dy.to(DEV))
            res[mode] = [z.detach().float().cpu(), xs.grad.float().cpu(), ss.grad.float().cpu()] + [t.grad.float().cpu() for t in ps]
    finally:
        ops.OWN_GEMM = prev
    for a, bb, nm in zip(res["1"], res["0"], ("z", "dx", "dskip", "dw", "db", "dwc", "dbc")):
        assert_close(a, bb, 2e-2, "own vs library " + nm)


def test_gemm_tuner_measures_once_and_the_model_follows_its_pick():
    """ops.GemmTuner: the first request of a new (rows bucket, n, k) class times hs_gemm_nt and the library on synthetic operands and
    remembers the faster; later requests (and neighbouring row counts of the same power of two) are dictionary hits; own_gemm_ok
    returns the pick for bias products; either pick gives the same Linear result to bf16 rounding."""
    from heal_swin_amd import _lib, ops
    from heal_swin_amd.ops import gemm as G
    tuner = G.GemmTuner()
    m, n, k = 40000, 640, 320  # not a shape of any benchmark workload
    dev = torch.device("cuda", torch.cuda.current_device())
    pick = tuner.pick(m, n, k, dev)
    assert pick in (True, False) and len(tuner.trials) == 1
    own_us, lib_us = tuner.trials[tuner.key(m, n, k)]
    assert own_us > 0 and lib_us > 0 and pick == (own_us <= lib_us)
    assert tuner.pick(m + 1000, n, k, dev) == pick and len(tuner.trials) == 1  # same power-of-two bucket: no second trial
    assert tuner.pick(64, 64, 64, dev) is None and len(tuner.trials) == 1     # too small to matter: the class rule answers
    prev_tuner, prev_mode = G.GEMM_TUNER, ops.GEMM_TUNE
    try:
        G.GEMM_TUNER = tuner
        ops.GEMM_TUNE = True
        assert ops.own_gemm_ok(_lib.HS_EPI_BIAS, n, k, torch.bfloat16, m=m) == pick
        g = torch.Generator(device=DEV).manual_seed(0)
        x = torch.randn((m, k), generator=g, device=DEV).to(torch.bfloat16)
        w = torch.randn((n, k), generator=g, device=DEV) * k ** -0.5
        b = torch.randn(n, generator=g, device=DEV)
        ys = {}
        for forced in (True, False):
            tuner.picks[tuner.key(m, n, k)] = forced
            ys[forced] = ops.linear(x, w, b).float()
        ref = x.float() @ w.t() + b
        for forced, y in ys.items():
            assert_close(y, ref, 6e-3, f"linear through {'hs_gemm_nt' if forced else 'the library'}")
        ops.GEMM_TUNE = False
        assert ops.own_gemm_ok(_lib.HS_EPI_BIAS, n, k, torch.bfloat16, m=m) == (k <= 128 or n <= 128 or (n <= 256 and k <= 256))
    finally:
        G.GEMM_TUNER, ops.GEMM_TUNE = prev_tuner, prev_mode
