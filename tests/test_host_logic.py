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


def test_param_cast_cache_survives_fused_optimizers_that_do_not_bump_versions():
    """`torch.optim.Adam(fused=True)` updates parameters WITHOUT bumping `_version` (asserted here, so that a PyTorch that
    changes this is noticed): a grad-enabled forward therefore refreshes unconditionally (`force`), and the first no-grad
    forward after one does too (`dirty`); later no-grad forwards stay free."""
    from heal_swin_amd import ops
    torch.manual_seed(0)
    p = torch.nn.Parameter(torch.randn(16, 8))
    try:
        opt = torch.optim.Adam([p], lr=0.1, fused=True)
    except (RuntimeError, ValueError):
        import pytest
        pytest.skip("no fused Adam for CPU tensors in this PyTorch")
    cache = ops.ParamCastCache([p], torch.bfloat16)
    calls = []
    orig = torch._foreach_copy_
    torch._foreach_copy_ = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        for _ in range(3):                      # three training steps
            cache.refresh(force=True)           # what SwinHPTransformerSys.forward does when grad is enabled
            assert torch.equal(cache.get(p, torch.bfloat16), p.detach().to(torch.bfloat16))
            v0 = p._version
            p.grad = torch.randn_like(p)
            opt.step()
            fused_is_silent = p._version == v0
        assert len(calls) == 3
        cache.refresh()                         # evaluation after training: `dirty` forces the copy even if the version is unchanged
        assert len(calls) == 4 and torch.equal(cache.get(p, torch.bfloat16), p.detach().to(torch.bfloat16))
        cache.refresh()                         # second evaluation: nothing happened in between
        assert len(calls) == 4
    finally:
        torch._foreach_copy_ = orig
    assert fused_is_silent, "fused Adam now bumps _version: the `force` rule in ParamCastCache could be relaxed"


def test_cast_param_falls_back_to_a_plain_cast_without_a_cache():
    from heal_swin_amd import ops
    p = torch.nn.Parameter(torch.randn(4, 4))
    assert ops.RT.cast_cache is None
    assert ops._cast_param(p, torch.float32) is p
    c = ops._cast_param(p, torch.bfloat16)
    assert c.dtype == torch.bfloat16 and torch.equal(c, p.detach().to(torch.bfloat16))


def test_npz_sample_format_round_trip(tmp_path):
    """The reference's on-disk sample format (`hp_img` uint8 [3, Npix], `hp_mask` uint8 [Npix] in one .npz,
    project_on_s2.py:365-372 / hp_datasets.py:92-98): write, list, read, collate, DataSpec."""
    import numpy as np
    from heal_swin_amd import data as D
    rng = np.random.default_rng(0)
    npix = 8 * 16 * 16
    samples = {}
    for name in ("00002_FV", "00001_RV", "00010_MVL"):
        img = rng.integers(0, 256, (3, npix), dtype=np.uint8)
        mask = rng.integers(0, 10, npix, dtype=np.uint8)
        D.write_sample(tmp_path / f"{name}.npz", img, mask)
        samples[name] = (img, mask)
    (tmp_path / "metadata.txt").write_text("not a sample")
    ds = D.HPSegmentationNpzDataset(str(tmp_path))
    assert len(ds) == 3 and ds.names == sorted(samples)
    raw = np.load(tmp_path / "00001_RV.npz")
    assert sorted(raw.files) == ["hp_img", "hp_mask"]            # exactly the reference's keys
    for i, name in enumerate(ds.names):
        img, mask = ds[i]
        assert img.dtype == np.uint8 and img.shape == (3, npix) and mask.dtype == np.uint8 and mask.shape == (npix,)
        assert np.array_equal(img, samples[name][0]) and np.array_equal(mask, samples[name][1])
    assert np.array_equal(ds.get_item_by_name("00010_MVL")[1], samples["00010_MVL"][1])
    imgs, masks = D.collate_uint8([ds[0], ds[2]])
    assert imgs.dtype == torch.uint8 and tuple(imgs.shape) == (2, 3, npix) and tuple(masks.shape) == (2, npix)
    spec = D.data_spec_of(ds[0], n_classes=10)
    assert (spec.dim_in, spec.f_in, spec.f_out, spec.base_pix) == (npix, 3, 10, 8)
    assert D.data_spec_of((np.zeros((3, 12 * 256 * 256), np.uint8), None), 12).base_pix == 12


def test_fold_head_ce_permutes_class_rows_for_the_backward_kernel():
    """`hs_ln_head_ce_bwd` wants accumulator register r < 8 of lane half h to be class 8 h + r; the MFMA puts row m = (r & 3) +
    8 (r >> 2) + 4 h there, so ops._fold_head_ce exchanges row blocks 4..7 and 8..11 of the folded weight (hi rows 0..31 and lo rows
    32..63 alike) and of the bias vector."""
    import torch
    from heal_swin_amd import ops

    torch.manual_seed(0)
    C, f_out = 64, 12
    gamma, beta, w = torch.randn(C), torch.randn(C), torch.randn(f_out, C, 1)
    wf, bv = ops._fold_head(gamma, beta, w, C, "cpu")
    wp, bp = ops._fold_head_ce(gamma, beta, w, C, "cpu")
    for h in range(2):
        for r in range(8):
            m = (r & 3) + 8 * (r >> 2) + 4 * h  # accumulator row of register r in lane half h
            cls = 8 * h + r
            assert torch.equal(wp[m], wf[cls]) and torch.equal(wp[32 + m], wf[32 + cls]) and bp[m] == bv[cls]
    assert not wp[16:32].any() and not wp[48:].any()


def test_bf16x3_policy_and_legality_helpers():
    import torch
    from heal_swin_amd import ops

    class T:  # shape / dtype / device stand-in (the helpers never touch data)
        def __init__(self, k, dtype=torch.float32, cuda=True):
            self.shape, self.dtype, self.is_cuda = (4, k), dtype, cuda
    prev, prev_probe = ops.FP32_GEMM, ops._MM_OUT_DTYPE[0]
    try:
        ops.FP32_GEMM = "bf16x3"
        ops._MM_OUT_DTYPE[0] = False  # a PyTorch build without mm(out_dtype=): every product stays exact fp32 instead of raising later
        assert not ops._bf16x3_ok(T(512), 2048, 512)
        ops._MM_OUT_DTYPE[0] = True
        assert ops._bf16x3_ok(T(512), 2048, 512) and ops._bf16x3_ok(T(512), 512, 512)   # stage 2 of HEAL-SWIN-B: MFMA-bound in fp32
        assert not ops._bf16x3_ok(T(128), 384, 128) and not ops._bf16x3_ok(T(96), 288, 96)  # stage 0: HBM-bound, stays exact fp32
        assert not ops._bf16x3_ok(T(512, torch.bfloat16), 2048, 512) and not ops._bf16x3_ok(T(516), 2048, 516)
        ops.FP32_GEMM = "strict"
        assert not ops._bf16x3_ok(T(512), 2048, 512)
    finally:
        ops.FP32_GEMM, ops._MM_OUT_DTYPE[0] = prev, prev_probe
    assert ops.own_gemm_legal(128, 512, torch.bfloat16) and not ops.own_gemm_legal(12, 128, torch.bfloat16)
    assert not ops.own_gemm_legal(128, 512, torch.float32)


def test_own_gemm_policy_class_rule_tuner_picks_and_overrides(monkeypatch):
    """ops.own_gemm_ok: GELU / GELU' epilogues by their K limits; bias / residual products by the first-call tuner's pick where one
    exists (ops.GemmTuner; its trials need a GPU -- here its dictionary is filled by hand), else the class rule (HBM-bound shapes:
    narrow or short); HS_OWN_GEMM = 0 / 1 and the data-parallel preference override everything."""
    import torch
    from heal_swin_amd import _lib, ops
    from heal_swin_amd.ops import gemm as G
    bf = torch.bfloat16
    monkeypatch.setattr(ops, "OWN_GEMM", "auto")
    monkeypatch.setattr(ops.RT, "prefer_own_gemm", False)
    assert ops.own_gemm_ok(_lib.HS_EPI_BIAS, 384, 96, bf) and ops.own_gemm_ok(_lib.HS_EPI_BIAS, 128, 512, bf)  # class rule: narrow / short
    assert not ops.own_gemm_ok(_lib.HS_EPI_BIAS, 512, 2048, bf) and not ops.own_gemm_ok(_lib.HS_EPI_BIAS, 1536, 512, bf)
    assert not ops.own_gemm_ok(_lib.HS_EPI_BIAS, 1152, 384, bf, m=65536)  # (no GPU here: no trial, the class rule answers)
    assert ops.own_gemm_ok(_lib.HS_EPI_GELU, 2048, 512, bf) and ops.own_gemm_ok(_lib.HS_EPI_DGELU, 2048, 512, bf)  # epilogue products: own
    assert not ops.own_gemm_ok(_lib.HS_EPI_BIAS, 1152, 384, torch.float32) and not ops.own_gemm_ok(_lib.HS_EPI_BIAS, 1150, 384, bf)
    # a remembered pick is keyed by (power-of-two bucket of the rows, n, k) and decides for every row count of that bucket
    tuner = G.GemmTuner()
    tuner.picks[tuner.key(65536, 1152, 384)] = True
    tuner.picks[tuner.key(1 << 20, 128, 128)] = False
    monkeypatch.setattr(G, "GEMM_TUNER", tuner)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    assert tuner.key(65536, 1152, 384) == tuner.key(100000, 1152, 384) != tuner.key(32768, 1152, 384)
    assert tuner.pick(100000, 1152, 384, None) is True and tuner.pick(64, 64, 64, None) is None  # hit / too small for a trial
    assert ops.own_gemm_ok(_lib.HS_EPI_BIAS, 1152, 384, bf, m=100000) and not ops.own_gemm_ok(_lib.HS_EPI_BIAS, 128, 128, bf, m=1 << 20)
    assert ops.own_gemm_ok(_lib.HS_EPI_BIAS, 128, 128, bf, m=1024)  # another bucket, below the trial size: class rule
    monkeypatch.setattr(ops, "GEMM_TUNE", False)
    assert not ops.own_gemm_ok(_lib.HS_EPI_BIAS, 1152, 384, bf, m=100000) and ops.own_gemm_ok(_lib.HS_EPI_BIAS, 128, 128, bf, m=1 << 20)
    monkeypatch.setattr(ops, "GEMM_TUNE", True)
    monkeypatch.setattr(ops, "OWN_GEMM", "0")
    assert not ops.own_gemm_ok(_lib.HS_EPI_BIAS, 1152, 384, bf, m=100000) and not ops.own_gemm_ok(_lib.HS_EPI_GELU, 2048, 512, bf)
    monkeypatch.setattr(ops, "OWN_GEMM", "1")
    assert ops.own_gemm_ok(_lib.HS_EPI_BIAS, 128, 128, bf, m=1 << 20) and ops.own_gemm_ok(_lib.HS_EPI_BIAS, 512, 2048, bf)


def test_weight_split_cache_hits_for_fresh_views_and_follows_the_weight_epoch(monkeypatch):
    """callers pass `w.view(n, k)` / `w.reshape(...)` -- a new tensor object per call -- so an identity check
    never hit.  The entry is now found by the storage it views; it is dropped when the parameter's version OR the weight epoch
    (every grad-enabled forward: fused optimizers do not bump versions) has moved."""
    from heal_swin_amd import ops
    launches = []

    class FakeLib:
        def hs_split_bf16x3(self, *a):
            launches.append(a)
            return 0
    monkeypatch.setattr(ops, "lib", FakeLib())
    monkeypatch.setattr(ops, "ptr", lambda t: 0)
    monkeypatch.setattr(ops, "stream_ptr", lambda d: 0)
    ops._WSPLIT.clear()
    w = torch.nn.Parameter(torch.randn(16, 8))
    a = ops._weight_split(w.view(16, 8), False)
    b = ops._weight_split(w.reshape(16, 8), False)      # another view object of the same storage: a hit
    assert a is b and len(launches) == 1
    t1 = ops._weight_split(w.view(16, 8), True)
    t2 = ops._weight_split(w.detach().view(16, 8), True)
    assert t1 is t2 and t1 is not a and len(launches) == 2
    with torch.no_grad():
        w.add_(1.0)                                      # a non-fused optimizer step bumps the shared version counter
    assert ops._weight_split(w.view(16, 8), False) is not a and len(launches) == 3
    n = len(launches)
    ops.note_forward(True)                               # a training forward: parameters may have been stepped silently
    ops._weight_split(w.view(16, 8), False)
    assert len(launches) == n + 1
    ops.note_forward(False)                              # first evaluation after training: one more epoch ...
    ops._weight_split(w.view(16, 8), False)
    ops.note_forward(False)                              # ... later evaluations re-use the operand
    ops._weight_split(w.view(16, 8), False)
    assert len(launches) == n + 2
    for i in range(ops._WSPLIT_CAPACITY + 8):            # bounded: dead models do not pin their weights forever
        ops._weight_split(torch.zeros(8, 8), False)
    assert len(ops._WSPLIT) <= ops._WSPLIT_CAPACITY
    ops._WSPLIT.clear()
