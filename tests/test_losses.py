"""Caller-side losses of the product (heal_swin_amd/losses.py) vs the golden vectors captured from
the reference's CrossEntropyLoss / loss_depth_regression / normalize_depth_data.  The depth losses are plain torch compositions (checked on CPU here and on the GPU); the segmentation
cross-entropy exists only as HIP kernels (GPU test; the oracle restatement is pinned in tests/test_oracle_model.py)."""
import numpy as np
import pytest
import torch

from _golden import load


def _check_depth(dev):
    from types import SimpleNamespace as NS

    from heal_swin_amd import losses as L
    z = load("losses")
    tgt = torch.from_numpy(z["depth/target"]).to(dev)
    cases = [("l1", L.get_depth_loss(NS(use_logvar=False, loss="l1", huber_delta=1)), "depth/pred"),
             ("l2", L.get_depth_loss(NS(use_logvar=False, loss="l2", huber_delta=1)), "depth/pred"),
             ("huber_d1", L.get_depth_loss(NS(use_logvar=False, loss="huber", huber_delta=1)), "depth/pred"),
             ("huber_d0p3", L.get_depth_loss(NS(use_logvar=False, loss="huber", huber_delta=0.3)), "depth/pred"),
             ("logvar", L.get_depth_loss(NS(use_logvar=True, loss="l1", huber_delta=1)), "depth/logvar/pred")]
    assert cases[0][1] is L.depth_l1_loss and cases[1][1] is L.depth_l2_loss and cases[4][1] is L.depth_mean_log_var_loss
    for tag, fn, pkey in cases:
        pred = torch.from_numpy(z[pkey]).to(dev).requires_grad_(True)
        loss = fn(pred, tgt)
        assert abs(float(loss) - float(z[f"depth/{tag}/loss"])) < 1e-6, tag
        loss.backward()
        assert np.abs(pred.grad.cpu().numpy() - z[f"depth/{tag}/dpred"]).max() < 2.5e-7, tag  # GPU exp: 1 ulp of the fp32 gradient
    d = torch.from_numpy(z["depth/standardize/in"]).to(dev)
    assert np.allclose(L.depth_standardize(d).cpu().numpy(), z["depth/standardize/out"], rtol=1e-6, atol=1e-6)
    assert np.allclose(L.depth_unstandardize(L.depth_standardize(d)).cpu().numpy(), z["depth/standardize/back"], rtol=1e-6, atol=1e-5)


def _check_seg(dev):
    from heal_swin_amd import losses as L
    z = load("losses")
    for tag in ("weighted", "uniform"):
        logits = torch.from_numpy(z["seg/logits"]).to(dev).requires_grad_(True)
        loss = L.seg_loss(logits, torch.from_numpy(z["seg/labels"]).to(dev), torch.from_numpy(z[f"seg/{tag}/weights"]))
        assert abs(float(loss) - float(z[f"seg/{tag}/loss"])) < 1e-6
        loss.backward()
        assert np.abs(logits.grad.cpu().numpy() - z[f"seg/{tag}/dlogits"]).max() < 1e-7
    assert np.array_equal(L.seg_predictions(torch.from_numpy(z["seg/logits"]).to(dev)).cpu().numpy(), z["seg/argmax"])
    # bf16 logits are up-cast: the loss of rounded logits equals the fp32 loss of the same rounded values
    lg = torch.from_numpy(z["seg/logits"]).to(dev).to(torch.bfloat16)
    a = L.seg_loss(lg, torch.from_numpy(z["seg/labels"]).to(dev))
    b = L.seg_loss(lg.float(), torch.from_numpy(z["seg/labels"]).to(dev))
    assert float(a) == float(b)


def test_depth_losses_cpu():
    _check_depth("cpu")  # thin torch compositions: device-agnostic host logic


def test_seg_loss_has_no_cpu_path():
    from heal_swin_amd import losses as L
    z = load("losses")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.seg_loss(torch.from_numpy(z["seg/logits"]), torch.from_numpy(z["seg/labels"]))


@pytest.mark.gpu
def test_losses_gpu():
    _check_seg("cuda")
    _check_depth("cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_seg_loss_padded_row_fast_path_equals_the_generic_kernels(dtype):
    """The model's own logits layout -- [B, Npix, 16] rows with K = 12 used columns, viewed as [B, K, Npix] -- takes the
    `seg_ce_*_row16` kernels (bf16: 32-byte rows, fp32: 64-byte rows since round 3); a class-major copy of the same values takes
    the generic strided kernels.  Same loss and the same gradient, which arrives in the padded buffer's layout with zero pads."""
    from heal_swin_amd import losses as L
    g = torch.Generator(device="cuda").manual_seed(1)
    B, P, K = 2, 5000, 12
    rows = (torch.randn(B, P, 16, generator=g, device="cuda") * 3).to(dtype)
    rows[..., K:] = 0
    labels = torch.randint(0, K, (B, P), generator=g, device="cuda", dtype=torch.uint8)
    w = torch.rand(K, generator=g, device="cuda") + 0.5
    fast = rows.clone().requires_grad_(True)
    lf = L.seg_loss(fast[..., :K].transpose(1, 2), labels, w)
    lf.backward()
    slow = rows[..., :K].transpose(1, 2).contiguous().requires_grad_(True)  # [B, K, P] class-major: the generic kernels
    ls = L.seg_loss(slow, labels, w)
    ls.backward()
    assert abs(float(lf) - float(ls)) <= 1e-6 * abs(float(ls))
    assert torch.equal(fast.grad[..., K:], torch.zeros_like(fast.grad[..., K:]))
    assert torch.allclose(fast.grad[..., :K].transpose(1, 2).float(), slow.grad.float(), rtol=0, atol=1e-8 if dtype == torch.float32 else 0)
    # a view that starts in the middle of a padded row must NOT take the whole-row loads
    mid = rows.clone()[..., 4:16].transpose(1, 2)
    lm = L.seg_loss(mid, labels, w)
    lr = L.seg_loss(mid.contiguous(), labels, w)
    assert abs(float(lm) - float(lr)) <= 1e-6 * abs(float(lr))
