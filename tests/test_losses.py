"""Caller-side losses of the product (heal_swin_amd/losses.py) vs the golden vectors captured from
the reference's CrossEntropyLoss / loss_depth_regression / normalize_depth_data.  Plain torch compositions: run on CPU
here and on the GPU in the gpu-marked variant."""
import numpy as np
import pytest
import torch

from _golden import load


def _check(dev):
    from heal_swin_amd import losses as L
    z = load("losses")
    for tag in ("weighted", "uniform"):
        logits = torch.from_numpy(z["seg/logits"]).to(dev).requires_grad_(True)
        loss = L.seg_loss(logits, torch.from_numpy(z["seg/labels"]).to(dev), torch.from_numpy(z[f"seg/{tag}/weights"]))
        assert abs(float(loss) - float(z[f"seg/{tag}/loss"])) < 1e-6
        loss.backward()
        assert np.abs(logits.grad.cpu().numpy() - z[f"seg/{tag}/dlogits"]).max() < 1e-7
    assert np.array_equal(L.seg_predictions(torch.from_numpy(z["seg/logits"]).to(dev)).cpu().numpy(), z["seg/argmax"])
    for tag, fn in (("l1", L.depth_l1_loss), ("l2", L.depth_l2_loss)):
        pred = torch.from_numpy(z["depth/pred"]).to(dev).requires_grad_(True)
        loss = fn(pred, torch.from_numpy(z["depth/target"]).to(dev))
        assert abs(float(loss) - float(z[f"depth/{tag}/loss"])) < 1e-6
        loss.backward()
        assert np.abs(pred.grad.cpu().numpy() - z[f"depth/{tag}/dpred"]).max() < 1e-7
    d = torch.from_numpy(z["depth/standardize/in"]).to(dev)
    assert np.allclose(L.depth_standardize(d).cpu().numpy(), z["depth/standardize/out"], rtol=1e-6, atol=1e-6)
    assert np.allclose(L.depth_unstandardize(L.depth_standardize(d)).cpu().numpy(), z["depth/standardize/back"], rtol=1e-6, atol=1e-5)
    # bf16 logits are up-cast: the loss of rounded logits equals the fp32 loss of the same rounded values
    lg = torch.from_numpy(z["seg/logits"]).to(dev).to(torch.bfloat16)
    a = L.seg_loss(lg, torch.from_numpy(z["seg/labels"]).to(dev))
    b = L.seg_loss(lg.float(), torch.from_numpy(z["seg/labels"]).to(dev))
    assert float(a) == float(b)


def test_losses_cpu():
    _check("cpu")


@pytest.mark.gpu
def test_losses_gpu():
    _check("cuda")
