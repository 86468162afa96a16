"""The floating-point oracle (oracle/model.py) vs golden vectors captured from the reference:
forward outputs, input gradients and every parameter gradient.  fp32 on CPU, tolerance 2e-5 relative to
the tensor's scale ("fp32 <= 1e-5 here"; slack for summation order)."""
import numpy as np
import pytest
import torch

from oracle import model as OM
from oracle import tables as T
from _golden import (case, load, state_dict, model_cfg_spec, ns, MODEL_CASES, REFINIT_MODEL_CASES, REFINIT_WA_CASES,
                     REFINIT_BLOCK_CASES, refinit_cfg_spec)


def close(a, b, tol=2e-5, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max()) / scale
    assert err <= tol, f"{what}: rel-to-scale err {err:.3e} > {tol}"


def run_and_check(c, fwd, tol=2e-5):
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in state_dict(c).items()}
    x = torch.from_numpy(c["x"]).requires_grad_(True)
    y = fwd(sd, x)
    close(y.detach().numpy(), c["y"], tol, "y")
    y.backward(torch.from_numpy(c["dy"]))
    close(x.grad.numpy(), c["dx"], tol, "dx")
    for k, g in c["grad"].items():
        got = sd[k].grad
        got = np.zeros_like(g) if got is None else got.numpy()
        close(got, g, tol, "grad " + k)


WA_CASES = [f"{a}_{m}" for a in ("scaled", "cos") for m in ("nomask", "rollmask", "ringmask")]


@pytest.mark.parametrize("name", WA_CASES)
def test_window_attention(name):
    c = case("modules", "window_attention/" + name)
    mask = torch.from_numpy(c["mask"].astype(np.float32)) if "mask" in c else None
    rel = torch.from_numpy(T.rel_pos_index(64))
    run_and_check(c, lambda sd, x: OM.window_attention(x, sd, "", 3, rel, mask, name.startswith("cos")))


@pytest.mark.parametrize("tag,nh", [("w16", 2), ("w4", 1)])
def test_window_attention_plain(tag, nh):
    c = case("modules", "window_attention/plain_" + tag)
    run_and_check(c, lambda sd, x: OM.window_attention(x, sd, "", nh, None, None, False))


def test_patch_merging():
    run_and_check(case("modules", "patch_merging"), lambda sd, x: OM.patch_merging(x, sd, ""))


def test_patch_expand():
    run_and_check(case("modules", "patch_expand"), lambda sd, x: OM.patch_expand(x, sd, ""))


def test_final_patch_expand():
    run_and_check(case("modules", "final_patch_expand"), lambda sd, x: OM.patch_expand(x, sd, "", p=4))


@pytest.mark.parametrize("v2", [False, True])
@pytest.mark.parametrize("sname,strat,shift", [("noshift", "nest_roll", 0), ("roll", "nest_roll", 8), ("ring", "ring_shift", 4),
                                               ("grid", "nest_grid_shift", 8)])
def test_block(v2, sname, strat, shift):
    c = case("modules", f"block/{'v2' if v2 else 'v1'}_{sname}")
    sh = OM.Shifter(strat, 512, 8, 16, shift)
    rel = torch.from_numpy(T.rel_pos_index(16))
    run_and_check(c, lambda sd, x: OM.swin_block(x, sd, "", 2, 16, sh, rel, v2, v2))
    # the oracle's mask equals the buffer the reference keeps in its state dict
    if shift:
        ref_mask = state_dict(c)["attn_mask"]
        assert np.array_equal(sh.attn_mask().numpy(), ref_mask.to(torch.float32).numpy())
        assert (str(c["sd_dtype"]["attn_mask"]) == "int64") == sh.mask_is_int


@pytest.mark.parametrize("name", list(MODEL_CASES))
def test_whole_model(name):
    c = case("models", "model/" + name)
    cfg, spec = model_cfg_spec(name)
    # ref_test_config (embed_dim = 2): LayerNorm over 2 channels of 0..255-scaled activations is ill-conditioned
    # in fp32 (the two fp32 evaluations differ by summation order only), hence the looser bound there.
    tol = 2e-3 if name == "ref_test_config" else 5e-5
    run_and_check(c, lambda sd, x: OM.forward(sd, ns(cfg), ns(spec), x), tol=tol)


# ----------------------------------------------------------------------------- reference-scale goldens (refinit.npz)
def close_own_scale(a, b, tol, what=""):
    """max|a-b| / max|b| with NO floor at 1: at the reference's init scale (sigma 0.02) gradients are 1e-3 .. 1e-6 in size."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = float(np.abs(b).max())
    err = float(np.abs(a - b).max())
    assert err <= tol * scale + 1e-12, f"{what}: err {err:.3e} vs scale {scale:.3e} (ratio {err / max(scale, 1e-300):.3e} > {tol})"


def run_and_check_own_scale(c, fwd, tol):
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in state_dict(c).items()}
    x = torch.from_numpy(c["x"]).requires_grad_(True)
    y = fwd(sd, x)
    close_own_scale(y.detach().numpy(), c["y"], tol, "y")
    y.backward(torch.from_numpy(c["dy"]))
    close_own_scale(x.grad.numpy(), c["dx"], tol, "dx")
    for k, g in c["grad"].items():
        got = sd[k].grad
        got = np.zeros_like(g) if got is None else got.numpy()
        close_own_scale(got, g, tol, "grad " + k)


@pytest.mark.parametrize("name", REFINIT_WA_CASES)
def test_refinit_window_attention(name):
    c = case("refinit", "window_attention/" + name)
    mask = torch.from_numpy(c["mask"].astype(np.float32)) if "mask" in c else None
    rel = torch.from_numpy(T.rel_pos_index(64))
    run_and_check_own_scale(c, lambda sd, x: OM.window_attention(x, sd, "", 4, rel, mask, name.startswith("cos")), 5e-5)


@pytest.mark.parametrize("v2,sname,strat,shift", REFINIT_BLOCK_CASES)
def test_refinit_block(v2, sname, strat, shift):
    c = case("refinit", f"block/{'v2' if v2 else 'v1'}_{sname}")
    sh = OM.Shifter(strat, 512, 8, 64, shift)
    rel = torch.from_numpy(T.rel_pos_index(64))
    run_and_check_own_scale(c, lambda sd, x: OM.swin_block(x, sd, "", 2, 64, sh, rel, v2, v2), 5e-5)


@pytest.mark.parametrize("name", list(REFINIT_MODEL_CASES))
def test_refinit_whole_model(name):
    c = case("refinit", "model/" + name)
    cfg, spec = refinit_cfg_spec(name)
    run_and_check_own_scale(c, lambda sd, x: OM.forward(sd, ns(cfg), ns(spec), x), 2e-4)


def test_seg_loss():
    z = load("losses")
    for tag in ("weighted", "uniform"):
        logits = torch.from_numpy(z["seg/logits"]).requires_grad_(True)
        loss = OM.seg_loss(logits, torch.from_numpy(z["seg/labels"]), torch.from_numpy(z[f"seg/{tag}/weights"]))
        close(loss.detach().numpy(), z[f"seg/{tag}/loss"], 1e-6, "loss")
        loss.backward()
        close(logits.grad.numpy(), z[f"seg/{tag}/dlogits"], 1e-6, "dlogits")
    assert np.array_equal(torch.from_numpy(z["seg/logits"]).argmax(1).numpy(), z["seg/argmax"])


def test_depth_losses():
    z = load("losses")
    cases = (("l1", OM.get_depth_loss("l1"), "depth/pred"), ("l2", OM.get_depth_loss("l2"), "depth/pred"),
             ("huber_d1", OM.get_depth_loss("huber", huber_delta=1), "depth/pred"),
             ("huber_d0p3", OM.get_depth_loss("huber", huber_delta=0.3), "depth/pred"),
             ("logvar", OM.get_depth_loss("l1", use_logvar=True), "depth/logvar/pred"))
    for tag, fn, pkey in cases:
        pred = torch.from_numpy(z[pkey]).requires_grad_(True)
        loss = fn(pred, torch.from_numpy(z["depth/target"]))
        close(loss.detach().numpy(), z[f"depth/{tag}/loss"], 1e-6, "loss")
        loss.backward()
        close(pred.grad.numpy(), z[f"depth/{tag}/dpred"], 1e-6, "dpred")
    d = torch.from_numpy(z["depth/standardize/in"])
    close(OM.depth_standardize(d).numpy(), z["depth/standardize/out"], 1e-6)
    close(OM.depth_unstandardize(OM.depth_standardize(d)).numpy(), z["depth/standardize/back"], 1e-6)
