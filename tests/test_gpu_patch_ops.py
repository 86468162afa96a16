"""`hs_patch_merge_fwd/bwd` and `hs_patch_expand_fwd/bwd` (one C-ABI call per module and direction) against
(i) the golden vectors captured from the reference's PatchMerging / PatchExpand / FinalPatchExpand_X4
(models_torch/swin_hp_transformer.py:364-452), forward and every gradient, and (ii) the oracle at BASELINE's stage widths."""
import pytest
import torch

from _golden import case, state_dict
from tests._util import GRAD_TOL, TOL, assert_close

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _golden(name):
    c = case("modules", name)
    sd = state_dict(c)
    x = torch.from_numpy(c["x"]).cuda().to(BF).requires_grad_(True)
    p = {k: v.cuda().float().requires_grad_(True) for k, v in sd.items()}
    return c, x, p


def _check(c, x, p, y, tag):
    assert_close(y, c["y"], TOL[BF], tag + " y")
    y.backward(torch.from_numpy(c["dy"]).cuda().to(BF))
    assert_close(x.grad, c["dx"], GRAD_TOL[BF], tag + " dx")
    for k, g in c["grad"].items():
        assert_close(p[k].grad, g, GRAD_TOL[BF], f"{tag} grad {k}")


def test_patch_merge_operator_matches_the_reference_golden():
    from heal_swin_amd import ops

    c, x, p = _golden("patch_merging")
    _check(c, x, p, ops.patch_merge(x, p["norm.weight"], p["norm.bias"], p["reduction.weight"]), "hs_patch_merge")


@pytest.mark.parametrize("name,children", [("patch_expand", 4), ("final_patch_expand", 4)])
def test_patch_expand_operator_matches_the_reference_golden(name, children):
    from heal_swin_amd import ops

    c, x, p = _golden(name)
    _check(c, x, p, ops.patch_expand(x, p["expand.weight"], p["norm.weight"], p["norm.bias"], children), "hs_" + name)


@pytest.mark.parametrize("B,N,C", [(2, 12288, 128), (1, 3072, 512), (3, 100, 96)])
def test_patch_ops_vs_oracle_at_stage_widths(B, N, C):
    """default-style parameters, BASELINE stage widths (C = 128 / 512 of HEAL-SWIN-B, 96 of -T with a ragged row count)."""
    from heal_swin_amd import ops
    from oracle import model as OM  # checker only

    torch.manual_seed(B * N + C)
    x = torch.randn(B, N, C).to(BF)
    sd = {"norm.weight": 1 + 0.2 * torch.randn(4 * C), "norm.bias": 0.1 * torch.randn(4 * C),
          "reduction.weight": torch.randn(2 * C, 4 * C) * 0.02}
    xr = x.float().requires_grad_(True)
    pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yr = OM.patch_merging(xr, pr, "")
    dy = torch.randn_like(yr).to(BF)
    yr.backward(dy.float())
    xg = x.cuda().requires_grad_(True)
    pg = {k: v.cuda().requires_grad_(True) for k, v in sd.items()}
    y = ops.patch_merge(xg, pg["norm.weight"], pg["norm.bias"], pg["reduction.weight"])
    y.backward(dy.cuda())
    tag = f"hs_patch_merge[{B}x{N}x{C}]"
    assert_close(y, yr, TOL[BF], tag + " y")
    assert_close(xg.grad, xr.grad, GRAD_TOL[BF], tag + " dx")
    for k in sd:
        assert_close(pg[k].grad, pr[k].grad, GRAD_TOL[BF], f"{tag} grad {k}")

    sd = {"expand.weight": torch.randn(2 * C, C) * 0.02, "norm.weight": 1 + 0.2 * torch.randn(C // 2), "norm.bias": 0.1 * torch.randn(C // 2)}
    xr = x.float().requires_grad_(True)
    pr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yr = OM.patch_expand(xr, pr, "")
    dy = torch.randn_like(yr).to(BF)
    yr.backward(dy.float())
    xg = x.cuda().requires_grad_(True)
    pg = {k: v.cuda().requires_grad_(True) for k, v in sd.items()}
    y = ops.patch_expand(xg, pg["expand.weight"], pg["norm.weight"], pg["norm.bias"], 4)
    y.backward(dy.cuda())
    tag = f"hs_patch_expand[{B}x{N}x{C}]"
    assert_close(y, yr, TOL[BF], tag + " y")
    assert_close(xg.grad, xr.grad, GRAD_TOL[BF], tag + " dx")
    for k in sd:
        assert_close(pg[k].grad, pr[k].grad, GRAD_TOL[BF], f"{tag} grad {k}")


def test_patch_ops_reject_fp32_loudly():
    from heal_swin_amd import ops

    x = torch.randn(1, 64, 32, device="cuda", requires_grad=True)
    with pytest.raises(RuntimeError, match="bf16"):
        ops.patch_merge(x, torch.ones(128, device="cuda"), torch.zeros(128, device="cuda"), torch.randn(64, 128, device="cuda"))
