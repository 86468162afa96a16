"""End-to-end training sanity on the GPU: a few Adam steps on a fixed synthetic batch must reduce the loss through every
custom forward/backward kernel, for both callers of the hot path:
segmentation (CE, bf16 activations, ring_shift + cosine + v2, paper drop rates) and depth regression (masked L1 over
non-inf targets + standardisation, fp32, nest_roll, f_out = 1: BASELINE config 5's path at a small size)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(cfg_kw, bp, nside, f_out):
    from heal_swin_amd.data_spec import DataSpec
    from heal_swin_amd.models_torch.swin_hp_transformer import SwinHPTransformerConfig, SwinHPTransformerSys
    cfg = dict(patch_size=4, window_size=64, rel_pos_bias="flat", embed_dim=64, depths=[2, 2], num_heads=[2, 4])
    cfg.update(cfg_kw)
    spec = DataSpec(dim_in=bp * nside * nside, f_in=3, f_out=f_out, base_pix=bp, class_names=[])
    torch.manual_seed(0)
    return SwinHPTransformerSys(SwinHPTransformerConfig(**cfg), spec).to(DEV), spec


def test_segmentation_training_reduces_loss_bf16_paper_style():
    from heal_swin_amd.losses import seg_loss, seg_predictions
    model, spec = _model(dict(shift_size=4, shift_strategy="ring_shift", use_cos_attn=True, use_v2_norm_placement=True,
                              drop_rate=0.1, attn_drop_rate=0.1, drop_path_rate=0.1), 8, 32, 6)
    model.compute_dtype = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randint(0, 256, (4, 3, spec.dim_in), generator=g, device=DEV).float()
    # labels correlated with the input so there is something to learn
    y = (x[:, 0] // 43).clamp(max=5).to(torch.uint8)
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    model.train()
    losses = []
    for _ in range(40):
        opt.zero_grad()
        loss = seg_loss(model(x), y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(l == l for l in losses), "NaN loss"
    assert losses[-1] < 0.6 * losses[0], losses[::8]
    model.eval()
    with torch.no_grad():
        acc = float((seg_predictions(model(x)) == y).float().mean())
    assert acc > 0.5, acc


def test_depth_training_reduces_masked_l1_fp32():
    from heal_swin_amd import losses as L
    model, spec = _model(dict(shift_size=32, shift_strategy="nest_roll", drop_path_rate=0.0), 12, 32, 1)
    g = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randint(0, 256, (2, 3, spec.dim_in), generator=g, device=DEV).float()
    depth = 5.0 + x.mean(1) / 16.0                      # metres, a smooth function of the input
    depth[torch.rand(depth.shape, generator=g, device=DEV) < 0.04] = float("inf")  # background, as in the reference data
    target = L.depth_standardize(depth)                 # inf stays inf
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    model.train()
    losses = []
    for _ in range(40):
        opt.zero_grad()
        loss = L.depth_l1_loss(model(x), target)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(l == l for l in losses), "NaN loss"
    assert losses[-1] < 0.5 * losses[0], losses[::8]
    with torch.no_grad():
        pred_m = L.depth_unstandardize(model(x)[:, 0])
    assert torch.isfinite(pred_m).all()
