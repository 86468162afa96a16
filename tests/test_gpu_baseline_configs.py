"""Parity on BASELINE.json's own architectures, default-initialised, against the oracle (forward AND backward):

  configs[1]  HEAL-SWIN-T, nside 128, 8 base pixels, window 64, nest_roll -- at its full size, bf16 and fp32
  configs[2]  HEAL-SWIN-B (embed 128, depths [2,2,18,2], heads [4,8,16,32]), 12 base pixels, window 64 -- at nside 64
              (the oracle's 46 blocks at nside 256 take minutes per image on the host)
  paper       HEAL-SWIN-T ring_shift(4) + cosine attention + v2 norm placement at nside 64, logit_scale at its init (ln 10)

Weights as `SwinHPTransformerSys.__init__` leaves them (trunc-normal 0.02 Linears, unit LayerNorms) except the
relative-position table, which the reference initialises to zero and is drawn N(0, 0.02) here so that the bias path
carries signal.  Inputs are raw 0..255 images and random labels; the loss is the segmentation caller's cross entropy.
north_star: logits within 1e-3 (fp32) / 1e-2 (bf16) of the reference -- asserted on max|a-b|/max|b|, with the observed
errors (also rms and element-relative) printed in the terminal summary.  The goldens with one attention head pinned at
the x100 logit clamp are a separate, explicitly labelled stress case (tests/test_gpu_model.py, test_gpu_kernels.py).
"""
import types

import pytest
import torch

import conftest
from _util import assert_close, errors, assert_unbiased

pytestmark = pytest.mark.gpu
DEV = "cuda"

T_CFG = dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24])
B_CFG = dict(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32])
CASES = {
    # name: (arch, nside, base_pix, batch, overrides)
    "configs1_T_nside128_bp8": (T_CFG, 128, 8, 1, dict(shift_strategy="nest_roll", shift_size=32)),  # (one image: the oracle's fwd + bwd is the suite's slowest item)
    "configs2_B_nside64_bp12": (B_CFG, 64, 12, 1, dict(shift_strategy="nest_roll", shift_size=32)),
    "paper_T_ring_cos_v2_nside64_bp8": (T_CFG, 64, 8, 1, dict(shift_strategy="ring_shift", shift_size=4, use_cos_attn=True,
                                                                use_v2_norm_placement=True)),
}
# logits: north_star; parameter / input gradients: two passes through every bf16 activation
LOGIT_TOL = {torch.float32: 1e-3, torch.bfloat16: 1e-2}
STAGE_TOL = {torch.float32: 1e-4, torch.bfloat16: 3.5e-2}  # per-stage activations of the full-size models (max |a - b| / max |b|)
GRAD_TOL = {torch.float32: 2e-3, torch.bfloat16: 5e-2}
_ORACLE = {}


def _setup(name):
    from heal_swin_amd.data_spec import DataSpec
    from heal_swin_amd.models_torch import swin_hp_transformer as M
    arch, nside, bp, batch, over = CASES[name]
    cfg = dict(patch_size=4, window_size=64, shift_size=32, shift_strategy="nest_roll", rel_pos_bias="flat", mlp_ratio=4.0,
               qkv_bias=True, qk_scale=None, use_cos_attn=False, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0,
               use_v2_norm_placement=False, ape=False)
    cfg.update(arch)
    cfg.update(over)
    spec = dict(dim_in=bp * nside * nside, f_in=3, f_out=12, base_pix=bp, class_names=[])
    torch.manual_seed(11)
    model = M.SwinHPTransformerSys(M.SwinHPTransformerConfig(**cfg), DataSpec(**spec))
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("relative_position_bias_table"):
                p.normal_(0, 0.02)
    g = torch.Generator().manual_seed(5)
    x = torch.randint(0, 256, (batch, 3, spec["dim_in"]), generator=g).float()
    y = torch.randint(0, 12, (batch, spec["dim_in"]), generator=g)
    return model, cfg, spec, x, y


def _oracle(name):
    """Oracle logits, loss and gradients of the case (CPU fp32, computed once per session)."""
    if name not in _ORACLE:
        from oracle import model as OM
        model, cfg, spec, x, y = _setup(name)
        sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in model.state_dict().items()
              if not k.endswith("attn_mask")}
        xr = x.clone().requires_grad_(True)
        logits = OM.forward(sd, types.SimpleNamespace(**cfg), types.SimpleNamespace(**spec), xr)
        loss = OM.seg_loss(logits, y)
        loss.backward()
        grads = {k: v.grad.detach() for k, v in sd.items() if v.requires_grad and v.grad is not None}
        _ORACLE[name] = (logits.detach(), float(loss), grads, xr.grad.detach())
    return _ORACLE[name]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("name", list(CASES))
def test_baseline_architecture_fwd_bwd_vs_oracle(name, dtype):
    from heal_swin_amd.losses import seg_loss
    ref_logits, ref_loss, ref_grads, ref_dx = _oracle(name)
    model, cfg, spec, x, y = _setup(name)
    model = model.to(DEV).train()
    model.compute_dtype = dtype
    xg = x.to(DEV).requires_grad_(True)
    logits = model(xg)
    assert logits.dtype == torch.float32 and logits.shape == ref_logits.shape  # logits leave the model in fp32 whatever the compute dtype
    loss = seg_loss(logits, y.to(DEV))
    loss.backward()
    torch.cuda.synchronize()

    tag = f"{name}[{'bf16' if dtype == torch.bfloat16 else 'fp32'}]"
    e = errors(logits, ref_logits)
    conftest.NOTES.append(f"{tag}: logits max|a-b|/max|b| {e['scale_err']:.2e} (scale {e['scale']:.2f}), rms {e['rms_err']:.2e}, "
                          f"elem99.9 {e['elem_err']:.2e}; loss {float(loss):.6f} vs oracle {ref_loss:.6f}")
    assert_close(logits, ref_logits, LOGIT_TOL[dtype], tag + " logits")
    assert abs(float(loss) - ref_loss) <= (1e-4 if dtype == torch.float32 else 2e-3) * max(1.0, abs(ref_loss)), (float(loss), ref_loss)

    params = dict(model.named_parameters())
    worst = ("", 0.0)
    rms = []
    ls_pairs = []  # (name, bf16 d logit_scale, oracle d logit_scale) of every cosine-attention module
    for k, g in ref_grads.items():
        got = params[k].grad
        got = torch.zeros_like(params[k]) if got is None else got
        eg = errors(got, g)
        rms.append(eg["rms_err"])
        if eg["scale_err"] > worst[1]:
            worst = (k, eg["scale_err"])
        # biases in front of a LayerNorm-free softmax (k bias of scaled attention) have an exactly-zero gradient in the
        # reference: compare those absolutely against the scale of their weight's gradient
        floor = 1e-6 * float(ref_grads[k.replace(".bias", ".weight")].abs().max()) if k.endswith(".bias") else 0.0
        tol = GRAD_TOL[dtype]
        if dtype == torch.bfloat16 and cfg["use_cos_attn"]:
            # cosine attention + v2 norm placement: the L2-normalisation Jacobian amplifies the bf16 rounding of q, k, and the
            # branch-end LayerNorm weights have gradients of scale 1e-5 (observed up to 8.9e-2 of the tensor's scale)
            tol = 0.15
        if k.endswith("logit_scale") and dtype == torch.bfloat16:
            # one scalar per head = sum over every (window, query, key) of dS * S_raw with terms of both signs: results of
            # 1e-7 .. 1e-5 from terms of 1e-2.  The kernel accumulates it in fp32 from fp32 scores (D = rowsum(P o dP) from the
            # same registers since round 3, so sum_k dS = 0 holds to fp32 rounding); what remains is the bf16 rounding of the
            # ACTIVATIONS feeding it.  Checked per tensor against the noise level of the reference's own scale, and over all
            # heads of the model as a direction (cosine similarity, below)
            assert float((got.float().cpu() - g).abs().max()) <= 1e-5, (k, float((got.float().cpu() - g).abs().max()))
            ls_pairs.append((k, got.float().cpu().reshape(-1), g.reshape(-1)))
            continue
        assert_close(got, g, tol, f"{tag} grad {k}", floor=floor + (1e-7 if dtype == torch.bfloat16 else 1e-9))
        assert_unbiased(got, g, f"{tag} grad {k}")  # (a systematic 10 % error, whatever the element bound above admits)
    if ls_pairs:
        a = torch.cat([p_[1] for p_ in ls_pairs]).double()
        b = torch.cat([p_[2] for p_ in ls_pairs]).double()
        cos = float((a * b).sum() / (a.norm() * b.norm()).clamp_min(1e-300))
        rel = float((a - b).norm() / b.norm().clamp_min(1e-300))
        worst_ls = max(ls_pairs, key=lambda t: float((t[1] - t[2]).abs().max() / t[2].abs().max().clamp_min(1e-30)))
        conftest.NOTES.append(f"{tag}: d logit_scale over {a.numel()} heads: cosine {cos:.4f}, ||a-b||/||b|| {rel:.3f}; worst tensor "
                              f"{worst_ls[0]} max|a-b|/max|b| {float((worst_ls[1] - worst_ls[2]).abs().max() / worst_ls[2].abs().max()):.2f}")
        # observed on MI355X (paper config, 228 heads): cosine 1.0000, ||a-b||/||b|| 9e-3, worst single tensor 0.10 of its scale.
        # A sign / scale error in the dscale or normalisation-Jacobian path would give a cosine near 0 or negative
        assert cos >= 0.995 and rel <= 5e-2, (cos, rel)
    edx = errors(xg.grad, ref_dx)
    assert_close(xg.grad, ref_dx, GRAD_TOL[dtype], tag + " dx")
    conftest.NOTES.append(f"{tag}: {len(ref_grads)} parameter gradients, worst max|a-b|/max|b| {worst[1]:.2e} ({worst[0]}), "
                          f"median rms {sorted(rms)[len(rms) // 2]:.2e}; input gradient {edx['scale_err']:.2e}")


# ----------------------------------------------------------------------------- configs[2] at its FULL size (forward), 3 seeds
_FULL = {}
FULL_SEEDS = [11, 12, 13]  # weights AND inputs are re-drawn per seed (seed 11 = the round-1/2 case)


def _setup_seeded(name, seed):
    """_setup with the weight seed `seed` and the input seed `seed - 6` (seed 11 reproduces _setup exactly)."""
    from heal_swin_amd.data_spec import DataSpec
    from heal_swin_amd.models_torch import swin_hp_transformer as M
    arch, nside, bp, batch, over = CASES[name]
    cfg = dict(patch_size=4, window_size=64, shift_size=32, shift_strategy="nest_roll", rel_pos_bias="flat", mlp_ratio=4.0,
               qkv_bias=True, qk_scale=None, use_cos_attn=False, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0,
               use_v2_norm_placement=False, ape=False)
    cfg.update(arch)
    cfg.update(over)
    spec = dict(dim_in=bp * nside * nside, f_in=3, f_out=12, base_pix=bp, class_names=[])
    torch.manual_seed(seed)
    model = M.SwinHPTransformerSys(M.SwinHPTransformerConfig(**cfg), DataSpec(**spec))
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("relative_position_bias_table"):
                p.normal_(0, 0.02)
    g = torch.Generator().manual_seed(seed - 6)
    x = torch.randint(0, 256, (batch, 3, spec["dim_in"]), generator=g).float()
    y = torch.randint(0, 12, (batch, spec["dim_in"]), generator=g)
    return model, cfg, spec, x, y


def _full_size_oracle(seed):
    """HEAL-SWIN-B, nside 256, 12 base pixels (786 432 pixels, 46 blocks), one image: oracle logits, loss and the per-stage
    activations, forward only under no_grad (the oracle's autograd graph of this size holds every [windows, heads, 64, 64]
    score tensor: tens of GB)."""
    if seed not in _FULL:
        from oracle import model as OM
        CASES["_full"] = (B_CFG, 256, 12, 1, dict(shift_strategy="nest_roll", shift_size=32))
        model, cfg, spec, x, y = _setup_seeded("_full", seed)
        del CASES["_full"]
        sd = {k: v.detach() for k, v in model.state_dict().items() if not k.endswith("attn_mask")}
        torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
        taps = {}
        with torch.no_grad():
            logits = OM.forward(sd, types.SimpleNamespace(**cfg), types.SimpleNamespace(**spec), x, taps=taps)
            loss = float(OM.seg_loss(logits, y))
        _FULL.clear()  # one seed at a time: the per-stage activations of a case are 0.5 GB
        _FULL[seed] = dict(model=model, x=x, y=y, logits=logits, loss=loss, taps=taps)
    return _FULL[seed]


def _stage_taps(model):
    """Forward hooks recording the product's activations at the oracle's tap points."""
    got, handles = {}, []

    def hook(name):
        return lambda mod, inp, out: got.__setitem__(name, (out[0] if isinstance(out, tuple) else out).detach())

    for i, layer in enumerate(model.layers):
        handles.append(layer.register_forward_hook(hook(f"layers.{i}")))
    handles.append(model.norm.register_forward_hook(hook("norm")))
    for k, layer in enumerate(model.decoder.layers_up):
        handles.append(layer.register_forward_hook(hook(f"decoder.layers_up.{k}")))
    return got, handles


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("seed", FULL_SEEDS)
def test_headline_config_full_size_logits_vs_oracle(seed, dtype):
    """BASELINE configs[2] exactly as bench.py runs it (HEAL-SWIN-B, nside 256, 12 base pixels, window 64, nest_roll 32), default
    initialisation + N(0, 0.02) bias tables, on THREE independent draws of weights and inputs: logits within north_star's 1e-3
    (fp32) / 1e-2 (bf16) of the oracle, CE loss equal.  The per-stage error profile is printed in the summary so that an
    excursion can be attributed to a layer instead of being tolerated."""
    from heal_swin_amd.losses import seg_loss
    f = _full_size_oracle(seed)
    model = f["model"].to(DEV).eval()
    model.compute_dtype = dtype
    for grad_mode in (True, False):  # the training kernels and the no-grad path (stage 0 through the one-launch module kernel)
        got, handles = _stage_taps(model)
        with torch.set_grad_enabled(grad_mode):
            logits = model(f["x"].to(DEV))
            loss = float(seg_loss(logits, f["y"].to(DEV)))
        for hd in handles:
            hd.remove()
        tag = f"configs2_B_nside256_bp12_FULL[seed {seed}, {'bf16' if dtype == torch.bfloat16 else 'fp32'}{'' if grad_mode else ', no_grad'}]"
        e = errors(logits, f["logits"])
        profile = " ".join(f"{k.replace('decoder.layers_up', 'up').replace('layers', 'enc')}={errors(got[k], f['taps'][k])['scale_err']:.1e}"
                           for k in f["taps"] if k in got)
        conftest.NOTES.append(f"{tag}: logits max|a-b|/max|b| {e['scale_err']:.2e} (scale {e['scale']:.2f}), rms {e['rms_err']:.2e}; "
                              f"loss {loss:.6f} vs oracle {f['loss']:.6f}; per stage: {profile}")
        assert_close(logits, f["logits"], LOGIT_TOL[dtype], tag + " logits")
        assert abs(loss - f["loss"]) <= (1e-4 if dtype == torch.float32 else 2e-3) * max(1.0, abs(f["loss"]))
        # the per-stage profile is a bound, not a print: max |a - b| over the stage's ~10^8 activations against their own scale
        # (observed on MI355X, three seeds: fp32 <= 1.8e-5; bf16 <= 2.7e-2 -- the residual stream of a stage is rounded at each of its
        # up to 36 adds; carried as hi + lo, test_headline_config_full_size_compensated_stream, it is <= 1.4e-2)
        for k in f["taps"]:
            if k in got:
                se = errors(got[k], f["taps"][k])["scale_err"]
                assert se <= STAGE_TOL[dtype], f"{tag}: stage {k} at {se:.2e} of its scale (bound {STAGE_TOL[dtype]:.1e})"
    f["model"].cpu()


def test_headline_config_full_size_compensated_stream():
    """The same model and inputs (the last seed: its oracle pass is still cached; bf16) with the residual stream carried as hi + lo (`ops.COMP_RESIDUAL`, HS_COMP_RESIDUAL=1):
    every stage within 1.5e-2 of the oracle's activations (plain stream: up to 2.7e-2), logits within north_star's 1e-2.  Measured cost
    of the option on MI355X: 144.5 -> 153.8 ms per HEAL-SWIN-B step (it excludes the fused stage-0 kernels and the residual epilogues),
    45.7 -> 48.8 ms on the paper config -- which is why it is an option and not the default (profiles/r05_comp_residual_full_size.txt)."""
    from heal_swin_amd import ops
    seed = FULL_SEEDS[-1]
    f = _full_size_oracle(seed)
    model = f["model"].to(DEV).eval()
    model.compute_dtype = torch.bfloat16
    prev = ops.COMP_RESIDUAL
    ops.COMP_RESIDUAL = True
    try:
        got, handles = _stage_taps(model)
        logits = model(f["x"].to(DEV))
        for hd in handles:
            hd.remove()
    finally:
        ops.COMP_RESIDUAL = prev
    tag = f"configs2_B_nside256_bp12_FULL[seed {seed}, bf16, compensated stream]"
    prof = {k: errors(got[k], f["taps"][k])["scale_err"] for k in f["taps"] if k in got}
    conftest.NOTES.append(f"{tag}: logits {errors(logits, f['logits'])['scale_err']:.2e}; per stage: " + " ".join(f"{k}={v:.1e}" for k, v in prof.items()))
    assert_close(logits, f["logits"], LOGIT_TOL[torch.bfloat16], tag + " logits")
    for k, v in prof.items():
        assert v <= 1.5e-2, f"{tag}: stage {k} at {v:.2e}"
    f["model"].cpu()


def test_headline_config_full_size_gradients_vs_oracle():
    """BASELINE configs[2] at its full size, the BACKWARD against the oracle itself: the oracle's autograd graph of this model holds
    every [windows, heads, 64, 64] score tensor (~0.1 TB of host memory with its saved intermediates), which the MI355X boxes have
    (the `cpu_baseline` leg of bench.py runs the same pass); skipped on hosts with less than 160 GiB available.  One image, last seed
    of the forward tests (model and inputs shared), CE loss; every parameter gradient of the fp32 kernels and of the bf16 training
    kernels against the oracle's, each tensor on its own scale."""
    import psutil
    from heal_swin_amd.losses import seg_loss
    from oracle import model as OM
    if psutil.virtual_memory().available < 160 * 2 ** 30:
        pytest.skip("the oracle's full-size autograd graph needs ~0.1 TB of host memory")
    seed = FULL_SEEDS[-1]
    f = _full_size_oracle(seed)
    model = f["model"].cpu()
    CASES["_full"] = (B_CFG, 256, 12, 1, dict(shift_strategy="nest_roll", shift_size=32))
    try:
        _, cfg, spec, _, _ = _setup_seeded("_full", seed)
    finally:
        del CASES["_full"]
    names = [n for n, _ in model.named_parameters()]
    sd = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.endswith("attn_mask")}
    for n in names:
        sd[n].requires_grad_(True)
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    logits = OM.forward(sd, types.SimpleNamespace(**cfg), types.SimpleNamespace(**spec), f["x"])
    loss_o = OM.seg_loss(logits, f["y"])
    loss_o.backward()
    ref = {n: sd[n].grad for n in names}
    del logits
    assert abs(float(loss_o.detach()) - f["loss"]) < 1e-5
    model = model.to(DEV).train()  # (all drop rates are 0: train() only selects the training kernels)
    xg, yg = f["x"].to(DEV), f["y"].to(DEV)
    report = []
    for dtype, tol_scale, tol_rms in ((torch.float32, 2e-4, 2e-4), (torch.bfloat16, 8e-2, 6e-2)):  # observed 4.2e-5 / 3.2e-5 and 4.6e-2 / 3.9e-2  # (bf16: the rel-pos tables of the deep stages)
        model.compute_dtype = dtype
        tag_ = "fp32" if dtype == torch.float32 else "bf16"
        model.zero_grad(set_to_none=True)
        loss = seg_loss(model(xg), yg)
        loss.backward()
        worst_s, worst_r, rms_all = ("", 0.0), ("", 0.0), []
        for n, p_ in model.named_parameters():
            assert p_.grad is not None, n
            e = errors(p_.grad.float().cpu(), ref[n])
            assert_unbiased(p_.grad.float().cpu(), ref[n], f"configs2_B_FULL {tag_} grad {n}")
            rms_all.append(e["rms_err"])
            if e["scale_err"] > worst_s[1]:
                worst_s = (n, e["scale_err"])
            if e["rms_err"] > worst_r[1]:
                worst_r = (n, e["rms_err"])
        tag = "fp32" if dtype == torch.float32 else "bf16"
        report.append(f"{tag}: worst max|a-b|/max|b| {worst_s[1]:.2e} ({worst_s[0]}), worst rms {worst_r[1]:.2e} ({worst_r[0]}), "
                      f"median rms {sorted(rms_all)[len(rms_all) // 2]:.2e}")
        assert worst_s[1] <= tol_scale, (tag, worst_s)
        assert worst_r[1] <= tol_rms, (tag, worst_r)
    conftest.NOTES.append(f"configs2_B_nside256_bp12_FULL[seed {seed}] parameter gradients vs the ORACLE's autograd ({len(names)} tensors): " + "; ".join(report))
    model.zero_grad(set_to_none=True)
    f["model"].cpu()


# ----------------------------------------------------------------------------- the paper's run config at its FULL size (forward)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_paper_config_full_size_logits_vs_oracle(dtype):
    """The paper's model exactly as its run config builds it (HEAL-SWIN-T, nside 256, 8 base pixels = 524 288 pixels, window 64,
    ring_shift 4, cosine attention, v2 norm placement; run_configs/segmentation/swin_hp_*_train_run_config.py:48-65) -- also
    BASELINE configs[3]'s non-trivial shift permutation at nside 256: the ring-shift tables of all four stages, the int64-mask
    semantics and the cosine path at full size.  Default initialisation + N(0, 0.02) bias tables, one image, forward under
    no_grad on the oracle side: logits within north_star's 1e-3 (fp32) / 1e-2 (bf16), CE loss equal."""
    from heal_swin_amd.losses import seg_loss
    from oracle import model as OM
    CASES["_paper_full"] = (T_CFG, 256, 8, 1, dict(shift_strategy="ring_shift", shift_size=4, use_cos_attn=True, use_v2_norm_placement=True))
    try:
        model, cfg, spec, x, y = _setup_seeded("_paper_full", 21)
    finally:
        del CASES["_paper_full"]
    key = "_paper_full_oracle"
    if key not in _FULL:
        sd = {k: v.detach() for k, v in model.state_dict().items() if not k.endswith("attn_mask")}
        torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
        with torch.no_grad():
            logits = OM.forward(sd, types.SimpleNamespace(**cfg), types.SimpleNamespace(**spec), x)
            loss = float(OM.seg_loss(logits, y))
        _FULL.clear()
        _FULL[key] = (logits, loss)
    ref_logits, ref_loss = _FULL[key]
    model = model.to(DEV).eval()
    model.compute_dtype = dtype
    for grad_mode in (True, False):
        with torch.set_grad_enabled(grad_mode):
            logits = model(x.to(DEV))
            loss = float(seg_loss(logits, y.to(DEV)))
        tag = f"paper_T_ring_cos_v2_nside256_bp8_FULL[{'bf16' if dtype == torch.bfloat16 else 'fp32'}{'' if grad_mode else ', no_grad'}]"
        e = errors(logits, ref_logits)
        conftest.NOTES.append(f"{tag}: logits max|a-b|/max|b| {e['scale_err']:.2e} (scale {e['scale']:.2f}), rms {e['rms_err']:.2e}; "
                              f"loss {loss:.6f} vs oracle {ref_loss:.6f}")
        assert_close(logits, ref_logits, LOGIT_TOL[dtype], tag + " logits")
        assert abs(loss - ref_loss) <= (1e-4 if dtype == torch.float32 else 2e-3) * max(1.0, abs(ref_loss))


def test_paper_config_full_size_gradients_vs_oracle():
    """The paper's model at its full size (HEAL-SWIN-T, nside 256, ring_shift 4, cosine attention, v2 norm placement), the BACKWARD
    against the oracle's autograd: every parameter gradient of one image's CE loss, fp32 and bf16 kernels, each tensor on its own
    scale.  (d logit_scale is a cancelling sum over all scores of a head: bounded separately, as in tests/test_gpu_model.py.)"""
    import psutil
    from heal_swin_amd.losses import seg_loss
    from oracle import model as OM
    if psutil.virtual_memory().available < 64 * 2 ** 30:
        pytest.skip("the oracle's full-size autograd graph of the paper config needs ~40 GB of host memory")
    CASES["_paper_full"] = (T_CFG, 256, 8, 1, dict(shift_strategy="ring_shift", shift_size=4, use_cos_attn=True, use_v2_norm_placement=True))
    try:
        model, cfg, spec, x, y = _setup_seeded("_paper_full", 21)
    finally:
        del CASES["_paper_full"]
    names = [n for n, _ in model.named_parameters()]
    sd = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.endswith("attn_mask")}
    for n in names:
        sd[n].requires_grad_(True)
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    OM.seg_loss(OM.forward(sd, types.SimpleNamespace(**cfg), types.SimpleNamespace(**spec), x), y).backward()
    ref = {n: sd[n].grad for n in names}
    model = model.to(DEV).train()
    xg, yg = x.to(DEV), y.to(DEV)
    report = []
    # three families: the two that are cancelling sums over every score of a head (d logit_scale; the relative-position table, whose
    # entries at the deep stages sum a few dozen windows of bf16-derived dS) are bounded on their own, as in tests/test_gpu_model.py
    for dtype, tol, tol_table, tol_scale_param in ((torch.float32, 3e-4, 5e-4, 5e-4), (torch.bfloat16, 8e-2, 0.2, 0.2)):  # observed 4.1e-5 / 1.4e-4 / 8.3e-5 and 5.2e-2 / 1.1e-1 / 7.0e-2
        model.compute_dtype = dtype
        model.zero_grad(set_to_none=True)
        seg_loss(model(xg), yg).backward()
        worst, worst_tb, worst_ls, rms_all = ("", 0.0), ("", 0.0), ("", 0.0), []
        for n, p_ in model.named_parameters():
            assert p_.grad is not None, n
            e = errors(p_.grad.float().cpu(), ref[n])
            assert_unbiased(p_.grad.float().cpu(), ref[n], f"paper_T_FULL {'fp32' if dtype == torch.float32 else 'bf16'} grad {n}")
            rms_all.append(e["rms_err"])
            if n.endswith("logit_scale"):
                if e["scale_err"] > worst_ls[1]:
                    worst_ls = (n, e["scale_err"])
            elif n.endswith("relative_position_bias_table"):
                if e["scale_err"] > worst_tb[1]:
                    worst_tb = (n, e["scale_err"])
            elif e["scale_err"] > worst[1]:
                worst = (n, e["scale_err"])
        tag = "fp32" if dtype == torch.float32 else "bf16"
        report.append(f"{tag}: worst max|a-b|/max|b| {worst[1]:.2e} ({worst[0]}), rel-pos tables {worst_tb[1]:.2e} ({worst_tb[0]}), "
                      f"d logit_scale {worst_ls[1]:.2e} ({worst_ls[0]}), median rms {sorted(rms_all)[len(rms_all) // 2]:.2e}")
        assert worst[1] <= tol, (tag, worst)
        assert worst_tb[1] <= tol_table, (tag, worst_tb)
        assert worst_ls[1] <= tol_scale_param, (tag, worst_ls)
    conftest.NOTES.append(f"paper_T_ring_cos_v2_nside256_bp8_FULL parameter gradients vs the ORACLE's autograd ({len(names)} tensors): " + "; ".join(report))


# ----------------------------------------------------------------------------- configs[2] at its FULL size: the BACKWARD
def _param_groups(model):
    """Parameter names grouped the way the network is staged: one finite-difference direction per group, so that an error in a
    layer with small gradients is not hidden behind the layers with large ones."""
    groups = {}
    for n, _ in model.named_parameters():
        parts = n.split(".")
        if parts[0] == "layers":
            key = f"enc{parts[1]}"
        elif parts[0] == "decoder" and parts[1] == "layers_up":
            key = f"up{parts[2]}"
        elif parts[0] == "decoder":
            key = "dec_rest"
        else:
            key = parts[0]
        groups.setdefault(key, []).append(n)
    return groups


FULL_BWD_CASES = {
    # name: (arch, nside, base_pix, overrides, f_out, seed, compare bf16 gradients?, bf16 gradient bound)
    "configs2_B_nside256_bp12": (B_CFG, 256, 12, dict(shift_strategy="nest_roll", shift_size=32), 12, 12, True, GRAD_TOL[torch.bfloat16]),
    # the paper's run config (BASELINE configs[3]'s ring-shift permutation at nside 256): cosine attention with its normalisation
    # Jacobian and d logit_scale, v2 norm placement, ring-shift tables and int64-mask semantics of all four stages
    "paper_T_ring_cos_v2_nside256_bp8": (T_CFG, 256, 8, dict(shift_strategy="ring_shift", shift_size=4, use_cos_attn=True,
                                                              use_v2_norm_placement=True), 12, 21, True, 0.15),
    # BASELINE configs[4]: depth head (f_out = 1), masked L1 over the finite targets, fp32 only
    "configs4_depth_T_nside256_bp8_fp32": (T_CFG, 256, 8, dict(shift_strategy="nest_roll", shift_size=32), 1, 31, False, None),
}


@pytest.fixture
def strict_fp32_gemm():
    """Exact-fp32 GEMMs for the finite-difference test: central differences divide the forward's error by the step, so the 4e-6
    of the default bf16x3 products (ops.FP32_GEMM) would drown the derivative; bf16x3 itself is held to the oracle by every other
    fp32 test of the suite."""
    from heal_swin_amd import ops
    prev, ops.FP32_GEMM = ops.FP32_GEMM, "strict"
    yield
    ops.FP32_GEMM = prev


@pytest.mark.parametrize("name", list(FULL_BWD_CASES))
def test_full_size_backward_is_the_derivative_of_the_forward(name, strict_fp32_gemm):
    """The oracle's autograd graph of a model at nside 256 needs ~0.1 TB of host memory (test_headline_config_full_size_gradients_vs_oracle
    compares with it directly where the host has that); independently of the host, the full-size backward is pinned by the
    size-independent property that defines it: for a direction d in parameter space,
    <grad L, d> equals the central difference (L(w + e d) - L(w - e d)) / 2e of the FORWARD -- which the tests above (and
    tests/test_gpu_model.py for the depth head) pin to the oracle at this size.  fp32 kernels (deterministic, 7e-7 forward
    accuracy), one direction per stage of the network (each direction = that group's own gradient with random element weights:
    a strong signal that still weighs every element differently), Richardson-extrapolated over halving steps.  Then the bf16
    training kernels' gradients are compared with the fp32 ones tensor by tensor at the bf16 gradient bound of the oracle tests."""
    from heal_swin_amd import losses as L
    arch, nside, bp, over, f_out, seed, with_bf16, bf16_tol = FULL_BWD_CASES[name]
    CASES["_full"] = (arch, nside, bp, 1, over)
    try:
        model, cfg, spec, x, y = _setup_seeded("_full", seed)
    finally:
        del CASES["_full"]
    if f_out != 12:  # depth head: rebuild with f_out = 1, positive targets with ~4 % infinite (background) pixels
        from heal_swin_amd.data_spec import DataSpec
        from heal_swin_amd.models_torch import swin_hp_transformer as M
        spec = dict(spec, f_out=f_out)
        torch.manual_seed(seed)
        model = M.SwinHPTransformerSys(M.SwinHPTransformerConfig(**cfg), DataSpec(**spec))
        with torch.no_grad():
            for n, p in model.named_parameters():
                if n.endswith("relative_position_bias_table"):
                    p.normal_(0, 0.02)
        g = torch.Generator().manual_seed(seed)
        y = torch.randn(1, spec["dim_in"], generator=g).abs() * 10
        y[torch.rand(1, spec["dim_in"], generator=g) < 0.04] = float("inf")
        loss_fn = L.depth_l1_loss
    else:
        loss_fn = L.seg_loss
    model = model.to(DEV).train()  # (all drop rates are 0: train() only selects the training kernels)
    xg, yg = x.to(DEV), y.to(DEV)

    def loss_and_grads(dtype):
        model.compute_dtype = dtype
        model.zero_grad(set_to_none=True)
        loss = loss_fn(model(xg), yg)
        loss.backward()
        return float(loss.detach()), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    def loss_only():
        with torch.enable_grad():  # the training kernels, as in the differentiated pass (no_grad selects the fused no-grad path)
            return float(loss_fn(model(xg), yg).detach().double())

    loss32, g32 = loss_and_grads(torch.float32)
    params = dict(model.named_parameters())
    assert set(g32) == set(params), sorted(set(params) - set(g32))
    gen = torch.Generator(device=DEV).manual_seed(3)
    model.compute_dtype = torch.float32
    worst = ("", 0.0)
    report, details = [], []
    # (the masked L1 loss of the depth head has a kink per pixel: a few predictions change sign inside a step; observed <= 9.7e-3)
    RHO0, MIN_DLOSS, FD_TOL = 8e-3, 8e-5, (1e-2 if f_out == 12 else 2e-2)
    for key, names in _param_groups(model).items():
        d = {n: g32[n] * (1.0 + 0.5 * torch.randn(g32[n].shape, generator=gen, device=DEV)) for n in names}
        slope = float(sum((g32[n].double() * d[n].double()).sum() for n in names))
        if slope <= 0:
            continue
        fds = []
        orig = {n: params[n].detach().clone() for n in names}
        pnorm = float(sum(orig[n].double().pow(2).sum() for n in names)) ** 0.5
        dnorm = float(sum(d[n].double().pow(2).sum() for n in names)) ** 0.5
        # steps: from 8e-3 |w_group| / |d| halving while the first-order change of the loss stays >= 8e-5 (the fp32 forward
        # reproduces the loss to a few 1e-8 under such perturbations: observed); the truncation error of a central difference is
        # O(eps^2) (third derivative) and large along a gradient direction of the deep stages, so the estimate is the Richardson
        # combination (4 f(e/2) - f(e)) / 3 of the two smallest steps
        eps = RHO0 * pnorm / dnorm
        while len(fds) < 2 or (eps * slope >= MIN_DLOSS and len(fds) < 14):
            vals = []
            for sgn in (+1.0, -1.0):
                with torch.no_grad():
                    for n in names:
                        params[n].copy_(orig[n]).add_(d[n], alpha=sgn * eps)
                vals.append(loss_only())
            fds.append((vals[0] - vals[1]) / (2 * eps))
            eps *= 0.5
        est = (4 * fds[-1] - fds[-2]) / 3
        rel = [abs(est - slope) / slope, abs(fds[-1] - slope) / slope]
        with torch.no_grad():
            for n in names:
                params[n].copy_(orig[n])
        report.append(f"{key}={rel[0]:.1e}")
        details.append(f"{key}: slope {slope:.3e} |w| {pnorm:.3e} |d| {dnorm:.3e} steps {len(fds)} last eps {2 * eps:.3e} "
                       f"rel err Richardson {rel[0]:.1e} plain {rel[1]:.1e} all {[f'{abs(f - slope) / slope:.1e}' for f in fds]}")
        if rel[0] > worst[1]:
            worst = (key, rel[0])
    import os
    if os.path.isdir("gpurun_out"):
        with open(f"gpurun_out/fullsize_bwd_fd_details_{name}.txt", "w") as fh:
            fh.write("\n".join(details) + "\n")
    # observed on MI355X: see the summary line; a wrong or missing term in any layer's backward shows as O(1)
    assert worst[1] <= FD_TOL, (worst, details)
    conftest.NOTES.append(f"{name}_FULL backward: |<grad, d> - central difference| / <grad, d> per group "
                          f"(fp32 kernels, loss {loss32:.6f}): {' '.join(report)}; worst {worst[0]} {worst[1]:.1e}")

    if not with_bf16:
        return
    loss16, g16 = loss_and_grads(torch.bfloat16)
    assert abs(loss16 - loss32) <= 2e-3 * max(1.0, abs(loss32)), (loss16, loss32)
    rms, per, ls = [], [], []
    for n, g in g32.items():
        e = errors(g16[n], g)
        rms.append(e["rms_err"])
        per.append((e["scale_err"], n))
        if n.endswith("logit_scale"):  # one noisy scalar per head: judged as a direction over all heads, as in the oracle test
            ls.append((g16[n].float().reshape(-1), g.reshape(-1)))
            continue
        floor = 1e-6 * float(g32[n.replace(".bias", ".weight")].abs().max()) if n.endswith(".bias") else 0.0
        # relative-position bias tables: each entry sums dS over every window of the image in bf16-rounded terms of both signs
        # (observed up to 4.9e-2 of the table's scale on the headline model; every other tensor <= 3e-2)
        tol = max(bf16_tol, 8e-2) if n.endswith("relative_position_bias_table") else bf16_tol
        assert_close(g16[n], g, tol, f"{name} full-size bf16 grad vs fp32 grad {n}", floor=floor + 1e-7)
        assert_unbiased(g16[n], g, f"{name} full-size bf16 grad vs fp32 grad {n}")
    if ls:
        a16 = torch.cat([t[0] for t in ls]).double()
        b32 = torch.cat([t[1] for t in ls]).double()
        cos = float((a16 * b32).sum() / (a16.norm() * b32.norm()).clamp_min(1e-300))
        conftest.NOTES.append(f"{name}_FULL backward: d logit_scale over {a16.numel()} heads, bf16 vs fp32 kernels: cosine {cos:.4f}, "
                              f"||a-b||/||b|| {float((a16 - b32).norm() / b32.norm().clamp_min(1e-300)):.3f}")
        assert cos >= 0.99, cos
    per.sort(reverse=True)
    conftest.NOTES.append(f"{name}_FULL backward: bf16 vs fp32 kernels, {len(g32)} parameter gradients, median rms "
                          f"{sorted(rms)[len(rms) // 2]:.2e}; largest max|a-b|/max|b|: " + ", ".join(f"{n} {v:.2e}" for v, n in per[:6]))
