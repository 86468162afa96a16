#!/usr/bin/env python3
"""Generate the committed golden fixtures by importing the REFERENCE (read-only, /root/reference).

Run in the build container only:   python tests/golden/make_golden.py
The reference cannot travel to the GPU box; what is committed are plain arrays (inputs, expected
outputs, gradients, state dicts as float32/int arrays) -- no reference source, bytecode or pickles.

Import stubs (own code, tests/golden/_stubs): `timm.models.layers.{DropPath,trunc_normal_}` and
`healpy.pixelfunc.{ring2nest,nest2ring}` (the latter = oracle/healpix.py, so RingShift goldens pin
everything EXCEPT healpy's own ring<->nest arithmetic, which stays "parity unpinned").
All drop rates are 0 so that the RNG-dependent parts of timm/torch contribute nothing.
"""
import hashlib
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(HERE, "_stubs"), "/root/reference", ROOT]
warnings.filterwarnings("ignore")

from heal_swin.models_torch import hp_shifting as S  # noqa: E402
from heal_swin.models_torch import hp_windowing as W  # noqa: E402
from heal_swin.models_torch import swin_hp_transformer as M  # noqa: E402
from heal_swin.data.segmentation.data_spec import DataSpec  # noqa: E402
from heal_swin.training import loss_depth_regression as LD  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def npy(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ tables
def make_tables():
    out = {}
    for ws in (4, 16, 64, 256):
        out[f"nest_win_idcs/{ws}"] = npy(W.get_nest_win_idcs(ws))
    for ws in (4, 16, 64):
        wa = M.WindowAttention(8, ws, 2, rel_pos_bias="flat")
        out[f"rel_pos_index/{ws}"] = npy(wa.relative_position_index).astype(np.int16)
    for (n, ws, s) in ((64, 16, 8), (1024, 16, 8), (2048, 64, 32), (512, 4, 2)):
        r = S.NestRollShift(s, n, ws)
        x = torch.arange(n)[None, :, None]
        out[f"nest_roll/{n}_{ws}_{s}/idx"] = npy(r.shift(x)).reshape(-1).astype(np.int32)
        out[f"nest_roll/{n}_{ws}_{s}/inv"] = npy(r.shift_back(x)).reshape(-1).astype(np.int32)
        m = r.get_mask()
        assert m.dtype == torch.float32
        out[f"nest_roll/{n}_{ws}_{s}/mask_sha"] = np.array(sha(npy(m)))
        out[f"nest_roll/{n}_{ws}_{s}/mask_nonzero_windows"] = np.flatnonzero(npy(m).reshape(m.shape[0], -1).any(1)).astype(np.int32)
    hashes = {}
    for ns in (4, 8, 16, 32, 64, 128):
        for ws in (16, 64):
            if ws > ns * ns:
                continue
            g = S.NestGridShift(ns, 8, ws)
            key = f"nest_grid/{ns}_{ws}"
            lab = npy(g.get_mask(False))
            m = g.get_mask()
            assert m.dtype == torch.float32
            if ns <= 32:
                out[key + "/idx"] = npy(g.shift_idcs).astype(np.int32)
                out[key + "/inv"] = npy(g.back_shift_idcs).astype(np.int32)
                out[key + "/labels"] = lab.astype(np.int8)
            hashes[key] = (sha(npy(g.shift_idcs)), sha(npy(g.back_shift_idcs)), sha(lab.astype(np.int64)), sha(npy(m)))
            for s in (4, ws // 2):
                r = S.RingShift(ns, 8, ws, s)
                key = f"ring/{ns}_{ws}_{s}"
                m = r.get_mask()
                assert m.dtype == torch.int64
                if ns <= 32:
                    out[key + "/idx"] = npy(r.shift_idcs).astype(np.int32)
                    out[key + "/inv"] = npy(r.back_shift_idcs).astype(np.int32)
                    out[key + "/labels"] = npy(r.mask).astype(np.int8)
                hashes[key] = (sha(npy(r.shift_idcs)), sha(npy(r.back_shift_idcs)), sha(npy(r.mask)), sha(npy(m)))
    for k, v in hashes.items():
        out[k + "/sha"] = np.array(v)  # sha256 of int64 idx, int64 inv, int64 labels, full [nW,Ws,Ws] mask
    # reference failure modes of RingShift for base_pix != 8
    fails = []
    for bp in (4, 5, 9, 12):
        try:
            S.RingShift(8, bp, 16, 4)
            fails.append("ok")
        except Exception as e:  # noqa: BLE001
            fails.append(type(e).__name__)
    out["ring/fail_modes_bp_4_5_9_12"] = np.array(fails)
    np.savez_compressed(os.path.join(HERE, "tables.npz"), **out)
    print("tables.npz", len(out), "arrays")


# ------------------------------------------------------------------ helpers for float fixtures
def randomize_(module, gen):
    """Give every parameter a non-trivial value (bias table and LN affine included: they init to 0 / 1)."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith("logit_scale"):
                # around log(10), with one head pushed past the log(100) clamp (:144-146)
                p.copy_(torch.log(torch.tensor(10.0)) + 0.5 * torch.randn(p.shape, generator=gen))
                p.view(-1)[0] = 5.0
            elif name.endswith("relative_position_bias_table"):
                p.copy_(0.5 * torch.randn(p.shape, generator=gen))
            elif "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=gen))
            elif name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=gen))
            else:
                p.copy_(torch.randn(p.shape, generator=gen) * (0.5 / max(1.0, float(p.shape[-1])) ** 0.5))


def run_case(module, x, fwd, gen, out, key, with_state=True):
    """forward + backward with a fixed random cotangent; stores x, y, dy, dx, state dict and param grads."""
    x = x.clone().requires_grad_(True)
    y = fwd(x)
    dy = torch.randn(y.shape, generator=gen)
    y.backward(dy)
    out[key + "/x"] = npy(x).astype(np.float32)
    out[key + "/y"] = npy(y)
    out[key + "/dy"] = npy(dy)
    out[key + "/dx"] = npy(x.grad)
    if with_state:
        for k, v in module.state_dict().items():
            a = npy(v)
            if k.endswith("attn_mask"):  # {0,-100}; reference dtype (float32 | int64 for ring) kept in a side entry
                out[key + "/sd_dtype/" + k] = np.array(str(a.dtype))
                a = a.astype(np.int8)
            elif k.endswith("relative_position_index"):
                a = a.astype(np.int16)  # reference dtype int64
            out[key + "/sd/" + k] = a
    for k, p in module.named_parameters():
        out[key + "/grad/" + k] = npy(p.grad) if p.grad is not None else np.zeros(p.shape, np.float32)
        p.grad = None


def make_modules():
    gen = torch.Generator().manual_seed(1234)
    out = {}
    # --- WindowAttention (B_=4 windows = 2 images x 2 windows, Ws=64, C=96, nH=3 -> hd=32) :47-174
    C, nH, Ws, nW, B = 96, 3, 64, 2, 2
    roll_mask = S.NestRollShift(32, nW * Ws, Ws).get_mask()  # float32 [2,64,64]
    ring = S.RingShift(4, 8, Ws, 4)  # nside 4: 128 tokens... use windows 0 and the most-masked one
    ring_full = ring.get_mask()  # int64 [2? ...]
    # pick 2 windows of a bigger ring shift that actually contain mixed labels
    ring16 = S.RingShift(16, 8, Ws, 4).get_mask()
    cnt = (ring16 != 0).reshape(ring16.shape[0], -1).sum(1)
    pick = torch.argsort(cnt, descending=True)[:nW]
    ring_mask = ring16[pick].contiguous()  # int64 [2,64,64]
    assert ring_mask.dtype == torch.int64 and (ring_mask != 0).any()
    del ring_full
    for cos in (False, True):
        for mname, mask in (("nomask", None), ("rollmask", roll_mask), ("ringmask", ring_mask)):
            wa = M.WindowAttention(C, Ws, nH, rel_pos_bias="flat", use_cos_attn=cos)
            randomize_(wa, gen)
            x = torch.randn((B if mask is not None else 1) * nW, Ws, C, generator=gen)
            key = f"window_attention/{'cos' if cos else 'scaled'}_{mname}"
            run_case(wa, x, lambda t: wa(t, mask=mask), gen, out, key)
            if mask is not None:
                out[key + "/mask"] = npy(mask).astype(np.int8)
    # no rel-pos bias, no qkv bias, small odd sizes (Ws=16, hd=8; Ws=4, hd=2 = the reference test config)
    for (C2, nH2, Ws2, tag) in ((16, 2, 16, "w16"), (2, 1, 4, "w4")):
        wa = M.WindowAttention(C2, Ws2, nH2, rel_pos_bias=None, qkv_bias=False, use_cos_attn=False)
        randomize_(wa, gen)
        x = torch.randn(6, Ws2, C2, generator=gen)
        run_case(wa, x, lambda t: wa(t), gen, out, f"window_attention/plain_{tag}")

    # --- PatchMerging / PatchExpand / FinalPatchExpand_X4 :364-452 at (B=2, N=256, C=32)
    pm = M.PatchMerging(32)
    randomize_(pm, gen)
    run_case(pm, torch.randn(2, 256, 32, generator=gen), pm, gen, out, "patch_merging")
    pe = M.PatchExpand(32)
    randomize_(pe, gen)
    run_case(pe, torch.randn(2, 256, 32, generator=gen), pe, gen, out, "patch_expand")
    fe = M.FinalPatchExpand_X4(4, 32)
    randomize_(fe, gen)
    run_case(fe, torch.randn(1, 256, 32, generator=gen), fe, gen, out, "final_patch_expand")

    # --- SwinTransformerBlock x {v1,v2} x {unshifted, nest_roll, ring} :193-340; 8 faces x nside 8 = 512 tokens
    for v2 in (False, True):
        for (sname, strat, shift) in (("noshift", "nest_roll", 0), ("roll", "nest_roll", 8), ("ring", "ring_shift", 4),
                                      ("grid", "nest_grid_shift", 8)):
            blk = M.SwinTransformerBlock(32, 512, 8, 2, window_size=16, shift_size=shift, shift_strategy=strat,
                                         rel_pos_bias="flat", use_v2_norm_placement=v2, use_cos_attn=v2)
            randomize_(blk, gen)
            key = f"block/{'v2' if v2 else 'v1'}_{sname}"
            run_case(blk, torch.randn(2 if (v2 and sname == "ring") else 1, 512, 32, generator=gen), blk, gen, out, key)
    np.savez_compressed(os.path.join(HERE, "modules.npz"), **out)
    print("modules.npz", len(out), "arrays")


MODEL_CASES = {
    # name: (base_pix, nside_in, config kwargs)
    "bp4_roll_v1": (4, 16, dict(shift_strategy="nest_roll", shift_size=8, use_cos_attn=False, use_v2_norm_placement=False)),
    "bp8_roll_v1_nobias": (8, 16, dict(shift_strategy="nest_roll", shift_size=8, rel_pos_bias=None, qkv_bias=False)),
    "bp12_roll_v2cos": (12, 16, dict(shift_strategy="nest_roll", shift_size=8, use_cos_attn=True, use_v2_norm_placement=True)),
    "bp8_grid_v1": (8, 16, dict(shift_strategy="nest_grid_shift", shift_size=8)),
    "bp8_ring_v2cos": (8, 16, dict(shift_strategy="ring_shift", shift_size=4, use_cos_attn=True, use_v2_norm_placement=True)),
    "bp8_ring_v1_ape_depth": (8, 16, dict(shift_strategy="ring_shift", shift_size=4, ape=True, _f_out=1)),
    # the reference's own test config shape (testing/swin_hp_test_run_config.py:24-55): nside 32, Ws 4, embed 2
    "ref_test_config": (8, 32, dict(window_size=4, shift_size=2, embed_dim=2, depths=[2, 1], num_heads=[1, 1], rel_pos_bias=None,
                                    shift_strategy="nest_roll", _f_out=3, _batch=1)),
}


def model_config(kw):
    kw = dict(kw)
    kw.pop("_f_out", None)
    kw.pop("_batch", None)
    base = dict(patch_size=4, window_size=16, shift_size=8, rel_pos_bias="flat", embed_dim=16, depths=[2, 2], num_heads=[2, 4],
                drop_path_rate=0.0)
    base.update(kw)
    return M.SwinHPTransformerConfig(**base)


def make_models():
    gen = torch.Generator().manual_seed(4321)
    out = {}
    for name, (bp, nside, kw) in MODEL_CASES.items():
        f_out = kw.get("_f_out", 5)
        cfg = model_config(kw)
        spec = DataSpec(dim_in=bp * nside * nside, f_in=3, f_out=f_out, base_pix=bp, class_names=[])
        model = M.SwinHPTransformerSys(cfg, spec)
        randomize_(model, gen)
        model.train()  # all drop rates are 0: train == eval numerically, but exercises the training path
        x = torch.randint(0, 256, (kw.get("_batch", 2), 3, spec.dim_in), generator=gen).float()  # raw 0..255 inputs, like the caller
        run_case(model, x, model, gen, out, f"model/{name}")
        print(name, "state entries", len(model.state_dict()), "params", sum(p.numel() for p in model.parameters()))
    np.savez_compressed(os.path.join(HERE, "models.npz"), **out)
    print("models.npz", len(out), "arrays")


def make_losses():
    gen = torch.Generator().manual_seed(99)
    out = {}
    # segmentation CE with class weights: model_lightning_swin_hp.py:39-45,:104-111
    logits = torch.randn(2, 6, 384, generator=gen).mul(3).requires_grad_(True)
    labels = torch.randint(0, 6, (2, 384), generator=gen).to(torch.uint8)
    weights = torch.tensor([0.5, 1.0, 2.0, 1.5, 0.1, 3.0])
    for tag, w in (("weighted", weights), ("uniform", torch.ones(6))):
        loss = torch.nn.CrossEntropyLoss(weight=w)(logits, labels.long())
        (g,) = torch.autograd.grad(loss, logits)
        out[f"seg/{tag}/loss"] = npy(loss)
        out[f"seg/{tag}/dlogits"] = npy(g)
        out[f"seg/{tag}/weights"] = npy(w)
    out["seg/logits"] = npy(logits)
    out["seg/labels"] = npy(labels)
    out["seg/argmax"] = npy(torch.max(logits, 1)[1]).astype(np.int8)
    # depth l1 / l2 with infinite (background) targets: training/loss_depth_regression.py:9-53
    pred = torch.randn(2, 1, 512, generator=gen).requires_grad_(True)
    target = torch.randn(2, 512, generator=gen).abs() * 10
    target[torch.rand(2, 512, generator=gen) < 0.04] = float("inf")
    for tag, fn in (("l1", LD.l1_loss), ("l2", LD.mse)):
        loss = fn(pred, target)
        (g,) = torch.autograd.grad(loss, pred)
        out[f"depth/{tag}/loss"] = npy(loss)
        out[f"depth/{tag}/dpred"] = npy(g)
    out["depth/pred"] = npy(pred)
    out["depth/target"] = npy(target)
    # standardize affine: data/depth_estimation/normalize_depth_data.py:133-158 with MaskedDepthDataStatistics
    from heal_swin.data.depth_estimation import normalize_depth_data as ND

    st = ND.MaskedDepthDataStatistics()
    d = torch.linspace(0.2, 900.0, 17)
    out["depth/standardize/in"] = npy(d)
    out["depth/standardize/out"] = npy(ND.normalize_data(d, st, "standardize"))
    out["depth/standardize/back"] = npy(ND.unnormalize_data(ND.normalize_data(d, st, "standardize"), st, "standardize"))
    # huber (delta 1 and 0.3) on the one-channel prediction and the mean / log-variance loss on a two-channel one:
    # training/loss_depth_regression.py:23-38, :56-83 (drawn AFTER everything above, so the earlier arrays keep their values)
    from heal_swin.models_lightning.depth_estimation.depth_common_config import CommonDepthConfig

    for tag, delta in (("huber_d1", 1), ("huber_d0p3", 0.3)):
        fn = LD.get_depth_loss(CommonDepthConfig(loss="huber", huber_delta=delta))
        loss = fn(pred, target)
        (g,) = torch.autograd.grad(loss, pred)
        out[f"depth/{tag}/loss"] = npy(loss)
        out[f"depth/{tag}/dpred"] = npy(g)
    assert LD.get_depth_loss(CommonDepthConfig(loss="l1")) is LD.l1_loss and LD.get_depth_loss(CommonDepthConfig(loss="l2")) is LD.mse
    pred2 = torch.randn(2, 2, 512, generator=gen).requires_grad_(True)
    fn = LD.get_depth_loss(CommonDepthConfig(loss="l1", use_logvar=True))
    assert fn is LD.mean_log_var_loss
    loss = fn(pred2, target)
    (g,) = torch.autograd.grad(loss, pred2)
    out["depth/logvar/loss"] = npy(loss)
    out["depth/logvar/dpred"] = npy(g)
    out["depth/logvar/pred"] = npy(pred2)
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)
    print("losses.npz", len(out), "arrays")



# ------------------------------------------------------------------ reference-scale initialisation (round 3)
def refinit_(module, gen):
    """The reference's OWN initialisation scale (swin_hp_transformer.py:912-919, :84-87, :92-96): Linear weights trunc-normal
    sigma 0.02 with zero biases, LayerNorm 1 / 0, logit_scale = ln 10, relative-position table N(0, 0.02) (the reference starts
    it at zero, which would leave the bias path without signal).  Conv1d layers keep the torch default init, as in the
    reference.  Drawn from `gen` so that the fixture is reproducible."""
    import math

    with torch.no_grad():
        for m in module.modules():
            if isinstance(m, torch.nn.Linear):
                w = torch.randn(m.weight.shape, generator=gen).clamp_(-2.0, 2.0) * 0.02  # trunc-normal(0.02), cut at 2 sigma
                m.weight.copy_(w)
                if m.bias is not None:
                    m.bias.zero_()
            elif isinstance(m, torch.nn.LayerNorm):
                m.weight.fill_(1.0)
                m.bias.zero_()
            elif isinstance(m, torch.nn.Conv1d):
                bound = 1.0 / math.sqrt(m.weight.shape[1] * m.weight.shape[2])
                m.weight.copy_((torch.rand(m.weight.shape, generator=gen) * 2 - 1) * bound)
                if m.bias is not None:
                    m.bias.copy_((torch.rand(m.bias.shape, generator=gen) * 2 - 1) * bound)
        for name, p in module.named_parameters():
            if name.endswith("logit_scale"):
                p.fill_(math.log(10.0))
            elif name.endswith("relative_position_bias_table"):
                p.copy_(0.02 * torch.randn(p.shape, generator=gen))
            elif name.endswith("absolute_pos_embed"):
                p.copy_(0.02 * torch.randn(p.shape, generator=gen).clamp_(-2.0, 2.0))


REFINIT_MODEL_CASES = {
    # production kernel shapes (window 64, head_dim 32) at the reference's own weight scale
    "bp12_roll_v1_scaled": (12, 16, dict(shift_strategy="nest_roll", shift_size=32, use_cos_attn=False, use_v2_norm_placement=False)),
    "bp8_ring_v2_cos": (8, 16, dict(shift_strategy="ring_shift", shift_size=4, use_cos_attn=True, use_v2_norm_placement=True)),
}


def refinit_model_config(kw):
    base = dict(patch_size=4, window_size=64, shift_size=32, rel_pos_bias="flat", embed_dim=32, depths=[2, 2], num_heads=[1, 2],
                drop_path_rate=0.0)
    base.update(kw)
    return M.SwinHPTransformerConfig(**base)


def make_refinit():
    """Second golden set: the same reference modules at the reference's OWN initialisation scale, on
    the production kernel shapes (window 64, head_dim 32), so that north_star's 1e-3 (fp32) / 1e-2 (bf16) can be asserted
    directly on reference-produced tensors without a stress multiplier."""
    gen = torch.Generator().manual_seed(20260930)
    out = {}
    C, nH, Ws, nW, B = 128, 4, 64, 2, 2
    roll_mask = S.NestRollShift(32, nW * Ws, Ws).get_mask()
    ring16 = S.RingShift(16, 8, Ws, 4).get_mask()
    cnt = (ring16 != 0).reshape(ring16.shape[0], -1).sum(1)
    ring_mask = ring16[torch.argsort(cnt, descending=True)[:nW]].contiguous()
    for cos in (False, True):
        for mname, mask in (("nomask", None), ("rollmask", roll_mask), ("ringmask", ring_mask)):
            if (cos, mname) in ((False, "ringmask"), (True, "rollmask")):
                continue  # (fixture size: each mask kind once per attention kind is enough)
            wa = M.WindowAttention(C, Ws, nH, rel_pos_bias="flat", use_cos_attn=cos)
            refinit_(wa, gen)
            # LayerNorm-scale activations, as the block feeds the module (norm1 output / block input)
            x = torch.randn((B if mask is not None else 1) * nW, Ws, C, generator=gen)
            key = f"window_attention/{'cos' if cos else 'scaled'}_{mname}"
            run_case(wa, x, lambda t: wa(t, mask=mask), gen, out, key)
            if mask is not None:
                out[key + "/mask"] = npy(mask).astype(np.int8)
    # one v1 (pre-norm, scaled attention, nest_roll) and one v2 (post-norm, cosine attention, ring_shift) block:
    # 8 faces x nside 8 = 512 tokens = 8 windows of 64, C = 64, 2 heads of 32
    for v2, sname, strat, shift in ((False, "roll", "nest_roll", 32), (True, "ring", "ring_shift", 4)):
        blk = M.SwinTransformerBlock(64, 512, 8, 2, window_size=64, shift_size=shift, shift_strategy=strat,
                                     rel_pos_bias="flat", use_v2_norm_placement=v2, use_cos_attn=v2)
        refinit_(blk, gen)
        run_case(blk, torch.randn(1, 512, 64, generator=gen), blk, gen, out, f"block/{'v2' if v2 else 'v1'}_{sname}")
    for name, (bp, nside, kw) in REFINIT_MODEL_CASES.items():
        cfg = refinit_model_config(kw)
        spec = DataSpec(dim_in=bp * nside * nside, f_in=3, f_out=12, base_pix=bp, class_names=[])
        model = M.SwinHPTransformerSys(cfg, spec)
        refinit_(model, gen)
        model.train()
        x = torch.randint(0, 256, (2, 3, spec.dim_in), generator=gen).float()
        run_case(model, x, model, gen, out, f"model/{name}")
        print(name, "state entries", len(model.state_dict()), "params", sum(p.numel() for p in model.parameters()))
    np.savez_compressed(os.path.join(HERE, "refinit.npz"), **out)
    print("refinit.npz", len(out), "arrays")


# ------------------------------------------------------------------ fisheye -> HEALPix projection
PROJ_CALS = {
    # WoodScape-like calibrations (polynomial fisheye model, quaternion scalar last); the small ones scale the optics to small
    # synthetic images so that the fixtures stay small
    "fv_966x1280": dict(name="FV", intrinsic=dict(aspect_ratio=1.0, cx_offset=3.942, cy_offset=-0.472, width=1280.0, height=966.0,
                                                  poly_order=4, k1=339.749, k2=-31.988, k3=48.275, k4=-7.201),
                        extrinsic=dict(quaternion=[0.5946970238045494, -0.5837953694518585, 0.39063952590941586, -0.39195666481783994])),
    "mvl_96x128": dict(name="MVL", intrinsic=dict(aspect_ratio=1.0005, cx_offset=0.731, cy_offset=-0.613, width=128.0, height=96.0,
                                                  poly_order=4, k1=33.97, k2=-3.2, k3=4.83, k4=-0.72),
                       extrinsic=dict(quaternion=[0.1215, -0.8817, 0.4417, 0.1153])),
    "rv_60x80": dict(name="RV", intrinsic=dict(aspect_ratio=0.9991, cx_offset=-1.25, cy_offset=0.75, width=80.0, height=60.0,
                                               poly_order=4, k1=21.2, k2=-2.0, k3=3.0, k4=-0.45),
                     extrinsic=dict(quaternion=[-0.4057, -0.4038, 0.5817, 0.5789])),
}


def _import_projection():
    """project_on_s2 imports the dataset / plotting stack (torchvision, matplotlib paths ...) at module level; none of it is
    used by the four functions exercised here, so those modules are replaced by empty ones for the import."""
    import types

    for name in ("heal_swin.data.segmentation.flat_datasets", "heal_swin.utils.utils", "heal_swin.utils.get_paths",
                 "heal_swin.utils.healpy_utils"):
        sys.modules.setdefault(name, types.ModuleType(name))
    import heal_swin.data.segmentation as DS
    import heal_swin.utils as U

    DS.flat_datasets = sys.modules["heal_swin.data.segmentation.flat_datasets"]
    for n in ("utils", "get_paths", "healpy_utils"):
        setattr(U, n, sys.modules["heal_swin.utils." + n])
    import heal_swin.data.segmentation.project_on_s2 as P

    return P


def make_projection():
    """Outputs of the reference's project_s2_points_to_img / sample_bilinear / sample_mask (project_on_s2.py:38-80, :141-183)
    on synthetic images.  The grid (theta, phi) is an INPUT here (healpy's pix2ang is absent: the grid comes from
    oracle/healpix.py:pix2ang_nest and is stored with the case)."""
    from oracle.healpix import pix2ang_nest

    P = _import_projection()
    rng = np.random.default_rng(20260929)
    out = {}
    for key, nside, base_pix, rotate in (("mvl_96x128", 16, 8, False), ("mvl_96x128", 16, 8, True), ("rv_60x80", 8, 12, True),
                                         ("rv_60x80", 32, 8, False), ("fv_966x1280", 32, 8, True)):
        cal = PROJ_CALS[key]
        H, W = int(cal["intrinsic"]["height"]), int(cal["intrinsic"]["width"])
        theta, phi = pix2ang_nest(nside, np.arange(nside * nside * base_pix))
        tag = f"{key}/n{nside}_bp{base_pix}_{'rot' if rotate else 'plain'}"
        if H <= 128:
            img = rng.integers(0, 256, (3, H, W), dtype=np.uint8)
            img[:, : H // 3, : W // 3] = 200  # a constant patch: the reference's truncation of c (1 - eps) to c - 1 shows here
            mask = rng.integers(0, 10, (H, W), dtype=np.uint8)
            out[tag + "/img"], out[tag + "/mask"] = img, mask
        else:  # the full-size case stores a seed instead of a 3.7 MB image
            r2 = np.random.default_rng(7)
            img = r2.integers(0, 256, (3, H, W), dtype=np.uint8)
            mask = r2.integers(0, 10, (H, W), dtype=np.uint8)
            out[tag + "/img_seed"] = np.array(7)
        u, v = P.project_s2_points_to_img(theta, phi, cal, rotate)
        out[tag + "/theta"], out[tag + "/phi"], out[tag + "/u"], out[tag + "/v"] = theta, phi, u, v
        out[tag + "/hp_img"] = P.sample_bilinear(torch.from_numpy(img), v, u).astype(np.uint8)
        out[tag + "/hp_mask"] = P.sample_mask(torch.from_numpy(mask), v, u, 3)
    # sampling edge cases on hand-made coordinates: integer coordinates (both weights zero), the image border, far outside,
    # exact .5 (round half to even in the mask), NaN
    img = rng.integers(1, 256, (3, 7, 9), dtype=np.uint8)
    mask = rng.integers(0, 10, (7, 9), dtype=np.uint8)
    rx = np.array([0.0, 2.0, 2.5, 3.5, 6.0, 6.2, -0.3, -1.0, 5.999999, 1e9, -1e9, 0.5, 1.5, np.nan, 3.25, 6.5, 2.0])
    ry = np.array([0.0, 3.0, 0.5, 1.5, 8.0, 8.4, 0.4, 2.0, 7.999999, 1.0, 1.0, 8.5, -0.5, 1.0, np.nan, 7.5, 4.75])
    out["edge/img"], out["edge/mask"], out["edge/rx"], out["edge/ry"] = img, mask, rx, ry
    out["edge/bilinear"] = P.sample_bilinear(torch.from_numpy(img), rx, ry)  # float64, before the uint8 truncation
    out["edge/hp_mask"] = P.sample_mask(torch.from_numpy(mask), rx, ry, 5)
    np.savez_compressed(os.path.join(HERE, "projection.npz"), **out)


if __name__ == "__main__":
    torch.set_num_threads(4)
    only = [a for a in sys.argv[1:] if not a.startswith("-")]  # e.g. `make_golden.py losses` regenerates one file
    for name, fn in (("tables", make_tables), ("modules", make_modules), ("models", make_models), ("losses", make_losses),
                     ("projection", make_projection), ("refinit", make_refinit)):
        if not only or name in only:
            fn()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
