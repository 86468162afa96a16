"""Helpers to read the committed golden fixtures (tests/golden/*.npz, made by make_golden.py)."""
import os
import types

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}


def load(name):
    if name not in _cache:
        _cache[name] = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return _cache[name]


def case(npz, key):
    """All arrays below `key/` as a nested-free dict: {'x':..., 'sd': {...}, 'grad': {...}}."""
    z = load(npz)
    out = {"sd": {}, "grad": {}, "sd_dtype": {}}
    pre = key + "/"
    for k in z.files:
        if not k.startswith(pre):
            continue
        rest = k[len(pre):]
        for grp in ("sd_dtype", "sd", "grad"):
            if rest.startswith(grp + "/"):
                out[grp][rest[len(grp) + 1:]] = z[k]
                break
        else:
            out[rest] = z[k]
    return out


def state_dict(c, restore_dtypes=True):
    """Golden state dict as torch tensors; compactly stored integer buffers restored to reference dtypes."""
    sd = {}
    for k, a in c["sd"].items():
        t = torch.from_numpy(np.array(a))
        if restore_dtypes:
            if k.endswith("attn_mask"):
                t = t.to(torch.int64 if str(c["sd_dtype"][k]) == "int64" else torch.float32)
            elif k.endswith("relative_position_index"):
                t = t.to(torch.int64)
        sd[k] = t
    return sd


# Config / DataSpec of the golden whole-model cases (mirrors MODEL_CASES in make_golden.py)
MODEL_CASES = {
    "bp4_roll_v1": (4, 16, dict(shift_strategy="nest_roll", shift_size=8)),
    "bp8_roll_v1_nobias": (8, 16, dict(shift_strategy="nest_roll", shift_size=8, rel_pos_bias=None, qkv_bias=False)),
    "bp12_roll_v2cos": (12, 16, dict(shift_strategy="nest_roll", shift_size=8, use_cos_attn=True, use_v2_norm_placement=True)),
    "bp8_grid_v1": (8, 16, dict(shift_strategy="nest_grid_shift", shift_size=8)),
    "bp8_ring_v2cos": (8, 16, dict(shift_strategy="ring_shift", shift_size=4, use_cos_attn=True, use_v2_norm_placement=True)),
    "bp8_ring_v1_ape_depth": (8, 16, dict(shift_strategy="ring_shift", shift_size=4, ape=True, _f_out=1)),
    "ref_test_config": (8, 32, dict(window_size=4, shift_size=2, embed_dim=2, depths=[2, 1], num_heads=[1, 1], rel_pos_bias=None,
                                    shift_strategy="nest_roll", _f_out=3)),
}


def model_cfg_spec(name):
    bp, nside, kw = MODEL_CASES[name]
    kw = dict(kw)
    f_out = kw.pop("_f_out", 5)
    cfg = dict(patch_size=4, window_size=16, shift_size=8, shift_strategy="nest_roll", rel_pos_bias="flat", embed_dim=16,
               depths=[2, 2], num_heads=[2, 4], mlp_ratio=4.0, qkv_bias=True, qk_scale=None, use_cos_attn=False,
               drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, use_v2_norm_placement=False, ape=False)
    cfg.update(kw)
    spec = dict(dim_in=bp * nside * nside, f_in=3, f_out=f_out, base_pix=bp, class_names=[])
    return cfg, spec


# The reference-scale set (refinit.npz, make_golden.py:make_refinit): production kernel shapes, the reference's own init scale
REFINIT_MODEL_CASES = {
    "bp12_roll_v1_scaled": (12, 16, dict(shift_strategy="nest_roll", shift_size=32, use_cos_attn=False, use_v2_norm_placement=False)),
    "bp8_ring_v2_cos": (8, 16, dict(shift_strategy="ring_shift", shift_size=4, use_cos_attn=True, use_v2_norm_placement=True)),
}
REFINIT_WA_CASES = ["scaled_nomask", "scaled_rollmask", "cos_nomask", "cos_ringmask"]
REFINIT_BLOCK_CASES = [(False, "roll", "nest_roll", 32), (True, "ring", "ring_shift", 4)]


def refinit_cfg_spec(name):
    bp, nside, kw = REFINIT_MODEL_CASES[name]
    cfg = dict(patch_size=4, window_size=64, shift_size=32, shift_strategy="nest_roll", rel_pos_bias="flat", embed_dim=32,
               depths=[2, 2], num_heads=[1, 2], mlp_ratio=4.0, qkv_bias=True, qk_scale=None, use_cos_attn=False,
               drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, use_v2_norm_placement=False, ape=False)
    cfg.update(kw)
    spec = dict(dim_in=bp * nside * nside, f_in=3, f_out=12, base_pix=bp, class_names=[])
    return cfg, spec


def ns(d):
    return types.SimpleNamespace(**d)
