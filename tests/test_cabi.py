"""The C-ABI library loads and exports every symbol include/healswin.h declares; host-side entry points (no GPU needed)
match the golden vectors captured from the reference bit-exactly."""
import ctypes
import hashlib
import os
import re

import numpy as np
import pytest

from _golden import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as g
    g.build()
    from heal_swin_amd import _lib
    return _lib


def header_symbols():
    src = open(os.path.join(ROOT, "include", "healswin.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hs_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(L):
    lib = ctypes.CDLL(L.LIB_PATH)
    declared = header_symbols()
    assert len(declared) >= 20
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    # and the Python binding covers exactly the declared set
    assert sorted(L.EXPORTED_SYMBOLS) == declared


def test_version_and_status_strings(L):
    assert "gfx950" in L.version()
    assert L.lib.hs_status_string(0) == b"ok"
    assert L.device_count() >= 0


def test_host_tables_match_golden(L):
    z = load("tables")
    for ws in (4, 16, 64, 256):
        assert np.array_equal(L.nest_win_idcs(ws), z[f"nest_win_idcs/{ws}"])
    for ws in (4, 16, 64):
        assert np.array_equal(L.rel_pos_index(ws), z[f"rel_pos_index/{ws}"].astype(np.int64))
    for (n, ws, s) in ((64, 16, 8), (1024, 16, 8), (2048, 64, 32), (512, 4, 2)):
        idx, inv, lab = L.build_nest_roll_shift(n, ws, s)
        assert np.array_equal(idx, z[f"nest_roll/{n}_{ws}_{s}/idx"]) and np.array_equal(inv, z[f"nest_roll/{n}_{ws}_{s}/inv"])
        assert sha(L.attn_mask_from_labels(lab, ws)) == str(z[f"nest_roll/{n}_{ws}_{s}/mask_sha"])
    for ns in (4, 8, 16, 32, 64, 128):
        for ws in (16, 64):
            if ws > ns * ns:
                continue
            idx, inv, lab = L.build_nest_grid_shift(ns, 8, ws)
            h = [str(x) for x in z[f"nest_grid/{ns}_{ws}/sha"]]
            assert [sha(idx.astype(np.int64)), sha(inv.astype(np.int64)), sha(lab.astype(np.int64))] == h[:3]
            assert sha(L.attn_mask_from_labels(lab, ws)) == h[3]
            for s in (4, ws // 2):
                idx, inv, lab = L.build_ring_shift(ns, 8, ws, s)
                h = [str(x) for x in z[f"ring/{ns}_{ws}_{s}/sha"]]
                assert [sha(idx.astype(np.int64)), sha(inv.astype(np.int64)), sha(lab.astype(np.int64))] == h[:3]
                assert sha(L.attn_mask_from_labels(lab, ws).astype(np.int64)) == h[3]


def test_healpix_known_answers_via_cabi(L):
    assert L.nest2ring(2, np.arange(10)).tolist() == [13, 5, 4, 0, 15, 7, 6, 1, 17, 9]
    assert L.ring2nest(2, np.arange(10)).tolist() == [3, 7, 11, 15, 2, 1, 6, 5, 10, 9]
    assert int(L.nest2ring(16, [1130])[0]) == 1504 and int(L.ring2nest(16, [1504])[0]) == 1130
    for ns in (1, 4, 64, 256):
        a = np.arange(12 * ns * ns)
        assert np.array_equal(L.ring2nest(ns, L.nest2ring(ns, a)), a)


def test_error_conventions(L):
    with pytest.raises(AssertionError):  # reference: assert base_pix == 8 (hp_shifting.py:78)
        L.build_nest_grid_shift(8, 12, 16)
    with pytest.raises(AssertionError):
        L.build_ring_shift(8, 4, 16, 4)
    with pytest.raises(AssertionError):
        L.nest2ring(3, [0])
    with pytest.raises(AssertionError):
        L.nest_win_idcs(8)  # not 4^k
    # device entry points fail loudly (no silent CPU fallback): no device in this container, or bad arguments
    st = L.lib.hs_layernorm_fwd(None, None, None, None, None, None, None, 4, 8, 0, None)
    assert st != 0 and L.lib.hs_last_error()


def test_model_mirrors_reference_surface(L):
    import torch
    from heal_swin_amd.data_spec import DataSpec
    from heal_swin_amd.models_torch import hp_shifting, hp_windowing
    from heal_swin_amd.models_torch.swin_hp_transformer import SwinHPTransformerConfig, SwinHPTransformerSys
    from _golden import MODEL_CASES, case, model_cfg_spec, state_dict

    for name in MODEL_CASES:
        cfg, spec = model_cfg_spec(name)
        m = SwinHPTransformerSys(SwinHPTransformerConfig(**cfg), DataSpec(**spec))
        ref = state_dict(case("models", "model/" + name))
        mine = m.state_dict()
        assert set(mine) == set(ref)
        for k, v in ref.items():
            assert mine[k].shape == v.shape and mine[k].dtype == v.dtype, k
            if k.endswith(("attn_mask", "relative_position_index")):
                assert torch.equal(mine[k], v), k
        m.load_state_dict(ref, strict=True)
    assert m.no_weight_decay() == {"absolute_pos_embed"} and m.no_weight_decay_keywords() == {"relative_position_bias_table"}
    c = SwinHPTransformerConfig()
    assert (c.patch_size, c.window_size, c.shift_size, c.shift_strategy, c.embed_dim, c.drop_path_rate) == (4, 4, 2, "nest_roll", 96, 0.1)
    # views and tables of hp_windowing
    x = torch.arange(2 * 32 * 3).reshape(2, 32, 3)
    w = hp_windowing.window_partition(x, 16)
    assert w.shape == (4, 16, 3) and torch.equal(hp_windowing.window_reverse(w, 16, 32), x)
    assert hp_windowing.get_nest_win_idcs(16).tolist() == [[5, 4, 1, 0], [7, 6, 3, 2], [13, 12, 9, 8], [15, 14, 11, 10]]
    with pytest.raises(AssertionError):
        hp_windowing.window_partition(x, 12)
    assert hp_shifting.NoShift().get_mask() is None
    # the product refuses CPU tensors instead of falling back
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, spec["dim_in"]))
