"""The oracle's integer tables vs the golden vectors captured from the reference (bit-exact), plus the
healpy known-answers that pin the ring<->nest restatement."""
import hashlib

import numpy as np
import pytest

from oracle import healpix as H
from oracle import tables as T
from _golden import load


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_healpix_known_answers():
    # healpy docstring examples (healpy.pixelfunc.nest2ring / ring2nest)
    assert H.nest2ring(2, np.arange(10)).tolist() == [13, 5, 4, 0, 15, 7, 6, 1, 17, 9]
    assert H.ring2nest(2, np.arange(10)).tolist() == [3, 7, 11, 15, 2, 1, 6, 5, 10, 9]
    assert int(H.nest2ring(16, [1130])[0]) == 1504
    assert int(H.ring2nest(16, [1504])[0]) == 1130
    assert [int(H.nest2ring(n, [11])[0]) for n in (1, 2, 4, 8)] == [11, 2, 12, 211]
    assert [int(H.ring2nest(n, [11])[0]) for n in (1, 2, 4, 8)] == [11, 13, 61, 253]


@pytest.mark.parametrize("nside", [1, 2, 4, 8, 16, 32, 64, 128, 256])
def test_healpix_bijection_and_inverse(nside):
    a = np.arange(12 * nside * nside)
    r = H.nest2ring(nside, a)
    assert np.array_equal(np.sort(r), a)
    assert np.array_equal(H.ring2nest(nside, r), a)
    # rings are contiguous in ring order and pixel 0 (nest) of face 0 sits on the equator-side corner
    ix, iy, f = H.ring2xyf(nside, a)
    assert f.min() == 0 and f.max() == 11 and ix.min() == 0 and ix.max() == nside - 1


def test_healpix_rejects_bad_input():
    with pytest.raises(ValueError):
        H.nest2ring(3, [0])
    with pytest.raises(ValueError):
        H.nest2ring(2, [48])


@pytest.mark.parametrize("ws", [4, 16, 64, 256])
def test_nest_win_idcs(ws):
    assert np.array_equal(T.nest_win_idcs(ws), load("tables")[f"nest_win_idcs/{ws}"])


def test_nest_win_idcs_survey_value():
    assert T.nest_win_idcs(16).tolist() == [[5, 4, 1, 0], [7, 6, 3, 2], [13, 12, 9, 8], [15, 14, 11, 10]]
    assert sha(T.nest_win_idcs(64))[:16] == "394ab6c6e4ca9864"


@pytest.mark.parametrize("ws", [4, 16, 64])
def test_rel_pos_index(ws):
    assert np.array_equal(T.rel_pos_index(ws), load("tables")[f"rel_pos_index/{ws}"].astype(np.int64))


def test_rel_pos_index_survey_value():
    assert T.rel_pos_index(4).tolist() == [[4, 5, 1, 2], [3, 4, 0, 1], [7, 8, 4, 5], [6, 7, 3, 4]]
    assert sha(T.rel_pos_index(64))[:16] == "04ed0e5a043ec346"


@pytest.mark.parametrize("n,ws,s", [(64, 16, 8), (1024, 16, 8), (2048, 64, 32), (512, 4, 2)])
def test_nest_roll(n, ws, s):
    z = load("tables")
    idx, inv, lab = T.nest_roll_shift(n, ws, s)
    assert np.array_equal(idx, z[f"nest_roll/{n}_{ws}_{s}/idx"])
    assert np.array_equal(inv, z[f"nest_roll/{n}_{ws}_{s}/inv"])
    m = T.attn_mask_from_labels(lab, ws).astype(np.float32)
    assert sha(m) == str(z[f"nest_roll/{n}_{ws}_{s}/mask_sha"])
    assert np.array_equal(np.flatnonzero(m.reshape(m.shape[0], -1).any(1)), z[f"nest_roll/{n}_{ws}_{s}/mask_nonzero_windows"])
    # only the last window is masked
    assert np.flatnonzero(m.reshape(m.shape[0], -1).any(1)).tolist() == [n // ws - 1]


GRID = [(ns, ws) for ns in (4, 8, 16, 32, 64, 128) for ws in (16, 64) if ws <= ns * ns]


@pytest.mark.parametrize("ns,ws", GRID)
def test_nest_grid_shift(ns, ws):
    z = load("tables")
    idx, inv, lab = T.nest_grid_shift(ns, 8, ws)
    key = f"nest_grid/{ns}_{ws}"
    if ns <= 32:
        assert np.array_equal(idx, z[key + "/idx"])
        assert np.array_equal(inv, z[key + "/inv"])
        assert np.array_equal(lab, z[key + "/labels"])
    h = [str(x) for x in z[key + "/sha"]]
    assert [sha(idx), sha(inv), sha(lab), sha(T.attn_mask_from_labels(lab, ws).astype(np.float32))] == h


def test_nest_grid_shift_survey_values():
    idx, inv, _ = T.nest_grid_shift(16, 8, 16)
    assert idx[:16].tolist() == [252, 253, 254, 255, 1448, 1449, 1450, 1451, 1108, 1109, 1110, 1111, 0, 1, 2, 3]
    assert sha(idx)[:16] == "1c8cd3b301fead90" and sha(inv)[:16] == "5cf53a54c2fd96fc"
    with pytest.raises(AssertionError):
        T.nest_grid_shift(16, 12, 16)


@pytest.mark.parametrize("ns,ws", GRID)
@pytest.mark.parametrize("half", [False, True])
def test_ring_shift(ns, ws, half):
    s = ws // 2 if half else 4
    z = load("tables")
    idx, inv, lab = T.ring_shift(ns, 8, ws, s)
    key = f"ring/{ns}_{ws}_{s}"
    if ns <= 32:
        assert np.array_equal(idx, z[key + "/idx"])
        assert np.array_equal(inv, z[key + "/inv"])
        assert np.array_equal(lab, z[key + "/labels"])
    h = [str(x) for x in z[key + "/sha"]]
    assert [sha(idx), sha(inv), sha(lab), sha(T.attn_mask_from_labels(lab, ws).astype(np.int64))] == h


def test_ring_shift_survey_values():
    idx, _, lab = T.ring_shift(16, 8, 16, 4)
    assert idx[:16].tolist() == list(range(1136, 1152))
    assert int((lab != 0).sum()) == 194 and sorted(set(lab.tolist())) == list(range(9))
    assert sha(idx)[:16] == "a767669f2993e66b"
    assert sha(T.ring_shift(128, 8, 64, 4)[0])[:16] == "bb3650d74f768b4b"
    # the reference raises for every base_pix != 8 (golden records which exception types)
    assert [str(x) for x in load("tables")["ring/fail_modes_bp_4_5_9_12"]] == ["ValueError", "IndexError", "KeyError", "KeyError"]
    for bp in (4, 5, 9, 12):
        with pytest.raises(ValueError):
            T.ring_shift(8, bp, 16, 4)
