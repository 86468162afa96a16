"""Fisheye -> HEALPix projection, CPU part: the oracle restatement against fixtures produced by the reference's
own functions (tests/golden/projection.npz <- heal_swin/data/segmentation/project_on_s2.py via tests/golden/make_golden.py),
pix2ang against healpy's documented values and an independent geometric derivation, and the host-side C++ / numpy of the
product (no GPU work here; the sampling kernels are checked in test_gpu_projection.py)."""
import os
import re

import numpy as np
import pytest

from oracle import healpix as H
from oracle import projection as OP
from tests.test_healpix_geometry import nested_centres

GOLD = os.path.join(os.path.dirname(__file__), "golden", "projection.npz")


def calibrations():
    """The PROJ_CALS dictionary of the generating script (read as data; the script itself imports the reference)."""
    src = open(os.path.join(os.path.dirname(__file__), "golden", "make_golden.py")).read()
    ns = {}
    exec(src[src.index("PROJ_CALS = {"):src.index("def _import_projection")], ns)
    return ns["PROJ_CALS"]


def cases(g):
    return sorted({k.rsplit("/", 1)[0] for k in g.files if k.count("/") == 2})


def images_of(g, tag, cal):
    if tag + "/img" in g.files:
        return g[tag + "/img"], g[tag + "/mask"]
    r = np.random.default_rng(int(g[tag + "/img_seed"]))
    h, w = int(cal["intrinsic"]["height"]), int(cal["intrinsic"]["width"])
    return r.integers(0, 256, (3, h, w), dtype=np.uint8), r.integers(0, 10, (h, w), dtype=np.uint8)


def test_pix2ang_matches_healpy_documented_values():
    # healpy.pixelfunc.pix2ang docstring (ring scheme)
    th, ph = H.pix2ang_ring(16, [1440])
    assert th[0] == pytest.approx(1.5291175943723188, abs=1e-15) and ph[0] == 0.0
    th, ph = H.pix2ang_ring(16, [1440, 427, 1520, 0])
    np.testing.assert_allclose(th, [1.52911759, 0.78550497, 1.57079633, 0.05103658], atol=5e-9)
    np.testing.assert_allclose(ph, [0.0, 0.78539816, 1.61988371, 0.78539816], atol=5e-9)
    for nside, (t, p) in zip((1, 2, 4, 8), ((2.30052398, 5.49778714), (0.84106867, 5.89048623), (0.41113786, 5.89048623), (0.2044802, 5.89048623))):
        th, ph = H.pix2ang_ring(nside, [11])
        assert th[0] == pytest.approx(t, abs=5e-9) and ph[0] == pytest.approx(p, abs=5e-9)


@pytest.mark.parametrize("nside", [1, 2, 4, 16, 64])
def test_pix2ang_matches_independent_pixel_geometry(nside):
    """Centres from the HEALPix projection plane (no ring index, jrll / jpll or kshift on that side)."""
    p = np.arange(12 * nside * nside)
    th, ph = H.pix2ang_nest(nside, p)
    z, phi = nested_centres(nside, p)
    np.testing.assert_allclose(np.cos(th), z, atol=1e-12)
    d = np.abs(np.mod(ph - phi + np.pi, 2 * np.pi) - np.pi)
    assert d.max() < 1e-12


@pytest.mark.parametrize("nside", [1, 2, 8, 32, 128])
def test_library_pix2ang_equals_the_oracle(nside):
    """Same formulas in C++ (glibc acos / atan2, as healpy's C++ uses) and numpy (its own vectorised arccos): phi bit-equal,
    theta within one unit in the last place of the two arccos implementations."""
    from heal_swin_amd import projection as P

    for bp in (8, 12):
        th, ph = P.hp_grid(nside, bp)
        tho, pho = OP.hp_grid(nside, bp)
        assert np.array_equal(ph, pho)
        assert np.abs(th - tho).max() <= 4.5e-16


def test_library_pix2ang_rejects_bad_arguments():
    from heal_swin_amd._lib import lib, np_ptr

    a = np.empty(4)
    assert lib.hs_pix2ang_nest(3, 0, 4, np_ptr(a), np_ptr(a)) != 0
    assert lib.hs_pix2ang_nest(2, 46, 4, np_ptr(a), np_ptr(a)) != 0


def test_oracle_coordinates_match_the_reference():
    g, cals = np.load(GOLD), calibrations()
    for tag in cases(g):
        cal, rot = cals[tag.split("/")[0]], tag.endswith("rot")
        u, v = OP.project_s2_points_to_img(g[tag + "/theta"], g[tag + "/phi"], cal, rot)
        if rot:  # the rotation goes through scipy in the reference: same mathematics, last-bit differences
            np.testing.assert_allclose(u, g[tag + "/u"], atol=1e-10, rtol=0)
            np.testing.assert_allclose(v, g[tag + "/v"], atol=1e-10, rtol=0)
        else:
            assert np.array_equal(u, g[tag + "/u"]) and np.array_equal(v, g[tag + "/v"]), tag


def test_product_host_coordinates_match_the_reference():
    from heal_swin_amd import projection as P

    g, cals = np.load(GOLD), calibrations()
    for tag in cases(g):
        cal, rot = cals[tag.split("/")[0]], tag.endswith("rot")
        u, v = P.project_s2_points_to_img(g[tag + "/theta"], g[tag + "/phi"], cal, rot)
        if rot:
            np.testing.assert_allclose(u, g[tag + "/u"], atol=1e-10, rtol=0)
            np.testing.assert_allclose(v, g[tag + "/v"], atol=1e-10, rtol=0)
        else:
            assert np.array_equal(u, g[tag + "/u"]) and np.array_equal(v, g[tag + "/v"]), tag
        th2, ph2 = P.rot_grid(*P.rot_grid(g[tag + "/theta"], g[tag + "/phi"], cal), cal, inv=True)  # rotation round trip
        np.testing.assert_allclose(np.cos(th2), np.cos(g[tag + "/theta"]), atol=1e-12)


def test_oracle_sampling_is_bit_equal_to_the_reference():
    g, cals = np.load(GOLD), calibrations()
    for tag in cases(g):
        img, mask = images_of(g, tag, cals[tag.split("/")[0]])
        with np.errstate(invalid="ignore"):
            hp_img = OP.sample_bilinear(img, g[tag + "/v"], g[tag + "/u"]).astype(np.uint8)
            hp_mask = OP.sample_mask(mask, g[tag + "/v"], g[tag + "/u"], 3)
        assert np.array_equal(hp_img, g[tag + "/hp_img"]) and np.array_equal(hp_mask, g[tag + "/hp_mask"]), tag
        assert (g[tag + "/hp_img"] > 0).mean() > 0.3  # the camera sees a good part of the grid: the case is not vacuous


def test_oracle_sampling_edge_cases():
    g = np.load(GOLD)
    with np.errstate(invalid="ignore"):
        b = OP.sample_bilinear(g["edge/img"], g["edge/rx"], g["edge/ry"])
        m = OP.sample_mask(g["edge/mask"], g["edge/rx"], g["edge/ry"], 5)
    assert np.array_equal(b, g["edge/bilinear"], equal_nan=True) and np.array_equal(m, g["edge/hp_mask"])
    assert b[0, 0] == 0 and b[0, 1] == 0  # integer coordinates: both weights vanish (reference behaviour, kept)


def test_end_to_end_oracle_projection():
    g, cals = np.load(GOLD), calibrations()
    tag = "rv_60x80/n32_bp8_plain"
    cal = cals["rv_60x80"]
    with np.errstate(invalid="ignore"):
        hp_img, hp_mask = OP.project_to_hp(g[tag + "/img"], g[tag + "/mask"], cal, 32, 8, False, 3)
    assert np.array_equal(hp_img, g[tag + "/hp_img"]) and np.array_equal(hp_mask, g[tag + "/hp_mask"])


def test_sampling_needs_the_gpu():
    import torch
    from heal_swin_amd import projection as P

    with pytest.raises(RuntimeError, match="GPU tensor"):
        P.sample_bilinear_u8(torch.zeros(3, 4, 4, dtype=torch.uint8), np.zeros(2), np.zeros(2))
    with pytest.raises(TypeError, match="uint8"):
        P.sample_mask(torch.zeros(4, 4), np.zeros(2), np.zeros(2))
