"""Fisheye -> HEALPix sampling kernels (`hs_sample_bilinear_u8`, `hs_sample_mask_u8`) through the C ABI and the host mirror
heal_swin_amd/projection.py: bit-exact against the reference's own outputs (tests/golden/projection.npz) and against the
oracle on seeded inputs at the full WoodScape / nside-256 size."""
import os

import numpy as np
import pytest
import torch

from oracle import projection as OP
from tests.test_projection import GOLD, calibrations, cases, images_of

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_sampling_is_bit_equal_to_the_reference_fixtures():
    from heal_swin_amd import projection as P

    g, cals = np.load(GOLD), calibrations()
    for tag in cases(g):
        img, mask = images_of(g, tag, cals[tag.split("/")[0]])
        hp_img = P.sample_bilinear_u8(dev(img), g[tag + "/v"], g[tag + "/u"]).cpu().numpy()
        hp_mask = P.sample_mask(dev(mask), g[tag + "/v"], g[tag + "/u"], 3).cpu().numpy()
        assert np.array_equal(hp_img, g[tag + "/hp_img"]), tag
        assert np.array_equal(hp_mask, g[tag + "/hp_mask"]), tag


def test_sampling_edge_cases():
    """Integer coordinates (0, as in the reference), borders, far outside, NaN, exact halves in the mask."""
    from heal_swin_amd import projection as P

    g = np.load(GOLD)
    want = g["edge/bilinear"]
    with np.errstate(invalid="ignore"):
        want = np.where(np.isnan(want), 0, want).astype(np.uint8)  # .astype(np.uint8) of the NaN samples is 0 on x86
    got = P.sample_bilinear_u8(dev(g["edge/img"]), g["edge/rx"], g["edge/ry"]).cpu().numpy()
    assert np.array_equal(got, want)
    got_m = P.sample_mask(dev(g["edge/mask"]), g["edge/rx"], g["edge/ry"], 5).cpu().numpy()
    assert np.array_equal(got_m, g["edge/hp_mask"])
    # empty coordinate list, and non-finite coordinates of every kind
    assert P.sample_bilinear_u8(dev(g["edge/img"]), np.zeros(0), np.zeros(0)).shape == (3, 0)
    bad = np.array([np.inf, -np.inf, np.nan, 1e300, -1e300])
    assert not P.sample_bilinear_u8(dev(g["edge/img"]), bad, np.ones(5)).any()
    assert (P.sample_mask(dev(g["edge/mask"]), np.ones(5), bad, 9).cpu().numpy() == 9).all()


def test_batched_full_size_projection_matches_the_oracle():
    """WoodScape frame size (966 x 1280), nside 256, 8 base pixels (524 288 pixels), a batch of 3 frames against one table."""
    from heal_swin_amd import projection as P

    cal = calibrations()["fv_966x1280"]
    rng = np.random.default_rng(5)
    imgs = rng.integers(0, 256, (3, 3, 966, 1280), dtype=np.uint8)
    imgs[1, :, 300:600, 400:900] = 117  # constant region: c vs c - 1 after truncation depends on every rounding
    masks = rng.integers(0, 10, (3, 966, 1280), dtype=np.uint8)
    proj = P.HPProjector(cal, 256, 8, rotate_pole=True, s2_bkgd_class=7)
    hp_img, hp_mask = proj(dev(imgs), dev(masks))
    assert hp_img.shape == (3, 3, 8 * 256 * 256) and hp_mask.shape == (3, 8 * 256 * 256)
    u, v = proj.u.cpu().numpy(), proj.v.cpu().numpy()
    import time
    import conftest
    t0 = time.perf_counter()
    with np.errstate(invalid="ignore"):
        OP.sample_bilinear(imgs[0], v, u).astype(np.uint8)
        OP.sample_mask(masks[0], v, u, 7)
    conftest.NOTES.append(f"projection: numpy oracle {1e3 * (time.perf_counter() - t0):.0f} ms per 966x1280 frame at nside 256 "
                          "(GPU kernels: tools/bench_projection.py, profiles/archive_r01_r04/r02_projection_sampling.json)")
    with np.errstate(invalid="ignore"):
        for b in range(3):
            assert np.array_equal(hp_img[b].cpu().numpy(), OP.sample_bilinear(imgs[b], v, u).astype(np.uint8))
            assert np.array_equal(hp_mask[b].cpu().numpy(), OP.sample_mask(masks[b], v, u, 7))
    assert (hp_img[1].cpu().numpy() == 116).any()  # the truncation effect is present and reproduced
    # the whole chain against the oracle's own grid: the two pix2ang differ by <= 1 ulp of arccos at some pixels, which can
    # move single samples by one grey level; nothing else may differ
    with np.errstate(invalid="ignore"):
        ref_img, ref_mask = OP.project_to_hp(imgs[0], masks[0], cal, 256, 8, True, 7)
    d = np.abs(hp_img[0].cpu().numpy().astype(int) - ref_img.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3
    assert (hp_mask[0].cpu().numpy() != ref_mask).mean() < 1e-5


def test_projected_sample_round_trips_through_the_npz_format(tmp_path):
    from heal_swin_amd import data, projection as P

    cal = calibrations()["mvl_96x128"]
    rng = np.random.default_rng(1)
    img, mask = rng.integers(0, 256, (3, 96, 128), dtype=np.uint8), rng.integers(0, 10, (96, 128), dtype=np.uint8)
    proj = P.HPProjector(cal, 16, 8)
    hp_img, hp_mask = proj(dev(img), dev(mask))
    data.write_sample(os.path.join(tmp_path, "s.npz"), hp_img.cpu().numpy(), hp_mask.cpu().numpy())
    ds = data.HPSegmentationNpzDataset(str(tmp_path))
    a, b = ds[0]
    assert np.array_equal(a, hp_img.cpu().numpy()) and np.array_equal(b, hp_mask.cpu().numpy())
