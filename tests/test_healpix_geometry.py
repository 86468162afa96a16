"""An INDEPENDENT pin of ring <-> nest (the one piece of arithmetic the reference takes from healpy, which is not available
here: SURVEY 8c "parity unpinned").  The oracle and the C++ library both restate HEALPix's integer index formulas; an error
shared by the two would be invisible to a comparison between them.  This test derives the pixel CENTRES (z = cos(colatitude),
phi) in two unrelated ways and requires them to coincide under our nest2ring:

  * RING scheme: the closed-form ring-index -> (z, phi) formulas of Gorski et al. 2005 (ApJ 622, 759), eqs. (2)-(9):
    polar caps  i = ring, j = position in ring:  z = 1 - i^2 / (3 nside^2),  phi = pi / (2 i) (j - 1/2);
    equatorial belt  z = 4/3 - 2 i / (3 nside),  phi = pi / (2 nside) (j - s / 2) for the shifted rings (s = (i - nside + 1)
    mod 2 = 1) and pi / (2 nside) (j - 1) for the others (first pixel of the ring AT phi = 0, the software convention);
  * NESTED scheme: geometry only -- pixel (face, ix, iy) is a point of the HEALPix PROJECTION plane (Calabretta & Roukema 2007,
    H = 4, K = 3): the twelve base faces are squares of diagonal pi/2 centred at (pi/4 + f pi/2, pi/4), (f pi/2, 0),
    (pi/4 + f pi/2, -pi/4); ix runs north-east, iy north-west from the face's southern corner; the inverse projection maps the
    plane point to (z, phi).  No ring index, jrll / jpll table or kshift enters this side.

Agreement to 1e-12 for every pixel at several nside (and the same for ring2nest as the inverse) pins both restatements to
the HEALPix definition itself rather than to each other.  (The bit interleaving nested index <-> (face, ix, iy) is the
definition of the NESTED scheme and is additionally pinned by healpy's docstring examples in test_oracle_tables.py.)
"""
import numpy as np
import pytest

from oracle import healpix as H


def ring_centres(nside, p):
    """(z, phi) of RING-scheme pixels p: Gorski et al. 2005."""
    p = np.asarray(p, dtype=np.int64)
    npix = 12 * nside * nside
    ncap = 2 * nside * (nside - 1)
    z = np.empty(p.shape, dtype=np.float64)
    phi = np.empty(p.shape, dtype=np.float64)
    north = p < ncap
    south = p >= npix - ncap
    belt = ~(north | south)
    # north polar cap
    ph = (p[north] + 1) / 2.0
    i = np.floor(np.sqrt(ph - np.sqrt(np.floor(ph)))).astype(np.int64) + 1
    j = p[north] + 1 - 2 * i * (i - 1)
    z[north] = 1.0 - i * i / (3.0 * nside * nside)
    phi[north] = np.pi / (2.0 * i) * (j - 0.5)
    # equatorial belt
    pe = p[belt] - ncap
    i = pe // (4 * nside) + nside
    j = pe % (4 * nside) + 1
    s = (i - nside + 1) % 2
    z[belt] = 4.0 / 3.0 - 2.0 * i / (3.0 * nside)
    # the paper writes phi = pi / (2 nside) (j - s / 2); the HEALPix software (and healpy, the reference's dependency) counts
    # the unshifted rings (s = 0) from phi = 0 rather than ending them there, i.e. (j - 1) instead of j: the same set of
    # centres, labelled from the pixel at phi = 0 -- face 4 is centred on phi = 0 and holds ring pixel 4 at nside 1
    phi[belt] = np.pi / (2.0 * nside) * (j - np.where(s == 1, 0.5, 1.0))
    # south polar cap: mirror image (ring i counted from the south pole, same position j along the ring)
    q = npix - 1 - p[south]  # index counted from the end
    ph = (q + 1) / 2.0
    i = np.floor(np.sqrt(ph - np.sqrt(np.floor(ph)))).astype(np.int64) + 1
    jrev = q + 1 - 2 * i * (i - 1)      # position counted from the END of the ring
    j = 4 * i + 1 - jrev
    z[south] = -(1.0 - i * i / (3.0 * nside * nside))
    phi[south] = np.pi / (2.0 * i) * (j - 0.5)
    return z, np.mod(phi, 2 * np.pi)


def nested_centres(nside, p):
    """(z, phi) of NESTED-scheme pixels p through the HEALPix projection plane."""
    p = np.asarray(p, dtype=np.int64)
    face = p // (nside * nside)
    within = p % (nside * nside)
    ix = np.zeros_like(within)
    iy = np.zeros_like(within)
    for b in range(32):  # de-interleave: even bits -> ix, odd bits -> iy (the definition of the nested order)
        ix |= ((within >> (2 * b)) & 1) << b
        iy |= ((within >> (2 * b + 1)) & 1) << b
    row = face // 4  # 0 north, 1 equatorial, 2 south
    col = face % 4
    xc = np.where(row == 1, col * np.pi / 2, np.pi / 4 + col * np.pi / 2)
    yc = np.where(row == 0, np.pi / 4, np.where(row == 1, 0.0, -np.pi / 4))
    x = xc + (np.pi / 4) * (ix - iy) / nside
    y = yc - np.pi / 4 + (np.pi / 4) * (ix + iy + 1) / nside
    z = np.empty(p.shape, dtype=np.float64)
    phi = np.empty(p.shape, dtype=np.float64)
    eq = np.abs(y) <= np.pi / 4 + 1e-15
    z[eq] = 8.0 * y[eq] / (3.0 * np.pi)
    phi[eq] = x[eq]
    po = ~eq
    sigma = 2.0 - 4.0 * np.abs(y[po]) / np.pi
    z[po] = np.sign(y[po]) * (1.0 - sigma * sigma / 3.0)
    phic = np.floor(np.mod(x[po], 2 * np.pi) / (np.pi / 2)) * (np.pi / 2) + np.pi / 4  # centre of the polar triangle
    xm = np.mod(x[po], 2 * np.pi)
    phi[po] = phic + (xm - phic) / sigma
    return z, np.mod(phi, 2 * np.pi)


def _assert_same_points(za, pa, zb, pb):
    assert np.abs(za - zb).max() < 1e-12
    d = np.abs(pa - pb)
    d = np.minimum(d, 2 * np.pi - d)
    assert d.max() < 1e-11


@pytest.mark.parametrize("nside", [1, 2, 4, 8, 16, 64, 256])
def test_oracle_nest2ring_matches_pixel_geometry(nside):
    a = np.arange(12 * nside * nside)
    zn, pn = nested_centres(nside, a)
    zr, pr = ring_centres(nside, H.nest2ring(nside, a))
    _assert_same_points(zn, pn, zr, pr)
    # and the inverse map
    zr2, pr2 = ring_centres(nside, a)
    zn2, pn2 = nested_centres(nside, H.ring2nest(nside, a))
    _assert_same_points(zr2, pr2, zn2, pn2)


@pytest.mark.parametrize("nside", [1, 2, 8, 32, 128, 256])
def test_library_nest2ring_matches_pixel_geometry(nside):
    """The same for the C++ host tables of libhealswin (hs_nest2ring / hs_ring2nest, called on the CPU)."""
    from heal_swin_amd import _lib
    a = np.arange(12 * nside * nside)
    zn, pn = nested_centres(nside, a)
    zr, pr = ring_centres(nside, _lib.nest2ring(nside, a))
    _assert_same_points(zn, pn, zr, pr)
    zr2, pr2 = ring_centres(nside, a)
    zn2, pn2 = nested_centres(nside, _lib.ring2nest(nside, a))
    _assert_same_points(zr2, pr2, zn2, pn2)


def test_ring_centres_are_ordered_as_the_ring_scheme_requires():
    """Sanity of the reference side itself: ring-scheme pixels run north to south, and eastwards inside a ring."""
    for nside in (1, 4, 32):
        z, phi = ring_centres(nside, np.arange(12 * nside * nside))
        assert np.all(np.diff(z) <= 1e-15)
        same_ring = np.abs(np.diff(z)) < 1e-15
        assert np.all(np.diff(phi)[same_ring] > 0)
