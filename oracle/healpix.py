"""HEALPix ring <-> nested pixel index conversion (numpy, int64).  Oracle / test infra.

The reference calls `healpy.pixelfunc.ring2nest` / `nest2ring` (healpy==1.15.2, pinned in
`/root/reference/setup.py:22`) from `heal_swin/models_torch/hp_shifting.py:329,333`.  healpy is a
third-party dependency that is absent here, so this restates the published HEALPix algorithm
(Gorski et al. 2005, "HEALPix: a framework for high-resolution discretization ...", sec. 4 and
the `nest2xyf / xyf2ring / ring2xyf / xyf2nest` decomposition of the HEALPix C++ `T_Healpix_Base`):

  nested index  = face * nside^2 + interleave(ix, iy)        (ix -> even bits, iy -> odd bits)
  ring number   jr = jrll[face]*nside - ix - iy - 1            (1 .. 4*nside-1, north to south)
  in-ring index jp = (jpll[face]*nr + ix - iy + 1 + kshift)/2  wrapped to 1 .. 4*nr

PARITY UNPINNED against healpy itself (it cannot be run here); pinned by healpy's docstring examples
(tests/test_oracle_tables.py::test_healpix_known_answers), by bijection/inverse properties, and by an independent geometric
derivation of the pixel centres in both schemes (tests/test_healpix_geometry.py: closed-form ring-index formulas vs the HEALPix
projection plane for the nested index agree to 1e-12 under these maps for every pixel at nside 1..256).
"""
import numpy as np

# ring index (in units of nside) of the top corner of each base pixel, and its phi offset
_JRLL = np.array([2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4], dtype=np.int64)
_JPLL = np.array([1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7], dtype=np.int64)


def _check_nside(nside):
    nside = int(nside)
    if nside < 1 or (nside & (nside - 1)) != 0:
        raise ValueError(f"nside must be a power of two for nested ordering, got {nside}")
    return nside


def _compact_bits(v):
    """Keep the even bits of v and squeeze them together (inverse of _spread_bits)."""
    v = v & 0x5555555555555555
    v = (v | (v >> 1)) & 0x3333333333333333
    v = (v | (v >> 2)) & 0x0F0F0F0F0F0F0F0F
    v = (v | (v >> 4)) & 0x00FF00FF00FF00FF
    v = (v | (v >> 8)) & 0x0000FFFF0000FFFF
    v = (v | (v >> 16)) & 0x00000000FFFFFFFF
    return v


def _spread_bits(v):
    """Insert a zero bit above every bit of v (v < 2^31)."""
    v = v & 0x00000000FFFFFFFF
    v = (v | (v << 16)) & 0x0000FFFF0000FFFF
    v = (v | (v << 8)) & 0x00FF00FF00FF00FF
    v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0F
    v = (v | (v << 2)) & 0x3333333333333333
    v = (v | (v << 1)) & 0x5555555555555555
    return v


def nest2xyf(nside, ipix):
    """nested index -> (ix, iy, face)."""
    nside = _check_nside(nside)
    ipix = np.asarray(ipix, dtype=np.int64)
    npface = nside * nside
    face = ipix // npface
    p = ipix % npface
    return _compact_bits(p), _compact_bits(p >> 1), face


def xyf2nest(nside, ix, iy, face):
    nside = _check_nside(nside)
    return face * (nside * nside) + _spread_bits(ix) + (_spread_bits(iy) << 1)


def nest2ring(nside, ipix):
    """Restates healpy.pixelfunc.nest2ring (used at reference hp_shifting.py:333)."""
    nside = _check_nside(nside)
    ipix = np.asarray(ipix, dtype=np.int64)
    npix = 12 * nside * nside
    if ipix.size and (ipix.min() < 0 or ipix.max() >= npix):
        raise ValueError("pixel index out of range")
    ix, iy, face = nest2xyf(nside, ipix)
    nl4 = 4 * nside
    ncap = 2 * nside * (nside - 1)
    jr = _JRLL[face] * nside - ix - iy - 1

    north = jr < nside
    south = jr > 3 * nside
    equat = ~(north | south)

    nr = np.where(north, jr, np.where(south, nl4 - jr, nside))
    n_before = np.where(
        north,
        2 * nr * (nr - 1),
        np.where(south, npix - 2 * (nr + 1) * nr, ncap + (jr - nside) * nl4),
    )
    kshift = np.where(equat, (jr - nside) & 1, 0)

    jp = (_JPLL[face] * nr + ix - iy + 1 + kshift) // 2
    jp = np.where(jp > nl4, jp - nl4, jp)
    jp = np.where(jp < 1, jp + nl4, jp)
    return n_before + jp - 1


def _isqrt(v):
    r = np.floor(np.sqrt(v.astype(np.float64))).astype(np.int64)
    # fix possible float rounding at perfect squares
    r = np.where(r * r > v, r - 1, r)
    r = np.where((r + 1) * (r + 1) <= v, r + 1, r)
    return r


def ring2xyf(nside, ipix):
    """ring index -> (ix, iy, face)."""
    nside = _check_nside(nside)
    ipix = np.asarray(ipix, dtype=np.int64)
    npix = 12 * nside * nside
    if ipix.size and (ipix.min() < 0 or ipix.max() >= npix):
        raise ValueError("pixel index out of range")
    nl2 = 2 * nside
    nl4 = 4 * nside
    ncap = 2 * nside * (nside - 1)

    north = ipix < ncap
    south = ipix >= npix - ncap
    equat = ~(north | south)

    # --- north polar cap
    iring_n = (1 + _isqrt(1 + 2 * np.where(north, ipix, 0))) >> 1
    iphi_n = ipix + 1 - 2 * iring_n * (iring_n - 1)
    face_n = (iphi_n - 1) // np.maximum(iring_n, 1)

    # --- equatorial belt
    ip_e = np.where(equat, ipix - ncap, 0)
    tmp = ip_e // nl4
    iring_e = tmp + nside
    iphi_e = ip_e - tmp * nl4 + 1
    kshift_e = (iring_e + nside) & 1
    ire = tmp + 1
    irm = nl2 + 1 - tmp
    ifm = (iphi_e - ire // 2 + nside - 1) // nside
    ifp = (iphi_e - irm // 2 + nside - 1) // nside
    face_e = np.where(ifp == ifm, ifp | 4, np.where(ifp < ifm, ifp, ifm + 8))

    # --- south polar cap
    ip_s = np.where(south, npix - ipix, 1)
    iring_s = (1 + _isqrt(2 * ip_s - 1)) >> 1
    iphi_s = 4 * iring_s + 1 - (ip_s - 2 * iring_s * (iring_s - 1))
    face_s = 8 + (iphi_s - 1) // np.maximum(iring_s, 1)

    iring = np.where(north, iring_n, np.where(south, 2 * nl2 - iring_s, iring_e))
    iphi = np.where(north, iphi_n, np.where(south, iphi_s, iphi_e))
    nr = np.where(north, iring_n, np.where(south, iring_s, nside))
    kshift = np.where(equat, kshift_e, 0)
    face = np.where(north, face_n, np.where(south, face_s, face_e))

    irt = iring - _JRLL[face] * nside + 1
    ipt = 2 * iphi - _JPLL[face] * nr - kshift - 1
    ipt = np.where(ipt >= nl2, ipt - 8 * nside, ipt)
    ix = (ipt - irt) >> 1
    iy = (-(ipt + irt)) >> 1
    return ix, iy, face


def ring2nest(nside, ipix):
    """Restates healpy.pixelfunc.ring2nest (used at reference hp_shifting.py:329)."""
    ix, iy, face = ring2xyf(nside, ipix)
    return xyf2nest(nside, ix, iy, face)


def pix2ang_nest(nside, ipix):
    """(theta, phi) of the centres of nested pixels, float64: restates healpy.pixelfunc.pix2ang(nside, ipix, nest=True)
    (reference data/segmentation/project_on_s2.py:350) after the HEALPix C++ `T_Healpix_Base::pix2loc`:
        ring jr = jrll[face] nside - ix - iy - 1;   polar caps: z = +-(1 - nr^2 fact2), nr = jr or 4 nside - jr;
        belt: z = (2 nside - jr) fact1;   phi = (pi/4) (jpll[face] nr + ix - iy) / nr;   theta = acos z, or
        atan2(sqrt(tmp (2 - tmp)), z) with tmp = nr^2 fact2 where |z| > 0.99 (the library's accurate form near the poles).
    PARITY UNPINNED against healpy (absent here); pinned by healpy's docstring examples (tests/test_projection.py) and by the
    independent ring-centre formulas of tests/test_healpix_geometry.py."""
    nside = _check_nside(nside)
    ipix = np.asarray(ipix, dtype=np.int64)
    npix = 12 * nside * nside
    if ipix.size and (ipix.min() < 0 or ipix.max() >= npix):
        raise ValueError("pixel index out of range")
    ix, iy, face = nest2xyf(nside, ipix)
    fact2 = 4.0 / npix
    fact1 = (nside << 1) * fact2
    jr = _JRLL[face] * nside - ix - iy - 1
    north = jr < nside
    south = jr > 3 * nside
    nr = np.where(north, jr, np.where(south, 4 * nside - jr, nside))
    tmp = (nr * nr).astype(np.float64) * fact2
    z = np.where(north, 1.0 - tmp, np.where(south, tmp - 1.0, (2 * nside - jr).astype(np.float64) * fact1))
    sth = np.sqrt(tmp * (2.0 - tmp))
    near_pole = (north & (z > 0.99)) | (south & (z < -0.99))
    theta = np.where(near_pole, np.arctan2(sth, z), np.arccos(z))
    t = _JPLL[face] * nr + ix - iy
    t = np.where(t < 0, t + 8 * nr, t)
    halfpi = 0.5 * np.pi
    phi = np.where(nr == nside, 0.75 * halfpi * t.astype(np.float64) * fact1, (0.5 * halfpi * t.astype(np.float64)) / nr)
    return theta, phi


def pix2ang_ring(nside, ipix):
    """healpy.pixelfunc.pix2ang(nside, ipix) in the (default) ring scheme, through ring2nest."""
    return pix2ang_nest(nside, ring2nest(nside, ipix))
