"""Fisheye image -> HEALPix projection (numpy, float64).  Oracle / test infrastructure: only tests/, smoke() and bench.py's
cpu_baseline may import this; the product path is heal_swin_amd/projection.py over `hs_sample_*` (HIP) and never calls it.

Restates the forward projection of the reference's data preparation, heal_swin/data/segmentation/project_on_s2.py:
  project_dataset_hp          :344-372   grid = pix2ang(nside, nest) of the first base_pix * nside^2 pixels; per image
                                         hp_img = sample_bilinear(img, v, u).astype(uint8), hp_mask = sample_mask(mask, v, u, bkgd)
  project_s2_points_to_img    :141-183   rho = sum_i k_i theta^i;  u = rho cos(phi) + cx + W/2 - 1/2,
                                         v = rho sin(phi) aspect + cy + H/2 - 1/2         (WoodScape polynomial fisheye model)
  rot_grid                    :108-136   optional rotation of the grid so that the camera axis is the pole
  sample_bilinear             :38-73     four floor / ceil neighbours, out-of-image neighbours contribute 0, weights
                                         (i1 - r) and (r - i0) -- BOTH zero when r is an integer: such samples are 0, as in
                                         the reference -- x-direction first, then y; truncation to uint8 by the caller
  sample_mask                 :76-80     nearest pixel by np.around (half to even), out-of-image -> background class
The arithmetic keeps the reference's operation order so that results are bit-identical given the same (theta, phi).
pix2ang is healpy's (absent): oracle/healpix.py:pix2ang_nest, parity unpinned against healpy (see there).
Pinned by tests/golden/projection.npz (outputs of the reference functions themselves, tests/golden/make_golden.py)."""
import numpy as np

from .healpix import pix2ang_nest

_EXT_REF = {"FV": (1.0, 0.0, 0.0), "RV": (-1.0, 0.0, 0.0), "MVL": (0.0, 1.0, 0.0), "MVR": (0.0, -1.0, 0.0)}


def hp_grid(nside, base_pix):
    """(theta, phi) of the first base_pix * nside^2 nested pixels (ref :347-354)."""
    return pix2ang_nest(nside, np.arange(nside * nside * base_pix, dtype=np.int64))


def _quat_matrix(q):
    """Rotation matrix of the unit quaternion q = (x, y, z, w) (scalar last, scipy's convention, ref :109)."""
    x, y, z, w = np.asarray(q, dtype=np.float64) / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def grid_rotation(cal_info, inv=False):
    """3x3 matrix applied to the grid's unit vectors by rot_grid (ref :108-125): extrinsic 'yz' Euler rotation
    R = Rz(phi_ref) Ry(theta_ref) that takes the pole to the camera's reference axis expressed in camera coordinates."""
    r = _quat_matrix(cal_info["extrinsic"]["quaternion"])
    int_ref = r.T @ np.asarray(_EXT_REF[cal_info["name"]])  # r.inv().apply(ext_ref)
    phi_ref = np.arctan2(int_ref[1], int_ref[0])
    theta_ref = np.arccos(int_ref[2])
    cy_, sy_ = np.cos(theta_ref), np.sin(theta_ref)
    cz_, sz_ = np.cos(phi_ref), np.sin(phi_ref)
    ry = np.array([[cy_, 0, sy_], [0, 1, 0], [-sy_, 0, cy_]])
    rz = np.array([[cz_, -sz_, 0], [sz_, cz_, 0], [0, 0, 1]])
    m = rz @ ry  # extrinsic: first about y, then about z
    return m.T if inv else m


def rot_grid(theta, phi, cal_info, inv=False):
    """ref :108-136."""
    m = grid_rotation(cal_info, inv)
    xyz = np.stack(((np.cos(phi) * np.sin(theta)).reshape(-1), (np.sin(phi) * np.sin(theta)).reshape(-1), np.cos(theta).reshape(-1)), axis=-1)
    rot = xyz @ m.T
    phi_rot = np.arctan2(rot[:, 1], rot[:, 0]).reshape(phi.shape)
    theta_rot = np.arccos(rot[:, 2]).reshape(theta.shape)
    return theta_rot, phi_rot


def project_s2_points_to_img(theta, phi, cal_info, rotate_pole=False):
    """Spherical points -> float image coordinates (u along the width, v along the height), ref :141-183."""
    if rotate_pole:
        theta, phi = rot_grid(theta, phi, cal_info, inv=False)
    it = cal_info["intrinsic"]
    rho = 0
    for order in range(1, it["poly_order"] + 1):
        rho += it["k" + str(order)] * theta**order
    u = rho * np.cos(phi)
    v = rho * np.sin(phi)
    u = u + it["cx_offset"] + int(it["width"]) / 2 - 0.5
    v = v * it["aspect_ratio"] + it["cy_offset"] + int(it["height"]) / 2 - 0.5
    return u, v


def _within(signal, x, y, background):
    """sample_within_bounds, ref :23-35 (x indexes the second-to-last axis, y the last)."""
    h, w = signal.shape[-2:]
    ok = (0 <= x) & (x < h) & (0 <= y) & (y < w)
    out = np.full(signal.shape[:-2] + x.shape, background)
    out[..., ok] = signal[..., x[ok], y[ok]]
    return out


def sample_bilinear(signal, rx, ry):
    """signal [C, H, W], rx along H, ry along W -> float64 [C, *rx.shape] (ref :38-73)."""
    signal = np.asarray(signal)
    ix0, iy0 = np.floor(rx).astype(int), np.floor(ry).astype(int)
    ix1, iy1 = np.ceil(rx).astype(int), np.ceil(ry).astype(int)
    s00, s10 = _within(signal, ix0, iy0, 0), _within(signal, ix1, iy0, 0)
    s01, s11 = _within(signal, ix0, iy1, 0), _within(signal, ix1, iy1, 0)
    fx1 = (ix1 - rx) * s00 + (rx - ix0) * s10
    fx2 = (ix1 - rx) * s01 + (rx - ix0) * s11
    return (iy1 - ry) * fx1 + (ry - iy0) * fx2


def sample_mask(mask, rx, ry, background):
    """mask [H, W] -> uint8 [*rx.shape], nearest neighbour by round-half-even (ref :76-80)."""
    xi, yi = np.around(rx, 0).astype(int), np.around(ry, 0).astype(int)
    return _within(np.asarray(mask), xi, yi, background).astype(np.uint8)


def project_to_hp(img, mask, cal_info, nside, base_pix=8, rotate_pole=False, s2_bkgd_class=0):
    """One sample of project_dataset_hp (ref :344-372): (hp_img uint8 [C, Npix], hp_mask uint8 [Npix])."""
    theta, phi = hp_grid(nside, base_pix)
    u, v = project_s2_points_to_img(theta, phi, cal_info, rotate_pole)
    return sample_bilinear(img, v, u).astype(np.uint8), sample_mask(mask, v, u, s2_bkgd_class)
