"""Fisheye image -> HEALPix projection on the GPU (second half): the step that produces the model's input
(`hp_img` uint8 [3, Npix], `hp_mask` uint8 [Npix], nested order on the first `base_pix` base pixels) from a calibrated
fisheye frame.  Mirrors heal_swin/data/segmentation/project_on_s2.py -- same function names, arguments and results:

  hp_grid(nside, base_pix)                          :347-354  pix2ang of the first base_pix * nside^2 nested pixels
                                                              (`hs_pix2ang_nest`, host C++; the reference calls healpy)
  rot_grid(theta, phi, cal_info, inv)               :108-136  grid rotated so that the camera axis is the pole
  project_s2_points_to_img(theta, phi, cal_info, rotate_pole)  :141-183  WoodScape polynomial fisheye model -> (u, v)
  sample_bilinear_u8(img, rx, ry) / sample_mask(mask, rx, ry, bkgd)      :38-80   `hs_sample_bilinear_u8` / `hs_sample_mask_u8`
                                                              (HIP, float64, bit-exact; device tensors only)
  HPProjector                                        :344-372  project_dataset_hp for one calibration: the coordinate table
                                                              is built once and stays on the GPU, every batch of frames is
                                                              two kernel launches

The coordinate table is host setup work done once per calibration (the reference caches it per calibration as well); it is
float64 numpy with the reference's expression order, so that (u, v) -- and with them every sampled byte -- equal the
reference's bit for bit.  The per-image work is on the GPU only: there is no CPU sampling path.
"""
import numpy as np
import torch

from ._lib import check, lib, np_ptr, ptr, stream_ptr

_EXT_REF = {"FV": (1.0, 0.0, 0.0), "RV": (-1.0, 0.0, 0.0), "MVL": (0.0, 1.0, 0.0), "MVR": (0.0, -1.0, 0.0)}


def hp_grid(nside, base_pix=8):
    """(theta, phi) float64 of nested pixels 0 .. base_pix * nside^2 - 1."""
    n = int(nside) * int(nside) * int(base_pix)
    theta, phi = np.empty(n, dtype=np.float64), np.empty(n, dtype=np.float64)
    check(lib.hs_pix2ang_nest(int(nside), 0, n, np_ptr(theta), np_ptr(phi)), "hs_pix2ang_nest")
    return theta, phi


def _camera_rotation(cal_info):
    x, y, z, w = np.asarray(cal_info["extrinsic"]["quaternion"], dtype=np.float64) / np.linalg.norm(cal_info["extrinsic"]["quaternion"])
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rot_grid(theta, phi, cal_info, inv=False):
    """The grid rotated by R = Rz(phi_ref) Ry(theta_ref), (theta_ref, phi_ref) the direction of the camera's reference axis
    (FV +x, RV -x, MVL +y, MVR -y of the vehicle frame) in camera coordinates; inv=True applies R^-1."""
    axis = _camera_rotation(cal_info).T @ np.asarray(_EXT_REF[cal_info["name"]])
    phi_ref, theta_ref = np.arctan2(axis[1], axis[0]), np.arccos(axis[2])
    ct, st, cp, sp = np.cos(theta_ref), np.sin(theta_ref), np.cos(phi_ref), np.sin(phi_ref)
    m = np.array([[cp, -sp, 0.0], [sp, cp, 0.0], [0.0, 0.0, 1.0]]) @ np.array([[ct, 0.0, st], [0.0, 1.0, 0.0], [-st, 0.0, ct]])
    if inv:
        m = m.T
    sin_t = np.sin(theta)
    xyz = np.stack(((np.cos(phi) * sin_t).reshape(-1), (np.sin(phi) * sin_t).reshape(-1), np.cos(theta).reshape(-1)), axis=-1) @ m.T
    with np.errstate(invalid="ignore"):
        return np.arccos(xyz[:, 2]).reshape(theta.shape), np.arctan2(xyz[:, 1], xyz[:, 0]).reshape(phi.shape)


def project_s2_points_to_img(theta, phi, cal_info, rotate_pole=False):
    """Float pixel coordinates (u along the width, v along the height) of spherical points."""
    if rotate_pole:
        theta, phi = rot_grid(theta, phi, cal_info, inv=False)
    intr = cal_info["intrinsic"]
    rho = 0
    for order in range(1, intr["poly_order"] + 1):
        rho += intr["k" + str(order)] * theta**order
    u = rho * np.cos(phi) + intr["cx_offset"] + int(intr["width"]) / 2 - 0.5
    v = rho * np.sin(phi) * intr["aspect_ratio"] + intr["cy_offset"] + int(intr["height"]) / 2 - 0.5
    return u, v


def _device_u8(t, ndim, what):
    if not torch.is_tensor(t):
        t = torch.from_numpy(np.ascontiguousarray(t))
    if t.dtype != torch.uint8:
        raise TypeError(f"{what} must be uint8 (the reference samples tv.io.read_image frames and class-id masks), got {t.dtype}")
    if not t.is_cuda:
        raise RuntimeError(f"{what} must be a GPU tensor: the sampling runs in the HIP kernels only (no CPU path)")
    lead = ndim - t.dim()
    if lead not in (0, 1):
        raise ValueError(f"{what}: expected {ndim - 1} or {ndim} dimensions, got {t.dim()}")
    return (t[None] if lead else t).contiguous(), bool(lead)


def _coords(r, device):
    if torch.is_tensor(r):
        return r.to(device=device, dtype=torch.float64).contiguous().reshape(-1)
    return torch.from_numpy(np.ascontiguousarray(r, dtype=np.float64).reshape(-1)).to(device)


def sample_bilinear_u8(img, rx, ry):
    """`sample_bilinear(img, rx, ry).astype(np.uint8)`: img uint8 [C, H, W] or [B, C, H, W] on the GPU, rx along H, ry along
    W (float64 arrays or tensors) -> uint8 [(B,) C, n]."""
    img, squeeze = _device_u8(img, 4, "img")
    rx, ry = _coords(rx, img.device), _coords(ry, img.device)
    assert rx.shape == ry.shape, "rx and ry must have the same shape"
    b, c, h, w = img.shape
    out = torch.empty((b, c, rx.numel()), dtype=torch.uint8, device=img.device)
    check(lib.hs_sample_bilinear_u8(ptr(img), b, c, h, w, ptr(rx), ptr(ry), rx.numel(), ptr(out), stream_ptr(img.device)),
          "hs_sample_bilinear_u8")
    return out[0] if squeeze else out


def sample_mask(mask, rx, ry, s2_bkgd_class=0):
    """Nearest-pixel (round half to even) class ids: mask uint8 [H, W] or [B, H, W] on the GPU -> uint8 [(B,) n]."""
    mask, squeeze = _device_u8(mask, 3, "mask")
    rx, ry = _coords(rx, mask.device), _coords(ry, mask.device)
    assert rx.shape == ry.shape, "rx and ry must have the same shape"
    b, h, w = mask.shape
    out = torch.empty((b, rx.numel()), dtype=torch.uint8, device=mask.device)
    check(lib.hs_sample_mask_u8(ptr(mask), b, h, w, ptr(rx), ptr(ry), rx.numel(), int(s2_bkgd_class), ptr(out),
                                stream_ptr(mask.device)), "hs_sample_mask_u8")
    return out[0] if squeeze else out


class HPProjector:
    """project_dataset_hp for the frames of one camera: `proj(imgs, masks)` -> (hp_img uint8 [B, C, Npix], hp_mask uint8
    [B, Npix]) on the GPU, ready for `SwinHPTransformerSys.forward` and `seg_loss` (or `data.write_sample`)."""

    def __init__(self, cal_info, nside, base_pix=8, rotate_pole=False, s2_bkgd_class=0, device="cuda"):
        self.nside, self.base_pix, self.s2_bkgd_class = int(nside), int(base_pix), int(s2_bkgd_class)
        theta, phi = hp_grid(nside, base_pix)
        u, v = project_s2_points_to_img(theta, phi, cal_info, rotate_pole)
        self.device = torch.device(device)
        self.u, self.v = _coords(u, self.device), _coords(v, self.device)  # resident coordinate table: 16 B per pixel

    @property
    def npix(self):
        return self.u.numel()

    def __call__(self, imgs, masks=None):
        hp_img = sample_bilinear_u8(imgs, self.v, self.u)
        if masks is None:
            return hp_img
        return hp_img, sample_mask(masks, self.v, self.u, self.s2_bkgd_class)
