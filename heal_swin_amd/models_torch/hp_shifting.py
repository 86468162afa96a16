"""Shift strategies -- mirrors heal_swin/models_torch/hp_shifting.py (NoShift, NestRollShift, NestGridShift,
RingShift, get_attn_mask_from_mask) with the same constructor signatures and attributes.

Differences in mechanism, not in results:
  * the permutations come from the host-side C++ builders (`hs_build_*_shift`), generated in HEALPix face
    coordinates instead of the reference's nested-index offset searches;
  * besides `shift_idcs` / `back_shift_idcs` (int64, as in the reference) every shifter carries the compact
    device tables the kernels read: `idx` / `inv` (int32) and `labels` (uint8, region label per shifted
    position) -- the [nW, Ws, Ws] mask is never materialised on the hot path;
  * `shift` / `shift_back` run the HIP row-gather kernel (`hs_gather_rows`); inside the model they are not
    called at all because the permutation is fused into the window-attention kernel.
"""
import numpy as np
import torch

from .. import _lib, ops
from .hp_windowing import window_partition


def get_attn_mask_from_mask(mask, window_size):
    """(N,) region labels -> (nW, Ws, Ws) mask in {0, -100} (reference hp_shifting.py:10-28).  Host-side."""
    m = mask.detach().cpu()
    lab = m.to(torch.int64).numpy()
    assert lab.min() >= 0 and lab.max() < 256, "region labels must fit uint8"
    out = torch.from_numpy(_lib.attn_mask_from_labels(lab.astype(np.uint8), window_size))
    return out.to(torch.int64) if not mask.is_floating_point() else out.to(mask.dtype)


class _ShifterBase:
    """Device tables shared by the three strategies."""

    mask_dtype = torch.float32  # dtype of the reference's attn_mask buffer

    def _set_tables(self, idx, inv, labels, window_size):
        self.window_size_ = window_size
        self._idx_np, self._inv_np, self._labels_np = idx, inv, labels
        self.shift_idcs = torch.from_numpy(idx.astype(np.int64))
        self.back_shift_idcs = torch.from_numpy(inv.astype(np.int64))
        self._dev = {}

    def tables(self, device):
        """(idx int32, inv int32, labels uint8) resident on `device` (uploaded once, unlike the reference's
        CPU-resident shift_idcs that are re-copied on every call)."""
        key = str(device)
        if key not in self._dev:
            self._dev[key] = tuple(torch.from_numpy(a).to(device) for a in (self._idx_np, self._inv_np, self._labels_np))
        return self._dev[key]

    def labels(self):
        return torch.from_numpy(self._labels_np.copy())

    def get_mask(self, get_attn_mask=True):
        lab = torch.from_numpy(self._labels_np.astype(np.int64)).to(self.mask_dtype)
        if not get_attn_mask:
            return lab
        return get_attn_mask_from_mask(lab, self.window_size_)

    def shift(self, x):
        idx, inv, _ = self.tables(x.device)
        return ops.gather_rows(x, idx, inv, 0)

    def shift_back(self, x):
        idx, inv, _ = self.tables(x.device)
        return ops.gather_rows(x, inv, idx, 0)


class NoShift:
    def get_mask(self):
        return None

    def shift(self, x):
        return x

    def shift_back(self, x):
        return x


class NestRollShift(_ShifterBase):
    """roll along the nested pixel axis (reference hp_shifting.py:42-73)."""

    def __init__(self, shift_size, input_resolution, window_size):
        self.shift_size = shift_size
        self.input_resolution = input_resolution
        self.window_size = window_size
        idx, inv, lab = _lib.build_nest_roll_shift(input_resolution, window_size, shift_size)
        self._set_tables(idx, inv, lab, window_size)

    def shift(self, x):
        return ops.gather_rows(x, None, None, self.shift_size % x.shape[1])

    def shift_back(self, x):
        return ops.gather_rows(x, None, None, (-self.shift_size) % x.shape[1])


class NestGridShift(_ShifterBase):
    """half-window diagonal shift on the 8-base-pixel grid (reference hp_shifting.py:76-306)."""

    def __init__(self, nside, base_pix, window_size):
        assert base_pix == 8, "NestGridShift is currently only implemented for 8 base pixels"
        self.nside = nside
        self.ws = window_size
        self.base_pix = base_pix
        self.npix = base_pix * nside**2
        self.n_windows = self.npix // self.ws
        idx, inv, lab = _lib.build_nest_grid_shift(nside, base_pix, window_size)
        self._set_tables(idx, inv, lab, window_size)


class RingShift(_ShifterBase):
    """roll in ring ordering of the full sphere, mapped through ring<->nest (reference hp_shifting.py:309-404).
    Like the reference it is only valid for base_pix == 8; the mask dtype is int64 (:380)."""

    mask_dtype = torch.int64

    def __init__(self, nside, base_pix, window_size, shift_size):
        self.nside = nside
        self.base_pix = base_pix
        self.npix = base_pix * nside**2
        self.ws = window_size
        self.shift_size = shift_size
        idx, inv, lab = _lib.build_ring_shift(nside, base_pix, window_size, shift_size)
        self._set_tables(idx, inv, lab, window_size)
        self.mask = torch.from_numpy(lab.astype(np.int64))
