"""Window views and the nested-window index table -- mirrors heal_swin/models_torch/hp_windowing.py.

In nested ordering a window is a contiguous run of `window_size` pixels, so partition / reverse are views
(reference hp_windowing.py:18-21, :37-40); the model never calls them on the hot path because the
fused attention kernel indexes windows directly.
"""
import math

import torch

from .. import _lib


def _assert_pow2(window_size):
    # reference: assert (math.log(window_size) / math.log(2)) % 1 == 0   (hp_windowing.py:16, :35)
    assert window_size > 0 and (math.log(window_size) / math.log(2)) % 1 == 0


def window_partition(x, window_size):
    """x: (B, N, C) -> (num_windows*B, window_size, C); window row index = b*nW + w."""
    _assert_pow2(window_size)
    B, N, C = x.shape
    return x.contiguous().view(B * (N // window_size), window_size, C)


def window_reverse(windows, window_size, N):
    """windows: (num_windows*B, window_size, C) -> (B, N, C)."""
    _assert_pow2(window_size)
    B = int(windows.shape[0] / (N // window_size))
    return windows.contiguous().view(B, N, -1)


def get_nest_win_idcs(window_size):
    """sqrt(Ws) x sqrt(Ws) int64 tensor of nested indices (reference hp_windowing.py:43-62), built by the
    host-side C++ table builder `hs_nest_win_idcs`."""
    return torch.from_numpy(_lib.nest_win_idcs(window_size))
