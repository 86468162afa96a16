"""Index tables of the HEAL-SWIN hot path (numpy int64).  Oracle / test infra -- see oracle/__init__.py.

Everything here is integer work and must match the reference BIT-EXACTLY; it is pinned against
golden vectors captured from the reference (tests/golden/tables.npz, tests/test_oracle_tables.py).

The reference builds these tables procedurally on nested indices; this restatement works in
HEALPix face coordinates (face, ix, iy) with `nested = face*nside^2 + interleave(ix, iy)`, which
turns the reference's offset searches into plain translations.
"""
import numpy as np

from . import healpix as hpx


def _isqrt_pow2(ws):
    side = int(round(ws ** 0.5))
    if side * side != ws or side & (side - 1):
        raise AssertionError(f"window_size must be 4^k, got {ws}")
    return side


def nest_win_idcs(window_size):
    """sqrt(Ws) x sqrt(Ws) table of nested indices of one window.

    Follows reference `get_nest_win_idcs`, hp_windowing.py:43-62.  The reference fills quadrants
    recursively as [[1, 0], [3, 2]]; that is exactly `ix = side-1-col`, `iy = row`, so
    table[row, col] = interleave(side-1-col, row).
    """
    side = _isqrt_pow2(window_size)
    row, col = np.meshgrid(np.arange(side), np.arange(side), indexing="ij")
    return hpx.xyf2nest(side, side - 1 - col, row, np.zeros_like(row)).astype(np.int64)


def rel_pos_index(window_size):
    """[Ws, Ws] index into the (2*side-1)^2 relative-position bias table, tokens in nested order.

    Follows reference WindowAttention.__init__, swin_hp_transformer.py:98-114: Swin's Cartesian
    index (drow + side-1) * (2*side-1) + (dcol + side-1) with rows/cols taken from the window
    table above, i.e. for nested token n: row = iy(n), col = side-1-ix(n).
    """
    side = _isqrt_pow2(window_size)
    n = np.arange(window_size)
    ix, iy, _ = hpx.nest2xyf(side, n)
    row, col = iy, side - 1 - ix
    drow = row[:, None] - row[None, :] + side - 1
    dcol = col[:, None] - col[None, :] + side - 1
    return (drow * (2 * side - 1) + dcol).astype(np.int64)


# ----------------------------------------------------------------------------- shifts
# Every shifter is described by
#   idx    int64 [N]   shift(x)[:, j] = x[:, idx[j]]           (reference .shift)
#   inv    int64 [N]   shift_back(y)[:, i] = y[:, inv[i]]      (reference .shift_back), inv = idx^-1
#   labels       [N]   region label of shifted position j; attention between two positions of a
#                      window is masked (-100) iff their labels differ (get_attn_mask_from_mask)


def invert_permutation(idx):
    """reference `_get_inverse_index_map` (hp_shifting.py:258-259, 390-391) = argsort."""
    idx = np.asarray(idx, dtype=np.int64)
    if not np.array_equal(np.sort(idx), np.arange(idx.size)):
        raise AssertionError("shift is not a permutation")  # reference _validate_shift_result
    inv = np.empty_like(idx)
    inv[idx] = np.arange(idx.size)
    return inv


def nest_roll_shift(n_pix, window_size, shift_size):
    """reference NestRollShift, hp_shifting.py:42-73.

    shift = roll(x, -s) -> idx[j] = (j + s) mod N; labels 0 | 1 | 2 on [0, N-Ws) | [N-Ws, N-s) | [N-s, N).
    labels are float32 in the reference (mask dtype float32).
    """
    j = np.arange(n_pix, dtype=np.int64)
    idx = (j + shift_size) % n_pix
    labels = np.zeros(n_pix, dtype=np.int64)
    labels[n_pix - window_size: n_pix - shift_size] = 1
    labels[n_pix - shift_size:] = 2
    return idx, invert_permutation(idx), labels


# base pixel that a half-window step in -y (resp. -x) leaves a face into (8-base-pixel layout);
# equivalent to reference BASE_PIX_OFFSETS, hp_shifting.py:126 and :196
_GRID_FACE_Y = np.array([5, 6, 7, 4, 0, 1, 2, 3], dtype=np.int64)
_GRID_FACE_X = np.array([4, 5, 6, 7, 0, 1, 2, 3], dtype=np.int64)


def nest_grid_shift(nside, base_pix, window_size):
    """reference NestGridShift, hp_shifting.py:76-306.

    In face coordinates the reference's two passes are translations by half a window side h:
      dir1 (`_get_shifted_idcs_dir1`, :162-182): source = (f, ix, iy - h), leaving face f through its
            iy = 0 edge into face _GRID_FACE_Y[f] at iy + nside - h;
      dir2 (`_get_shifted_idcs_dir2`, :225-251): source = (f, ix - h, iy), leaving through ix = 0 into
            _GRID_FACE_X[f];
    composed as shift_idcs = dir1[dir2] (:90-91).
    labels (`get_mask(get_attn_mask=False)`, :261-300), faces 4..7 only:
      bottom window row (window iy == 0), lower half (iy_in < h)  -> f + 1
      left window column (window ix == 0), left half (ix_in < h)  -> f + 5   (overrides)
      faces 0..3: first quarter (ix_in < h and iy_in < h) of window (0, 0) -> f + 5
    """
    assert base_pix == 8, "NestGridShift is currently only implemented for 8 base pixels"
    side = _isqrt_pow2(window_size)
    h = side // 2
    npix = base_pix * nside * nside
    j = np.arange(npix, dtype=np.int64)

    def step(p, axis):
        ix, iy, f = hpx.nest2xyf(nside, p)
        if axis == "y":
            cross = iy < h
            f2 = np.where(cross, _GRID_FACE_Y[f], f)
            return hpx.xyf2nest(nside, ix, (iy - h) % nside, f2)
        cross = ix < h
        f2 = np.where(cross, _GRID_FACE_X[f], f)
        return hpx.xyf2nest(nside, (ix - h) % nside, iy, f2)

    idx = step(step(j, "x"), "y")  # dir1[dir2[j]]
    inv = invert_permutation(idx)

    ix, iy, f = hpx.nest2xyf(nside, j)
    wx, wy = ix // side, iy // side
    px, py = ix % side, iy % side
    labels = np.zeros(npix, dtype=np.int64)
    upper = f >= 4
    m_bottom = upper & (wy == 0) & (py < h)
    labels[m_bottom] = f[m_bottom] + 1
    m_left = upper & (wx == 0) & (px < h)
    labels[m_left] = f[m_left] + 5
    m_co = (~upper) & (wx == 0) & (wy == 0) & (px < h) & (py < h)
    labels[m_co] = f[m_co] + 5
    return idx, inv, labels


_RING_LOST_FROM = {4: 7, 5: 4, 6: 5, 7: 6}  # reference hp_shifting.py:354


def ring_shift(nside, base_pix, window_size, shift_size):
    """reference RingShift, hp_shifting.py:309-404.

    Roll by `shift_size` along the RING ordering of the full 12*nside^2 sphere (:327-334):
        source(j) = ring2nest((nest2ring(j) - s) mod 12 nside^2)
    Positions whose source lies outside the first `base_pix` faces get label face+1 (:336-344) and
    are re-filled, in increasing position order, from the sorted pixels nothing maps to
    ("lost" pixels, :346-378): faces 4..7 take from face _RING_LOST_FROM[face]; faces 0..3 take
    the left-overs in order.  The reference only works for base_pix == 8 (the dict has no other keys).
    labels are int64 in the reference (mask dtype int64, :380).
    """
    if base_pix != 8:
        # reference behaviour: ValueError / IndexError / KeyError depending on base_pix
        raise ValueError("RingShift is only valid for base_pix == 8")
    nfull = 12 * nside * nside
    npface = nside * nside
    npix = base_pix * npface
    j = np.arange(npix, dtype=np.int64)
    src = hpx.ring2nest(nside, (hpx.nest2ring(nside, j) - shift_size) % nfull)
    face = j // npface
    outside = src >= npix
    labels = np.where(outside, face + 1, 0).astype(np.int64)

    used = np.zeros(nfull, dtype=bool)
    used[src] = True
    lost = [np.flatnonzero(~used[f * npface:(f + 1) * npface]) + f * npface for f in range(base_pix)]

    idx = src.copy()
    leftovers = []
    for f in range(4, base_pix):
        pos = np.flatnonzero(outside & (face == f))
        pool = lost[_RING_LOST_FROM[f]]
        assert pos.size <= pool.size, f"for base pixel {f}, there were not enough source pixel"
        idx[pos] = pool[: pos.size]
        leftovers.append(pool[pos.size:])
    leftovers = np.concatenate(leftovers)
    pos = np.flatnonzero(outside & (face < 4))
    assert leftovers.size == pos.size, "unused source pixels do not match pixels to be filled"
    idx[pos] = leftovers
    return idx, invert_permutation(idx), labels


def attn_mask_from_labels(labels, window_size):
    """reference `get_attn_mask_from_mask`, hp_shifting.py:10-28: [nW, Ws, Ws], -100 where labels differ."""
    lab = np.asarray(labels).reshape(-1, window_size)
    diff = lab[:, None, :] - lab[:, :, None]
    return np.where(diff != 0, -100, 0)
