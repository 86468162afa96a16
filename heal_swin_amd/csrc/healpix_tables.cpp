// Host-side HEALPix index tables of the HEAL-SWIN hot path (built once per model, plain C++).
//
// The reference builds its shift permutations procedurally on nested indices (offset searches over
// the Z-order tree, models_torch/hp_shifting.py:101-251).  Here every table is generated in face
// coordinates (face, ix, iy), nested = face*nside^2 + interleave(ix, iy), where the shifts are plain
// translations and the ring<->nest maps are the standard HEALPix formulas.
#include <algorithm>
#include <cmath>
#include <vector>

#include "hs_common.h"

namespace {

// ring index (units of nside) of each base pixel's northern corner, and its longitude offset
const int kJrll[12] = {2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4};
const int kJpll[12] = {1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7};

inline uint64_t spread_bits(uint64_t v) {  // abc -> 0a0b0c
    v &= 0xFFFFFFFFull;
    v = (v | (v << 16)) & 0x0000FFFF0000FFFFull;
    v = (v | (v << 8)) & 0x00FF00FF00FF00FFull;
    v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}
inline uint64_t compact_bits(uint64_t v) {  // keeps the even bits
    v &= 0x5555555555555555ull;
    v = (v | (v >> 1)) & 0x3333333333333333ull;
    v = (v | (v >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    v = (v | (v >> 4)) & 0x00FF00FF00FF00FFull;
    v = (v | (v >> 8)) & 0x0000FFFF0000FFFFull;
    v = (v | (v >> 16)) & 0x00000000FFFFFFFFull;
    return v;
}

struct Xyf {
    int64_t ix, iy;
    int face;
};

inline Xyf nest_to_xyf(int64_t nside, int64_t p) {
    const int64_t npface = nside * nside;
    const int64_t in_face = p % npface;
    return {(int64_t)compact_bits((uint64_t)in_face), (int64_t)compact_bits((uint64_t)in_face >> 1), (int)(p / npface)};
}
inline int64_t xyf_to_nest(int64_t nside, int64_t ix, int64_t iy, int face) {
    return (int64_t)face * nside * nside + (int64_t)(spread_bits((uint64_t)ix) | (spread_bits((uint64_t)iy) << 1));
}

inline int64_t xyf_to_ring(int64_t nside, const Xyf& c) {
    const int64_t nl4 = 4 * nside, npix = 12 * nside * nside, ncap = 2 * nside * (nside - 1);
    const int64_t jr = (int64_t)kJrll[c.face] * nside - c.ix - c.iy - 1;  // ring number, 1..4nside-1
    int64_t nr, n_before, kshift;
    if (jr < nside) {  // north polar cap
        nr = jr;
        n_before = 2 * nr * (nr - 1);
        kshift = 0;
    } else if (jr > 3 * nside) {  // south polar cap
        nr = nl4 - jr;
        n_before = npix - 2 * (nr + 1) * nr;
        kshift = 0;
    } else {  // equatorial belt
        nr = nside;
        n_before = ncap + (jr - nside) * nl4;
        kshift = (jr - nside) & 1;
    }
    int64_t jp = ((int64_t)kJpll[c.face] * nr + c.ix - c.iy + 1 + kshift) / 2;
    if (jp > nl4) jp -= nl4;
    if (jp < 1) jp += nl4;
    return n_before + jp - 1;
}

inline int64_t isqrt64(int64_t v) {
    int64_t r = (int64_t)std::sqrt((double)v);
    while (r * r > v) --r;
    while ((r + 1) * (r + 1) <= v) ++r;
    return r;
}

inline Xyf ring_to_xyf(int64_t nside, int64_t pix) {
    const int64_t nl2 = 2 * nside, nl4 = 4 * nside, npix = 12 * nside * nside, ncap = 2 * nside * (nside - 1);
    int64_t iring, iphi, kshift, nr;
    int face;
    if (pix < ncap) {
        iring = (1 + isqrt64(1 + 2 * pix)) >> 1;
        iphi = pix + 1 - 2 * iring * (iring - 1);
        kshift = 0;
        nr = iring;
        face = (int)((iphi - 1) / nr);
    } else if (pix < npix - ncap) {
        const int64_t ip = pix - ncap;
        const int64_t row = ip / nl4;
        iring = row + nside;
        iphi = ip - row * nl4 + 1;
        kshift = (iring + nside) & 1;
        nr = nside;
        const int64_t ire = row + 1, irm = nl2 + 1 - row;
        const int64_t ifm = (iphi - ire / 2 + nside - 1) / nside;
        const int64_t ifp = (iphi - irm / 2 + nside - 1) / nside;
        face = (int)((ifp == ifm) ? (ifp | 4) : ((ifp < ifm) ? ifp : (ifm + 8)));
    } else {
        const int64_t ip = npix - pix;
        const int64_t ir = (1 + isqrt64(2 * ip - 1)) >> 1;
        iphi = 4 * ir + 1 - (ip - 2 * ir * (ir - 1));
        kshift = 0;
        nr = ir;
        iring = 2 * nl2 - ir;
        face = 8 + (int)((iphi - 1) / nr);
    }
    const int64_t irt = iring - (int64_t)kJrll[face] * nside + 1;
    int64_t ipt = 2 * iphi - (int64_t)kJpll[face] * nr - kshift - 1;
    if (ipt >= nl2) ipt -= 8 * nside;
    // arithmetic shifts on possibly negative numerators floor, as in the HEALPix formulas
    return {(ipt - irt) >> 1, (-(ipt + irt)) >> 1, face};
}

int check_nside(int nside) {
    if (nside < 1 || !hs::is_pow2(nside)) return hs::fail(HS_ERR_INVALID_ARG, "nside must be a power of two, got %d", nside);
    return HS_OK;
}

// inverse permutation + the reference's permutation check (_validate_shift_result)
int finish_shift(const std::vector<int64_t>& src, int32_t* idx, int32_t* inv) {
    const int64_t n = (int64_t)src.size();
    std::vector<int32_t> tmp(n, -1);
    for (int64_t j = 0; j < n; ++j) {
        const int64_t s = src[j];
        if (s < 0 || s >= n || tmp[s] != -1) return hs::fail(HS_ERR_NOT_PERMUTATION, "shift validation failed at position %lld", (long long)j);
        tmp[s] = (int32_t)j;
    }
    if (idx)
        for (int64_t j = 0; j < n; ++j) idx[j] = (int32_t)src[j];
    if (inv) std::copy(tmp.begin(), tmp.end(), inv);
    return HS_OK;
}

// base pixel entered when a half-window step in -y / -x leaves face f (8-base-pixel layout of the
// reference, equivalent to BASE_PIX_OFFSETS at hp_shifting.py:126 and :196)
const int kGridFaceY[8] = {5, 6, 7, 4, 0, 1, 2, 3};
const int kGridFaceX[8] = {4, 5, 6, 7, 0, 1, 2, 3};
// which face's unmapped pixels refill face f (GET_LOST_FROM, hp_shifting.py:354)
inline int ring_lost_from(int f) { return f == 4 ? 7 : f - 1; }

}  // namespace

extern "C" {

int hs_nest2ring(int nside, const int64_t* in, int64_t* out, int64_t n) {
    if (int st = check_nside(nside)) return st;
    HS_CHECK_ARG(in && out && n >= 0, "null pointer or negative count");
    const int64_t npix = 12ll * nside * nside;
    for (int64_t i = 0; i < n; ++i) {
        HS_CHECK_ARG(in[i] >= 0 && in[i] < npix, "pixel index %lld out of range for nside %d", (long long)in[i], nside);
        out[i] = xyf_to_ring(nside, nest_to_xyf(nside, in[i]));
    }
    return HS_OK;
}

int hs_ring2nest(int nside, const int64_t* in, int64_t* out, int64_t n) {
    if (int st = check_nside(nside)) return st;
    HS_CHECK_ARG(in && out && n >= 0, "null pointer or negative count");
    const int64_t npix = 12ll * nside * nside;
    for (int64_t i = 0; i < n; ++i) {
        HS_CHECK_ARG(in[i] >= 0 && in[i] < npix, "pixel index %lld out of range for nside %d", (long long)in[i], nside);
        const Xyf c = ring_to_xyf(nside, in[i]);
        out[i] = xyf_to_nest(nside, c.ix, c.iy, c.face);
    }
    return HS_OK;
}

int hs_nest_win_idcs(int window_size, int64_t* out) {
    const int side = hs::isqrt_pow2_window(window_size);
    HS_CHECK_ARG(side > 0 && out, "window_size must be 4^k, got %d", window_size);
    // the reference's recursive quadrant fill [[1,0],[3,2]] is ix = side-1-col, iy = row
    for (int r = 0; r < side; ++r)
        for (int c = 0; c < side; ++c) out[r * side + c] = xyf_to_nest(side, side - 1 - c, r, 0);
    return HS_OK;
}

int hs_rel_pos_index(int window_size, int64_t* out) {
    const int side = hs::isqrt_pow2_window(window_size);
    HS_CHECK_ARG(side > 0 && out, "window_size must be 4^k, got %d", window_size);
    const int span = 2 * side - 1;
    for (int a = 0; a < window_size; ++a) {
        const Xyf ca = nest_to_xyf(side, a);
        for (int b = 0; b < window_size; ++b) {
            const Xyf cb = nest_to_xyf(side, b);
            const int64_t drow = ca.iy - cb.iy + side - 1;
            const int64_t dcol = (side - 1 - ca.ix) - (side - 1 - cb.ix) + side - 1;
            out[(int64_t)a * window_size + b] = drow * span + dcol;
        }
    }
    return HS_OK;
}

int hs_build_nest_roll_shift(int64_t n_pix, int window_size, int shift_size, int32_t* idx, int32_t* inv, uint8_t* labels) {
    HS_CHECK_ARG(n_pix > 0 && n_pix < (1ll << 31), "n_pix out of range");
    HS_CHECK_ARG(window_size > 0 && window_size <= n_pix, "window_size %d does not fit %lld pixels", window_size, (long long)n_pix);
    HS_CHECK_ARG(shift_size > 0 && shift_size < window_size, "shift_size must be in (0, window_size)");
    for (int64_t j = 0; j < n_pix; ++j) {
        if (idx) idx[j] = (int32_t)((j + shift_size) % n_pix);
        if (inv) inv[j] = (int32_t)((j - shift_size + n_pix) % n_pix);
        if (labels) labels[j] = j < n_pix - window_size ? 0 : (j < n_pix - shift_size ? 1 : 2);
    }
    return HS_OK;
}

int hs_build_nest_grid_shift(int nside, int base_pix, int window_size, int32_t* idx, int32_t* inv, uint8_t* labels) {
    if (int st = check_nside(nside)) return st;
    HS_CHECK_ARG(base_pix == 8, "NestGridShift is currently only implemented for 8 base pixels");
    const int side = hs::isqrt_pow2_window(window_size);
    HS_CHECK_ARG(side >= 2 && side <= nside, "window_size %d invalid for nside %d", window_size, nside);
    const int64_t h = side / 2, npix = (int64_t)base_pix * nside * nside;
    std::vector<int64_t> src(npix);
    for (int64_t j = 0; j < npix; ++j) {
        Xyf c = nest_to_xyf(nside, j);
        // second pass of the reference (dir2, -x) is applied to the destination first: idx = dir1[dir2[j]]
        if (c.ix < h) c.face = kGridFaceX[c.face];
        c.ix = (c.ix - h + nside) % nside;
        if (c.iy < h) c.face = kGridFaceY[c.face];
        c.iy = (c.iy - h + nside) % nside;
        src[j] = xyf_to_nest(nside, c.ix, c.iy, c.face);
    }
    if (int st = finish_shift(src, idx, inv)) return st;
    if (labels) {
        for (int64_t j = 0; j < npix; ++j) {
            const Xyf c = nest_to_xyf(nside, j);
            const int64_t wx = c.ix / side, wy = c.iy / side, px = c.ix % side, py = c.iy % side;
            uint8_t lab = 0;
            if (c.face >= 4) {
                if (wy == 0 && py < h) lab = (uint8_t)(c.face + 1);
                if (wx == 0 && px < h) lab = (uint8_t)(c.face + 5);
            } else if (wx == 0 && wy == 0 && px < h && py < h) {
                lab = (uint8_t)(c.face + 5);
            }
            labels[j] = lab;
        }
    }
    return HS_OK;
}

int hs_build_ring_shift(int nside, int base_pix, int window_size, int shift_size, int32_t* idx, int32_t* inv, uint8_t* labels) {
    if (int st = check_nside(nside)) return st;
    HS_CHECK_ARG(base_pix == 8, "RingShift is only valid for base_pix == 8 (the reference fails for every other value)");
    HS_CHECK_ARG(window_size > 0 && (window_size & (window_size - 1)) == 0, "window_size must be a power of two, got %d", window_size);
    HS_CHECK_ARG(shift_size > 0, "shift_size must be positive");
    const int64_t npface = (int64_t)nside * nside, npix = base_pix * npface, nfull = 12 * npface;
    std::vector<int64_t> src(npix);
    std::vector<uint8_t> used(nfull, 0);
    for (int64_t j = 0; j < npix; ++j) {
        const int64_t ring = xyf_to_ring(nside, nest_to_xyf(nside, j));
        const int64_t from = ((ring - shift_size) % nfull + nfull) % nfull;
        const Xyf c = ring_to_xyf(nside, from);
        src[j] = xyf_to_nest(nside, c.ix, c.iy, c.face);
        used[src[j]] = 1;
    }
    // pixels of each used face that no position reads ("lost"), ascending
    std::vector<std::vector<int64_t>> lost(base_pix);
    for (int f = 0; f < base_pix; ++f)
        for (int64_t p = f * npface; p < (f + 1) * npface; ++p)
            if (!used[p]) lost[f].push_back(p);
    if (labels)
        for (int64_t j = 0; j < npix; ++j) labels[j] = src[j] >= npix ? (uint8_t)(j / npface + 1) : 0;
    std::vector<int64_t> leftovers;
    for (int f = 4; f < base_pix; ++f) {
        const std::vector<int64_t>& pool = lost[ring_lost_from(f)];
        size_t k = 0;
        for (int64_t j = f * npface; j < (f + 1) * npface; ++j)
            if (src[j] >= npix) {
                if (k >= pool.size()) return hs::fail(HS_ERR_INVALID_ARG, "for base pixel %d, there were not enough source pixel", f);
                src[j] = pool[k++];
            }
        leftovers.insert(leftovers.end(), pool.begin() + k, pool.end());
    }
    size_t k = 0;
    for (int64_t j = 0; j < 4 * npface; ++j)
        if (src[j] >= npix) {
            if (k >= leftovers.size()) return hs::fail(HS_ERR_INVALID_ARG, "unused source pixels do not match the pixels to be filled");
            src[j] = leftovers[k++];
        }
    if (k != leftovers.size()) return hs::fail(HS_ERR_INVALID_ARG, "unused source pixels do not match the pixels to be filled");
    return finish_shift(src, idx, inv);
}

int hs_attn_mask_from_labels(const uint8_t* labels, int64_t n, int window_size, float* out) {
    HS_CHECK_ARG(labels && out && window_size > 0 && n % window_size == 0, "bad arguments");
    const int64_t nw = n / window_size;
    for (int64_t w = 0; w < nw; ++w)
        for (int i = 0; i < window_size; ++i)
            for (int j = 0; j < window_size; ++j)
                out[(w * window_size + i) * window_size + j] =
                    labels[w * window_size + i] != labels[w * window_size + j] ? -100.0f : 0.0f;
    return HS_OK;
}

/* healpy.pixelfunc.pix2ang(nside, ipix, nest=True) for ipix = first .. first + count - 1 (reference
 * data/segmentation/project_on_s2.py:347-354): pixel centres after the HEALPix C++ pix2loc -- polar caps
 * z = +-(1 - nr^2 fact2), belt z = (2 nside - jr) fact1, phi = (pi/4) (jpll nr + ix - iy) / nr, theta = acos z or
 * atan2(sqrt(tmp (2 - tmp)), z) where |z| > 0.99.  Same operation order as oracle/healpix.py:pix2ang_nest. */
int hs_pix2ang_nest(int nside, int64_t first, int64_t count, double* theta, double* phi) {
    if (int st = check_nside(nside)) return st;
    const int64_t npix = 12 * (int64_t)nside * nside;
    HS_CHECK_ARG(theta && phi && first >= 0 && count >= 0 && first + count <= npix, "pixel range outside [0, 12 nside^2)");
    const double fact2 = 4.0 / (double)npix, fact1 = (double)(nside << 1) * fact2, halfpi = 0.5 * M_PI;
    for (int64_t k = 0; k < count; ++k) {
        const Xyf c = nest_to_xyf(nside, first + k);
        const int64_t jr = (int64_t)kJrll[c.face] * nside - c.ix - c.iy - 1;
        int64_t nr;
        double z, th;
        if (jr < nside) {
            nr = jr;
            const double tmp = (double)(nr * nr) * fact2;
            z = 1.0 - tmp;
            th = z > 0.99 ? std::atan2(std::sqrt(tmp * (2.0 - tmp)), z) : std::acos(z);
        } else if (jr > 3 * (int64_t)nside) {
            nr = 4 * (int64_t)nside - jr;
            const double tmp = (double)(nr * nr) * fact2;
            z = tmp - 1.0;
            th = z < -0.99 ? std::atan2(std::sqrt(tmp * (2.0 - tmp)), z) : std::acos(z);
        } else {
            nr = nside;
            z = (double)(2 * (int64_t)nside - jr) * fact1;
            th = std::acos(z);
        }
        int64_t t = (int64_t)kJpll[c.face] * nr + c.ix - c.iy;
        if (t < 0) t += 8 * nr;
        theta[k] = th;
        phi[k] = nr == nside ? 0.75 * halfpi * (double)t * fact1 : (0.5 * halfpi * (double)t) / (double)nr;
    }
    return HS_OK;
}

}  // extern "C"
