"""ctypes binding of libhealswin.so (C ABI declared in include/healswin.h).

There is no fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libhealswin.so")

HS_F32, HS_BF16 = 0, 1
HS_ATTN_COSINE = 1
HS_ATTN_FORCE_VALU = 2
HS_ATTN_RESIDUAL = 4
HS_ATTN_OVERWRITE_GRADS = 8
HS_EPI_BIAS, HS_EPI_GELU, HS_EPI_DGELU, HS_EPI_RESID = 0, 1, 2, 3
HS_ACC_DEFER = 2
HS_MLP_NORM_AFTER = 16

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_uint = ctypes.c_uint
c_ptr = ctypes.c_void_p
c_float = ctypes.c_float

# name -> argtypes; every function returns int status unless listed in _OTHER_RESTYPE
_SIGNATURES = {
    "hs_nest2ring": [c_int, c_ptr, c_ptr, c_i64],
    "hs_ring2nest": [c_int, c_ptr, c_ptr, c_i64],
    "hs_nest_win_idcs": [c_int, c_ptr],
    "hs_rel_pos_index": [c_int, c_ptr],
    "hs_build_nest_roll_shift": [c_i64, c_int, c_int, c_ptr, c_ptr, c_ptr],
    "hs_build_nest_grid_shift": [c_int, c_int, c_int, c_ptr, c_ptr, c_ptr],
    "hs_build_ring_shift": [c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_ptr],
    "hs_attn_mask_from_labels": [c_ptr, c_i64, c_int, c_ptr],
    "hs_rel_bias_gather": [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr],
    "hs_rel_bias_scatter_grad_sorted_add": [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr],
    "hs_cos_head_scale_fwd": [c_ptr, c_ptr, c_int, c_ptr],
    "hs_cos_head_scale_bwd": [c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr],
    "hs_rel_bias_gather_many": [c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_int, c_ptr],
    "hs_rel_bias_scatter_grad_sorted_many": [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_int, c_ptr],
    "hs_cos_head_scale_many": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr],
    "hs_rel_bias_scatter_grad": [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr],
    "hs_rel_bias_scatter_grad_sorted": [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr],
    "hs_window_attn_fwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr,
                           c_int, c_i64, c_int, c_int, c_int, c_uint, ctypes.c_float, ctypes.c_uint64, c_int, c_ptr],
    "hs_window_attn_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr,
                           c_int, c_i64, c_int, c_int, c_int, c_uint, ctypes.c_float, ctypes.c_uint64, c_int, c_ptr],
    "hs_gather_rows": [c_ptr, c_ptr, c_ptr, c_i64, c_int, c_i64, c_i64, c_ptr],
    "hs_pix2ang_nest": [c_int, c_i64, c_i64, c_ptr, c_ptr],
    "hs_ln_head_supported": [c_int, c_int, c_int],
    "hs_expand_ln_head_supported": [c_int, c_int, c_int, c_int],
    "hs_expand_ln_head_fwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr],
    "hs_expand_ln_head_ce_fwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int,
                                 c_int, c_ptr],
    "hs_ln_head_ce_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr],
    "hs_ln_head_fwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr],
    "hs_ln_head_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr],
    "hs_sample_bilinear_u8": [c_ptr, c_int, c_int, c_int, c_int, c_ptr, c_ptr, c_i64, c_ptr, c_ptr],
    "hs_sample_mask_u8": [c_ptr, c_int, c_int, c_int, c_ptr, c_ptr, c_i64, c_int, c_ptr, c_ptr],
    "hs_gelu_fwd": [c_ptr, c_ptr, c_i64, ctypes.c_float, ctypes.c_uint64, c_int, c_ptr],
    "hs_gelu_bwd": [c_ptr, c_ptr, c_ptr, c_i64, ctypes.c_float, ctypes.c_uint64, c_int, c_ptr],
    "hs_residual_drop": [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, ctypes.c_float, ctypes.c_uint64, c_int, c_ptr],
    "hs_linear_wgrad": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_ptr],
    "hs_reduce_flush": [c_ptr],
    "hs_set_seed_epoch": [c_ptr],
    "hs_linear_wgrad_gelu": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_ptr],
    "hs_linear_wgrad_gelu_supported": [c_i64, c_int, c_int, c_int],
    "hs_mlp_fused_supported": [c_int, c_int, c_int],
    "hs_mlp_fused_fwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_uint,
                         c_int, c_ptr],
    "hs_mlp_fused_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr],
    "hs_mlp_fused_drop_fwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_float,
                              ctypes.c_uint64, ctypes.c_uint64, c_i64, c_int, c_int, c_uint, c_int, c_ptr],
    "hs_mlp_fused_drop_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_float, ctypes.c_uint64, c_i64, c_int, c_int, c_int, c_ptr],
    "hs_adam_step": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_float, c_ptr, c_float, c_float, c_float, c_float, c_int, c_ptr, c_ptr],
    "hs_adam_advance": [c_ptr, c_ptr],
    "hs_split_bf16x3": [c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr],
    "hs_gelu_split3": [c_ptr, c_ptr, c_ptr, c_i64, c_int, ctypes.c_float, ctypes.c_uint64, c_ptr],
    "hs_linear_wgrad_ld": [c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr],
    "hs_layernorm_fwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr],
    "hs_add_layernorm_fwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_ptr],
    "hs_add_layernorm_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_i64, c_int, c_int, c_ptr],
    "hs_layernorm_drop_fwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, ctypes.c_float, ctypes.c_uint64,
                              c_i64, c_int, c_int, c_ptr],
    "hs_layernorm_drop_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_i64, ctypes.c_float,
                              ctypes.c_uint64, c_i64, c_int, c_int, c_ptr],
    "hs_add_layernorm_drop_fwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, ctypes.c_float,
                                  ctypes.c_uint64, c_i64, c_int, c_int, c_ptr],
    "hs_layernorm_fwd_ex": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, ctypes.c_float,
                            ctypes.c_uint64, c_i64, c_int, c_int, c_ptr],
    "hs_add_layernorm_drop_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_i64,
                                  ctypes.c_float, ctypes.c_uint64, c_i64, c_int, c_int, c_ptr],
    "hs_seg_ce_fwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_i64, c_i64, c_i64, c_int, c_i64, c_int, c_ptr],
    "hs_seg_ce_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_int, c_i64,
                      c_int, c_ptr],
    "hs_gemm_nt": [c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_i64, c_ptr, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int,
                   ctypes.c_float, ctypes.c_uint64, c_int, c_ptr],
    "hs_gemm_nt_set_tile": [c_int],
    "hs_set_reserved_cus": [c_int],
    "hs_transpose_many_16": [c_ptr, c_int, c_int, c_ptr],
    "hs_debug_occupy_cus": [c_int, c_int, c_int, ctypes.c_double, c_ptr],
    "hs_debug_buffer_soffset_probe": [c_ptr, c_int, c_int, c_ptr, c_ptr],
    "hs_window_attn_module_supported": [c_int, c_int, c_int, c_int],
    "hs_window_attn_module_fwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_ptr, c_int, c_i64,
                                  c_int, c_int, c_int, c_uint, c_int, c_ptr],
    "hs_window_attn_module_fwd_train": [c_ptr] * 17 + [c_i64] + [c_ptr] * 6 + [c_int, c_i64, c_int, c_int, c_int, c_uint, c_int, c_ptr],
    "hs_window_attn_module_bwd_chain": [c_ptr] * 14 + [c_i64] + [c_ptr] * 11 + [c_int, c_int, c_i64, c_int, c_int, c_int, c_uint, c_int, c_ptr],
    "hs_layernorm_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_i64, c_int, c_int, c_ptr],
    "hs_patch_merge_fwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_ptr],
    "hs_patch_merge_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_i64,
                           c_int, c_int, c_int, c_ptr],
    "hs_patch_expand_fwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_int, c_int, c_int, c_int, c_ptr],
    "hs_patch_expand_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_i64,
                            c_int, c_int, c_int, c_int, c_ptr],
}
_OTHER = {
    "hs_version": ([], ctypes.c_char_p),
    "hs_last_error": ([], ctypes.c_char_p),
    "hs_status_string": ([c_int], ctypes.c_char_p),
    "hs_device_count": ([], c_int),
    "hs_get_reserved_cus": ([], c_int),
    "hs_get_seed_epoch": ([], c_ptr),
    "hs_reduce_pending": ([c_ptr], c_int),
    "hs_layernorm_bwd_workspace": ([c_i64, c_int], c_i64),
    "hs_seg_ce_partials": ([c_i64, c_i64], c_i64),
    "hs_ln_head_partials": ([c_i64], c_i64),
    "hs_expand_ln_head_blocks": ([c_i64], c_i64),
    "hs_linear_wgrad_workspace": ([c_i64, c_int, c_int], c_i64),
    "hs_window_attn_bwd_workspace": ([c_int, c_i64, c_int, c_int, c_int, c_int], c_i64),
    "hs_patch_merge_bwd_workspace": ([c_i64, c_int, c_int], c_i64),
    "hs_window_attn_module_bwd_chain_workspace": ([c_int, c_i64, c_int, c_int, c_int], c_i64),
    "hs_patch_expand_bwd_workspace": ([c_i64, c_int, c_int, c_int], c_i64),
}
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + list(_OTHER))


def _load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libhealswin.so not found at {LIB_PATH}: build it first (python heal_swin_amd/build.py, or "
            "__graft_entry__.build()).  heal_swin_amd has no CPU or PyTorch fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    for name, (argtypes, restype) in _OTHER.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    return lib


lib = _load()


def check(status, what):
    if status != 0:
        msg = lib.hs_last_error().decode()
        kind = lib.hs_status_string(status).decode()
        if status == 1:  # HS_ERR_INVALID_ARG mirrors the reference's bare asserts
            raise AssertionError(f"{what}: {msg}")
        raise RuntimeError(f"{what} failed ({kind}): {msg}")


def version():
    return lib.hs_version().decode()


def device_count():
    return int(lib.hs_device_count())


def np_ptr(a):
    return None if a is None else a.ctypes.data_as(c_ptr)


def ptr(t):
    """Raw device/host pointer of a torch tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr(device):
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dtype_code(dtype):
    import torch

    if dtype == torch.float32:
        return HS_F32
    if dtype == torch.bfloat16:
        return HS_BF16
    raise TypeError(f"heal_swin_amd kernels support float32 and bfloat16 activations, got {dtype}")


# ------------------------------------------------------------------ host tables (numpy in / out)
def nest2ring(nside, ipix):
    a = np.ascontiguousarray(ipix, dtype=np.int64)
    out = np.empty_like(a)
    check(lib.hs_nest2ring(int(nside), np_ptr(a), np_ptr(out), a.size), "hs_nest2ring")
    return out


def ring2nest(nside, ipix):
    a = np.ascontiguousarray(ipix, dtype=np.int64)
    out = np.empty_like(a)
    check(lib.hs_ring2nest(int(nside), np_ptr(a), np_ptr(out), a.size), "hs_ring2nest")
    return out


def nest_win_idcs(window_size):
    side = int(round(window_size ** 0.5))
    out = np.empty((max(side, 1), max(side, 1)), dtype=np.int64)
    check(lib.hs_nest_win_idcs(int(window_size), np_ptr(out)), "hs_nest_win_idcs")
    return out


def rel_pos_index(window_size):
    out = np.empty((window_size, window_size), dtype=np.int64)
    check(lib.hs_rel_pos_index(int(window_size), np_ptr(out)), "hs_rel_pos_index")
    return out


def _shift_out(n):
    return np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.uint8)


def build_nest_roll_shift(n_pix, window_size, shift_size):
    idx, inv, lab = _shift_out(n_pix)
    check(lib.hs_build_nest_roll_shift(int(n_pix), int(window_size), int(shift_size), np_ptr(idx), np_ptr(inv), np_ptr(lab)),
          "hs_build_nest_roll_shift")
    return idx, inv, lab


def build_nest_grid_shift(nside, base_pix, window_size):
    idx, inv, lab = _shift_out(max(base_pix, 0) * nside * nside)
    check(lib.hs_build_nest_grid_shift(int(nside), int(base_pix), int(window_size), np_ptr(idx), np_ptr(inv), np_ptr(lab)),
          "hs_build_nest_grid_shift")
    return idx, inv, lab


def build_ring_shift(nside, base_pix, window_size, shift_size):
    idx, inv, lab = _shift_out(max(base_pix, 0) * nside * nside)
    check(lib.hs_build_ring_shift(int(nside), int(base_pix), int(window_size), int(shift_size), np_ptr(idx), np_ptr(inv),
                                  np_ptr(lab)), "hs_build_ring_shift")
    return idx, inv, lab


def attn_mask_from_labels(labels, window_size):
    lab = np.ascontiguousarray(labels, dtype=np.uint8)
    out = np.empty((lab.size // window_size, window_size, window_size), dtype=np.float32)
    check(lib.hs_attn_mask_from_labels(np_ptr(lab), lab.size, int(window_size), np_ptr(out)), "hs_attn_mask_from_labels")
    return out
