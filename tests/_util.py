"""Shared helpers for the GPU parity tests."""
import numpy as np
import torch

# north_star tolerances: activations within 1e-3 (fp32) / 1e-2 (bf16) of the reference, measured
# relative to the tensor's scale: max|a-b| / max(1, max|b|)
TOL = {torch.float32: 1e-3, torch.bfloat16: 1e-2}
# gradients of bf16 runs accumulate more rounding (two passes through every bf16 activation)
GRAD_TOL = {torch.float32: 1e-3, torch.bfloat16: 3e-2}


def rel_err(a, b):
    a = np.asarray(a.detach().float().cpu().numpy() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b.detach().float().cpu().numpy() if torch.is_tensor(b) else b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max()))


def assert_close(a, b, tol, what=""):
    e = rel_err(a, b)
    assert e <= tol, f"{what}: rel-to-scale err {e:.3e} > {tol}"
    return e
