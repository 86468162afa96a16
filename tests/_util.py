"""Shared helpers for the GPU parity tests.

Error measures (all against the reference / oracle tensor b):
  scale_err  max|a-b| / max|b|          -- absolute error relative to the tensor's own scale (NO floor at 1: a tensor of
                                            scale 1e-4, e.g. a parameter gradient, is judged at 1e-4)
  rms_err    ||a-b||_2 / ||b||_2        -- average relative error
  elem_err   99.9th percentile of |a-b| / max(|b|, 1e-2 max|b|)   -- element-relative error, elements below 1 % of the scale
                                            judged against 1 % of the scale (relative error is meaningless at zero crossings)
north_star tolerances: activations within 1e-3 (fp32) / 1e-2 (bf16) of the reference; asserted on scale_err.
"""
import numpy as np
import torch

TOL = {torch.float32: 1e-3, torch.bfloat16: 1e-2}
# gradients of bf16 runs accumulate more rounding (two passes through every bf16 activation)
GRAD_TOL = {torch.float32: 1e-3, torch.bfloat16: 3e-2}

REPORT = []  # (what, scale_err, rms_err, elem_err, scale) of every comparison made; printed at the end of the session
SLOPES = []  # (what, slope - 1, cosine, n) of every assert_unbiased call


def _np(a):
    return np.asarray(a.detach().float().cpu().numpy() if torch.is_tensor(a) else a, dtype=np.float64)


def _errors_gpu(a, b):
    """The same measures computed on the GPU in float64 (large tensors: the numpy path converts both operands to float64 on the
    host and sorts for the quantile -- seconds per comparison at the full-size shapes, minutes on a slow host)."""
    dev = a.device if (torch.is_tensor(a) and a.is_cuda) else b.device
    a = (a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))).detach().to(dev).to(torch.float64)
    b = (b if torch.is_tensor(b) else torch.from_numpy(np.ascontiguousarray(b))).detach().to(dev).to(torch.float64)
    assert a.shape == b.shape, (tuple(a.shape), tuple(b.shape))
    d = (a - b).abs()
    scale = float(b.abs().max()) if b.numel() else 0.0
    if scale == 0.0:
        m = float(d.max()) if d.numel() else 0.0
        return dict(scale=0.0, abs=m, scale_err=m, rms_err=0.0, elem_err=0.0)
    rms = float(torch.sqrt((d * d).sum()) / torch.sqrt((b * b).sum()).clamp_min(1e-300))
    el = (d / torch.maximum(b.abs(), torch.full((), 1e-2 * scale, dtype=torch.float64, device=dev))).flatten()
    if el.numel() > (1 << 22):  # the percentile is a diagnostic: a strided sample of <= 4 M elements
        el = el[:: (el.numel() + (1 << 22) - 1) // (1 << 22)]
    elem = float(torch.quantile(el, 0.999)) if el.numel() > 1000 else float(el.max())
    dmax = float(d.max())
    return dict(scale=scale, abs=dmax, scale_err=dmax / scale, rms_err=rms, elem_err=elem)


def errors(a, b):
    if (torch.is_tensor(a) and a.is_cuda) or (torch.is_tensor(b) and b.is_cuda):
        return _errors_gpu(a, b)
    a, b = _np(a), _np(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    d = np.abs(a - b)
    scale = float(np.abs(b).max()) if b.size else 0.0
    if scale == 0.0:
        return dict(scale=0.0, abs=float(d.max()) if d.size else 0.0, scale_err=float(d.max()) if d.size else 0.0, rms_err=0.0, elem_err=0.0)
    rms = float(np.sqrt((d * d).sum()) / max(np.sqrt((b * b).sum()), 1e-300))
    el = d / np.maximum(np.abs(b), 1e-2 * scale)
    return dict(scale=scale, abs=float(d.max()), scale_err=float(d.max()) / scale, rms_err=rms,
                elem_err=float(np.quantile(el, 0.999)) if el.size > 1000 else float(el.max()))


def rel_err(a, b):
    """max|a-b| / max|b| (an all-zero reference is compared absolutely)."""
    return errors(a, b)["scale_err"]


def assert_close(a, b, tol, what="", floor=0.0):
    """scale_err <= tol.  `floor`: absolute error always accepted (for tensors that are exactly or nearly zero in the
    reference, e.g. the gradient of a bias behind a LayerNorm)."""
    e = errors(a, b)
    REPORT.append((what, e["scale_err"], e["rms_err"], e["elem_err"], e["scale"]))
    assert e["scale_err"] <= tol or e["abs"] <= floor, (
        f"{what}: max|a-b|/max|b| = {e['scale_err']:.3e} > {tol} (scale {e['scale']:.3e}, abs {e['abs']:.3e}, "
        f"rms {e['rms_err']:.3e}, elem99.9 {e['elem_err']:.3e})")
    return e["scale_err"]


def worst(prefix=""):
    """(worst scale_err, worst rms_err, worst elem_err) over the comparisons whose label starts with prefix."""
    rows = [r for r in REPORT if r[0].startswith(prefix)]
    if not rows:
        return 0.0, 0.0, 0.0
    return max(r[1] for r in rows), max(r[2] for r in rows), max(r[3] for r in rows)


def assert_unbiased(a, b, what="", slope_tol=0.05, cos_min=None, min_elems=256):
    """A SYSTEMATIC error shows in the least-squares slope of a against the reference b,  s = <a, b> / <b, b>,  however noisy the
    elements are: zero-mean rounding noise of relative rms r moves s by ~ r / sqrt(n), a dropped product / a wrong factor / a missed
    accumulation of 10 % moves it by 0.1.  This is what lets the bf16 GRADIENT tests -- whose element bounds must admit the
    rounding noise of cancelling sums (up to 0.2-0.5 of a tensor's scale on its worst element) -- still catch a 10 % error:
    |s - 1| <= slope_tol on every tensor with >= min_elems elements (smaller ones are left to their element bound).
    cos_min: optional bound on the cosine between a and b."""
    if torch.is_tensor(a):
        a = a.detach().to(torch.float64).flatten()
        b = (b if torch.is_tensor(b) else torch.from_numpy(np.ascontiguousarray(b))).detach().to(a.device).to(torch.float64).flatten()
        bb, ab, aa = float(b @ b), float(a @ b), float(a @ a)
    else:
        a, b = _np(a).ravel(), _np(b).ravel()
        bb, ab, aa = float(b @ b), float(a @ b), float(a @ a)
    n = int(b.shape[0])
    if n < min_elems or bb == 0.0:
        return None
    slope = ab / bb
    cos = ab / max((aa * bb) ** 0.5, 1e-300)
    SLOPES.append((what, slope - 1.0, cos, n))
    assert abs(slope - 1.0) <= slope_tol, f"{what}: slope of the result against the reference = {slope:.4f} (|s - 1| > {slope_tol}; cosine {cos:.4f}, {n} elements)"
    if cos_min is not None:
        assert cos >= cos_min, f"{what}: cosine with the reference {cos:.4f} < {cos_min}"
    return slope
