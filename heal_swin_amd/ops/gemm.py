"""Linear layers: the own bf16 GEMM with fused epilogues (hs_gemm_nt) and its policy against the library GEMMs, the bf16 x 3 products
of fp32 activations, weight / bias gradients (hs_linear_wgrad), `LinearFn` (nn.Linear forward + backward)."""
import math
import os

import torch

from .. import _lib
from .._lib import check, lib, ptr, stream_ptr
from .runtime import RT, _cast_param, _defer_flag, _defer_keep, _require_gpu, _sink_buffer, _timed  # noqa: F401


# ----------------------------------------------------------------------------- hs_gemm_nt (own bf16 GEMM with fused epilogues)
OWN_GEMM = os.environ.get("HS_OWN_GEMM", "auto")  # "auto": per-shape choice below; "0": library GEMMs only; "1": own kernel wherever legal


# Set by parallel.GradBucketAllReduce while compute units are reserved for a co-resident gradient exchange (world > 1): every
# bf16 Linear product then runs on hs_gemm_nt, whose persistent grids honour hs_set_reserved_cus.  The library GEMMs fill all
# 256 CUs and cannot be masked: with 8 foreign workgroups resident they lose 64 % (256 -> 420 us, profiles/archive_r01_r04/r03_cu_contention.json).
# Costs ~3 ms per step on an idle chip (HS_OWN_GEMM=1 measurement of round 3), saves ~27 ms under contention (r04_cu_contention.json).
OWN_GELU_MAX_K = 4096
OWN_DGELU_MAX_K = 1024
OWN_BIAS_MAX_K = 0  # (> 0 would send every bias / residual product with k <= this to hs_gemm_nt: measured, slower -- profiles/archive_r01_r04/r03_gemm_policy_ab.txt)


class GemmTuner:
    """First-call micro-tuner of the own-kernel-or-library question for the bias / residual products (`own_gemm_ok`): the class rule
    and the table above were measured at the bench's shapes (batch 8, nside 256); any other batch / nside / width meets shapes
    nobody measured.  The first time a product of a new (rows bucket, n, k) class is asked for, both implementations run on
    synthetic operands of that shape (two operand sets in rotation, 1 + 3 launches each, interleaved, best time counts) and the
    faster one is remembered for the process; ~1-2 ms per new shape, during the first (warm-up) step.  Never inside a stream capture
    (the class rule answers there) and only for products large enough for the choice to matter.  It replaces the per-shape table of
    rounds 4-5 (four (n, k) pairs measured by hand in situ): on the bench's shapes it takes the same decisions and the step is the same
    to +-0.5 ms (profiles/r06_gemm_tuner_ab.txt).  `GEMM_TUNE = False` (an attribute, for tests and A/B runs): class rule only."""
    MIN_FLOP = 1 << 33  # below ~8 GFLOP a product is a few microseconds either way

    def __init__(self):
        self.picks = {}   # (rows bucket, n, k) -> True (hs_gemm_nt) / False (library)
        self.trials = {}  # the same key -> (own us, library us)

    @staticmethod
    def key(m, n, k):
        return (max(1, int(m)).bit_length(), int(n), int(k))  # rows in powers of two: one entry serves neighbouring batch sizes

    def pick(self, m, n, k, device):
        key = self.key(m, n, k)
        if key in self.picks:
            return self.picks[key]  # (None: a trial that failed -- the class rule answers from then on)
        if 2 * m * n * k < self.MIN_FLOP or not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
            return None
        try:
            return self._trial(key, m, n, k, device)
        except Exception:  # noqa: BLE001  (out of memory beside a large model, a library error: never the caller's problem)
            self.picks[key] = None
            return None

    def _trial(self, key, m, n, k, device):
        with torch.no_grad():
            g = torch.Generator(device=device).manual_seed(1)
            sets = [(torch.randn((m, k), generator=g, device=device).to(torch.bfloat16),
                     (torch.randn((n, k), generator=g, device=device) * k ** -0.5).to(torch.bfloat16)) for _ in range(2)]
            bias32 = torch.zeros(n, device=device)
            bias16 = bias32.to(torch.bfloat16)
            best = {True: float("inf"), False: float("inf")}
            for rep in range(4):  # (the first round is the warm-up: kernel attributes, library heuristics / TunableOp lookups)
                for own in (True, False):
                    a, w = sets[rep & 1]
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    if own:
                        gemm_nt(a, w, bias32)
                    else:
                        torch.nn.functional.linear(a, w, bias16)
                    e1.record()
                    e1.synchronize()
                    if rep:
                        best[own] = min(best[own], e0.elapsed_time(e1))
            del sets
        self.trials[key] = (round(1e3 * best[True], 1), round(1e3 * best[False], 1))
        self.picks[key] = best[True] <= best[False]
        return self.picks[key]


GEMM_TUNE = True
GEMM_TUNER = GemmTuner()


def own_gemm_ok(epi, n, k, dtype, k2=0, m=None):
    """Whether `hs_gemm_nt` should run this product (else the library GEMM + the standalone elementwise kernel).
    Measured on MI355X against hipBLASLt on the B / nside 256 / batch 8 shapes (tools/bench_gemm_nt.py,
    profiles/archive_r01_r04/r02_gemm_nt_vs_library.*): the own kernel wins where the product is HBM-bound (short reductions, narrow outputs:
    stages 0-1), ties the untuned hipBLASLt on the K = 512 shapes (and loses to the TunableOp-selected solutions bench.py
    loads) and loses the long reductions (K >= 1024: 0.96-1.06 vs 1.26 PFLOP/s).  A GELU forward epilogue pays while the
    product is HBM-bound (it has to write h AND gelu(h): at K = 512 the 256x256 tile needs 355-368 us against 197 us tuned
    library GEMM + 141 us standalone GELU pass); the GELU-gradient epilogue (reads h, writes once) wins at every stage
    (K = 1024: 265 us against 175-188 us library GEMM + 105 us GELU' pass)."""
    if dtype != torch.bfloat16 or OWN_GEMM == "0" or k % 8 or k2 % 8 or n % 8 or n < 16:
        return False  # (n % 8: whole-row-segment stores; the model pads the 12-class head to 16 rows)
    if OWN_GEMM == "1" or RT.prefer_own_gemm:
        return True
    kk = k + k2
    if epi == _lib.HS_EPI_DGELU:
        return kk <= OWN_DGELU_MAX_K
    if epi == _lib.HS_EPI_GELU:
        return kk <= OWN_GELU_MAX_K
    if OWN_BIAS_MAX_K > 0:
        return kk <= OWN_BIAS_MAX_K
    if GEMM_TUNE and m is not None and k2 == 0 and torch.cuda.is_available():
        pick = GEMM_TUNER.pick(int(m), n, kk, torch.device("cuda", torch.cuda.current_device()))
        if pick is not None:
            return pick
    return kk <= 128 or n <= 128 or (n <= 256 and kk <= 256)


RESID_DGRAD_OWN = True  # (T@256 paper config, same box: 46.2 vs 46.9 ms per step)


def own_gemm_legal(n, k, dtype):
    """Whether `hs_gemm_nt` CAN run an [*, k] x [n, k]^T product (the policy question is own_gemm_ok)."""
    return dtype == torch.bfloat16 and OWN_GEMM != "0" and k % 8 == 0 and n % 8 == 0 and n >= 16


# Residual adds in the GEMM epilogue (v1 blocks without stochastic regularisers): x1 = x + proj(o) and x2 = x1 + fc2(act) leave the
# proj / fc2 product's epilogue (EPI_RESID: acc + bias + residual, ONE rounding), so the LayerNorm that follows is a plain
# LayerNorm (reads 1, writes 1) instead of the fused add + LayerNorm (reads 2, writes 2): 2 of 8 tensor-units per block.
RESID_EPILOGUE = True


def gemm_nt(a2d, w, bias=None, epi=0, aux=None, a2=None, w2=None, want_c=True, drop_p=0.0, seed=0):
    """c = epilogue(a2d @ w^T (+ a2 @ w2^T) + bias) through `hs_gemm_nt`; returns (c, aux).  a2d [m, k] bf16 (row stride free),
    w [n, k] bf16 (row stride free), bias fp32 [n] or None."""
    m, k = a2d.shape
    n = w.shape[0]
    assert a2d.stride(1) == 1 and w.stride(1) == 1 and a2d.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
    c = torch.empty((m, n), dtype=torch.bfloat16, device=a2d.device) if want_c else None
    if epi == _lib.HS_EPI_GELU:
        aux = torch.empty((m, n), dtype=torch.bfloat16, device=a2d.device)
    k2 = 0 if a2 is None else a2.shape[1]
    if bias is not None and bias.dtype != torch.float32:
        bias = bias.float()
    with _timed(f"hs_gemm_nt epi={epi} m={m} n={n} k={k + k2}", a2d.device, 2 * (m * (k + k2) + n * (k + k2) + m * n * (2 if (epi and want_c) else 1)), 2 * m * n * (k + k2)):
        check(lib.hs_gemm_nt(ptr(a2d), a2d.stride(0), ptr(w), w.stride(0), k, ptr(a2), 0 if a2 is None else a2.stride(0), ptr(w2),
                             0 if w2 is None else w2.stride(0), k2, ptr(bias), ptr(c), ptr(aux), m, n, epi, float(drop_p), int(seed),
                             _lib.HS_BF16, stream_ptr(a2d.device)), "hs_gemm_nt")
    return c, aux


def _lib_tag(kind, m, n, k):
    """Tag of a library-GEMM call in KERNEL_TIMINGS (bench.py --kernel-table): which shapes hipBLASLt still runs, and how fast."""
    return f"lib {kind} m={m} n={n} k={k}"


# fp32 activations (the reference's precision): "bf16x3" forms every Linear product as ONE bf16 GEMM of three-fold depth over
# hi / lo splits of both operands with fp32 accumulation (csrc/split3.hip: a_hi b_hi + a_hi b_lo + a_lo b_hi, ~1e-5 relative to the
# fp32 product, 3/16 of the fp32-MFMA time); "strict" keeps exact-fp32 GEMMs (library fp32 GEMM, v_mfma_f32_32x32x2_f32 weight
# gradients) -- the reference form, used by the finite-difference tests.
FP32_GEMM = os.environ.get("HS_FP32_GEMM", "bf16x3")
_BF16X3_MIN = 160  # n k / (n + k) from which a product takes the bf16x3 form (sweep: see _bf16x3_ok)
_SPLIT_MEMO = []  # the last few splits (key, tensor): dy is split once for the input- and the weight-gradient product
_MM_OUT_DTYPE = [None]  # whether torch.mm(..., out_dtype=) is available in this build (probed on first use)


def _bf16x3_ok(x, n=None, k=None):
    """bf16x3 for this fp32 product?  Only where the exact-fp32 product is MFMA-bound: n k / (n + k) >= 160 (e.g. 512 x 256; sweep 64 / 100 /
    128 / 192 / 300 on B@256 fp32: 635 / 625 / 607 / 606 / 632 ms).  Narrow products (C = 96 / 128: stage 0, the whole first stages of HEAL-SWIN-T) move 4 (n + k) bytes per row against
    2 n k flops at 110 TFLOP/s -- they are HBM-bound in fp32 already and the split passes would only add traffic (measured: the T
    depth-head companion 35.9 -> 33.2 images/s with bf16x3 everywhere)."""
    ok = FP32_GEMM == "bf16x3" and x.dtype == torch.float32 and x.is_cuda and x.shape[-1] % 8 == 0
    ok = ok and (n is None or n * k >= _BF16X3_MIN * (n + k))
    if ok and _MM_OUT_DTYPE[0] is None:
        _probe_mm_out_dtype(x.device)
    return ok and _MM_OUT_DTYPE[0] is not False


def _probe_mm_out_dtype(device):
    """bf16x3 needs `torch.mm(bf16, bf16, out_dtype=float32)`; a build without it runs the exact-fp32 products instead (one warning)."""
    try:
        a = torch.zeros((8, 8), dtype=torch.bfloat16, device=device)
        torch.mm(a, a, out_dtype=torch.float32)
        _MM_OUT_DTYPE[0] = True
    except Exception:  # noqa: BLE001  (a build without mm.dtype)
        _MM_OUT_DTYPE[0] = False
        import warnings
        warnings.warn("heal_swin_amd: torch.mm(..., out_dtype=) is unavailable in this PyTorch build; fp32 Linear products run as "
                      "exact fp32 GEMMs (HS_FP32_GEMM=strict behaviour) instead of bf16x3")


def split3(x2d, mode):
    """bf16 [rows, 3 k] = [hi | hi | lo] (mode 0) or [hi | lo | hi] (mode 1) of fp32 x2d [rows, k] (`hs_split_bf16x3`)."""
    x2d = x2d.contiguous()
    key = (x2d.data_ptr(), tuple(x2d.shape), x2d._version, mode)
    for kk, _, t in _SPLIT_MEMO:
        if kk == key:
            return t
    rows, k = x2d.shape
    out = torch.empty((rows, 3 * k), dtype=torch.bfloat16, device=x2d.device)
    check(lib.hs_split_bf16x3(ptr(x2d), ptr(out), rows, k, mode, stream_ptr(x2d.device)), "hs_split_bf16x3")
    if mode == 0:
        # (the entry keeps the SOURCE alive: its address cannot be recycled for another tensor while the key is in the memo)
        _SPLIT_MEMO.append((key, x2d, out))
        del _SPLIT_MEMO[:-2]
    return out


class _Split:
    """An fp32 [rows, k] operand that exists ONLY as its bf16x3 split t3 [rows, 3 k] (`hs_gelu_split3`): stands in for the tensor
    in _lib_linear / _lib_matmul / _param_grads (duck-typed: shape, dtype, device, is_contiguous, record_stream)."""

    def __init__(self, t3, k):
        self.t3, self.shape, self.dtype, self.device = t3, (t3.shape[0], k), torch.float32, t3.device

    def is_contiguous(self):
        return True

    def record_stream(self, stream):
        self.t3.record_stream(stream)

    def dim(self):
        return 2


def _split_of(x2d):
    """The [hi | hi | lo] split of x2d a forward product just made (still in the memo), or None: the Linear keeps it for its weight
    gradient instead of splitting the same activations again in the backward (a third of the fp32 step's split passes)."""
    if isinstance(x2d, _Split):
        return x2d.t3
    if x2d is None or x2d.dtype != torch.float32:
        return None
    key = (x2d.data_ptr(), tuple(x2d.shape), x2d._version, 0)
    for kk, _, t in _SPLIT_MEMO:
        if kk == key:
            return t
    return None


_WSPLIT = {}  # (storage address, storage offset, shape, transposed) -> (version, weight epoch, view of the source, [hi | lo | hi] split)
_WSPLIT_CAPACITY = 1024


def _weight_split(w2d, transposed):
    """The [hi | lo | hi] operand (mode 1) of the fp32 weight w2d [n, k] -- or of its transpose [k, n] -- for the bf16x3 products.
    Cached per weight: callers hand over fresh VIEWS of the parameter (`w.view(n, k)`), so an entry is identified by the storage
    it views (address + offset + shape; the entry keeps a view alive, so the address cannot be recycled while it is cached) and is
    valid while the parameter's `_version` (shared by all its views) AND `RT.weight_epoch` are unchanged.  The epoch moves with
    every grad-enabled model forward, because fused optimizers update parameters WITHOUT bumping `_version` (see ParamCastCache):
    in a training loop every weight is therefore split once per step and direction (forward, transposed for the input gradient);
    evaluation loops, gradient accumulation under no_grad re-forwards and activation checkpointing re-use the cached operand."""
    key = (w2d.untyped_storage().data_ptr(), w2d.storage_offset(), tuple(w2d.shape), tuple(w2d.stride()), bool(transposed))
    hit = _WSPLIT.get(key)
    if hit is not None and hit[0] == w2d._version and hit[1] == RT.weight_epoch:
        return hit[3]
    src = w2d.t().contiguous() if transposed else w2d.contiguous()
    rows, k = src.shape
    out = torch.empty((rows, 3 * k), dtype=torch.bfloat16, device=src.device)
    check(lib.hs_split_bf16x3(ptr(src), ptr(out), rows, k, 1, stream_ptr(src.device)), "hs_split_bf16x3")
    if len(_WSPLIT) >= _WSPLIT_CAPACITY:  # (models come and go in a test session: bounded, oldest entries first)
        for old in list(_WSPLIT)[:_WSPLIT_CAPACITY // 2]:
            del _WSPLIT[old]
    _WSPLIT[key] = (w2d._version, RT.weight_epoch, w2d.detach(), out)
    return out


def _mm_f32(a3, b3t, bias=None):
    """fp32 result of the bf16 product a3 @ b3t (+ bias): hipBLASLt with an fp32 output (`out_dtype`)."""
    if _MM_OUT_DTYPE[0] is None:
        _probe_mm_out_dtype(a3.device)
    if not _MM_OUT_DTYPE[0]:
        raise RuntimeError("HS_FP32_GEMM=bf16x3 needs torch.mm(..., out_dtype=torch.float32); set HS_FP32_GEMM=strict")
    if bias is None:
        return torch.mm(a3, b3t, out_dtype=torch.float32)
    if _MM_OUT_DTYPE[0] is True:  # addend (bias vector or residual matrix, fp32) in the GEMM's epilogue where the build has addmm.dtype
        try:
            y = torch.addmm(bias, a3, b3t, out_dtype=torch.float32)
            _MM_OUT_DTYPE[0] = "addmm"
            return y
        except Exception:  # noqa: BLE001
            _MM_OUT_DTYPE[0] = "mm"
    if _MM_OUT_DTYPE[0] == "addmm":
        return torch.addmm(bias, a3, b3t, out_dtype=torch.float32)
    return torch.mm(a3, b3t, out_dtype=torch.float32).add_(bias)


def _lib_linear(x2, w, b):
    if isinstance(x2, _Split):
        m, k = x2.shape
        with _timed(_lib_tag("fwd bf16x3", m, w.shape[0], 3 * k), x2.device, 4 * (m * k + m * w.shape[0]), 6 * m * k * w.shape[0]):
            return _mm_f32(x2.t3, _weight_split(w.reshape(w.shape[0], k), False).t(), b)
    m, k = x2.shape[0] if x2.dim() == 2 else x2.numel() // x2.shape[-1], x2.shape[-1]
    if _bf16x3_ok(x2, w.shape[0], k) and w.dtype == torch.float32 and w.shape[0] % 8 == 0:
        with _timed(_lib_tag("fwd bf16x3", m, w.shape[0], 3 * k), x2.device, 4 * (m * k + m * w.shape[0]), 6 * m * k * w.shape[0]):
            y = _mm_f32(split3(x2.reshape(m, k), 0), _weight_split(w.reshape(w.shape[0], k), False).t(), b)
        return y.view(x2.shape[:-1] + (w.shape[0],))
    with _timed(_lib_tag("fwd", m, w.shape[0], k), x2.device, 2 * (m * k + m * w.shape[0]), 2 * m * k * w.shape[0]):
        return torch.nn.functional.linear(x2, w, b)


def _lib_matmul(dy2, w, res=None):
    m, n = dy2.shape
    if isinstance(dy2, _Split):
        with _timed(_lib_tag("dgrad bf16x3", m, w.shape[1], 3 * n), dy2.device, 4 * (m * n + m * w.shape[1]), 6 * m * n * w.shape[1]):
            return _mm_f32(dy2.t3, _weight_split(w, True).t(), res)
    if _bf16x3_ok(dy2, w.shape[1], n) and w.dtype == torch.float32 and w.shape[1] % 8 == 0:
        with _timed(_lib_tag("dgrad bf16x3", m, w.shape[1], 3 * n), dy2.device, 4 * (m * n + m * w.shape[1]), 6 * m * n * w.shape[1]):
            dx = _mm_f32(split3(dy2, 0), _weight_split(w, True).t(), res)
        return dx
    with _timed(_lib_tag("dgrad", m, w.shape[1], n), dy2.device, 2 * (m * n + m * w.shape[1]), 2 * m * n * w.shape[1]):
        return dy2 @ w if res is None else torch.addmm(res, dy2, w)


def _cast_param_t(p, dtype, cache=None):
    """[in, out] copy of weight p ([out, in, ...]) in `dtype` for the input-gradient product.  `cache`: the ParamCastCache the
    FORWARD of this autograd node ran under (kept on its ctx, so that several models in one process each take their own
    copies); falls back to the cache of the most recent forward."""
    if cache is None:
        cache = RT.cast_cache if RT.cast_cache is not None else RT.last_cast_cache
    c = cache.get_t(p, dtype) if (cache is not None and p.dim() == 2 and cache.current(p)) else None
    if c is None:
        n_out = p.shape[0]
        c = p.detach().to(dtype).view(n_out, -1).t().contiguous()
    return c


def _param_grads(dy2, x2, weight, bias, want_w, want_b, x3=None, gelu_x=False):
    """Weight / bias gradient of y = x W^T + b from dy2 [rows, n_out], x2 [rows, k_in]: deposited straight into the gradient
    sink's buffers when one knows the parameters (returns (None, None)), else returned in the parameters' dtype.
    gelu_x: the Linear's input was gelu(x2) and only the pre-activation x2 was kept (`hs_linear_wgrad_gelu`)."""
    n_out = weight.shape[0]
    k_in = weight.numel() // n_out
    if not (want_w or want_b):
        return None, None
    if x2 is None:  # (fp32 runs) only the bf16x3 split of the input was kept: the three-product weight gradient reads nothing else
        assert x3 is not None and dy2.dtype == torch.float32
        hip_ok = dy2.is_contiguous() and n_out % 8 == 0
        assert hip_ok
    else:
        hip_ok = (x2.is_contiguous() and dy2.is_contiguous() and n_out % 4 == 0 and
                  ((x2.dtype == torch.bfloat16 and k_in % 8 == 0) or (x2.dtype == torch.float32 and k_in % 4 == 0)))
    wbuf = _sink_buffer(weight) if (hip_ok and want_w) else None
    bbuf = _sink_buffer(bias) if (wbuf is not None and want_b) else None
    if wbuf is not None and (not want_b or bbuf is not None):
        # accumulate dW (and db) straight into the sink's gradient buffers (no autograd AccumulateGrad kernels, no dtype
        # round trip); optionally on the side stream
        aw = RT.async_wgrad
        wbuf = wbuf.view(n_out, k_in)
        if aw is not None:
            cur = torch.cuda.current_stream(dy2.device)
            aw.stream.wait_stream(cur)
            dy2.record_stream(aw.stream)
            (x2 if x2 is not None else x3).record_stream(aw.stream)
            with torch.cuda.stream(aw.stream):
                LinearFn._wgrad_hip(dy2, x2, n_out, k_in, want_b, wbuf, bbuf, x3, gelu_x)
        else:
            LinearFn._wgrad_hip(dy2, x2, n_out, k_in, want_b, wbuf, bbuf, x3, gelu_x)
        RT.grad_sink.deposited(weight)
        if want_b:
            RT.grad_sink.deposited(bias)
        return None, None
    dw = db = None
    if hip_ok:
        dw32, db32 = LinearFn._wgrad_hip(dy2, x2, n_out, k_in, want_b, x3=x3, gelu_x=gelu_x)
        dw = dw32.to(weight.dtype).view(weight.shape) if want_w else None
        db = db32.to(bias.dtype) if want_b else None
    else:  # odd widths: library GEMM
        assert not gelu_x
        if want_w:
            dw = (dy2.t() @ x2).to(weight.dtype).view(weight.shape)
        if want_b:
            db = dy2.sum(0).to(bias.dtype)
    return dw, db


class LinearFn(torch.autograd.Function):
    """y = x W^T + b with fp32 master parameters and activations in x.dtype.
    forward / input gradient: library GEMM; weight + bias gradient: `hs_linear_wgrad` (split over the token axis, fp32
    results straight into the master dtype).  `weight` may carry trailing singleton dimensions (the decoder's 1x1 Conv1d head,
    [f_out, C, 1]): it is used as the [n_out, k_in] matrix it is, so the PARAMETER itself (a leaf) receives the gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias, passthrough=False, residual=None, pre=None):
        """pre = (y,): the product the fused module kernel already wrote (window_attn_module_train); nothing is launched."""
        _require_gpu(x, weight, bias, residual)
        n_out = weight.shape[0]
        k_in = weight.numel() // n_out
        w = _cast_param(weight, x.dtype).view(n_out, k_in)
        ctx.x_shape = x.shape
        ctx.bias_param = bias
        ctx.w_cast = w if w.dtype != weight.dtype else None  # activation-dtype copy, reused by the input-gradient GEMM
        ctx.cast_cache = RT.cast_cache
        ctx.passthrough = passthrough
        ctx.has_residual = residual is not None
        ctx.x3 = None
        if pre is not None:
            y = pre[0]
        elif residual is not None:
            # y = x W^T + b + residual: the add rides on the product's epilogue (one rounding); its gradient is dy itself
            assert not passthrough
            if own_gemm_legal(n_out, k_in, x.dtype) and x.is_contiguous():
                res2 = residual.reshape(-1, n_out)
                res2 = res2 if res2.is_contiguous() else res2.contiguous()
                y = gemm_nt(x.reshape(-1, k_in), w, bias, _lib.HS_EPI_RESID, aux=res2)[0].view(x.shape[:-1] + (n_out,))
            else:
                y = _lib_linear(x, w, None if bias is None else _cast_param(bias, x.dtype)) + residual
                ctx.x3 = _split_of(x.reshape(-1, k_in)) if x.is_contiguous() else None
        elif own_gemm_ok(_lib.HS_EPI_BIAS, n_out, k_in, x.dtype, m=x.numel() // k_in) and x.is_contiguous():
            y = gemm_nt(x.reshape(-1, k_in), w, bias)[0].view(x.shape[:-1] + (n_out,))  # fp32 master bias added in the epilogue
        else:
            y = _lib_linear(x, w, None if bias is None else _cast_param(bias, x.dtype))
            ctx.x3 = _split_of(x.reshape(-1, k_in)) if x.is_contiguous() else None
        # (fp32 runs) where the forward product made a bf16x3 split of x, the weight gradient reads THAT (6 bytes per element) and x
        # itself (4) is not kept for it
        ctx.save_for_backward(None if ctx.x3 is not None else x, weight)
        # passthrough: also hand x back (an alias) for the block's residual connection.  The gradient of that second use then
        # arrives HERE together with dy, and the input-gradient GEMM adds it as its beta * C term instead of autograd
        # launching a separate add over the whole activation (v2 norm placement: x + LN(branch(x)), ref :334-335)
        return (y, x.view_as(x)) if passthrough else y

    @staticmethod
    def _wgrad_hip(dy2, x2, n_out, k_in, want_b, dw_out=None, db_out=None, x3=None, gelu_x=False):
        """dW (and db) of one Linear.  With dw_out/db_out (existing fp32 gradient buffers) the result is ADDED there."""
        rows = dy2.shape[0]
        dev = dy2.device
        accumulate = 1 if dw_out is not None else 0
        # deposits into the gradient sink's buffers queue their slice sums (one launch per ~32 layers, ops.flush_reductions)
        defer = _defer_flag(dev) if (dw_out is not None and (db_out is not None or not want_b)) else 0
        dw32 = dw_out if dw_out is not None else torch.empty((n_out, k_in), dtype=torch.float32, device=dev)
        db32 = None
        if want_b:
            db32 = db_out if db_out is not None else torch.empty(n_out, dtype=torch.float32, device=dev)
        nws = int(lib.hs_linear_wgrad_workspace(rows, n_out, k_in))
        ws = torch.empty(nws, dtype=torch.float32, device=dev)
        if x3 is not None or (_bf16x3_ok(x2, n_out, k_in) and n_out % 8 == 0 and x2.dtype == dy2.dtype):
            # dW = dY^T X as three bf16 weight-gradient products over the hi / lo column blocks of the [hi | hi | lo] splits
            # (the split of dY is shared with the input-gradient product): hi^T hi + hi^T lo + lo^T hi; the bias gradient takes
            # the column sums of dY_hi and dY_lo
            dy3 = dy2.t3 if isinstance(dy2, _Split) else split3(dy2, 0)
            x3 = x3 if x3 is not None else split3(x2, 0)
            aw = RT.async_wgrad
            if aw is not None and torch.cuda.current_stream(dev) == aw.stream:
                # the splits were allocated on the main stream (memo / forward) and are read here on the side stream: tell the
                # caching allocator, or a block evicted from the memo could be recycled under the lagging weight-gradient kernels
                dy3.record_stream(aw.stream)
                x3.record_stream(aw.stream)
            with _timed("linear_wgrad bf16x3", dev, 3 * 2 * rows * (n_out + k_in), 6 * rows * n_out * k_in):
                for i, (yo, xo, dbp) in enumerate(((0, 0, db32), (0, 2 * k_in, None), (2 * n_out, 0, db32))):
                    # the three sums share dw32 (two of them db32): jobs of one hs_reduce_flush launch run side by side and would
                    # race on it, so only the LAST product's sum is queued -- the first two land at once, in stream order, and
                    # the workspace is free again when the next product writes it
                    d = defer if i == 2 else 0
                    check(lib.hs_linear_wgrad_ld(ptr(dy3), 3 * n_out, yo, ptr(x3), 3 * k_in, xo, ptr(dw32), ptr(dbp), ptr(ws), rows,
                                                 n_out, k_in, (1 if (accumulate or i) else 0) | d, stream_ptr(dev)), "hs_linear_wgrad_ld")
                if defer:
                    _defer_keep(dev, ws)
            return dw32, db32
        with _timed("linear_wgrad", dev, x2.element_size() * rows * (n_out + k_in), 2 * rows * n_out * k_in):
            if gelu_x:  # dW = dY^T gelu(x2): the activation is applied to the operand fragments inside the kernel
                check(lib.hs_linear_wgrad_gelu(ptr(dy2), ptr(x2), ptr(dw32), ptr(db32), ptr(ws), rows, n_out, k_in, accumulate | defer,
                                               _lib.dtype_code(x2.dtype), stream_ptr(dev)), "hs_linear_wgrad_gelu")
            else:
                check(lib.hs_linear_wgrad(ptr(dy2), ptr(x2), ptr(dw32), ptr(db32), ptr(ws), rows, n_out, k_in, accumulate | defer,
                                          _lib.dtype_code(x2.dtype), stream_ptr(dev)), "hs_linear_wgrad")
        if defer:
            _defer_keep(dev, ws)
        return dw32, db32

    @staticmethod
    def backward(ctx, dy, dx_res=None):
        x, weight = ctx.saved_tensors
        bias = ctx.bias_param
        n_out = weight.shape[0]
        k_in = weight.numel() // n_out
        if dy is None:  # only the passthrough alias was used downstream
            return dx_res, None, None, None, None, None
        dy2 = dy.reshape(-1, n_out)
        x2 = None if x is None else x.reshape(-1, k_in)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _input_grad(dy2, weight, ctx.w_cast, None if dx_res is None else dx_res.reshape(-1, k_in), ctx.cast_cache).reshape(ctx.x_shape)
        ctx.w_cast = ctx.cast_cache = None
        x3, ctx.x3 = ctx.x3, None
        dw, db = _param_grads(dy2, x2, weight, bias, ctx.needs_input_grad[1], bias is not None and ctx.needs_input_grad[2], x3)
        return dx, dw, db, None, (dy if ctx.has_residual else None), None


def _input_grad(dy2, weight, w_cast, dx_res2=None, cache=None):
    """dx = dy2 @ W (+ dx_res2): `hs_gemm_nt` on the transposed weight copy where that wins, else the library GEMM."""
    n_out = weight.shape[0]
    k_in = weight.numel() // n_out
    epi = _lib.HS_EPI_BIAS if dx_res2 is None else _lib.HS_EPI_RESID
    # with a residual-path gradient to add (v2 placement), the library form is torch.addmm(res, dy, W): a device-to-device copy of
    # `res` into the result and THEN the product with beta = 1 -- a whole extra pass (36 copies, ~1 ms per HEAL-SWIN-T @ 256 step);
    # hs_gemm_nt reads the addend in its epilogue instead
    if own_gemm_ok(epi, k_in, n_out, dy2.dtype, m=dy2.shape[0]) or (dx_res2 is not None and RESID_DGRAD_OWN and own_gemm_legal(k_in, n_out, dy2.dtype)):
        res = None if dx_res2 is None else dx_res2.to(dy2.dtype).contiguous()
        return gemm_nt(dy2, _cast_param_t(weight, dy2.dtype, cache), None, epi, aux=res)[0]
    w = w_cast if (w_cast is not None and w_cast.dtype == dy2.dtype) else (
        weight if weight.dtype == dy2.dtype else weight.to(dy2.dtype)).view(n_out, k_in)
    if dx_res2 is not None:
        return _lib_matmul(dy2, w, dx_res2.to(dy2.dtype))
    return _lib_matmul(dy2, w)


def linear(x, weight, bias=None):
    return LinearFn.apply(x, weight, bias)


def linear_residual(x, weight, bias, residual):
    """x W^T + b + residual with the add in the product's epilogue (LinearFn)."""
    return LinearFn.apply(x, weight, bias, False, residual)


def linear_passthrough(x, weight, bias=None):
    """(x W^T + b, alias of x): use the alias for a residual connection around the branch this Linear opens."""
    return LinearFn.apply(x, weight, bias, True)


# data_ptr -> zero-padded gradient buffer written by losses.seg_loss' backward (see PadSliceFn).  WEAK values: an entry exists only
# while the buffer itself is alive (i.e. while autograd still holds the gradient view into it), so nothing is retained when no
# PadSliceFn consumes it, and a recycled address cannot resurrect a dead buffer.


class PadSliceFn(torch.autograd.Function):
    """x[..., :n] of the padded head output.  Backward: when the incoming gradient is the [..., :n] view of a zero-padded
    buffer of x's shape (losses.seg_loss writes its gradient that way), that buffer IS the gradient of x; otherwise the
    gradient is copied into a zeroed buffer, as autograd's slice backward does."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.shape, ctx.n = x.shape, n
        return x[..., :n]

    @staticmethod
    def backward(ctx, g):
        full = RT.zero_padded_grads.pop(g.data_ptr(), None)
        if (full is not None and g._base is full and full.numel() == math.prod(ctx.shape) and full.dtype == g.dtype and
                g.shape == ctx.shape[:-1] + (ctx.n,) and g.stride() == full.view(ctx.shape)[..., :ctx.n].stride()):
            return full.view(ctx.shape), None
        out = g.new_zeros(ctx.shape)
        out[..., :ctx.n] = g
        return out, None


def pad_slice(x, n):
    return PadSliceFn.apply(x, n)
