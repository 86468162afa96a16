#!/usr/bin/env python3
"""Timing of hs_gemm_nt's 256x256 tile in the HS_GEMM_EXP measurement builds (see csrc/gemm_nt.hip): does the main loop keep its
rate when (1) only half of the waves issue the operand DMA, (2) the other half stores an output tile's worth of bytes from inside
the k-steps, (3) both?  Run through tools/gemm_overlap_premise.sh, which swaps the library builds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_swin_amd import _lib, ops  # noqa: E402

lib, ptr, check = _lib.lib, ops.ptr, _lib.check
dev = torch.device("cuda")
tag = sys.argv[1] if len(sys.argv) > 1 else "?"


def med(fn, reps=20):
    ts = []
    for it in range(reps + 3):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if it >= 3:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for name, m, n, k in (("s2 fc1", 98304, 2048, 512), ("s2 fc2", 98304, 512, 2048), ("s3 fc1", 24576, 4096, 1024)):
    a = torch.randn(m, k, device=dev).bfloat16()
    w = (torch.randn(n, k, device=dev) * 0.02).bfloat16()
    b = torch.randn(n, device=dev)
    c = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    aux = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    st = ops.stream_ptr(dev)
    lib.hs_gemm_nt_set_tile(3)

    def run(epi):
        check(lib.hs_gemm_nt(ptr(a), a.stride(0), ptr(w), w.stride(0), k, None, 0, None, 0, 0, ptr(b), ptr(c), ptr(aux), m, n, epi,
                             0.0, 0, _lib.dtype_code(torch.bfloat16), st), "hs_gemm_nt")

    run(_lib.HS_EPI_BIAS)
    ref = torch.nn.functional.linear(a[:4096], w).float() + b
    err = float((c[:4096].float() - ref).abs().max() / ref.abs().max())
    t_b = med(lambda: run(_lib.HS_EPI_BIAS))
    t_g = med(lambda: run(_lib.HS_EPI_GELU))
    fl = 2.0 * m * n * k
    print(f"{tag:34s} {name} {m}x{n}x{k}: bias {t_b:7.1f} us {fl / t_b / 1e6:6.0f} TF/s | gelu {t_g:7.1f} us | max err of the bias result {err:.1e}")
lib.hs_gemm_nt_set_tile(0)
