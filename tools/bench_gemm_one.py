#!/usr/bin/env python3
"""Run hs_gemm_nt on one shape / tile variant / epilogue (for rocprofv3 counter passes).
usage: bench_gemm_one.py M N K [variant] [epilogue] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_swin_amd._lib import check, lib, ptr  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
epi = int(sys.argv[5]) if len(sys.argv) > 5 else 0
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
bias = torch.randn(N, device="cuda")
c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
aux = torch.randn(M, N, device="cuda").to(torch.bfloat16)
lib.hs_gemm_nt_set_tile(variant)
for _ in range(iters):
    check(lib.hs_gemm_nt(ptr(a), K, ptr(w), K, K, None, 0, None, 0, 0, ptr(bias), ptr(c), ptr(aux), M, N, epi, 0.0, 0, 1, None), "gemm")
torch.cuda.synchronize()
