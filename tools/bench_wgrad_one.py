#!/usr/bin/env python3
"""Run hs_linear_wgrad on one shape (for rocprofv3 counter passes).  usage: bench_wgrad_one.py M N K [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_swin_amd._lib import check, lib, ptr  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
dw = torch.empty(N, K, device="cuda")
db = torch.empty(N, device="cuda")
ws = torch.empty(int(lib.hs_linear_wgrad_workspace(M, N, K)), device="cuda")
for _ in range(iters):
    check(lib.hs_linear_wgrad(ptr(dy), ptr(x), ptr(dw), ptr(db), ptr(ws), M, N, K, 0, 1, None), "wgrad")
torch.cuda.synchronize()
