#!/usr/bin/env python3
"""Premise check for a 'chasing pass': can an HBM-bound elementwise kernel run CO-RESIDENT with the own GEMM (two streams, no
dependency between them) so that the pair costs about max(GEMM, pass) instead of their sum?  Stage-2 fc1 shape
(98 304 x 2048 x 512, bf16), hs_gemm_nt bias epilogue with the 256x128 (157 VGPRs) and 256x256 (245 VGPRs) tiles and the library
GEMM, beside hs_gelu_fwd on an unrelated 98 304 x 2048 tensor (42 VGPRs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_swin_amd import _lib, ops  # noqa: E402

lib, ptr, check = _lib.lib, ops.ptr, _lib.check
dev = torch.device("cuda")
m, c, hid = 98304, 512, 2048
x = torch.randn(m, c, device=dev).bfloat16()
w1 = (torch.randn(hid, c, device=dev) * 0.02).bfloat16()
b1 = torch.zeros(hid, device=dev)
b1h = b1.bfloat16()
h2 = torch.randn(m, hid, device=dev).bfloat16()
a2 = torch.empty_like(h2)
side = torch.cuda.Stream()


def gemm(tile):
    if tile == 0:
        return torch.nn.functional.linear(x, w1, b1h)
    lib.hs_gemm_nt_set_tile(tile)
    return ops.gemm_nt(x, w1, b1)[0]


def gelu(stream):
    check(lib.hs_gelu_fwd(ptr(h2), ptr(a2), h2.numel(), 0.0, 0, _lib.dtype_code(torch.bfloat16), stream.cuda_stream), "gelu")


def timed(fn, reps=20):
    ts = []
    for it in range(reps + 3):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if it >= 3:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


main = torch.cuda.current_stream()


def both(tile, pass_first):
    side.wait_stream(main)
    if pass_first:
        gelu(side)
        gemm(tile)
    else:
        gemm(tile)
        gelu(side)
    main.wait_stream(side)


print(f"gelu pass alone          {timed(lambda: gelu(main)):7.1f} us")
for tile, name in ((2, "hs 256x128"), (3, "hs 256x256"), (0, "library   ")):
    t_g = timed(lambda: gemm(tile))
    t_b = timed(lambda: both(tile, False))
    t_b2 = timed(lambda: both(tile, True))
    print(f"{name}: GEMM alone {t_g:7.1f} us | GEMM then pass on the side stream {t_b:7.1f} us | pass launched first {t_b2:7.1f} us")
lib.hs_gemm_nt_set_tile(0)
