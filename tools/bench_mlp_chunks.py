#!/usr/bin/env python3
"""Does the MLP forward of a stage-2 block (98 304 rows, 512 -> 2048 -> 512, bf16) gain from running in row chunks whose hidden
tensors (h, act: 403 MB each at full size) fit the 256 MB Infinity Cache?  library fc1 -> hs_gelu_fwd -> library fc2, whole vs
2 / 4 / 8 chunks; a 1 GB buffer is streamed between repetitions so that every variant starts from a cold cache."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_swin_amd import _lib, ops  # noqa: E402

lib, ptr, check = _lib.lib, ops.ptr, _lib.check
dev = "cuda"
m, c, hid = 98304, 512, 2048
x = torch.randn(m, c, device=dev).bfloat16()
w1 = (torch.randn(hid, c, device=dev) * 0.02).bfloat16()
w2 = (torch.randn(c, hid, device=dev) * 0.02).bfloat16()
b1 = torch.zeros(hid, device=dev).bfloat16()
b2 = torch.zeros(c, device=dev).bfloat16()
h = torch.empty(m, hid, device=dev, dtype=torch.bfloat16)
a = torch.empty_like(h)
y = torch.empty(m, c, device=dev, dtype=torch.bfloat16)
thrash = torch.empty(1 << 29, device=dev, dtype=torch.bfloat16)
st = ops.stream_ptr(torch.device(dev))


def run(chunks, fused_order):
    rows = m // chunks
    if fused_order:  # fc1 -> gelu -> fc2 per chunk
        for i in range(chunks):
            s = slice(i * rows, (i + 1) * rows)
            torch.addmm(b1, x[s], w1.t(), out=h[s])
            check(lib.hs_gelu_fwd(ptr(h[s]), ptr(a[s]), h[s].numel(), 0.0, 0, _lib.dtype_code(torch.bfloat16), st), "gelu")
            torch.addmm(b2, a[s], w2.t(), out=y[s])
    else:  # all fc1, all gelu, all fc2 (chunked launches, no locality)
        for i in range(chunks):
            s = slice(i * rows, (i + 1) * rows)
            torch.addmm(b1, x[s], w1.t(), out=h[s])
        for i in range(chunks):
            s = slice(i * rows, (i + 1) * rows)
            check(lib.hs_gelu_fwd(ptr(h[s]), ptr(a[s]), h[s].numel(), 0.0, 0, _lib.dtype_code(torch.bfloat16), st), "gelu")
        for i in range(chunks):
            s = slice(i * rows, (i + 1) * rows)
            torch.addmm(b2, a[s], w2.t(), out=y[s])


def timed(chunks, fused_order, reps=12):
    ts = []
    for it in range(reps + 3):
        thrash.add_(1)  # 1 GB read + write: evicts L2 and the Infinity Cache
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        run(chunks, fused_order)
        e1.record()
        torch.cuda.synchronize()
        if it >= 3:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


for rnd in range(2):
    for chunks in (1, 2, 4, 8):
        for order in ((True,) if chunks == 1 else (True, False)):
            med, best = timed(chunks, order)
            print(f"round {rnd} chunks {chunks} {'per-chunk fc1->gelu->fc2' if order else 'phase by phase':26s} median {med:7.1f} us  min {best:7.1f} us")
