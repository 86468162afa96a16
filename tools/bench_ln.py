#!/usr/bin/env python3
"""LayerNorm kernel bandwidth at the stage shapes of HEAL-SWIN-B @ nside 256, batch 8 (bf16)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_swin_amd import ops  # noqa: E402


def t_of(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


for rows, width in [(1572864, 128), (393216, 256), (98304, 512), (24576, 1024), (393216, 512), (98304, 2048), (6291456, 128)]:
    x = torch.randn(rows, width, device="cuda").to(torch.bfloat16).requires_grad_(True)
    b = torch.randn(rows, width, device="cuda").to(torch.bfloat16).requires_grad_(True)
    w = torch.ones(width, device="cuda", requires_grad=True)
    be = torch.zeros(width, device="cuda", requires_grad=True)
    dy = torch.randn(rows, width, device="cuda").to(torch.bfloat16)
    E = rows * width * 2
    with torch.no_grad():
        tf = t_of(lambda: ops.layer_norm(x, w, be))
        taf = t_of(lambda: ops.add_layer_norm(x, b, w, be))
    y = ops.layer_norm(x, w, be)
    tb = t_of(lambda: torch.autograd.grad(y, (x, w, be), dy, retain_graph=True))
    s, y2 = ops.add_layer_norm(x, b, w, be)
    tab = t_of(lambda: torch.autograd.grad((s, y2), (x, w, be), (dy, dy), retain_graph=True))
    print(f"rows {rows:8d} width {width:5d} | ln fwd {tf*1e6:7.1f} us {2*E/tf/1e9:5.0f} GB/s | add_ln fwd {taf*1e6:7.1f} us {4*E/taf/1e9:5.0f} GB/s"
          f" | ln bwd {tb*1e6:7.1f} us {3*E/tb/1e9:5.0f} GB/s | add_ln bwd {tab*1e6:7.1f} us {4*E/tab/1e9:5.0f} GB/s")
