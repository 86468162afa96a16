#!/usr/bin/env python3
"""GELU kernel bandwidth vs torch's at the MLP hidden sizes of HEAL-SWIN-B @ 256, batch 8 (bf16)."""
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


def main():
    for rows, width in [(1572864, 512), (393216, 1024), (98304, 2048), (24576, 4096)]:
        x = torch.randn(rows, width, device="cuda").to(torch.bfloat16).requires_grad_(True)
        dy = torch.randn(rows, width, device="cuda").to(torch.bfloat16)
        E = rows * width * 2
        with torch.no_grad():
            tf = t_of(lambda: ops.gelu_dropout(x))
            tt = t_of(lambda: torch.nn.functional.gelu(x))
            tfd = t_of(lambda: ops.gelu_dropout(x, 0.1, seed=1))
        y = ops.gelu_dropout(x)
        tb = t_of(lambda: torch.autograd.grad(y, x, dy, retain_graph=True))
        yt = torch.nn.functional.gelu(x)
        tbt = t_of(lambda: torch.autograd.grad(yt, x, dy, retain_graph=True))
        err = float((ops.gelu_dropout(x.detach().float()) - torch.nn.functional.gelu(x.detach().float())).abs().max())
        print(f"{rows:8d}x{width:5d} fwd hs {tf*1e6:7.1f} us {2*E/tf/1e9:5.0f} GB/s (torch {tt*1e6:7.1f}) +drop {tfd*1e6:7.1f} | bwd hs {tb*1e6:7.1f} us {3*E/tb/1e9:5.0f} GB/s (torch {tbt*1e6:7.1f}) | max|hs-torch| fp32 {err:.2e}")


if __name__ == "__main__":
    main()
