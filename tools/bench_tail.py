#!/usr/bin/env python3
"""Decoder tail + segmentation loss at the headline shape (HEAL-SWIN-B, nside 256, 12 base pixels, batch 8: 1 572 864 tokens x 128
channels -> 6 291 456 pixel rows x 12 classes), forward + backward, event-timed:
   fused    ops.expand_ln_head_ce      (hs_expand_ln_head_ce_fwd, hs_ln_head_ce_bwd: no logits tensor)
   unfused  ops.expand_ln_head + losses.seg_loss   (fp32 logits written, CE forward / backward kernels, dlogits read back)
usage: bench_tail.py [--mode fused|unfused|both] [--iters 5]
Under `rocprofv3 --pmc WRITE_SIZE --kernel-trace` with one --mode, tools/pmc_db.py sums the bytes the tail's kernels write."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_swin_amd import ops  # noqa: E402
from heal_swin_amd.losses import seg_loss  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="both")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--tokens", type=int, default=8 * 196608)
    ap.add_argument("--width", type=int, default=128)
    a = ap.parse_args()
    dev, C, f_out, B = "cuda", a.width, 12, 8
    g = torch.Generator(device=dev).manual_seed(0)
    xn = torch.randn(a.tokens, C, device=dev, generator=g).to(torch.bfloat16).requires_grad_(True)
    wexp = (torch.randn(4 * C, C, device=dev, generator=g) * C ** -0.5).requires_grad_(True)
    gamma = torch.ones(C, device=dev, requires_grad=True)
    beta = torch.zeros(C, device=dev, requires_grad=True)
    w = (torch.randn(f_out, C, 1, device=dev, generator=g) * C ** -0.5).requires_grad_(True)
    labels = torch.randint(0, f_out, (B, 4 * a.tokens // B), device=dev, dtype=torch.uint8, generator=g)

    def fused():
        ops.expand_ln_head_ce(xn, wexp, gamma, beta, w, labels, None).backward()

    def unfused():
        lg = ops.expand_ln_head(xn, wexp, gamma, beta, w)
        logits = ops.pad_slice(lg.view(B, -1, 16), f_out).transpose(1, 2)
        seg_loss(logits, labels).backward()

    for name, fn in (("fused", fused), ("unfused", unfused)):
        if a.mode not in ("both", name):
            continue
        ts = []
        for it in range(a.iters + 2):
            for t in (xn, wexp, gamma, beta, w):
                t.grad = None
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            if it >= 2:
                ts.append(e0.elapsed_time(e1))
        print(f"{name:8s} tail fwd + loss + bwd: min {min(ts):.3f} ms  median {sorted(ts)[len(ts) // 2]:.3f} ms  ({a.tokens} tokens x {C})", flush=True)


if __name__ == "__main__":
    main()
