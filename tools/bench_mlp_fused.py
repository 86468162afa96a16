#!/usr/bin/env python3
"""Fused Mlp block (hs_mlp_fused_fwd / _bwd) against the composition it replaces, at the stage-0 shapes of the bench workloads.
usage: python tools/bench_mlp_fused.py [--iters 10]   -> one JSON line per shape (us per launch, GB/s over the algorithmic bytes)"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_swin_amd import ops  # noqa: E402


def timed(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    dev = "cuda"
    for name, rows, C in (("B256 stage 0", 8 * 196608, 128), ("T256 stage 0", 8 * 131072, 96), ("T128 stage 0", 8 * 32768, 96)):
        H = 4 * C
        g = torch.Generator(device=dev).manual_seed(0)
        x = torch.randn((rows, C), generator=g, device=dev).to(torch.bfloat16).requires_grad_(True)
        ps = [torch.ones(C, device=dev, requires_grad=True), torch.zeros(C, device=dev, requires_grad=True),
              (torch.randn((H, C), generator=g, device=dev) * 0.05).requires_grad_(True), torch.zeros(H, device=dev, requires_grad=True),
              (torch.randn((C, H), generator=g, device=dev) * 0.05).requires_grad_(True), torch.zeros(C, device=dev, requires_grad=True)]
        dy = torch.randn((rows, C), generator=g, device=dev).to(torch.bfloat16)
        res = {"shape": name, "rows": rows, "C": C}
        for mode in ("fused", "composed"):
            ops.FUSED_MLP = mode == "fused"

            def fwd():
                if mode == "fused":
                    return ops.fused_mlp_block(x, *ps)
                n2, xa = ops.layer_norm_passthrough(x, ps[0], ps[1])
                return ops.mlp(n2, ps[2], ps[3], ps[4], ps[5], residual=xa)
            t_f = timed(fwd, args.iters)

            def fb():
                for p in [x] + ps:
                    p.grad = None
                fwd().backward(dy)
            t_fb = timed(fb, args.iters)
            res[mode] = {"fwd_us": round(t_f, 1), "fwd_bwd_us": round(t_fb, 1), "bwd_us": round(t_fb - t_f, 1)}
        # the two kernels alone, through the autograd node's own launches
        ops.FUSED_MLP = True
        ops.KERNEL_TIMINGS, ops.TIMED_PREFIXES = [], ("mlp_fused",)
        for _ in range(args.iters):
            for p in [x] + ps:
                p.grad = None
            ops.fused_mlp_block(x, *ps).backward(dy)
        torch.cuda.synchronize()
        agg = {}
        for tag, s, e, nbytes, flops in ops.KERNEL_TIMINGS:
            a = agg.setdefault(tag, [0.0, 0, nbytes])
            a[0] += s.elapsed_time(e) * 1e3
            a[1] += 1
        ops.KERNEL_TIMINGS = None
        res["kernels"] = {k: {"us": round(v[0] / v[1], 1), "GBps": round(v[2] / (v[0] / v[1]) / 1e3, 0)} for k, v in agg.items()}
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
