#!/usr/bin/env python3
"""hs_gemm_nt vs the library GEMM (F.linear -> hipBLASLt) on the Linear shapes of a workload (default: HEAL-SWIN-B, nside 256,
12 base pixels, batch 8): TFLOP/s and algorithmic GB/s per shape, tile variant and epilogue.  Interleaved rounds, median."""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heal_swin_amd import _lib  # noqa: E402
from heal_swin_amd._lib import check, lib, ptr, stream_ptr  # noqa: E402


def shapes(embed, tokens0, batch):
    out = []
    for s in range(4):
        c, m = embed << s, batch * (tokens0 >> (2 * s))
        out += [(f"s{s} qkv", m, 3 * c, c), (f"s{s} proj", m, c, c), (f"s{s} fc1", m, 4 * c, c), (f"s{s} fc2", m, c, 4 * c)]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--embed", type=int, default=128)
    ap.add_argument("--tokens0", type=int, default=12 * 256 * 256 // 4)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--only", default="")
    ap.add_argument("--variants", default="1,2,3")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    global VARIANTS
    VARIANTS = [int(v) for v in args.variants.split(",")]
    dev = torch.device("cuda")
    rows = []
    for name, m, n, k in shapes(args.embed, args.tokens0, args.batch):
        if args.only and args.only not in name:
            continue
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
        bias = torch.randn(n, device=dev)
        biasb = bias.to(torch.bfloat16)
        c = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        aux = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        flops = 2.0 * m * n * k
        variants = {"lib": lambda: torch.nn.functional.linear(a, w, biasb)}
        if "fc1" in name:
            variants["lib+gelu"] = lambda: torch.nn.functional.gelu(torch.nn.functional.linear(a, w, biasb))
        for tile in VARIANTS:
            for epi, tag in ((_lib.HS_EPI_BIAS, "bias"),) + (((_lib.HS_EPI_GELU, "gelu"),) if "fc1" in name else ()) + (
                    ((_lib.HS_EPI_DGELU, "dgelu"),) if "fc1" in name else ()):
                def run(tile=tile, epi=epi):
                    lib.hs_gemm_nt_set_tile(tile)
                    check(lib.hs_gemm_nt(ptr(a), k, ptr(w), k, k, None, 0, None, 0, 0, ptr(bias), ptr(c), ptr(aux), m, n, epi, 0.0, 0,
                                         _lib.HS_BF16, stream_ptr(dev)), "hs_gemm_nt")
                variants[f"hs t{tile} {tag}"] = run
        times = {v: [] for v in variants}
        for v, fn in variants.items():
            fn()
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for v, fn in variants.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                fn()
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / 2)
        line = {"shape": name, "m": m, "n": n, "k": k}
        txt = f"{name:8s} m={m:8d} n={n:5d} k={k:5d} |"
        for v in variants:
            ms = statistics.median(times[v])
            line[v] = {"ms": ms, "TFLOPs": flops / ms / 1e9}
            txt += f" {v}: {ms * 1e3:7.1f}us {flops / ms / 1e9:6.0f}TF |"
        print(txt, flush=True)
        rows.append(line)
        del a, w, c, aux
    lib.hs_gemm_nt_set_tile(0)
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
