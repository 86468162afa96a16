#!/usr/bin/env python3
"""Bandwidth of the standalone shift gather / scatter `hs_gather_rows` (BASELINE config 4: the reference's shifter.shift /
shift_back, models_torch/hp_shifting.py:69-73, :302-306, :400-404) for the three strategies at nside 256, window 64:
nest_roll on the 12-base-pixel sphere (N = 196 608 tokens, C = 128) and nest_grid_shift / ring_shift on 8 base pixels
(N = 131 072, C = 96), batch 8, bf16.  Bytes = 2 * E * elt (one read + one write of every element); peak 8 TB/s."""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heal_swin_amd import ops  # noqa: E402
from heal_swin_amd.models_torch import hp_shifting as S  # noqa: E402

HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--json", default="")
    ap.add_argument("--once", action="store_true", help="one warm-up + one timed launch per case (for rocprofv3 passes)")
    args = ap.parse_args()
    dev = torch.device("cuda")
    cases = [
        ("nest_roll shift 32, bp 12", S.NestRollShift(32, 12 * 128 * 128, 64), 12 * 128 * 128, 128, True),
        ("nest_grid_shift, bp 8", S.NestGridShift(128, 8, 64), 8 * 128 * 128, 96, False),
        ("ring_shift 4, bp 8", S.RingShift(128, 8, 64, 4), 8 * 128 * 128, 96, False),
    ]
    out = []
    for name, shifter, n, c, is_roll in cases:
        x = torch.randn(args.batch, n, c, device=dev).to(torch.bfloat16)
        idx, inv, _ = shifter.tables(dev)
        for direction, (i1, i2, roll) in (("shift (gather)", (None, None, 32) if is_roll else (idx, inv, 0)),
                                          ("shift_back (scatter as gather by the inverse table)",
                                           (None, None, n - 32) if is_roll else (inv, idx, 0))):
            fn = lambda: ops.gather_rows(x, i1, i2, roll)  # noqa: E731
            y = fn()
            torch.cuda.synchronize()
            times = []
            for _ in range(1 if args.once else args.iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1) * 1e-3)
            t = statistics.median(times)
            nbytes = 2 * x.numel() * x.element_size()
            rec = {"case": name, "direction": direction, "tokens": n, "channels": c, "batch": args.batch, "row_bytes": c * 2,
                   "algorithmic_bytes": nbytes, "us": t * 1e6, "GB/s": nbytes / t / 1e9, "frac_of_8TBs": nbytes / t / 1e9 / HBM_PEAK_GBS}
            print(f"{name:28s} {direction[:14]:14s} N={n} C={c}: {t * 1e6:8.1f} us  {rec['GB/s']:7.0f} GB/s  ({rec['frac_of_8TBs']:.2f} of 8 TB/s)", flush=True)
            out.append(rec)
            del y
    if args.json:
        json.dump({"note": "hs_gather_rows, bf16, batch %d; bytes = 2 * elements * 2 B; median of %d launches (HIP events)" % (args.batch, args.iters),
                   "records": out}, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
