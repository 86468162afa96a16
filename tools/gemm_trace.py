#!/usr/bin/env python3
"""Per-wave timeline of hs_gemm_nt's persistent loop (library built with -DHS_GEMM_TRACE): shader-clock stamps at the top of
every k-step (3), before / after its barrier (1 / 2), at the epilogue's start (10), after its barrier (11) and after each of its
row blocks (12 + i), for the 8 waves of workgroups 0 and 9.  Prints where a tile's time goes."""
import argparse
import ctypes
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heal_swin_amd import _lib  # noqa: E402
from heal_swin_amd._lib import check, lib, ptr, stream_ptr  # noqa: E402

CAP = 240


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=98304)
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--k", type=int, default=512)
    ap.add_argument("--tile", type=int, default=3)
    ap.add_argument("--epi", type=int, default=2)
    ap.add_argument("--waves", type=int, default=8)
    args = ap.parse_args()
    dev = torch.device("cuda")
    raw = ctypes.CDLL(lib._name)
    if not hasattr(raw, "hs_gemm_nt_set_trace"):
        sys.exit("library built without -DHS_GEMM_TRACE")
    m, n, k = args.m, args.n, args.k
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
    bias = torch.randn(n, device=dev)
    c = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    aux = torch.randn(m, n, device=dev).to(torch.bfloat16)
    trace = torch.zeros(2 * args.waves * CAP, device=dev, dtype=torch.int64)

    def run():
        lib.hs_gemm_nt_set_tile(args.tile)
        check(lib.hs_gemm_nt(ptr(a), k, ptr(w), k, k, None, 0, None, 0, 0, ptr(bias), ptr(c), ptr(aux), m, n, args.epi, 0.0, 0,
                             _lib.HS_BF16, stream_ptr(dev)), "hs_gemm_nt")

    raw.hs_gemm_nt_set_trace(ctypes.c_void_p(0))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    t_plain = e0.elapsed_time(e1) * 1e3
    raw.hs_gemm_nt_set_trace(ctypes.c_void_p(trace.data_ptr()))
    run(); torch.cuda.synchronize()
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    t_traced = e0.elapsed_time(e1) * 1e3
    raw.hs_gemm_nt_set_trace(ctypes.c_void_p(0))
    tr = trace.cpu().view(2, args.waves, CAP)
    print(f"shape m={m} n={n} k={k} tile={args.tile} epi={args.epi}: {t_plain:.1f} us untraced, {t_traced:.1f} us traced")
    for wg in range(2):
        for wv in (0, args.waves - 1):
            ev = [(int(v) >> 8, int(v) & 0xff) for v in tr[wg, wv].tolist() if v]
            if not ev:
                continue
            t0 = ev[0][0]
            # segments: name by (code_prev -> code_next)
            seg = {}
            for (ta, ca), (tb, cb) in zip(ev, ev[1:]):
                seg.setdefault((ca, cb), []).append(tb - ta)
            print(f"workgroup {'0' if wg == 0 else '9'} wave {wv}: {len(ev)} events, {ev[-1][0] - t0} cycles in all")
            names = {(3, 1): "top of step -> operands landed (vmcnt wait)", (1, 2): "barrier", (2, 3): "k-step compute (32 MFMAs + next DMA)",
                     (2, 10): "last k-step compute", (10, 11): "epilogue: alias barrier", (11, 12): "epilogue row block 0",
                     (12, 13): "epilogue row block 1", (13, 14): "epilogue row block 2", (14, 15): "epilogue row block 3",
                     (15, 3): "epilogue end -> next step", (13, 3): "epilogue end -> next step", (0, 3): "start"}
            names.update({(11, 20): "row block 0: input wait + arithmetic", (12, 20): "row block 1: input wait + arithmetic",
                          (13, 20): "row block 2: input wait + arithmetic", (14, 20): "row block 3: input wait + arithmetic",
                          (20, 22): "arithmetic done -> patch round trip of the first output done", (22, 21): "4 row-segment stores issued (first output)",
                          (21, 22): "patch round trip of the second output", (22, 12): "stores issued -> end of block 0",
                          (22, 13): "stores issued -> end of block 1", (22, 14): "stores issued -> end of block 2", (22, 15): "stores issued -> end of block 3"})
            for key in sorted(seg):
                v = seg[key]
                print(f"   {key[0]:2d}->{key[1]:2d} {names.get(key, ''):48s} n={len(v):3d} median {statistics.median(v):8.0f} mean {statistics.mean(v):8.0f} max {max(v):7d} cycles")
            # first tile's raw timeline
            line = " ".join(f"{c}:{t - t0}" for t, c in ev[:40])
            print("   first events:", line)


if __name__ == "__main__":
    main()
