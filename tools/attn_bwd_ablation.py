#!/usr/bin/env python3
"""Timing-only ablations of attn_bwd_mfma_kernel (HS_ATTN_BWD_ABLATE bit mask; results are wrong by construction): what does each
part of the backward cost at the HEAL-SWIN-B stage shapes?  One subprocess per mask (the mask is read once per process).
Needs a library built with the hooks compiled in:   HS_EXTRA_CXXFLAGS=-DHS_ATTN_ABLATION python heal_swin_amd/build.py --force
(the product build compiles them away: they cost the dropout instantiations 7-12 spilled registers).
   python tools/attn_bwd_ablation.py            -> table on stdout
   python tools/attn_bwd_ablation.py --one S    (internal) time stage S under the current environment"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASKS = [(0, "full kernel"), (1, "no global stores"), (16, "no global loads after the first window"), (17, "no global traffic"),
         (2, "no softmax / dS arithmetic"), (8, "no X / dK / dV products"), (4, "no partial-sum exchange"), (10, "no arithmetic, no products"),
         (14, "loads + staging + stores only"), (31, "barriers and staging only")]


def one(stage):
    import torch
    sys.path.insert(0, ROOT)
    from heal_swin_amd import ops
    B, N, C, nh = 8, 196608 // 4 ** stage, 128 * 2 ** stage, 4 * 2 ** stage
    qkv = torch.randn(B, N, 3 * C, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    bias = torch.randn(nh, 64, 64, device="cuda", requires_grad=True)
    hs = torch.full((nh,), 0.17, device="cuda")
    dout = torch.randn(B, N, C, device="cuda", dtype=torch.bfloat16)
    ts = []
    for it in range(8):
        qkv.grad = None
        o = ops.window_attn_core(qkv, bias, hs, None, 0, None, nh, 64, False)
        ops.KERNEL_TIMINGS = []
        o.backward(dout)
        torch.cuda.synchronize()
        (_, e0, e1, _, _), = [t for t in ops.KERNEL_TIMINGS if t[0] == "window_attn_bwd"]
        ops.KERNEL_TIMINGS = None
        if it >= 3:
            ts.append(e0.elapsed_time(e1) * 1e3)
    print(f"{sum(ts) / len(ts):.1f}")


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--one":
        return one(int(sys.argv[2]))
    print("event-timed launch of hs_window_attn_bwd (incl. the partial reduce), HEAL-SWIN-B nside 256 batch 8, microseconds")
    print(f"{'mask':>4}  {'stage 0':>9} {'stage 2':>9}  what is removed")
    for mask, what in MASKS:
        row = []
        for stage in (0, 2):
            env = dict(os.environ, HS_ATTN_BWD_ABLATE=str(mask))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(stage)], env=env, capture_output=True, text=True)
            row.append(r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "fail")
        print(f"{mask:>4}  {row[0]:>9} {row[1]:>9}  {what}", flush=True)


if __name__ == "__main__":
    main()
