#!/usr/bin/env python3
"""Micro-benchmark of the fused shift+window-attention kernels at the stage shapes of a workload.
   python tools/bench_attn.py [--workload B256] [--batch 8] [--bwd]
Prints per-stage launch time, algorithmic GB/s (q,k,v in + o out; bwd: qkv,do in + dqkv out: the bf16 MFMA backward no longer reads o) and TFLOP/s."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, full_cfg  # noqa: E402
from heal_swin_amd import _lib, ops  # noqa: E402
from heal_swin_amd.models_torch import hp_shifting as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="B256")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    wl = WORKLOADS[a.workload]
    cfg = full_cfg(wl["cfg"])
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = "cuda"
    N0 = wl["base_pix"] * wl["nside"] ** 2 // cfg["patch_size"]
    Ws = cfg["window_size"]
    for s, nh in enumerate(cfg["num_heads"]):
        N, C = N0 // 4 ** s, cfg["embed_dim"] * 2 ** s
        if N < Ws:
            continue
        nside = int(round((N // wl["base_pix"]) ** 0.5))
        for shifted in (False, True):
            idx = labels = None
            roll = 0
            if shifted:
                if cfg["shift_strategy"] == "nest_roll":
                    sh = S.NestRollShift(cfg["shift_size"], N, Ws)
                    roll = cfg["shift_size"]
                    _, _, labels = sh.tables(dev)
                elif cfg["shift_strategy"] == "ring_shift":
                    sh = S.RingShift(nside, wl["base_pix"], Ws, cfg["shift_size"])
                    idx, _, labels = sh.tables(dev)
                else:
                    sh = S.NestGridShift(nside, wl["base_pix"], Ws)
                    idx, _, labels = sh.tables(dev)
            qkv = torch.randn(a.batch, N, 3 * C, device=dev, dtype=dt, requires_grad=True)
            bias = torch.randn(nh, Ws, Ws, device=dev, requires_grad=True)
            hs = torch.full((nh,), 0.17, device=dev, requires_grad=cfg["use_cos_attn"])
            dout = torch.randn(a.batch, N, C, device=dev, dtype=dt)
            res = {}
            for mode in ("fwd", "bwd"):
                ts = []
                for it in range(a.iters + 2):
                    qkv.grad = None
                    if mode == "fwd":
                        with torch.no_grad():
                            e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
                            e0.record()
                            ops.window_attn_core(qkv, bias, hs, idx, roll, labels, nh, Ws, cfg["use_cos_attn"])
                            e1.record()
                    else:
                        o = ops.window_attn_core(qkv, bias, hs, idx, roll, labels, nh, Ws, cfg["use_cos_attn"])
                        ops.KERNEL_TIMINGS = []
                        o.backward(dout)
                        (_, e0, e1, _, _), = [t for t in ops.KERNEL_TIMINGS if t[0] == "window_attn_bwd"]
                        ops.KERNEL_TIMINGS = None
                    torch.cuda.synchronize()
                    if it >= 2:
                        ts.append(e0.elapsed_time(e1) * 1e-3)
                res[mode] = min(ts)
            E = a.batch * N * C * qkv.element_size()
            fl = 4 * a.batch * N * C * Ws
            print(f"stage {s} N={N:7d} C={C:4d} nH={nh:2d} shifted={int(shifted)}  "
                  f"fwd {res['fwd']*1e6:8.1f} us {4*E/res['fwd']/1e9:7.0f} GB/s {fl/res['fwd']/1e12:6.1f} TF/s   "
                  f"bwd {res['bwd']*1e6:8.1f} us {(7 if a.dtype == 'bf16' else 8)*E/res['bwd']/1e9:7.0f} GB/s {2.5*fl/res['bwd']/1e12:6.1f} TF/s")


if __name__ == "__main__":
    main()
