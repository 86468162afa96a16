#!/usr/bin/env python3
"""hs_linear_wgrad on the Linear shapes of a workload: time, TFLOP/s, GB/s and the per-step total.
   python tools/bench_wgrad.py [--workload B256] [--batch 8] [--check]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, full_cfg  # noqa: E402
from heal_swin_amd._lib import check, lib, ptr  # noqa: E402
from tools.bench_gemm import t_of  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="B256")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--check", action="store_true", help="compare with a float32 matmul")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    a = ap.parse_args()
    wl = WORKLOADS[a.workload]
    cfg = full_cfg(wl["cfg"])
    N0 = wl["base_pix"] * wl["nside"] ** 2 // cfg["patch_size"]
    L = len(cfg["depths"])
    shapes = []
    for s in range(L):
        M, C = a.batch * N0 // 4 ** s, cfg["embed_dim"] * 2 ** s
        nblk = cfg["depths"][s] * (2 if s < L - 1 else 1)
        shapes += [(f"s{s} qkv", M, C, 3 * C, nblk), (f"s{s} proj", M, C, C, nblk), (f"s{s} fc1", M, C, 4 * C, nblk),
                   (f"s{s} fc2", M, 4 * C, C, nblk)]
        if s < L - 1:
            shapes += [(f"s{s} merge", M // 4, 4 * C, 2 * C, 1), (f"s{s} concat", M, 2 * C, C, 1), (f"s{s+1} expand", M // 4, 2 * C, 4 * C, 1)]
    shapes += [("final expand", a.batch * N0, cfg["embed_dim"], 4 * cfg["embed_dim"], 1)]
    tot = 0.0
    for name, M, K, N, cnt in shapes:
        dt, code = (torch.bfloat16, 1) if a.dtype == "bf16" else (torch.float32, 0)
        x = torch.randn(M, K, device="cuda", dtype=dt)
        dy = torch.randn(M, N, device="cuda", dtype=dt)
        dw = torch.empty(N, K, device="cuda")
        db = torch.empty(N, device="cuda")
        ws = torch.empty(int(lib.hs_linear_wgrad_workspace(M, N, K)), device="cuda")
        th = t_of(lambda: check(lib.hs_linear_wgrad(ptr(dy), ptr(x), ptr(dw), ptr(db), ptr(ws), M, N, K, 0, code, None), "wgrad"))
        fl, hbm = 2.0 * M * K * N, x.element_size() * M * (N + K)
        err = ""
        if a.check:
            Mc = min(M, 65536)
            check(lib.hs_linear_wgrad(ptr(dy), ptr(x), ptr(dw), ptr(db), ptr(ws), Mc, N, K, 0, code, None), "wgrad")
            ref = dy[:Mc].float().t() @ x[:Mc].float()
            eb = float((db - dy[:Mc].float().sum(0)).abs().max())
            err = f"  max|dw err| {float((dw - ref).abs().max()):.2e} (scale {float(ref.abs().max()):.1f})  max|db err| {eb:.2e}"
        print(f"{name:14s} M={M:8d} K={K:5d} N={N:5d} x{cnt:2d}  {th*1e6:8.1f} us {fl/th/1e12:6.0f} TF/s {hbm/th/1e9:6.0f} GB/s  = {th*cnt*1e3:6.2f} ms{err}")
        tot += th * cnt * 1e3
        del x, dy
    print(f"hs wgrad total {tot:.1f} ms/step")


if __name__ == "__main__":
    main()
