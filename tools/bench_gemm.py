#!/usr/bin/env python3
"""Library-GEMM efficiency on the Linear shapes of a workload (fwd, dgrad, wgrad), bf16.
   python tools/bench_gemm.py [--workload B256] [--batch 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, full_cfg  # noqa: E402
from heal_swin_amd._lib import check, lib, ptr  # noqa: E402


def t_of(fn, iters=5):
    for _ in range(2):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="B256")
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    wl = WORKLOADS[a.workload]
    cfg = full_cfg(wl["cfg"])
    N0 = wl["base_pix"] * wl["nside"] ** 2 // cfg["patch_size"]
    L = len(cfg["depths"])
    shapes = []  # (name, M tokens, K in, N out, count per step)
    for s in range(L):
        M, C = a.batch * N0 // 4 ** s, cfg["embed_dim"] * 2 ** s
        nblk = cfg["depths"][s] * (2 if s < L - 1 else 1)
        shapes += [(f"s{s} qkv", M, C, 3 * C, nblk), (f"s{s} proj", M, C, C, nblk), (f"s{s} fc1", M, C, 4 * C, nblk),
                   (f"s{s} fc2", M, 4 * C, C, nblk)]
        if s < L - 1:
            shapes += [(f"s{s} merge", M // 4, 4 * C, 2 * C, 1), (f"s{s} concat", M, 2 * C, C, 1), (f"s{s+1} expand", M // 4, 2 * C, 4 * C, 1)]
    shapes += [("final expand", a.batch * N0, cfg["embed_dim"], 4 * cfg["embed_dim"], 1)]
    tot_t = tot_f = tot_h = 0.0
    print(f"{'gemm':14s} {'M':>8s} {'K':>5s} {'N':>5s} cnt | fwd TF/s  dgrad TF/s  wgrad TF/s | ms/step(all 3)")
    for name, M, K, N, cnt in shapes:
        x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
        dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
        fl = 2.0 * M * K * N
        tf = t_of(lambda: torch.nn.functional.linear(x, w))
        td = t_of(lambda: dy @ w)
        tw = t_of(lambda: dy.t() @ x)
        dw = torch.empty(N, K, device="cuda"); db = torch.empty(N, device="cuda")
        ws = torch.empty(int(lib.hs_linear_wgrad_workspace(M, N, K)), device="cuda")
        th = t_of(lambda: check(lib.hs_linear_wgrad(ptr(dy), ptr(x), ptr(dw), ptr(db), ptr(ws), M, N, K, 0, 1, None), "wgrad"))
        hbm = 2.0 * M * (N + K)
        ms = (tf + td + tw) * cnt * 1e3
        tot_t += ms
        tot_f += 3 * fl * cnt
        print(f"{name:14s} {M:8d} {K:5d} {N:5d} {cnt:3d} | {fl/tf/1e12:7.0f}  {fl/td/1e12:9.0f}  {fl/tw/1e12:9.0f}  | {ms:8.2f} | hs wgrad {th*1e6:8.1f} us {fl/th/1e12:6.0f} TF/s {hbm/th/1e9:6.0f} GB/s  x{cnt} = {th*cnt*1e3:6.2f} ms")
        tot_h += th * cnt * 1e3
        del x, w, dy
    print(f"total {tot_t:.1f} ms/step for {tot_f/1e12:.1f} TF -> {tot_f/tot_t/1e9:.0f} TF/s ; hs wgrad total {tot_h:.1f} ms/step")


if __name__ == "__main__":
    main()
