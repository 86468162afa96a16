#!/usr/bin/env python3
"""Multi-GPU readiness measured on ONE GPU: what do long-lived kernels of another stream -- RCCL's
all-reduce ring kernels during the backward of a data-parallel step -- cost the launches of this library whose grids are sized
to exactly one resident round of all 256 CUs, and what does hs_set_reserved_cus() buy back?

A stand-in (`hs_debug_occupy_cus`: k workgroups of 256 threads, 128 VGPRs, 16 KB LDS, resident for the whole measurement on a side
stream) plays the communication kernels.  Measured: (1) the chip-filling kernels one by one at the HEAL-SWIN-B stage-2 shapes,
(2) the whole B / nside 256 / batch 8 training step.  Round 4: a duty-cycled occupier shaped like the real exchange beside the always-resident one, and the step with every GEMM on
hs_gemm_nt (ops.RT.prefer_own_gemm).  Writes JSON to stdout (-> profiles/archive_r01_r04/r04_cu_contention.json)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heal_swin_amd import _lib, ops  # noqa: E402
from heal_swin_amd._lib import check, lib  # noqa: E402

DEV = torch.device("cuda", 0)
side = torch.cuda.Stream(device=DEV)


def occupy(k, micros):
    if k:
        check(lib.hs_debug_occupy_cus(k, 256, 16 * 1024, float(micros), side.cuda_stream), "hs_debug_occupy_cus")


def timed(fn, iters, k, est_us):
    """average microseconds of fn() with k occupier workgroups resident during all `iters` calls"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    occupy(k, 2.0 * est_us * iters + 3000)
    time.sleep(0.002)  # the occupier is resident before the first launch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    torch.cuda.synchronize()
    return us


def kernel_cases():
    g = torch.Generator(device=DEV).manual_seed(0)
    rows, C, H = 98304, 512, 2048  # stage 2 of HEAL-SWIN-B at nside 256, batch 8
    x = torch.randn(rows, C, device=DEV, generator=g).to(torch.bfloat16)
    dy = torch.randn(rows, H, device=DEV, generator=g).to(torch.bfloat16)
    w1 = (torch.randn(H, C, device=DEV, generator=g) * 0.02).to(torch.bfloat16)
    h = torch.randn(rows, H, device=DEV, generator=g).to(torch.bfloat16)
    dyc = torch.randn(rows, C, device=DEV, generator=g).to(torch.bfloat16)
    qkv = torch.randn(8, 12288, 3 * C, device=DEV, generator=g).to(torch.bfloat16).requires_grad_(True)
    bias = torch.randn(16, 64, 64, device=DEV, generator=g)
    hs = torch.full((16,), 32 ** -0.5, device=DEV)
    do = torch.randn(8, 12288, C, device=DEV, generator=g).to(torch.bfloat16)

    def wgrad():
        ops.LinearFn._wgrad_hip(dy, x, H, C, True)

    def gemm_dgelu():
        ops.gemm_nt(dyc, w1, None, _lib.HS_EPI_DGELU, aux=h)  # dh = (dy W) o gelu'(h): m = 98304, n = 2048, k = 512

    def attn():
        o = ops.window_attn_core(qkv, bias, hs, None, 32, None, 16, 64, False)
        o.backward(do)
        qkv.grad = None

    def lib_gemm():
        torch.nn.functional.linear(x, w1)

    return [("hs_linear_wgrad s2 fc1 (98304 x 2048 x 512)", wgrad, 200), ("hs_gemm_nt GELU' epilogue s2 (98304 x 2048 x 512)", gemm_dgelu, 350),
            ("hs_window_attn fwd+bwd s2 (8 x 12288 x 512)", attn, 350), ("hipBLASLt s2 fc1 forward (98304 x 2048 x 512)", lib_gemm, 200)]


def step_case():
    import bench
    import types
    from heal_swin_amd.losses import seg_loss
    from heal_swin_amd.parallel import GradBucketAllReduce
    wl = bench.WORKLOADS["B256"]
    model, cfg, spec = bench.build_model(wl)
    model = model.to(DEV).train()
    model.compute_dtype = torch.bfloat16
    dp = GradBucketAllReduce(model.parameters(), reserved_cus=int(lib.hs_get_reserved_cus()))
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
    g = torch.Generator(device=DEV).manual_seed(1)
    imgs = torch.randint(0, 256, (8, 3, spec["dim_in"]), generator=g, device=DEV, dtype=torch.uint8)
    labels = torch.randint(0, 12, (8, spec["dim_in"]), generator=g, device=DEV, dtype=torch.uint8)

    def step():
        dp.zero_grad()
        seg_loss(model(imgs.float()), labels).backward()
        dp.finish()
        opt.step()

    return step, dp


def timed_duty(fn, iters, k, step_ms, bursts=10, burst_us=400.0):
    """the same with a DUTY-CYCLED occupier shaped like the real exchange of a data-parallel step: `bursts` launches of k
    workgroups x `burst_us` per step, evenly spaced, issued from a second host thread onto the side stream"""
    import threading
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    stop = threading.Event()

    def pulse():
        gap = step_ms * 1e-3 / bursts
        while not stop.is_set():
            occupy(k, burst_us)
            time.sleep(gap)

    th = threading.Thread(target=pulse, daemon=True)
    th.start()
    time.sleep(0.005)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    stop.set()
    th.join()
    us = 1e3 * e0.elapsed_time(e1) / iters
    torch.cuda.synchronize()
    return us


def main():
    out = {"device": torch.cuda.get_device_name(0), "occupier": "k workgroups x 256 threads, 128 VGPRs, 16 KB LDS on a side stream: "
           "`resident` for the whole measurement, or `duty` = 10 bursts x 0.4 ms per step (the shape of the real gradient exchange: "
           "596 MB of fp32 buckets per 160 ms step)", "kernels": [], "step": []}
    cases = kernel_cases()
    for reserved in (0, 16):
        check(lib.hs_set_reserved_cus(reserved), "hs_set_reserved_cus")
        for name, fn, est in cases:
            row = {"kernel": name, "reserved_cus": reserved, "us": {}}
            for k in (0, 8, 16):
                row["us"][str(k)] = round(timed(fn, 10, k, est), 1)
            out["kernels"].append(row)
            print(row, file=sys.stderr, flush=True)
    del cases
    torch.cuda.empty_cache()
    # whole step: library GEMMs where the per-shape policy picks them (idle-chip default) vs every bf16 Linear on hs_gemm_nt
    # (what GradBucketAllReduce switches on when CUs are reserved, ops.RT.prefer_own_gemm)
    for reserved, own in ((0, False), (16, False), (16, True)):
        check(lib.hs_set_reserved_cus(reserved), "hs_set_reserved_cus")
        step, dp = step_case()
        ops.RT.prefer_own_gemm = own
        row = {"workload": "HEAL-SWIN-B nside 256 batch 8 bf16 train step", "reserved_cus": reserved,
               "gemms": "all bf16 Linear products on hs_gemm_nt" if own else "per-shape policy (hipBLASLt for the MFMA-bound products)", "ms": {}}
        row["ms"]["idle"] = round(timed(step, 5, 0, 170000) / 1e3, 2)
        for k in (8, 16):
            row["ms"][f"resident k={k}"] = round(timed(step, 5, k, 170000) / 1e3, 2)
        for k in (8, 16):
            row["ms"][f"duty k={k}"] = round(timed_duty(step, 5, k, row["ms"]["idle"]) / 1e3, 2)
        out["step"].append(row)
        print(row, file=sys.stderr, flush=True)
        dp.remove()
        ops.RT.prefer_own_gemm = False
        del step, dp
        torch.cuda.empty_cache()
    check(lib.hs_set_reserved_cus(0), "hs_set_reserved_cus")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
