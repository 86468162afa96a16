#!/usr/bin/env python3
"""The one-launch WindowAttention module forward (`hs_window_attn_module_fwd`) against the three-kernel composition
(qkv GEMM -> hs_window_attn_fwd -> proj GEMM) at the stage-0 shapes of HEAL-SWIN-B (C = 128, N = 196 608, 12 base pixels) and
HEAL-SWIN-T (C = 96, N = 131 072, 8 base pixels) at nside 256, batch 8, bf16, no-grad.  Reports time, MFMA TFLOP/s of the
module flops (8 C^2 + 4 Ws C per token) against the 2.5 PFLOP/s dense bf16 peak, and algorithmic HBM GB/s (x in + out)."""
import argparse
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heal_swin_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=15)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--json", default="")
    ap.add_argument("--only-fused", action="store_true", help="(for rocprofv3 counter passes)")
    ap.add_argument("--case", default="")
    args = ap.parse_args()
    dev = torch.device("cuda")
    cases = [("B stage 0", 128, 4, 12 * 128 * 128, 32), ("T stage 0", 96, 3, 8 * 128 * 128, 32)]
    out = []
    for name, C, nH, N, shift in cases:
        if args.case and args.case not in name:
            continue
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(args.batch, N, C, device=dev, generator=g).to(torch.bfloat16)
        wqkv = (torch.randn(3 * C, C, device=dev, generator=g) * C ** -0.5).to(torch.bfloat16)
        wp = (torch.randn(C, C, device=dev, generator=g) * C ** -0.5).to(torch.bfloat16)
        bqkv, bp = torch.randn(3 * C, device=dev, generator=g) * 0.1, torch.randn(C, device=dev, generator=g) * 0.1
        bias = torch.randn(nH, 64, 64, device=dev, generator=g)
        hs = torch.full((nH,), 32 ** -0.5, device=dev)
        ln_g, ln_b = torch.rand(C, device=dev, generator=g) + 0.5, torch.randn(C, device=dev, generator=g) * 0.1
        labels = torch.zeros(N, dtype=torch.uint8, device=dev)
        labels[N - 64:N - shift] = 1
        labels[N - shift:] = 2
        flops = args.batch * N * (8 * C * C + 4 * 64 * C)

        def fused(ln=False, res=False):
            return ops.window_attn_module(x, wqkv, bqkv, wp, bp, bias, hs, None, shift, labels, nH, 64, False,
                                          ln_weight=ln_g if ln else None, ln_bias=ln_b if ln else None, residual=res)

        def composed():
            qkv = ops.gemm_nt(x.view(-1, C), wqkv, bqkv)[0].view(args.batch, N, 3 * C)
            o = ops.window_attn_core(qkv, bias, hs, None, shift, labels, nH, 64, False)
            return ops.gemm_nt(o.view(-1, C), wp, bp)[0]

        # training forms: what the block runs with gradients enabled (saves LayerNorm(x), qkv, the attention output, statistics)
        B = args.batch
        tr = dict(out=torch.empty_like(x), xn=torch.empty_like(x), o=torch.empty_like(x),
                  qkv=torch.empty(B, N, 3 * C, dtype=torch.bfloat16, device=dev), mean=torch.empty(B * N, device=dev),
                  rstd=torch.empty(B * N, device=dev), lse=torch.empty(B, nH, N, device=dev))

        def fused_train():
            from heal_swin_amd import _lib
            from heal_swin_amd._lib import check, lib, ptr
            check(lib.hs_window_attn_module_fwd_train(ptr(x), ptr(tr["out"]), ptr(tr["xn"]), ptr(tr["mean"]), ptr(tr["rstd"]), ptr(tr["qkv"]),
                                                      ptr(tr["o"]), ptr(tr["lse"]), ptr(wqkv), ptr(bqkv), ptr(wp), ptr(bp), ptr(ln_g), ptr(ln_b),
                                                      ptr(bias), ptr(hs), None, shift, ptr(labels), None, None, None, None, None, B, N, C, nH, 64, _lib.HS_ATTN_RESIDUAL,
                                                      _lib.HS_BF16, None), "train")

        def composed_train():
            from heal_swin_amd import _lib
            from heal_swin_amd._lib import check, lib, ptr
            check(lib.hs_layernorm_fwd(ptr(x), None, ptr(ln_g), ptr(ln_b), ptr(tr["xn"]), ptr(tr["mean"]), ptr(tr["rstd"]), B * N, C,
                                       _lib.HS_BF16, None), "ln")
            qkv = ops.gemm_nt(tr["xn"].view(-1, C), wqkv, bqkv)[0].view(B, N, 3 * C)
            check(lib.hs_window_attn_fwd(ptr(qkv), ptr(tr["o"]), ptr(tr["lse"]), ptr(bias), ptr(hs), None, shift, ptr(labels), B, N, C, nH, 64,
                                         0, 0.0, 0, _lib.HS_BF16, None), "core")
            return ops.gemm_nt(tr["o"].view(-1, C), wp, bp, _lib.HS_EPI_RESID, aux=x.view(-1, C))[0]

        variants = {"fused module": fused, "fused module + LN + residual": lambda: fused(True, True),
                    "TRAIN form: fused module + LN + residual, saves xn / qkv / O / statistics": fused_train}
        if not args.only_fused:
            variants["qkv GEMM + attn core + proj GEMM (hs_gemm_nt)"] = composed
            variants["TRAIN composition: LN -> qkv GEMM -> attn core (+ lse) -> proj GEMM + residual"] = composed_train
        with torch.no_grad():
            for fn in variants.values():
                fn()
            torch.cuda.synchronize()
            times = {k: [] for k in variants}
            for _ in range(args.iters):
                for k, fn in variants.items():
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    fn()
                    e1.record()
                    torch.cuda.synchronize()
                    times[k].append(e0.elapsed_time(e1) * 1e-3)
        for k in variants:
            t = statistics.median(times[k])
            nbytes = (9 if "TRAIN" in k else 3 if "residual" in k else 2) * x.numel() * 2  # (TRAIN: x in twice, out + 5 C saved)
            rec = {"case": name, "variant": k, "C": C, "tokens": N, "batch": args.batch, "us": t * 1e6, "module_TFLOPs": flops / t / 1e12,
                   "frac_of_2.5PF": flops / t / 2.5e15, "algorithmic_GBs": nbytes / t / 1e9}
            print(f"{name} C={C}: {k:82s} {t * 1e6:8.1f} us  {rec['module_TFLOPs']:7.0f} TF/s ({rec['frac_of_2.5PF']:.3f} of 2.5 PF)  "
                  f"{rec['algorithmic_GBs']:6.0f} GB/s", flush=True)
            out.append(rec)
    if args.json:
        json.dump({"records": out}, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
