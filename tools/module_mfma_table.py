#!/usr/bin/env python3
"""Row X in SURVEY 8d terms: MFMA fraction of the WindowAttention MODULE (qkv Linear -> window attention -> proj Linear,
swin_hp_transformer.py:124-174), forward + backward with parameter gradients, per stage of HEAL-SWIN-B at nside 256 / 12 base pixels,
batch 8, bf16 -- measured on the package's own nn.Module (`WindowAttention.attend`), with the per-kernel event brackets of
ops.KERNEL_TIMINGS splitting the time into the Linear products (forward, input gradient, weight gradient) and the attention core.

    module flops per token, forward = 8 C^2 + 4 Ws C ; forward + backward = 3 x          
    MFMA fraction = flops / time / 2.5 PFLOP/s (dense bf16 peak)

Two bounds are printed beside the measurement: the module with the core's time REMOVED (what a perfect qkv -> attention -> proj
fusion could reach if its GEMM parts kept their rate), and the rate of the GEMM parts alone."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heal_swin_amd import ops  # noqa: E402
from heal_swin_amd.models_torch.swin_hp_transformer import WindowAttention  # noqa: E402

PEAK = 2.5e15


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="B", choices=["B", "T"])
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--json", default="")
    ap.add_argument("--no-tuned-gemm", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda")
    # the library GEMMs with the solution picks the bench loads (PyTorch TunableOp results for these shapes), not the default heuristic
    tuned = os.path.join(ROOT, "heal_swin_amd", "tuning", f"tunableop_gfx950_{args.model}256_bs{args.batch}_bf16.csv")
    library = "default heuristic"
    if os.path.exists(tuned) and not args.no_tuned_gemm:
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(False)
        torch.cuda.tunable.set_filename(tuned, insert_device_ordinal=False)
        if torch.cuda.tunable.read_file(tuned):
            library = f"TunableOp picks ({os.path.basename(tuned)})"
    embed, heads0, bp = (128, 4, 12) if args.model == "B" else (96, 3, 8)
    depths = [2, 2, 18, 2] if args.model == "B" else [2, 2, 6, 2]
    rows = []
    tot = dict(flops=0.0, t=0.0, core=0.0)
    print(f"HEAL-SWIN-{args.model} nside 256, {bp} base pixels, batch {args.batch}, bf16: WindowAttention module, forward + backward; library GEMMs: {library}")
    print(f"{'stage':>5} {'C':>5} {'tokens':>8} {'blocks':>6} | {'module us':>10} {'TF/s':>6} {'of 2.5PF':>8} | {'core us':>8} {'GEMM us':>8} "
          f"{'GEMM TF/s':>9} | {'core-free bound':>15}")
    for s in range(4):
        C, nH, N = embed << s, heads0 << s, bp * 128 * 128 >> (2 * s)  # tokens per image: Npix / 4 at stage 0
        torch.manual_seed(s)
        attn = WindowAttention(C, 64, nH, rel_pos_bias="flat").to(dev).train()
        with torch.no_grad():
            attn.relative_position_bias_table.normal_(0, 0.2)
        x = torch.randn(args.batch, N, C, device=dev).to(torch.bfloat16).requires_grad_(True)
        dy = torch.randn(args.batch, N, C, device=dev).to(torch.bfloat16)
        labels = torch.zeros(N, dtype=torch.uint8, device=dev)
        labels[N - 64:N - 32] = 1
        labels[N - 32:] = 2

        def step():
            x.grad = None
            for p in attn.parameters():
                p.grad = None
            y = attn.attend(x, 64, None, 32, labels)
            y.backward(dy)

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        ops.KERNEL_TIMINGS, ops.TIMED_PREFIXES = [], None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            step()
        e1.record()
        torch.cuda.synchronize()
        wall = e0.elapsed_time(e1) * 1e-3 / args.iters
        fam = {}
        for tag, a, b, _nb, _fl in ops.KERNEL_TIMINGS:
            key = "core" if tag.startswith("window_attn") else "gemm" if (tag.startswith(("hs_gemm_nt", "lib ", "linear_wgrad"))) else "other"
            fam[key] = fam.get(key, 0.0) + a.elapsed_time(b) * 1e-3 / args.iters
        ops.KERNEL_TIMINGS = None
        flops = 3.0 * args.batch * N * (8 * C * C + 4 * 64 * C)
        gemm_flops = 3.0 * args.batch * N * 8 * C * C
        busy = fam.get("core", 0) + fam.get("gemm", 0)  # GPU time of the module's kernels (the wall time adds the brackets' gaps)
        rec = dict(stage=s, C=C, tokens=N, blocks=depths[s], module_us=busy * 1e6, wall_us=wall * 1e6, TFLOPs=flops / busy / 1e12,
                   frac=flops / busy / PEAK, core_us=fam.get("core", 0) * 1e6, gemm_us=fam.get("gemm", 0) * 1e6,
                   gemm_TFLOPs=gemm_flops / fam["gemm"] / 1e12, core_free_frac=flops / fam["gemm"] / PEAK)
        rows.append(rec)
        tot["flops"] += depths[s] * flops
        tot["t"] += depths[s] * busy
        tot["core"] += depths[s] * fam.get("core", 0)
        print(f"{s:>5} {C:>5} {N:>8} {depths[s]:>6} | {rec['module_us']:>10.0f} {rec['TFLOPs']:>6.0f} {rec['frac']:>8.3f} | {rec['core_us']:>8.0f} "
              f"{rec['gemm_us']:>8.0f} {rec['gemm_TFLOPs']:>9.0f} | {rec['core_free_frac']:>15.3f}")
        del attn, x, dy
        torch.cuda.empty_cache()
    print(f"all blocks of one network side, weighted by depth: {tot['flops'] / tot['t'] / 1e12:.0f} TF/s = {tot['flops'] / tot['t'] / PEAK:.3f} of 2.5 PF; "
          f"with the core's time removed {tot['flops'] / (tot['t'] - tot['core']) / PEAK:.3f}")
    if args.json:
        json.dump(dict(model=args.model, batch=args.batch, rows=rows,
                       weighted=dict(frac=tot["flops"] / tot["t"] / PEAK, core_free_frac=tot["flops"] / (tot["t"] - tot["core"]) / PEAK)),
                  open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
