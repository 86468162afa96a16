#!/usr/bin/env python3
"""Group a rocprofv3 *_kernel_stats.csv by kernel family.  usage: prof_summary.py stats.csv [steps]"""
import collections
import csv
import sys


def family(n):
    if "attn_bwd_f32" in n: return "hs attn_bwd_mfma_f32"
    if "attn_fwd_f32" in n: return "hs attn_fwd_mfma_f32"
    if "attn_bwd_mfma" in n: return "hs attn_bwd_mfma"
    if "attn_fwd_mfma" in n: return "hs attn_fwd_mfma"
    if "attn_bwd_generic" in n: return "hs attn_bwd_generic"
    if "attn_fwd_generic" in n: return "hs attn_fwd_generic"
    if "reduce_partials" in n: return "hs attn partial reduce"
    if "layernorm" in n: return "hs layernorm*"
    if "rel_bias" in n: return "hs rel_bias*"
    if "gather_rows" in n: return "hs gather_rows"
    if "wgrad" in n: return "hs linear_wgrad"
    if "reduce_slices" in n: return "hs linear_wgrad slice reduce"
    if "reduce_many" in n: return "hs parameter-gradient sums (batched)"
    if "adam_" in n: return "optimizer"
    if "transpose_many" in n: return "hs weight transposes"
    if "gemm_nt" in n: return "hs gemm_nt (own GEMM + epilogues)"
    if "gelu" in n: return "hs gelu fwd/bwd"
    if "hs::" in n: return "hs other"
    if n.startswith("Cijk") or n.startswith("Custom_Cijk"): return "hipBLASLt GEMM"
    if "Gelu" in n: return "torch GELU fwd/bwd"
    if "CUDAFunctor_add" in n: return "torch add"
    if "reduce_kernel" in n: return "torch reduce (bias grads etc)"
    if "copy" in n.lower(): return "torch copy/cast"
    if "cat" in n.lower(): return "torch cat"
    if "adam" in n.lower() or "multi_tensor" in n or "foreach" in n.lower(): return "optimizer"
    if "softmax" in n.lower() or "nll" in n.lower(): return "loss (CE)"
    return "other"


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    groups = collections.Counter()
    calls = collections.Counter()
    for r in rows:
        groups[family(r["Name"])] += int(r["TotalDurationNs"])
        calls[family(r["Name"])] += int(r["Calls"])
    print(f"{'family':34s} {'ms/step':>9s} {'%':>6s} {'calls/step':>10s}")
    for g, v in groups.most_common():
        print(f"{g:34s} {v / 1e6 / steps:9.2f} {100 * v / tot:6.1f} {calls[g] / steps:10.0f}")
    print(f"{'total GPU busy':34s} {tot / 1e6 / steps:9.2f}")


if __name__ == "__main__":
    main()
