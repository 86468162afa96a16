#!/usr/bin/env python3
"""bf16 logit error of the headline config at full size (HEAL-SWIN-B, nside 256, 12 base pixels, one image) over MORE draws of
weights and inputs than the test suite runs (tests/test_gpu_baseline_configs.py uses seeds 11, 12, 13): the spread behind the
margin to north_star's 1e-2.  Needs an MI355X.  usage: headline_seed_spread.py [first_seed] [count]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_gpu_baseline_configs as T  # noqa: E402
from _util import errors  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    worst = 0.0
    for seed in range(first, first + count):
        f = T._full_size_oracle(seed)
        model = f["model"].to("cuda").eval()
        model.compute_dtype = torch.bfloat16
        with torch.no_grad():
            e_ng = errors(model(f["x"].to("cuda")), f["logits"])["scale_err"]
        e_tr = errors(model(f["x"].to("cuda")).detach(), f["logits"])["scale_err"]
        worst = max(worst, e_ng, e_tr)
        print(f"seed {seed}: bf16 logits max|a-b|/max|b| training kernels {e_tr:.2e}, no-grad path {e_ng:.2e} (scale {float(f['logits'].abs().max()):.2f})", flush=True)
        f["model"].cpu()
    print(f"worst over {count} seeds: {worst:.2e}")


if __name__ == "__main__":
    main()
