#!/usr/bin/env python3
"""Count aten-level ops (and their GPU kernels) of one train step of a bench workload: finds stray elementwise launches.
   python tools/op_counts.py [--workload B256] [--batch 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_model  # noqa: E402
from heal_swin_amd.losses import seg_loss  # noqa: E402
from heal_swin_amd.parallel import GradBucketAllReduce  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="B256")
ap.add_argument("--batch", type=int, default=8)
a = ap.parse_args()
wl = WORKLOADS[a.workload]
model, cfg, spec = build_model(wl)
model = model.cuda().train()
model.compute_dtype = torch.bfloat16
dp = GradBucketAllReduce(model.parameters())
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
imgs = torch.randint(0, 256, (a.batch, 3, spec["dim_in"]), device="cuda", dtype=torch.uint8)
labels = torch.randint(0, spec["f_out"], (a.batch, spec["dim_in"]), device="cuda", dtype=torch.uint8)


def step():
    dp.zero_grad()
    seg_loss(model(imgs.float()), labels).backward()
    dp.finish()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=4)
rows = [(k.count, k.key, k.device_time_total if hasattr(k, "device_time_total") else 0, k.stack) for k in ka]
want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add_", "aten::add", "aten::mul", "aten::mul_", "aten::to", "aten::_to_copy",
        "aten::zeros_like", "aten::zeros", "aten::contiguous", "aten::clone", "aten::cat", "aten::sum")
rows = [r for r in rows if r[1] in want]
rows.sort(key=lambda r: -r[2])
for c, k, t, st in rows[:40]:
    print(f"{c:5d} {k:18s} {t/1e3:8.2f} ms  | " + " <- ".join(s.split("/")[-1] for s in st[:4]))
