#!/usr/bin/env python3
"""Which Python call sites launch the small torch copy / cast / fill kernels of a training step?  One T@128 step under torch.profiler
with stacks; aten::copy_ / aten::to / aten::clone / aten::contiguous / aten::fill_ / aten::zero_ events grouped by their innermost
frames inside this repository.  usage: python tools/trace_copies.py [workload]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from heal_swin_amd.optim import FlatAdam  # noqa: E402
from heal_swin_amd.parallel import GradBucketAllReduce  # noqa: E402

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "T128"]
dev = torch.device("cuda")
model, cfg, spec = bench.build_model(wl)
model = model.to(dev).train()
model.compute_dtype = torch.bfloat16
dp = GradBucketAllReduce(model.parameters())
opt = FlatAdam(model.parameters(), dp, lr=1e-4, model=model)
g = torch.Generator(device=dev).manual_seed(1)
imgs = torch.randint(0, 256, (8, 3, spec["dim_in"]), generator=g, device=dev, dtype=torch.uint8)
labels = torch.randint(0, spec["f_out"], (8, spec["dim_in"]), generator=g, device=dev, dtype=torch.uint8)


def step():
    dp.zero_grad()
    loss = model.forward_seg_loss(imgs.float(), labels)
    loss.backward()
    dp.finish()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
groups = collections.Counter()
times = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA or not ev.kernels:
        continue
    kn = [k.name for k in ev.kernels]
    if not any(("Memcpy" in k or "copyBuffer" in k or "elementwise" in k or "fillBuffer" in k or "Memset" in k) for k in kn):
        continue
    chain, cur = [], ev
    while cur is not None and len(chain) < 8:
        chain.append(cur.name)
        cur = cur.cpu_parent
    key = (kn[0][:40], " <- ".join(chain))
    groups[key] += 1
    times[key] += sum(k.duration for k in ev.kernels)
for key, n in sorted(groups.items(), key=lambda kv: -times[kv[0]])[:45]:
    print(f"{n:5d} {times[key]:9.1f} us  {key[0]:40s} {key[1][:260]}")
