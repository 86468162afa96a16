#!/usr/bin/env python3
"""Run the fused attention forward + backward at one stage shape of a bench workload (for rocprofv3 counter passes).
   usage: bench_attn_one.py [stage 0..3] [iters] [bf16|fp32] [workload = B256 | T256 | D256 ...] [shifted 0|1]
   (B256: nest_roll + scaled attention; T256: the paper config, ring_shift tables + cosine attention; D256: T's shapes with
   nest_roll + scaled attention -- the like-for-like partner of T256)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, full_cfg  # noqa: E402
from heal_swin_amd import ops  # noqa: E402
from heal_swin_amd.models_torch import hp_shifting as S  # noqa: E402

stage = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dtype = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "fp32") else torch.bfloat16
wname = sys.argv[4] if len(sys.argv) > 4 else ("D256" if dtype == torch.float32 else "B256")
shifted = (len(sys.argv) > 5 and sys.argv[5] == "1")
wl = WORKLOADS[wname]
cfg = full_cfg(wl["cfg"])
B = 2 if dtype == torch.float32 else 8
N = wl["base_pix"] * wl["nside"] ** 2 // cfg["patch_size"] // 4 ** stage
C, nh, Ws = cfg["embed_dim"] * 2 ** stage, cfg["num_heads"][stage], cfg["window_size"]
idx = labels = None
roll = 0
if shifted:
    nside = int(round((N // wl["base_pix"]) ** 0.5))
    if cfg["shift_strategy"] == "nest_roll":
        roll = cfg["shift_size"]
        _, _, labels = S.NestRollShift(cfg["shift_size"], N, Ws).tables("cuda")
    elif cfg["shift_strategy"] == "ring_shift":
        idx, _, labels = S.RingShift(nside, wl["base_pix"], Ws, cfg["shift_size"]).tables("cuda")
    else:
        idx, _, labels = S.NestGridShift(nside, wl["base_pix"], Ws).tables("cuda")
qkv = torch.randn(B, N, 3 * C, device="cuda", dtype=dtype, requires_grad=True)
bias = torch.randn(nh, 64, 64, device="cuda", requires_grad=True)
hs = torch.full((nh,), 0.17, device="cuda", requires_grad=bool(cfg["use_cos_attn"]))
dout = torch.randn(B, N, C, device="cuda", dtype=dtype)
for _ in range(iters):
    qkv.grad = None
    ops.window_attn_core(qkv, bias, hs, idx, roll, labels, nh, Ws, bool(cfg["use_cos_attn"])).backward(dout)
torch.cuda.synchronize()
