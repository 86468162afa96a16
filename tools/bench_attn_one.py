#!/usr/bin/env python3
"""Run the fused attention forward + backward at one stage shape of HEAL-SWIN-B @ 256 (for rocprofv3 counter passes).
   usage: bench_attn_one.py [stage 0..3] [iters] [bf16|fp32]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_swin_amd import ops  # noqa: E402

stage = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dtype = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "fp32") else torch.bfloat16
B, N, C, nh = 8, 196608 // 4 ** stage, 128 * 2 ** stage, 4 * 2 ** stage
if dtype == torch.float32:  # HEAL-SWIN-T @ 256 / 8 base pixels, batch 2 (the fp32 depth-regression shapes)
    B, N, C, nh = 2, 131072 // 4 ** stage, 96 * 2 ** stage, 3 * 2 ** stage
qkv = torch.randn(B, N, 3 * C, device="cuda", dtype=dtype, requires_grad=True)
bias = torch.randn(nh, 64, 64, device="cuda", requires_grad=True)
hs = torch.full((nh,), 0.17, device="cuda")
dout = torch.randn(B, N, C, device="cuda", dtype=dtype)
for _ in range(iters):
    qkv.grad = None
    ops.window_attn_core(qkv, bias, hs, None, 0, None, nh, 64, False).backward(dout)
torch.cuda.synchronize()
