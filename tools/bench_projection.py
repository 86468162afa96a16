#!/usr/bin/env python3
"""Fisheye -> HEALPix sampling kernels at WoodScape size (966 x 1280 frames, nside 256, 8 base pixels): frames/s and the
algorithmic bandwidth (per output pixel: 16 B of coordinates once per launch, per plane 4 neighbour bytes + 1 byte written).
(The numpy oracle's time for the same frame is printed by tests/test_gpu_projection.py -- only tests may import the oracle.)
python tools/bench_projection.py [--batch 16] [--json out.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heal_swin_amd import projection as P  # noqa: E402

CAL = dict(name="FV", intrinsic=dict(aspect_ratio=1.0, cx_offset=3.942, cy_offset=-0.472, width=1280.0, height=966.0, poly_order=4,
                                     k1=339.749, k2=-31.988, k3=48.275, k4=-7.201),
           extrinsic=dict(quaternion=[0.5946970238045494, -0.5837953694518585, 0.39063952590941586, -0.39195666481783994]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--nside", type=int, default=256)
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    imgs = torch.from_numpy(rng.integers(0, 256, (a.batch, 3, 966, 1280), dtype=np.uint8)).cuda()
    masks = torch.from_numpy(rng.integers(0, 10, (a.batch, 966, 1280), dtype=np.uint8)).cuda()
    t0 = time.perf_counter()
    proj = P.HPProjector(CAL, a.nside, 8, rotate_pole=True)
    table_s = time.perf_counter() - t0
    for _ in range(3):
        proj(imgs, masks)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    iters = 20
    ev[0].record()
    for _ in range(iters):
        proj(imgs, masks)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / iters
    n = proj.npix
    alg = 2 * 16 * n + a.batch * n * (3 * 5 + 2)  # two launches read the table; 3 image planes (4 in, 1 out) + mask (1 in, 1 out)
    res = dict(nside=a.nside, base_pix=8, npix=n, batch=a.batch, frame="3x966x1280 uint8", ms_per_batch=ms, frames_per_s=a.batch / ms * 1e3,
               algorithmic_GBps=alg / ms / 1e6, table_build_s_host_once_per_calibration=table_s)
    print(json.dumps(res, indent=1))
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
