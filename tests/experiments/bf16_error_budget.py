#!/usr/bin/env python3
"""Where does the bf16 logit error of HEAL-SWIN-B come from?  CPU experiment with the oracle (test infrastructure): the fp32
oracle forward is re-run with bf16 ROUNDING injected at one class of tensors at a time -- (s) the residual stream written by
every block, (n) LayerNorm outputs, (l) Linear outputs (qkv, attention output, proj, fc1 / GELU, fc2), (o) the logits -- and the logits
are compared with the un-rounded run (max|a-b| / max|b|, as the parity tests measure).  Usage: bf16_error_budget.py [nside] [seed]"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tests/experiments/ -> repo root
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from oracle import model as OM  # noqa: E402

r = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
MODE = set()
_ln, _lin, _blk, _gelu = OM.layer_norm, OM.linear, OM.swin_block, OM.gelu


def ln(x, w, b, eps=OM.LN_EPS):
    y = _ln(x, w, b, eps)
    return r(y) if "n" in MODE else y


def lin(x, w, b=None):
    y = _lin(x, w, b)
    return r(y) if "l" in MODE else y


def gelu(x):
    y = _gelu(x)
    return r(y) if "l" in MODE else y


def blk(x, *a, **k):
    y = _blk(x, *a, **k)
    return r(y) if "s" in MODE else y


OM.layer_norm, OM.linear, OM.swin_block, OM.gelu = ln, lin, blk, gelu


def main():
    nside = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 13
    import test_gpu_baseline_configs as T
    T.CASES["_x"] = (T.B_CFG, nside, 12, 1, dict(shift_strategy="nest_roll", shift_size=32))
    model, cfg, spec, x, y = T._setup_seeded("_x", seed)
    sd = {k: v.detach() for k, v in model.state_dict().items() if not k.endswith("attn_mask")}
    torch.set_num_threads(8)
    out = {}
    with torch.no_grad():
        for mode in ("", "s", "n", "l", "o", "nl", "snl", "nlo", "snlo"):
            MODE.clear()
            MODE.update(mode)
            y = OM.forward(sd, types.SimpleNamespace(**cfg), types.SimpleNamespace(**spec), r(x) if mode else x)
            out[mode] = r(y) if "o" in MODE else y  # (o) the logits themselves stored in bf16
    ref = out[""]
    for mode in ("s", "n", "l", "o", "nl", "snl", "nlo", "snlo"):
        e = (out[mode] - ref).abs().max() / ref.abs().max()
        print(f"nside {nside} seed {seed} rounding {mode:4s}: logits max|a-b|/max|b| = {float(e):.2e}")


if __name__ == "__main__":
    main()
