#!/usr/bin/env python3
"""Folds the per-group counter files of tools/collect_attn_pmc.sh (gpurun_out/<R>_attn_pmc_busy_s<stage>_g<group>.json) into one
record per (stage, kernel) with the derived ratios: usage  python tools/attn_pmc_summary.py r04b > profiles/r04b_attn_pmc_mfma_busy.json"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r04"
recs = {}
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"{R}_attn_pmc_busy_s*_g*.json"))):
    stage = int(re.search(r"_s(\d)_g", f).group(1))
    for kern, counters in json.load(open(f)).items():
        m = re.search(r"(attn_(?:fwd|bwd)_mfma_kernel)ILi(\d)", kern)
        name, hg = (m.group(1), int(m.group(2))) if m else (kern, None)
        r = recs.setdefault((stage, name), {"stage": stage, "kernel": name, "heads_per_workgroup": hg})
        for c, v in counters.items():
            r[c] = v["avg"]
out = []
for (_, _), r in sorted(recs.items()):
    g = r.get
    if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("GRBM_GUI_ACTIVE"):
        r["mfma_busy_frac"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / 32 / g("GRBM_GUI_ACTIVE")
    if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES"):
        r["wave_cycles_waiting_frac"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
    if g("SQ_INSTS_VALU") and g("SQ_INSTS_MFMA"):
        r["valu_insts_per_mfma"] = g("SQ_INSTS_VALU") / g("SQ_INSTS_MFMA")
    if g("SQ_LDS_BANK_CONFLICT") and g("SQ_LDS_IDX_ACTIVE"):
        r["lds_conflict_cycles_per_lds_cycle"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    out.append(r)
print(json.dumps({"note": "rocprofv3 --pmc passes (four counter groups, one per run, --kernel-trace only beside them: tools/collect_attn_pmc.sh) over "
                          "tools/bench_attn_one.py <stage> 3 on the shipped attention kernels (HEAL-SWIN-B @ nside 256 / 12 base pixels, batch 8, bf16). "
                          "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / 32 / GRBM_GUI_ACTIVE; wave_cycles_waiting_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES",
                  "records": out}, indent=1))
