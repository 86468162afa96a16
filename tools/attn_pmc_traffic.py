#!/usr/bin/env python3
"""HBM traffic of the fused attention kernels from two rocprofv3 PMC passes over `tools/bench_attn.py --iters 2`:
   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d A -o f -- python tools/bench_attn.py --iters 2
   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d B -o w -- python tools/bench_attn.py --iters 2
   python tools/attn_pmc_traffic.py A/.../f_counter_collection.csv B/.../w_counter_collection.csv A/.../f_kernel_trace.csv > out.json
Corrections per MI355X_MICROARCH.md (HBM section): both counters are in KiB; FETCH_SIZE counts half of wide coalesced reads
on gfx950 (x2).  bench_attn.py launches, per (stage, shifted): 4 forward, then 4 x (forward, backward)."""
import collections
import csv
import json
import sys


def per_dispatch(path, counter):
    vals = collections.OrderedDict()
    names = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        d = int(r["Dispatch_Id"])
        vals[d] = vals.get(d, 0.0) + float(r["Counter_Value"])
        names[d] = r["Kernel_Name"]
    return vals, names


def attn_sequence(vals, names):
    seq = []
    for d in sorted(vals):
        n = names[d]
        if "attn_fwd_mfma" in n or "attn_fwd2" in n:
            seq.append(("hs_window_attn_fwd", vals[d], d))
        elif "attn_bwd_mfma" in n or "attn_bwd2" in n:
            seq.append(("hs_window_attn_bwd", vals[d], d))
    return seq


def records(fetch_csv, write_csv, trace_csv, wl=None, batch=8):
    """wl: a bench.py WORKLOADS entry (default: HEAL-SWIN-B @ nside 256, 12 base pixels)."""
    fetch, fn = per_dispatch(fetch_csv, "FETCH_SIZE")
    write, wn = per_dispatch(write_csv, "WRITE_SIZE")
    dur = {}
    for r in csv.DictReader(open(trace_csv)):
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    fs, ws = attn_sequence(fetch, fn), attn_sequence(write, wn)
    assert len(fs) == len(ws) and len(fs) % 12 == 0, (len(fs), len(ws))
    if wl is None:
        N0, C0, nstage = 196608, 128, 4
    else:
        N0, C0 = wl["base_pix"] * wl["nside"] ** 2 // 4, wl["cfg"]["embed_dim"]
        nstage = len(wl["cfg"]["depths"])
    shapes = [(N0 // 4 ** s, C0 * 2 ** s) for s in range(nstage) if N0 // 4 ** s >= 64]
    out = []
    for cfg in range(len(fs) // 12):
        stage, shifted = cfg // 2, cfg % 2
        N, C = shapes[stage]
        E = batch * N * C * 2  # bytes of one [B, N, C] bf16 tensor
        for kern, mult in (("hs_window_attn_fwd", 4), ("hs_window_attn_bwd", 7)):  # bwd: q, k, v, dO in, dq, dk, dv out (no O read since round 3)
            f = [v for k, v, _ in fs[cfg * 12:(cfg + 1) * 12] if k == kern]
            w = [v for k, v, _ in ws[cfg * 12:(cfg + 1) * 12] if k == kern]
            t = [dur[d] for k, _, d in fs[cfg * 12:(cfg + 1) * 12] if k == kern and d in dur]
            fk, wk = sum(f) / len(f), sum(w) / len(w)
            hbm = (2 * fk + wk) * 1024
            out.append({"stage": stage, "shifted": shifted, "kernel": kern, "launches": len(f), "FETCH_SIZE_KiB": fk,
                        "WRITE_SIZE_KiB": wk, "hbm_bytes_corrected": hbm, "algorithmic_bytes": mult * E,
                        "traffic_over_algorithmic": hbm / (mult * E), "avg_us_under_pmc": sum(t) / max(1, len(t))})
    return out


def main():
    wl, name = None, "HEAL-SWIN-B"
    if len(sys.argv) > 4:  # a bench.py workload name: its stage shapes instead of the default's
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import WORKLOADS, full_cfg
        wl = dict(WORKLOADS[sys.argv[4]])
        wl["cfg"] = full_cfg(wl["cfg"])
        name = sys.argv[4]
    recs = records(sys.argv[1], sys.argv[2], sys.argv[3], wl)
    print(json.dumps({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/bench_attn.py --iters 2 "
                              "(" + name + " stage shapes, batch 8, bf16); corrected per MI355X_MICROARCH.md HBM section: KiB units, "
                              "FETCH_SIZE x2 on gfx950 for wide coalesced reads (tools/attn_pmc_traffic.py)", "records": recs}, indent=1))


if __name__ == "__main__":
    main()
