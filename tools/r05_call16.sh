#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
Q="--workload T256 --no-cpu-baseline --no-fp32-companion --no-pmc-traffic --no-graph-companion --no-companions"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_pd && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pd -o t -- python $GRAFT_REPO_ROOT/bench.py $Q --steps 10 --warmup 3 --paper-drop-rates > /dev/null 2>&1
cp $(find /tmp/prof_pd -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/r05_d_T256_paperdrop_kernel_stats.csv
python $GRAFT_REPO_ROOT/tools/prof_summary.py $GRAFT_REPO_ROOT/gpurun_out/r05_d_T256_paperdrop_kernel_stats.csv 13 | head -8
python - <<PY
import csv
rows=sorted(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/r05_d_T256_paperdrop_kernel_stats.csv")), key=lambda r:-int(r["TotalDurationNs"]))
for r in rows[:40]:
    if "layernorm" in r["Name"] or "gemm_nt" in r["Name"]:
        print(f"{int(r['TotalDurationNs'])/13e6:8.3f} {int(r['Calls'])/13:7.1f} {int(r['TotalDurationNs'])/int(r['Calls'])/1e3:9.1f}  {r['Name'][:110]}")
PY
