#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
Q="--workload T256 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-companion --no-pmc-traffic --no-graph-companion --no-companions"
python bench.py $Q 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('T256 eager', round(d['ms_per_step'],2), round(d['value'],1))"
python bench.py $Q --paper-drop-rates 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('T256 paper-drop eager', round(d['ms_per_step'],2), round(d['value'],1))"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_pd && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pd -o t -- python $GRAFT_REPO_ROOT/bench.py $Q --steps 10 --warmup 3 --paper-drop-rates > /dev/null 2>&1
cp $(find /tmp/prof_pd -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/r05_c_T256_paperdrop_kernel_stats.csv
python $GRAFT_REPO_ROOT/tools/prof_summary.py $GRAFT_REPO_ROOT/gpurun_out/r05_c_T256_paperdrop_kernel_stats.csv 13 | tee $GRAFT_REPO_ROOT/gpurun_out/r05_c_T256_paperdrop_summary.txt
python - <<PY
import csv
rows=sorted(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/r05_c_T256_paperdrop_kernel_stats.csv")), key=lambda r:-int(r["TotalDurationNs"]))
for r in rows[:22]:
    print(f"{int(r['TotalDurationNs'])/13e6:8.3f} {int(r['Calls'])/13:7.1f} {int(r['TotalDurationNs'])/int(r['Calls'])/1e3:9.1f}  {r['Name'][:120]}")
PY
