#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
Q="--workload T256 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-companion --no-pmc-traffic --no-graph-companion --no-companions --graph"
for V in 1 0 1 0; do
HS_TMP_RESID_OWN=$V python bench.py $Q 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('T256 resid_own=$V', round(d['ms_per_step'],2), round(d['value'],1))"
done
