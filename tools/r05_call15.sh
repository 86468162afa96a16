#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_training.py tests/test_gpu_hygiene.py tests/test_gpu_model.py -q 2>&1 | grep -E "^E  .*Assert|passed|failed|^FAILED|Error" | cut -c1-220 | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
Q="--workload T256 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-companion --no-pmc-traffic --no-graph-companion --no-companions"
python bench.py $Q 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('T256 eager', round(d['ms_per_step'],2), round(d['value'],1))"
python bench.py $Q --paper-drop-rates 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('T256 paper-drop eager', round(d['ms_per_step'],2), round(d['value'],1))"
