#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_baseline_configs.py -q -k "attn or attention or configs1 or paper" 2>&1 | grep -E "^E  .*AssertionError|passed|failed|^FAILED|Error" | cut -c1-220 | head -20
python tools/bench_attn.py --workload T128 --iters 10 2>/dev/null | head -3
Q="--no-cpu-baseline --no-fp32-companion --no-pmc-traffic --no-graph-companion"
timeout 600 python bench.py --steps 20 --warmup 5 $Q 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('companions',{})
print('B', round(d['ms_per_step'],2), d['config']['peak_device_memory_GB'], {k:(round(v['value'],1), round(v.get('ratio_to_no_drop_eager',0),3)) for k,v in c.items() if isinstance(v,dict) and 'value' in v})"
