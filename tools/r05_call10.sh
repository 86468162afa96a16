#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_mlp_fused.py tests/test_gpu_model.py tests/test_gpu_deferred.py -q 2>&1 | grep -E "^E  .*AssertionError|passed|failed|^FAILED|Error" | cut -c1-220 > gpurun_out/r05_c10_tests.log 2>&1
cat gpurun_out/r05_c10_tests.log | head -30
timeout 600 python tools/bench_mlp_fused.py 2>/dev/null | tee gpurun_out/r05_c10_mlp_bench.txt
Q="--no-cpu-baseline --no-fp32-companion --no-pmc-traffic --no-graph-companion"
timeout 600 python bench.py --steps 20 --warmup 5 $Q 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('companions',{})
print('B', round(d['ms_per_step'],2), d['config']['peak_device_memory_GB'], {k:(round(v['value'],1)) for k,v in c.items() if isinstance(v,dict) and 'value' in v})"
