#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_mlp_fused.py tests/test_gpu_training.py tests/test_gpu_model.py tests/test_gpu_baseline_configs.py -q 2>&1 | grep -E "^E  .*AssertionError|passed|failed|^FAILED" | cut -c1-220 > gpurun_out/r05_c6_tests.log 2>&1
cat gpurun_out/r05_c6_tests.log | head -30
timeout 600 python tools/bench_mlp_fused.py 2>/dev/null | tee gpurun_out/r05_c6_mlp_bench.txt
Q="--no-cpu-baseline --no-fp32-companion --no-pmc-traffic --no-graph-companion --no-companions"
for W in T256; do
for F in 1 0; do
HS_FUSED_MLP=$F timeout 300 python bench.py --workload $W --steps 20 --warmup 5 $Q --graph 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W fused=$F graph', round(d['ms_per_step'],2), round(d['value'],1))"
done; done
