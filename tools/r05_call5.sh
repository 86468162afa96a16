#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r05_c5_tests.log
Q="--no-cpu-baseline --no-fp32-companion --no-pmc-traffic --no-graph-companion"
timeout 600 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/r05_c5_bench_fused.json 2> gpurun_out/r05_c5_bench_fused.err
HS_FUSED_MLP=0 timeout 600 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/r05_c5_bench_nofused.json 2> gpurun_out/r05_c5_bench_nofused.err
timeout 600 python bench.py --steps 20 --warmup 5 $Q --no-companions > gpurun_out/r05_c5_bench_fused2.json 2> /dev/null
tail -4 gpurun_out/r05_c5_tests.log
for f in fused nofused fused2; do python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05_c5_bench_$f.json").read().strip().splitlines()[-1])
    c=d.get("companions",{})
    print("$f", round(d["ms_per_step"],2), d["config"].get("peak_device_memory_GB"), {k:(round(v["value"],1), round(v["ms_per_step_eager"],2), round(v["ms_per_step_graph"],2)) for k,v in c.items() if isinstance(v,dict) and "value" in v})
except Exception as e: print("$f", "ERR", e)
PY
done
