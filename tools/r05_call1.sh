#!/bin/bash
# round 5, GPU call 1: new tests, whole GPU suite, default bench line + A/B of the flat optimizer / deferred reductions
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
{
echo "=== new tests"; timeout 900 python -m pytest tests/test_gpu_deferred.py tests/test_gpu_optim.py -x -q 2>&1 | tail -15
echo "=== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
} > gpurun_out/r05_c1_tests.log 2>&1
Q="--no-cpu-baseline --no-fp32-companion --no-pmc-traffic --no-graph-companion"
timeout 600 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/r05_c1_bench_new.json 2> gpurun_out/r05_c1_bench_new.err
timeout 600 python bench.py --steps 20 --warmup 5 $Q --torch-adam > gpurun_out/r05_c1_bench_torchadam.json 2> gpurun_out/r05_c1_bench_torchadam.err
HS_DEFER_REDUCE=0 timeout 600 python bench.py --steps 20 --warmup 5 $Q --torch-adam > gpurun_out/r05_c1_bench_old.json 2> gpurun_out/r05_c1_bench_old.err
timeout 600 python bench.py --steps 20 --warmup 5 $Q --no-companions > gpurun_out/r05_c1_bench_new2.json 2> gpurun_out/r05_c1_bench_new2.err
tail -3 gpurun_out/r05_c1_tests.log
for f in new torchadam old new2; do python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r05_c1_bench_$f.json").read().strip().splitlines()[-1])
    c=d.get("companions",{})
    print("$f", round(d["ms_per_step"],2), {k:(round(v["value"],1), round(v["ms_per_step_eager"],2), round(v["ms_per_step_graph"],2)) for k,v in c.items() if isinstance(v,dict) and "value" in v})
except Exception as e: print("$f", "ERR", e)
PY
done
