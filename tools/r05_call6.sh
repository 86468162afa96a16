#!/bin/bash
# chunk-keyed dropout generator: tests, then the paper-drop line against the no-drop line
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05_call6; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_gemm.py tests/test_gpu_hygiene.py -x -q -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt
X="--workload T256 --steps 10 --warmup 3 --no-companions --no-cpu-baseline --no-fp32-companion --no-graph-companion --no-pmc-traffic --kernel-table"
python bench.py $X > $O/t256_nodrop.json 2> $O/t256_nodrop.err
python bench.py $X --paper-drop-rates > $O/t256_drop.json 2> $O/t256_drop.err
for f in t256_nodrop t256_drop; do python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1])
    print("$f", d["ms_per_step"], d["value"])
except Exception as e:
    print("$f", "failed", e)
PY
done > $O/summary.txt
cat $O/summary.txt
