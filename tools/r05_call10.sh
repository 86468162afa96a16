#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05_call10; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_hygiene.py tests/test_gpu_attn_module.py -x -q -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt
X="--steps 10 --warmup 3 --no-companions --no-cpu-baseline --no-fp32-companion --no-graph-companion --no-pmc-traffic --kernel-table"
run() { tag=$1; shift; "$@" 2>$O/$tag.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'])"; }
run t256_nodrop python bench.py --workload T256 $X
run t256_drop python bench.py --workload T256 $X --paper-drop-rates
grep "window_attn" $O/t256_nodrop.err $O/t256_drop.err | cut -c1-150
