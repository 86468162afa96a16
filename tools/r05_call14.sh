#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05_call14; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_mlp_fused.py -x -q -m gpu > $O/tests.txt 2>&1
tail -2 $O/tests.txt
bash tools/profile_workload.sh r05_h T256 --paper-drop-rates > /dev/null 2>&1
head -8 gpurun_out/r05_h_T256_summary.txt; grep "layernorm" gpurun_out/r05_h_T256_summary.txt | cut -c1-150
X="--steps 10 --warmup 3 --no-companions --no-cpu-baseline --no-fp32-companion --no-graph-companion --no-pmc-traffic --no-kernel-timing"
run() { tag=$1; shift; "$@" 2>$O/$tag.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'])"; }
run t256_nodrop python bench.py --workload T256 $X
run t256_drop python bench.py --workload T256 $X --paper-drop-rates
