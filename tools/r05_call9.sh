#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05_call9; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_mlp_fused.py -x -q -m gpu > $O/tests.txt 2>&1
tail -15 $O/tests.txt
python tools/bench_mlp_fused.py > $O/mlp_bench.txt 2>&1; tail -12 $O/mlp_bench.txt
X="--steps 10 --warmup 3 --no-companions --no-cpu-baseline --no-fp32-companion --no-graph-companion --no-pmc-traffic"
run() { tag=$1; shift; "$@" 2>$O/$tag.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'], d.get('final_loss'))"; }
run t256_nodrop python bench.py --workload T256 $X
run t256_drop python bench.py --workload T256 $X --paper-drop-rates
HS_FUSED_MLP=0 run t256_drop_composed python bench.py --workload T256 $X --paper-drop-rates
