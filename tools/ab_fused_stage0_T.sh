#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/ab_fused_stage0_T; mkdir -p $O
X="--steps 10 --warmup 3 --no-companions --no-cpu-baseline --no-fp32-companion --no-graph-companion --no-pmc-traffic"
run() { tag=$1; shift; "$@" 2>$O/$tag.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'])"; }
run t256_fused python bench.py --workload T256 $X
HS_FUSED_ATTN_TRAIN=0 run t256_composed_attn python bench.py --workload T256 $X
HS_FUSED_MLP=0 run t256_composed_mlp python bench.py --workload T256 $X
run t128_fused python bench.py --workload T128 $X
HS_FUSED_ATTN_TRAIN=0 run t128_composed_attn python bench.py --workload T128 $X
run t256_fused2 python bench.py --workload T256 $X
run t256_drop python bench.py --workload T256 $X --paper-drop-rates
