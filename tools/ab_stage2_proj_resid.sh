#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/ab_stage2_proj_resid; mkdir -p $O
X="--workload B256 --steps 10 --warmup 3 --no-companions --no-cpu-baseline --no-fp32-companion --no-graph-companion --no-pmc-traffic --kernel-table"
run() { tag=$1; shift; "$@" 2>$O/$tag.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'])"; }
run b256_base python bench.py $X
run b256_proj512 python tools/policy_ab.py "OWN_SHAPE_TABLE={(576,192):True,(1152,384):True,(384,384):True,(768,256):True,(512,512):True}" -- $X
run b256_base2 python bench.py $X
grep "n=512 k=512\|layernorm\|LayerNorm" $O/b256_base.err $O/b256_proj512.err | cut -c1-160
