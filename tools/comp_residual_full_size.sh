#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/comp_residual_full_size; mkdir -p $O
HS_COMP_RESIDUAL=1 timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py -q -m gpu -k "headline_config_full_size_logits or paper_config_full_size_logits" > $O/comp_fullsize.txt 2>&1
grep "FULL\[" $O/comp_fullsize.txt | grep "bf16\]" | cut -c1-420
tail -2 $O/comp_fullsize.txt
X="--steps 10 --warmup 3 --no-companions --no-cpu-baseline --no-fp32-companion --no-graph-companion --no-pmc-traffic"
run() { tag=$1; shift; "$@" 2>$O/$tag.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'])"; }
run b256_plain python bench.py --workload B256 $X
HS_COMP_RESIDUAL=1 run b256_comp python bench.py --workload B256 $X
run t256_plain python bench.py --workload T256 $X
HS_COMP_RESIDUAL=1 run t256_comp python bench.py --workload T256 $X
