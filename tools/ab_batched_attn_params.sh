#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/ab_batched_attn_params; mkdir -p $O
timeout 2000 python -m pytest tests/test_gpu_model.py tests/test_gpu_graphs.py tests/test_gpu_parallel.py tests/test_gpu_training.py tests/test_gpu_optim.py tests/test_gpu_deferred.py tests/test_gpu_attn_module.py -x -q -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt
X="--steps 20 --warmup 3 --no-companions --no-cpu-baseline --no-fp32-companion --no-graph-companion --no-pmc-traffic --no-kernel-timing"
run() { tag=$1; shift; "$@" 2>$O/$tag.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'])"; }
for rep in 1 2; do
run t128_graph_batched python bench.py --workload T128 $X --graph
run t128_graph_perblock python tools/policy_ab.py BATCH_ATTN_PARAMS=False -- --workload T128 $X --graph
done
run t128_eager_batched python bench.py --workload T128 $X
run t128_eager_perblock python tools/policy_ab.py BATCH_ATTN_PARAMS=False -- --workload T128 $X
run t256_batched python bench.py --workload T256 $X
run t256_perblock python tools/policy_ab.py BATCH_ATTN_PARAMS=False -- --workload T256 $X
run b256_batched python bench.py --workload B256 $X
run b256_perblock python tools/policy_ab.py BATCH_ATTN_PARAMS=False -- --workload B256 $X
