#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05_call13; mkdir -p $O
X="--steps 10 --warmup 3 --no-companions --no-cpu-baseline --no-fp32-companion --no-graph-companion --no-pmc-traffic --no-kernel-timing"
run() { tag=$1; shift; "$@" 2>$O/$tag.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'], d['config'].get('final_loss'))"; }
run t256_drop_eager python bench.py --workload T256 $X --paper-drop-rates
run t256_drop_graph python bench.py --workload T256 $X --paper-drop-rates --graph
run t128_drop_eager python bench.py --workload T128 $X --paper-drop-rates
run t128_drop_graph python bench.py --workload T128 $X --paper-drop-rates --graph
run t128_nodrop_graph python bench.py --workload T128 $X --graph
tail -3 $O/t256_drop_graph.err
