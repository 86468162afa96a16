#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r05_call15; mkdir -p $O
X="--workload T256 --steps 10 --warmup 3 --no-companions --no-cpu-baseline --no-fp32-companion --no-graph-companion --no-pmc-traffic --kernel-table --paper-drop-rates"
run() { tag=$1; shift; "$@" 2>$O/$tag.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['value'])"; }
run drop_auto python bench.py $X
run drop_tile2 python tools/policy_ab.py @hs_gemm_nt_set_tile=2 -- $X
run drop_tile3 python tools/policy_ab.py @hs_gemm_nt_set_tile=3 -- $X
for t in drop_auto drop_tile2 drop_tile3; do echo == $t; grep "hs_gemm_nt epi=[12]" $O/$t.err | cut -c1-130; done
