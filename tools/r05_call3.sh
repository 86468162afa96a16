#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
for W in T128 T256 B256; do bash tools/profile_workload.sh r05_a $W > /dev/null 2>&1; done
cd "$GRAFT_REPO_ROOT"
timeout 300 python bench.py --workload T128 --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-companion --no-graph-companion --no-companions --no-pmc-traffic --kernel-table > gpurun_out/r05_a_T128_bench.json 2> gpurun_out/r05_a_T128_kernel_table.txt
timeout 300 python bench.py --workload T256 --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-companion --no-graph-companion --no-companions --no-pmc-traffic --kernel-table > gpurun_out/r05_a_T256_bench.json 2> gpurun_out/r05_a_T256_kernel_table.txt
head -20 gpurun_out/r05_a_T128_summary.txt
