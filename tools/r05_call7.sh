#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
bash tools/end_of_round_profile.sh r05_b > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
for W in T128 T256; do bash tools/profile_workload.sh r05_b $W > /dev/null 2>&1; done
cd "$GRAFT_REPO_ROOT"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-companion --no-graph-companion --no-companions --no-pmc-traffic --kernel-table > gpurun_out/r05_b_B256_bench_kt.json 2> gpurun_out/r05_b_B256_kernel_table.txt
head -22 gpurun_out/r05_b_B256_summary.txt; cut -c1-300 gpurun_out/r05_b_B256_bench_default.json
