#!/bin/bash
# End-of-round evidence: the default bench line and a rocprofv3 kernel-stats pass of the same command (10 timed steps).
# usage: bash tools/end_of_round_profile.sh TAG   -> gpurun_out/TAG_B256_{bench_default.json,bench_under_rocprof.json,kernel_stats.csv,summary.txt}
TAG=${1:-rXX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out
mkdir -p $OUT
python bench.py 2> $OUT/${TAG}_bench_stderr.txt | tail -1 > $OUT/${TAG}_B256_bench_default.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o t -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-companion --no-graph-companion --no-companions --no-pmc-traffic 2>/dev/null | tail -1 > $OUT/${TAG}_B256_bench_under_rocprof.json
cp $(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_B256_kernel_stats.csv
python $ROOT/tools/prof_summary.py $OUT/${TAG}_B256_kernel_stats.csv 13 > $OUT/${TAG}_B256_summary.txt
cat $OUT/${TAG}_B256_summary.txt; cut -c1-400 $OUT/${TAG}_B256_bench_default.json
