#!/bin/bash
# rocprofv3 kernel-stats pass of `bench.py --workload W` (10 timed + 3 warm-up steps): family table + the 25 heaviest kernels.
# usage: bash tools/profile_workload.sh TAG W [extra bench args]  -> gpurun_out/TAG_W_{kernel_stats.csv,summary.txt,bench_under_rocprof.json}
TAG=${1:-rXX}; W=${2:-T256}; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_${TAG}_$W
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$W -o t -- python $ROOT/bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-companion --no-graph-companion --no-companions --no-pmc-traffic "$@" 2>/dev/null | tail -1 > $OUT/${TAG}_${W}_bench_under_rocprof.json
cp $(find /tmp/prof_${TAG}_$W -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_${W}_kernel_stats.csv
python $ROOT/tools/prof_summary.py $OUT/${TAG}_${W}_kernel_stats.csv 13 > $OUT/${TAG}_${W}_summary.txt
python - <<PY >> $OUT/${TAG}_${W}_summary.txt
import csv
rows=sorted(csv.DictReader(open("$OUT/${TAG}_${W}_kernel_stats.csv")), key=lambda r:-int(r["TotalDurationNs"]))
print("\nheaviest kernels (ms/step over 13 steps, calls/step, avg us):")
for r in rows[:25]:
    print(f"{int(r['TotalDurationNs'])/13e6:8.3f} {int(r['Calls'])/13:7.1f} {int(r['TotalDurationNs'])/int(r['Calls'])/1e3:9.1f}  {r['Name'][:150]}")
PY
cat $OUT/${TAG}_${W}_summary.txt
