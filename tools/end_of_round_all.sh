#!/bin/bash
# End-of-round evidence on one box: full GPU suite, smoke, default bench line + rocprof (B256), rocprof of the companions' workloads
TAG=${1:-r05_g}
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gpu_tests.txt 2>&1; tail -3 gpurun_out/${TAG}_gpu_tests.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; tail -4 gpurun_out/${TAG}_smoke.txt
bash tools/end_of_round_profile.sh $TAG > /dev/null 2>&1
python bench.py --kernel-table --no-companions --no-cpu-baseline --no-fp32-companion --no-graph-companion --no-pmc-traffic --steps 10 --warmup 3 2> gpurun_out/${TAG}_B256_kernel_table.txt > /dev/null
bash tools/profile_workload.sh $TAG T256 > /dev/null 2>&1
bash tools/profile_workload.sh $TAG T128 > /dev/null 2>&1
bash tools/profile_workload.sh ${TAG}_paperdrop T256 --paper-drop-rates > /dev/null 2>&1
head -20 gpurun_out/${TAG}_B256_summary.txt
python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_B256_bench_default.json").read().strip().splitlines()[-1])
print("B256", d["ms_per_step"], d["value"], "frac", d["roofline"]["frac"])
c=d.get("companions",{})
for k,v in c.items(): print(k, v.get("value"), v.get("ms_per_step_eager"), v.get("ms_per_step_graph"), v.get("ratio_to_no_drop"))
for k in ("fp32","depth_fp32","graph_replay"): print(k, (d.get(k) or {}).get("value"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][:100])
PY
